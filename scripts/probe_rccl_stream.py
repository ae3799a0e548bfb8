"""Which stream does torch's NCCL(=RCCL) process group run a collective on?  (dist.py / DESIGN section 6 rely on: a synchronous collective runs on
the caller's CURRENT stream, so pack -> all-reduce -> unpack need no host involvement and no fifth stream exists.)  One GPU is enough to see it: with
one rank RCCL turns an OUT-OF-PLACE collective into a device copy on the stream it was handed, which a kernel trace shows with its stream id next to
marker kernels this script launches on the caller's stream.
    rocprofv3 --kernel-trace --output-format csv -d out -o rccl -- python scripts/probe_rccl_stream.py
then  python scripts/probe_rccl_stream.py --read out/rccl_kernel_trace.csv"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    keep = [r for r in rows if "spin_kernel" in r["Kernel_Name"] or "copyBuffer" in r["Kernel_Name"] or "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()]
    for r in keep[-16:]:
        print(r["Kernel_Name"][:60].ljust(60), "stream", r.get("Stream_Id"), "queue", r.get("Queue_Id"), "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    sys.exit(0)

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

from gnn_pretraining_amd import _lib as L

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
lib = L.lib()
inp, out = torch.ones(1 << 20, device=dev), torch.empty(1 << 20, device=dev)
side = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
for rep in range(3):
    with torch.cuda.stream(side):
        h = side.cuda_stream
        L.check(lib.gmp_spin_us(300, h), "spin")          # marker A on the caller's stream
        dist.all_gather_into_tensor(out, inp)              # synchronous collective, one rank: a device copy on the stream RCCL was given
        L.check(lib.gmp_spin_us(100, h), "spin")          # marker B
    torch.cuda.synchronize()
with torch.cuda.stream(side):
    L.check(lib.gmp_spin_us(300, side.cuda_stream), "spin")
    w = dist.all_gather_into_tensor(out, inp, async_op=True)       # for comparison: the asynchronous form
    L.check(lib.gmp_spin_us(100, side.cuda_stream), "spin")
    w.wait()
torch.cuda.synchronize()
print("torch", torch.__version__, "caller stream handle", hex(side.cuda_stream), "default stream", hex(torch.cuda.default_stream(dev).cuda_stream))
dist.destroy_process_group()
