"""(probe) does the gloo backend run reduce_scatter_tensor / all_gather_into_tensor on CUDA tensors?  two ranks sharing GPU 0"""
import os, sys, subprocess
if len(sys.argv) == 1:
    ps = [subprocess.Popen([sys.executable, __file__, str(r)]) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
r = int(sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29593", RANK=str(r), WORLD_SIZE="2")
import torch, torch.distributed as dist
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
inp = (torch.arange(8.) + r).to(dev); out = torch.empty(4, device=dev)
for name, f in (("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(out, inp)),
                ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.empty(8, device=dev), out))):
    try:
        f(); torch.cuda.synchronize(); print(r, name, "ok", out.tolist())
    except Exception as e:
        print(r, name, "FAILED", repr(e)[:160])
dist.destroy_process_group()
