"""GPU time of each kernel of one forward backbone layer at the s4 step's shape, measured with the GPU kept busy ahead of the
loop (so the host is never the limiter): each kernel alone back to back, and the five of them chained."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gnn_pretraining_amd import _lib as L
from gnn_pretraining_amd.engine import StepEngine, NT
from gnn_pretraining_amd.models.pretrain_model import PretrainableGNN
from gnn_pretraining_amd.pretrain import pretrain as PT

dev = torch.device("cuda:0")
torch.set_num_threads(1)
model = PretrainableGNN(device=dev, domain_names=PT.PRETRAIN_DOMAINS["s4"], task_names=PT.ACTIVE_TASKS["s4"])
model.train()
eng = StepEngine(model, PT.ACTIVE_TASKS["s4"], PT.PRETRAIN_DOMAINS["s4"], dev, seed=0, rng_mode="vectorized")
pool = B.make_pool(0, dev, eng.dpad)
gen = torch.Generator().manual_seed(0)
for k in range(5):
    eng.step(pool[k], gen)
p = eng.last_plan
lib, st, N, H = eng.lib, eng._st(), p.N, 256
P = eng._P
l = 2
pre = f"gnn_backbone.layers.{l}."
c = eng.csr
big = torch.zeros(256 * 1024 * 1024, device=dev)
cfg1, cfg2 = eng._bn_cfg(True, False, 0), eng._bn_cfg(True, True, 10 + l)


def k_agg(): eng._chk(lib.gmp_gin_aggregate_fwd(eng.h[l].data_ptr(), c[0].data_ptr(), c[1].data_ptr(), P(pre + "gin_conv.eps"), eng.a[l].data_ptr(), N, H, st), "a")
def k_g1(): eng._gemm(NT, eng.a[l].data_ptr(), P(pre + "gin_conv.nn.0.weight"), P(pre + "gin_conv.nn.0.bias"), eng.z1[l].data_ptr(), N, 2 * H, H, H, H, 2 * H)
def k_b1(): eng._chk(lib.gmp_bn_fwd(eng.z1[l].data_ptr(), None, p.d32["seg_ptr"], None, p.S, p.max_seg, N, 2 * H, P(pre + "gin_conv.nn.1.weight"), P(pre + "gin_conv.nn.1.bias"), None, None,
                                    eng.stat["m1"][l].data_ptr(), eng.stat["s1"][l].data_ptr(), eng.r1[l].data_ptr(), C.byref(cfg1), eng.bn_ws.data_ptr(), eng.bn_ws.numel(), st), "b1")
def k_g2(): eng._gemm(NT, eng.r1[l].data_ptr(), P(pre + "gin_conv.nn.3.weight"), P(pre + "gin_conv.nn.3.bias"), eng.z2[l].data_ptr(), N, H, 2 * H, 2 * H, 2 * H, H)
def k_b2(): eng._chk(lib.gmp_bn_fwd(eng.z2[l].data_ptr(), eng.h[l].data_ptr(), p.d32["seg_ptr"], None, p.S, p.max_seg, N, H, P(pre + "batch_norm.weight"), P(pre + "batch_norm.bias"), None, None,
                                    eng.stat["m2"][l].data_ptr(), eng.stat["s2"][l].data_ptr(), eng.h[l + 1].data_ptr(), C.byref(cfg2), eng.bn_ws.data_ptr(), eng.bn_ws.numel(), st), "b2")


def timed(label, fn, n=400):
    torch.cuda.synchronize()
    for _ in range(120): big.add_(1.0)                 # ~50 ms of GPU work: the host enqueues the loop below meanwhile
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    per = a.elapsed_time(b) / n * 1e3
    print(f"{label:60s} {per:8.2f} us" + (f"  ({per / 5:.1f} per layer)" if "layers" in label or "flush" in label else ""))

print("rows", N, "segments", p.S, "max segment", p.max_seg)
for lab, fn in (("aggregate", k_agg), ("gemm 256->512 (+bias)", k_g1), ("bn 512 + relu", k_b1), ("gemm 512->256 (+bias)", k_g2), ("bn 256 + res + relu + dropout", k_b2)):
    timed(lab, fn)
timed("the five chained (one layer)", lambda: (k_agg(), k_g1(), k_b1(), k_g2(), k_b2()), n=100)


def layer(l):
    pre = f"gnn_backbone.layers.{l}."
    c1, c2 = eng._bn_cfg(True, False, 0), eng._bn_cfg(True, True, 10 + l)
    eng._chk(lib.gmp_gin_aggregate_fwd(eng.h[l].data_ptr(), c[0].data_ptr(), c[1].data_ptr(), P(pre + "gin_conv.eps"), eng.a[l].data_ptr(), N, H, st), "a")
    eng._gemm(NT, eng.a[l].data_ptr(), P(pre + "gin_conv.nn.0.weight"), P(pre + "gin_conv.nn.0.bias"), eng.z1[l].data_ptr(), N, 2 * H, H, H, H, 2 * H)
    eng._chk(lib.gmp_bn_fwd(eng.z1[l].data_ptr(), None, p.d32["seg_ptr"], None, p.S, p.max_seg, N, 2 * H, P(pre + "gin_conv.nn.1.weight"), P(pre + "gin_conv.nn.1.bias"), None, None,
                            eng.stat["m1"][l].data_ptr(), eng.stat["s1"][l].data_ptr(), eng.r1[l].data_ptr(), C.byref(c1), eng.bn_ws.data_ptr(), eng.bn_ws.numel(), st), "b1")
    eng._gemm(NT, eng.r1[l].data_ptr(), P(pre + "gin_conv.nn.3.weight"), P(pre + "gin_conv.nn.3.bias"), eng.z2[l].data_ptr(), N, H, 2 * H, 2 * H, 2 * H, H)
    eng._chk(lib.gmp_bn_fwd(eng.z2[l].data_ptr(), eng.h[l].data_ptr(), p.d32["seg_ptr"], None, p.S, p.max_seg, N, H, P(pre + "batch_norm.weight"), P(pre + "batch_norm.bias"), None, None,
                            eng.stat["m2"][l].data_ptr(), eng.stat["s2"][l].data_ptr(), eng.h[l + 1].data_ptr(), C.byref(c2), eng.bn_ws.data_ptr(), eng.bn_ws.numel(), st), "b2")


timed("five layers chained, per layer", lambda: [layer(i) for i in range(5)], n=40)
junk = torch.zeros(96 * 1024 * 1024, device=dev)          # 384 MB touched between passes: evicts L2 + MALL
timed("same, caches flushed between passes (incl. flush kernel ~0.15 ms / 5)", lambda: ([layer(i) for i in range(5)], junk.add_(1.0)), n=40)
timed("flush kernel alone / 5", lambda: junk.add_(1.0), n=40)

inp = pool[4 % len(pool)]

timed("engine._forward (python path: encoders + CSR wait + 5 layers)", lambda: eng._forward(p, inp), n=40)
