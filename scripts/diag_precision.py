"""Diagnostic (not a test): is the HIP-vs-oracle gradient gap fp32 rounding noise?
Compares the fp32 oracle and the HIP path against an fp64 run of the oracle."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gnn_pretraining_amd import synthetic as S
from gnn_pretraining_amd.models import GINBackbone
from oracle import models as OM
from parity_util import copy_state, set_dropout

DEV = torch.device("cuda:0")
for training in (True, False):
    gen = torch.Generator().manual_seed(3); torch.manual_seed(3)
    ob = OM.GINBackbone()
    for m in ob.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = torch.rand(m.weight.shape, generator=gen) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=gen) * 0.1
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=gen) * 0.1
            m.running_var.data = torch.rand(m.running_var.shape, generator=gen) + 0.5
    for l in ob.layers: l.gin_conv.eps.data.fill_(0.1)
    hb = GINBackbone(); copy_state(hb, ob); hb.to(DEV)
    o64 = copy.deepcopy(ob).double()
    for m in (ob, hb, o64):
        set_dropout(m, 0.0); m.train(training)
    b = S.domain_batch(gen, 21, 8)
    h0 = torch.randn(b.num_nodes, 256, generator=gen); g = torch.randn(b.num_nodes, 256, generator=gen)
    def run(model, x, ei, gg):
        x = x.clone().requires_grad_(); y = model(x, ei); y.backward(gg)
        return y.detach(), x.grad, {n: p.grad for n, p in model.named_parameters()}
    y32, gx32, gp32 = run(ob, h0, b.edge_index, g)
    y64, gx64, gp64 = run(o64, h0.double(), b.edge_index, g.double())
    yh, gxh, gph = run(hb, h0.to(DEV), b.edge_index.to(DEV), g.to(DEV))
    def e(a, ref): return ((a.cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    print(f"training={training}")
    print(f"  out     oracle32 {e(y32,y64):.2e}  hip {e(yh,y64):.2e}")
    print(f"  grad_h0 oracle32 {e(gx32,gx64):.2e}  hip {e(gxh,gx64):.2e}")
    d = (gxh.cpu().double() - gx64).abs() / gx64.abs().max()
    print("  grad_h0 elements off by >1e-5:", int((d > 1e-5).sum()), "rows:", (d > 1e-5).any(1).nonzero().flatten().tolist()[:10], "ptr", b.ptr_host)
    worst = sorted(((e(gph[n], gp64[n]), e(gp32[n], gp64[n]), n) for n in gp64), reverse=True)[:6]
    for eh, eo, n in worst:
        print(f"  {n:45s} hip {eh:.2e} oracle32 {eo:.2e}  |g|max {gp64[n].abs().max().item():.2e}")
