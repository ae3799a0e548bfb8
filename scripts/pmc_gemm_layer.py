"""20 launches of the backbone-layer GEMM (M x 256 -> 512, fp32 MFMA, bias) for a rocprofv3 --pmc pass (profiles/README.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import ops
M = int(os.environ.get("M", 7392))
A, B, b = torch.randn(M, 256, device="cuda:0"), torch.randn(512, 256, device="cuda:0"), torch.randn(512, device="cuda:0")
out = torch.empty(M, 512, device="cuda:0")
for _ in range(20):
    ops.gemm(ops.NT, A, B, b, out)
torch.cuda.synchronize()
