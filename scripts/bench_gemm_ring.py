"""Ring variants of the pipelined GEMM (GMP_GEMM_PIPE_STAGES: 0 = 4 stages / 4 buffers, 3 = 3 / 3, 32 = 3-stage schedule on 2 buffers) at the step's shapes: correctness against fp64, then interleaved timing rounds (min of 3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench_gemm_pipe as G
from gnn_pretraining_amd import ops

dev = "cuda:0"
VARIANTS = ["0", "3", "32"]
os.environ["GMP_GEMM_PIPE_TILE"] = "3"
for v in VARIANTS:
    os.environ["GMP_GEMM_PIPE_STAGES"] = v
    worst = 0.0
    for (M, N, K) in ((7392, 512, 256), (6507, 256, 512), (2049, 132, 768), (1025, 64, 64), (3700, 256, 96)):
        g = torch.Generator().manual_seed(M)
        A, W, b = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
        ref = A.double() @ W.double().t() + b.double()
        worst = max(worst, ((ops.gemm(ops.NT, A, W, b).double() - ref).abs().max() / ref.abs().max()).item())
        Gm = torch.randn(M, N, generator=g).to(dev)
        refn = Gm.double() @ W.double()
        worst = max(worst, ((ops.gemm(ops.NN, Gm, W).double() - refn).abs().max() / refn.abs().max()).item())
        Ai = torch.randint(-4, 5, (M, K), generator=g).float().to(dev)
        Wi = torch.randint(-4, 5, (N, K), generator=g).float().to(dev)
        assert torch.equal(ops.gemm(ops.NT, Ai, Wi).double(), Ai.double() @ Wi.double().t()), (v, M, N, K)
    R = 7391
    rows = [0, R // 7, 2 * R // 7 + 3, R // 2 + 1, R - 300, R]
    Gm, X = torch.randn(R, 512, device=dev), torch.randn(R, 256, device=dev)
    out, bo = torch.empty(5, 512, 256, device=dev), torch.empty(5, 512, device=dev)
    G.grouped_tn(Gm, X, rows, out, bo, torch.empty(32 << 20, dtype=torch.uint8, device=dev))
    for i in range(5):
        ref = Gm[rows[i]:rows[i + 1]].double().t() @ X[rows[i]:rows[i + 1]].double()
        worst = max(worst, ((out[i].double() - ref).abs().max() / ref.abs().max()).item())
    print(f"stages {v}: worst rel err {worst:.2e}")
    assert worst < 2e-5

cases = []
for (mode, name, m, n, k) in ((ops.NT, "NT 7392x512x256", 7392, 512, 256), (ops.NT, "NT 3700x512x256", 3700, 512, 256), (ops.NT, "NT 7392x256x512", 7392, 256, 512),
                              (ops.NN, "NN 7392x512x256", 7392, 512, 256), (ops.NN, "NN 7392x256x512", 7392, 256, 512),
                              (ops.NT, "NT 15000x256x768", 15000, 256, 768), (ops.NN, "NN 15000x768x256", 15000, 768, 256)):
    A = torch.randn(m, k, device=dev)
    B = torch.randn(n, k, device=dev) if mode == ops.NT else torch.randn(k, n, device=dev)
    out = torch.empty(m, n, device=dev)
    cases.append((name, 2.0 * m * n * k, (lambda mode=mode, A=A, B=B, out=out: ops.gemm(mode, A, B, out=out))))
R = 7392
rows = [0, R // 7, 2 * R // 7, (2 * R + R * 8 // 5) // 7, (2 * R + R * 16 // 5) // 7, R]
ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
for (Mo, No) in ((256, 512), (512, 256)):
    Gm, X = torch.randn(R, Mo, device=dev), torch.randn(R, No, device=dev)
    out, bo = torch.empty(5, Mo, No, device=dev), torch.empty(5, Mo, device=dev)
    cases.append((f"TN 5 tasks [{R},{Mo}]^T [{R},{No}]", 2.0 * R * Mo * No, (lambda Gm=Gm, X=X, out=out, bo=bo: G.grouped_tn(Gm, X, rows, out, bo, ws))))
res = {}
for rnd in range(3):
    for v in VARIANTS:
        os.environ["GMP_GEMM_PIPE_STAGES"] = v
        for (name, flops, f) in cases:
            res.setdefault((name, v), []).append(G.timeit(f))
for (name, flops, f) in cases:
    print(f"{name:34s}" + "".join(f" | st {v:>2s}: {min(res[(name, v)]):6.1f} us {flops / min(res[(name, v)]) / 1e6:5.1f} TF" for v in VARIANTS))
