"""K sweep of the pipelined GEMM at the layer's M, N: time = fixed + per-K-step (tuning aid).  Run under
rocprofv3 --kernel-trace --stats to read kernel durations instead of launch-to-launch times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import ops
dev = "cuda:0"
M, N = int(os.environ.get("M", 7392)), int(os.environ.get("N", 512))
def timeit(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for tile in os.environ.get("TILES", "0,1,3").split(","):
    os.environ["GMP_GEMM_PIPE_TILE"] = tile
    row = []
    for K in (64, 128, 256, 512, 1024, 2048):
        A, B, out = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        row.append((K, min(timeit(lambda: ops.gemm(ops.NT, A, B, out=out)) for _ in range(3))))
    print(f"tile {tile}: " + "  ".join(f"K={k}: {us:.1f}us" for k, us in row) +
          f"  | per 32-step {(row[-1][1] - row[2][1]) / ((2048 - 256) / 32):.3f} us, fixed {row[2][1] - 8 * (row[-1][1] - row[2][1]) / 56:.1f} us")
