"""Layer-GEMM kernel time without Python wrapper overhead: the C entry point called in a tight loop with prebuilt arguments (host ~2 us per call), HIP
events around 200 launches.  python scripts/bench_gemm_raw.py  -> us per launch for each ring variant / shape."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import torch

from gnn_pretraining_amd import _lib as L

dev = "cuda:0"
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
NT, NN = 0, 1


def run(mode, M, N, K, iters=200):
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev) if mode == NT else torch.randn(K, N, device=dev)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    f = lib.gmp_gemm_f32
    args = (mode, A.data_ptr(), B.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, K, B.size(1), N, 1.0, 0, 0, None, 0, st)
    for _ in range(20):
        f(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        f(*args)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, (t1 - t0) / iters * 1e6


shapes = [(NT, 7392, 512, 256), (NT, 3700, 512, 256), (NT, 7392, 256, 512), (NN, 7392, 512, 256), (NN, 7392, 256, 512), (NT, 15000, 256, 768)]
variants = sys.argv[1:] or ["0", "3", "32"]
os.environ["GMP_GEMM_PIPE_TILE"] = os.environ.get("GMP_GEMM_PIPE_TILE", "3")
res = {}
for rnd in range(3):
    for v in variants:
        os.environ["GMP_GEMM_PIPE_STAGES"] = v
        for s in shapes:
            us, host = run(*s)
            res.setdefault((s, v), []).append((us, host))
for s in shapes:
    mode, M, N, K = s
    fl = 2.0 * M * N * K
    print(f"{'NT' if mode == NT else 'NN'} {M}x{N}x{K}:" + "".join(
        f" | st {v:>2s}: {min(x[0] for x in res[(s, v)]):6.1f} us {fl / min(x[0] for x in res[(s, v)]) / 1e6:5.1f} TF (host {min(x[1] for x in res[(s, v)]):4.1f})" for v in variants))
