"""K-slices of the NT form (gmp_gemm_f32 with a workspace) against the unsliced launch on shapes with few output tiles: where does the rule of
gmp_gemm_f32_workspace_bytes pay?  python scripts/bench_gemm_slices.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pretraining_amd import _lib as L, ops
dev = torch.device("cuda:0")
lib = L.lib()
def bench(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for (M, K, N) in ((2708, 1440, 256), (2708, 512, 256), (2708, 1024, 256), (3700, 512, 256), (1100, 512, 256), (1100, 1440, 256), (1100, 256, 512), (1500, 768, 256), (600, 512, 256)):
    A, B, bias, out = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev), torch.empty(M, N, device=dev)
    wsb = lib.gmp_gemm_f32_workspace_bytes(0, M, N, K)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    def call(use_ws):
        L.check(lib.gmp_gemm_f32(0, A.data_ptr(), B.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, K, K, N, 1.0, 0, 0,
                                 ws.data_ptr() if use_ws else None, wsb if use_ws else 0, st), "gemm")
    t0, t1 = bench(lambda: call(False)), bench(lambda: call(True))
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    print(f"NT {M}x{K}->{N}: {tiles} tiles, slices {wsb // (M * N * 4) if wsb else 1}: unsliced {t0:.1f} us, sliced {t1:.1f} us")
