"""The pipelined fp32 MFMA GEMM (csrc/gemm_pipe.h) against the first kernel and against rocBLAS, at the stacked step's shapes:
correctness vs an fp64 product first, then interleaved A/B timing rounds in ONE process (GMP_GEMM_IMPL is read per call).
Tuning aid, not a test (tests/test_gpu_ops.py holds the parity tests)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gnn_pretraining_amd import _lib as L, ops

dev = "cuda:0"
torch.manual_seed(0)
lib = L.lib()


def P(t):
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def i32(xs):
    return (C.c_int32 * len(xs))(*xs)


def i64(xs):
    return (C.c_int64 * len(xs))(*xs)


def grouped_tn(G, X, rows, out, bias_out, ws):
    """out[g] = G[rows[g]:rows[g+1]]^T X[rows[g]:rows[g+1]] ; bias_out[g] = column sums of G over the group's rows."""
    ng = len(rows) - 1
    Mo, No = G.size(1), X.size(1)
    L.check(lib.gmp_gemm_f32_grouped(ops.TN, P(G), P(X), None, P(out), ng, i32(rows), None, None, i64([g * Mo * No for g in range(ng)]),
                                     P(bias_out), i64([g * Mo for g in range(ng)]), Mo, No, 0, Mo, No, No, 1.0, 0, 0,
                                     P(ws) if ws is not None else None, ws.numel() if ws is not None else 0, stream()), "grouped TN")


def check():
    worst = 0.0
    for impl in ("new", "old"):
        os.environ["GMP_GEMM_IMPL"] = impl
        for tile in ("-1", "0", "1", "2", "3"):
            if impl == "old" and tile != "-1":
                continue
            os.environ["GMP_GEMM_PIPE_TILE"] = tile
            for (M, N, K) in ((7392, 512, 256), (6507, 256, 512), (2049, 130, 768), (1025, 64, 64), (7700, 256, 768)):
                A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
                bias = torch.randn(N, device=dev)
                ref = (A.double() @ W.double().t() + bias.double())
                got = ops.gemm(ops.NT, A, W, bias)
                e = ((got.double() - ref).abs().max() / ref.abs().max()).item()
                got_r = ops.gemm(ops.NT, A, W, bias, relu=True)
                e = max(e, ((got_r.double() - ref.clamp_min(0)).abs().max() / ref.abs().max()).item())
                Gm = torch.randn(M, N, device=dev)
                if N % 4 == 0:
                    refn = Gm.double() @ W.double()
                    gotn = ops.gemm(ops.NN, Gm, W)
                    e = max(e, ((gotn.double() - refn).abs().max() / refn.abs().max()).item())
                    acc0 = torch.randn(M, K, device=dev)
                    gota = ops.gemm(ops.NN, Gm, W, out=acc0.clone(), alpha=0.5, accumulate=True)
                    e = max(e, ((gota.double() - (acc0.double() + 0.5 * refn)).abs().max() / refn.abs().max()).item())
                worst = max(worst, e)
                print(f"impl {impl} tile {tile:>2s}  NT/NN {M}x{N}x{K}: max rel err {e:.2e}")
            # grouped weight gradient over uneven task row ranges, with and without a workspace (row slices), ragged tails
            for (R, Mo, No) in ((7391, 512, 256), (6507, 256, 512), (1500, 128, 256)):
                rows = [0, R // 7, 2 * R // 7 + 3, R // 2 + 1, R - 300, R]
                Gm, X = torch.randn(R, Mo, device=dev), torch.randn(R, No, device=dev)
                for ws in (torch.empty(32 << 20, dtype=torch.uint8, device=dev), None):
                    out = torch.full((5, Mo, No), 7.0, device=dev)
                    bo = torch.full((5, Mo), 7.0, device=dev)
                    grouped_tn(Gm, X, rows, out, bo, ws)
                    e = 0.0
                    for g in range(5):
                        a, b = rows[g], rows[g + 1]
                        ref = Gm[a:b].double().t() @ X[a:b].double()
                        e = max(e, ((out[g].double() - ref).abs().max() / ref.abs().max()).item())
                        rb = Gm[a:b].double().sum(0)
                        e = max(e, ((bo[g].double() - rb).abs().max() / rb.abs().max()).item())
                    worst = max(worst, e)
                    print(f"impl {impl} tile {tile:>2s}  grouped TN {R} rows -> {Mo}x{No}, workspace {ws is not None}: max rel err {e:.2e}")
    os.environ["GMP_GEMM_PIPE_TILE"] = "-1"
    assert worst < 2e-5, worst
    print("correctness ok, worst", worst)


def timeit(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3        # us


def bench():
    M = int(os.environ.get("M", 7392))
    cases = []
    for (mode, name, m, n, k) in ((ops.NT, "NT fwd  a[M,256] W1[512,256]", M, 512, 256), (ops.NT, "NT fwd  r1[M,512] W2[256,512]", M, 256, 512),
                                  (ops.NN, "NN dgrad g[M,256] W2[256,512]", M, 512, 256), (ops.NN, "NN dgrad g[M,512] W1[512,256]", M, 256, 512),
                                  (ops.NT, "NT lp   feat[7936,768] W[256,768]", 7936, 256, 768), (ops.NT, "NT lp   feat[37000,768] W[256,768]", 37000, 256, 768)):
        A = torch.randn(m, k, device=dev)
        B = torch.randn(n, k, device=dev) if mode == ops.NT else torch.randn(k, n, device=dev)
        out = torch.empty(m, n, device=dev)
        cases.append((name, 2.0 * m * n * k, (lambda mode=mode, A=A, B=B, out=out: ops.gemm(mode, A, B, out=out))))
    R = M
    rows = [0, R // 7, 2 * R // 7, (2 * R + R * 8 // 5) // 7, (2 * R + R * 16 // 5) // 7, R]
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    for (Mo, No) in ((256, 512), (512, 256)):
        Gm, X = torch.randn(R, Mo, device=dev), torch.randn(R, No, device=dev)
        out, bo = torch.empty(5, Mo, No, device=dev), torch.empty(5, Mo, device=dev)
        cases.append((f"TN wgrad 5 tasks g[M,{Mo}]^T x[M,{No}]", 2.0 * R * Mo * No, (lambda Gm=Gm, X=X, out=out, bo=bo: grouped_tn(Gm, X, rows, out, bo, ws))))
    variants = [("old", "old", "-1"), ("new auto", "new", "-1"), ("new 128x128", "new", "0"), ("new 64x128", "new", "1"), ("new 128x64", "new", "2"), ("new 64x64", "new", "3")]
    res = {}
    for rnd in range(3):
        for (vn, impl, tile) in variants:
            os.environ["GMP_GEMM_IMPL"], os.environ["GMP_GEMM_PIPE_TILE"] = impl, tile
            for (name, flops, f) in cases:
                res.setdefault((name, vn), []).append(timeit(f))
    for (name, flops, f) in cases:
        line = f"{name:40s}"
        for (vn, _, _) in variants:
            us = min(res[(name, vn)])
            line += f" | {vn} {us:6.1f} us {flops / us / 1e6:5.1f} TF"
        print(line)
    os.environ["GMP_GEMM_IMPL"], os.environ["GMP_GEMM_PIPE_TILE"] = "new", "-1"
    A, B = torch.randn(M, 256, device=dev), torch.randn(512, 256, device=dev)
    us = timeit(lambda: torch.mm(A, B.t()))
    print(f"{'torch.mm (rocBLAS) M x512x256':40s} {us:8.1f} us  {2 * M * 512 * 256 / us / 1e6:7.1f} TF/s")
    A, B = torch.randn(M, 512, device=dev), torch.randn(256, 512, device=dev)
    us = timeit(lambda: torch.mm(A, B.t()))
    print(f"{'torch.mm (rocBLAS) M x256x512':40s} {us:8.1f} us  {2 * M * 512 * 256 / us / 1e6:7.1f} TF/s")


if __name__ == "__main__":
    if "nocheck" not in sys.argv:
        check()
    bench()
