"""The pipelined fp32 MFMA GEMM (csrc/gemm_pipe.h: LDS-DMA ring, counted waits, float4 epilogue) behind gmp_gemm_f32 /
gmp_gemm_f32_grouped -- the nn.Linear of the reference's GIN MLP and heads (src/models/gnn.py:29-37, heads.py:35-67) at the
stacked step's row counts.  Checked for every tile shape against an fp64 product (tolerance: fp32 accumulation over K,
2e-5 of the largest output), against exact integer data (bit-exact: catches a swapped fragment / output map that a
tolerance would hide), with ragged M / N, the accumulate / alpha / ReLU epilogue, and the grouped weight-gradient form
over uneven row ranges with partial last K-steps, row slices and the bias-gradient rider."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from gnn_pretraining_amd import _lib as L, ops          # noqa: E402

DEV = torch.device("cuda:0")
TILES = ["-1", "0", "1", "2", "3"]


def _rel(got, ref):
    return ((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def _grouped_tn(G, X, rows, out, bias_out, ws):
    lib, ng, Mo, No = L.lib(), len(rows) - 1, G.size(1), X.size(1)
    L.check(lib.gmp_gemm_f32_grouped(ops.TN, G.data_ptr(), X.data_ptr(), None, out.data_ptr(), ng, (C.c_int32 * len(rows))(*rows), None, None,
                                     (C.c_int64 * ng)(*[g * Mo * No for g in range(ng)]), bias_out.data_ptr(),
                                     (C.c_int64 * ng)(*[g * Mo for g in range(ng)]), Mo, No, 0, Mo, No, No, 1.0, 0, 0,
                                     ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                     torch.cuda.current_stream().cuda_stream), "grouped TN")


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K", [(7392, 512, 256), (6507, 256, 512), (2049, 130, 768), (1025, 64, 64), (2708, 256, 1440), (1500, 128, 2048)])
def test_pipelined_nt_nn_against_fp64(monkeypatch, tile, M, N, K):
    monkeypatch.setenv("GMP_GEMM_PIPE_TILE", tile)
    g = torch.Generator().manual_seed(M + N + K)
    A, W, bias = torch.randn(M, K, generator=g).to(DEV), torch.randn(N, K, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    ref = A.double() @ W.double().t() + bias.double()
    assert _rel(ops.gemm(ops.NT, A, W, bias), ref) < 2e-5
    assert _rel(ops.gemm(ops.NT, A, W, bias, relu=True), ref.clamp_min(0)) < 2e-5
    if N % 4 == 0:                                   # input-gradient form: B is the k-major weight matrix
        G = torch.randn(M, N, generator=g).to(DEV)
        refn = G.double() @ W.double()
        assert _rel(ops.gemm(ops.NN, G, W), refn) < 2e-5
        acc = torch.randn(M, K, generator=g).to(DEV)
        got = ops.gemm(ops.NN, G, W, out=acc.clone(), alpha=0.5, accumulate=True)
        assert _rel(got, acc.double() + 0.5 * refn) < 2e-5


@pytest.mark.parametrize("tile", TILES)
def test_pipelined_gemm_is_exact_on_integer_data(monkeypatch, tile):
    """small-integer operands: every product and partial sum is exact in fp32, so the result must equal the fp64 product bit for
    bit whatever the k order -- and an asymmetric B exposes a transposed or permuted fragment / output map."""
    monkeypatch.setenv("GMP_GEMM_PIPE_TILE", tile)
    g = torch.Generator().manual_seed(5)
    M, N, K = 1283, 196, 96
    A = torch.randint(-4, 5, (M, K), generator=g).float().to(DEV)
    W = torch.randint(-4, 5, (N, K), generator=g).float().to(DEV)
    assert torch.equal(ops.gemm(ops.NT, A, W).double(), A.double() @ W.double().t())
    G = torch.randint(-4, 5, (M, N), generator=g).float().to(DEV)
    assert torch.equal(ops.gemm(ops.NN, G, W).double(), G.double() @ W.double())
    # few output tiles and a long K (the fine-tune encoder's shape class): ops.gemm hands over a workspace and the pipelined kernel runs as
    # K-slices summed in slice order -- exact on integers whatever the slicing
    A2 = torch.randint(-3, 4, (1100, 1440), generator=g).float().to(DEV)
    W2 = torch.randint(-3, 4, (128, 1440), generator=g).float().to(DEV)
    b2 = torch.randint(-3, 4, (128,), generator=g).float().to(DEV)
    assert L.lib().gmp_gemm_f32_workspace_bytes(ops.NT, 1100, 128, 1440) > 0
    assert torch.equal(ops.gemm(ops.NT, A2, W2, b2, relu=True).double(), (A2.double() @ W2.double().t() + b2.double()).clamp_min(0))
    rows = [0, 130, 131, 700, M]                     # a one-row group, ragged tails everywhere
    out, bo = torch.zeros(4, N, K, device=DEV), torch.zeros(4, N, device=DEV)
    _grouped_tn(G, A, rows, out, bo, torch.empty(16 << 20, dtype=torch.uint8, device=DEV))
    for i in range(4):
        a, b = rows[i], rows[i + 1]
        assert torch.equal(out[i].double(), G[a:b].double().t() @ A[a:b].double()), f"group {i}"
        assert torch.equal(bo[i].double(), G[a:b].double().sum(0)), f"group {i} bias gradient"


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("R,Mo,No", [(7391, 512, 256), (6507, 256, 512), (1500, 128, 256)])
@pytest.mark.parametrize("workspace", [True, False])
def test_pipelined_grouped_weight_gradient(monkeypatch, tile, R, Mo, No, workspace):
    """dW_t = g_t^T x_t and db_t = colsum(g_t) per task row range (the stacked backward, engine._backbone_backward): uneven ranges
    whose lengths are not multiples of the 32-deep K-step; with a workspace the ranges are cut into row slices and reduced in
    slice order, without one the first kernel runs (callers without a workspace never reach the pipelined path)."""
    monkeypatch.setenv("GMP_GEMM_PIPE_TILE", tile)
    g = torch.Generator().manual_seed(R)
    rows = [0, R // 7, 2 * R // 7 + 3, R // 2 + 1, R - 300, R]
    G, X = torch.randn(R, Mo, generator=g).to(DEV), torch.randn(R, No, generator=g).to(DEV)
    out, bo = torch.full((5, Mo, No), 7.0, device=DEV), torch.full((5, Mo), 7.0, device=DEV)
    _grouped_tn(G, X, rows, out, bo, torch.empty(32 << 20, dtype=torch.uint8, device=DEV) if workspace else None)
    for i in range(5):
        a, b = rows[i], rows[i + 1]
        assert _rel(out[i], G[a:b].double().t() @ X[a:b].double()) < 2e-5, f"group {i}"
        assert _rel(bo[i], G[a:b].double().sum(0)) < 2e-5, f"group {i} bias gradient"


def test_old_and_pipelined_kernels_agree_to_rounding(monkeypatch):
    """GMP_GEMM_IMPL=old keeps every problem on the first kernel (A/B aid): same products, another k order."""
    g = torch.Generator().manual_seed(1)
    A, W = torch.randn(4096, 512, generator=g).to(DEV), torch.randn(256, 512, generator=g).to(DEV)
    new = ops.gemm(ops.NT, A, W)
    monkeypatch.setenv("GMP_GEMM_IMPL", "old")
    old = ops.gemm(ops.NT, A, W)
    assert _rel(new, old.double()) < 2e-5
