"""Linear + BatchNorm of short segments in one launch (csrc/gemm_pipe.h gemm_seg_bn_kernel behind gmp_linear_bn_fwd /
gmp_linear_bn_bwd_input): the Linear -> BatchNorm1d pairs of the reference's GINLayer (src/models/gnn.py:27-45) over the stacked
step's segments (one segment = the rows of one reference forward() call).  Checked against (a) the library's separate launches
(gmp_gemm_f32 + gmp_bn_fwd / gmp_bn_bwd): u bit-identical, the rest to rounding; (b) an fp64 torch restatement per segment
(nn.functional.batch_norm semantics, autograd for the backward).  Tolerances: fp32 sums over K <= 512 and over <= 384 rows."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gnn_pretraining_amd import ops          # noqa: E402

DEV = torch.device("cuda:0")


def _segments(gen, S, lo, hi, empty_at=None):
    n = torch.randint(lo, hi + 1, (S,), generator=gen).tolist()
    if empty_at is not None:
        n[empty_at] = 0
    ptr = [0]
    for v in n:
        ptr.append(ptr[-1] + v)
    return ptr, max(n)


def _rel(got, ref, floor=0.0):
    return ((got.double() - ref.double()).abs().max() / max(ref.double().abs().max().item(), floor, 1e-30)).item()


CASES = [  # S, rows lo..hi, K, N, residual, dropout
    (28, 180, 300, 256, 512, False, 0.0),      # the step's first pair: wide tiles (64 columns), 5 row blocks per wave pair
    (28, 180, 300, 512, 256, True, 0.2),       # second pair: narrow tiles (32 columns), residual + dropout
    (28, 1, 64, 256, 512, False, 0.0),         # one row block; single-row segments (variance 0)
    (7, 100, 384, 512, 256, True, 0.0),        # few segments: narrow tiles, 3 row blocks per wave
    (40, 65, 128, 64, 64, False, 0.0),         # smallest K the ring takes
    (12, 200, 320, 256, 512, False, 0.0),      # < 160 wide tiles -> narrow
]


@pytest.mark.parametrize("S,lo,hi,K,N,res,pdrop", CASES)
def test_linear_bn_forward(S, lo, hi, K, N, res, pdrop):
    gen = torch.Generator().manual_seed(S * 1000 + K)
    ptr, mx = _segments(gen, S, lo, hi, empty_at=3 if S > 4 else None)
    rows = ptr[-1]
    assert ops.linear_bn_supported(S, mx, K, N)
    x = torch.randn(rows, K, generator=gen).to(DEV)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=gen).to(DEV)
    r = torch.randn(rows, N, generator=gen).to(DEV) if res else None
    gam = (1 + 0.2 * torch.randn(N, generator=gen)).to(DEV)
    bet = (0.3 * torch.randn(N, generator=gen)).to(DEV)
    seg = torch.tensor(ptr, dtype=torch.int32, device=DEV)
    cfg = ops.make_bn_config(True, True, pdrop, seed=77, stream_id=5)
    y, u, sm, sr = ops.linear_bn_fwd(x, w, b, r, seg, mx, gam, bet, cfg)
    # (a) the separate launches
    z = ops.gemm(ops.NT, x, w, b)
    y2, sm2, sr2 = ops.bn_fwd(z, r, seg, mx, gam, bet, None, None, cfg)
    u2 = z + r if res else z
    live = torch.tensor([ptr[i + 1] > ptr[i] for i in range(S)], device=DEV)
    if rows >= 4096:            # the plain GEMM runs on the pipelined kernel too (no split-K at this size): same K order, u bit-identical
        assert torch.equal(u, u2)
        assert _rel(sm[live], sm2[live], 1e-3) < 2e-6 and _rel(sr[live], sr2[live]) < 2e-5
        assert _rel(y, y2) < 2e-5
    # (b) fp64: the product, then the normalisation of the u the kernel stored, per segment
    u64 = x.double() @ w.double().t() + b.double() + (r.double() if res else 0)
    assert _rel(u, u64) < 2e-5
    ud = u.double()
    for s in range(S):
        a, e = ptr[s], ptr[s + 1]
        if e == a:
            continue
        m = ud[a:e].mean(0)
        v = ud[a:e].var(0, unbiased=False)
        assert _rel(sm[s], m, 1e-2) < 1e-5 and _rel(sr[s], (v + 1e-5).rsqrt()) < 1e-5, s
        ref = torch.relu(gam.double() * (ud[a:e] - m) * (v + 1e-5).rsqrt() + bet.double())
        got = y[a:e].double()
        if pdrop > 0:
            keep = y2[a:e] != 0                      # gmp_bn_fwd's mask: same Philox key (seed, site, element quad)
            got, ref = got * keep, ref / (1 - pdrop) * keep
            assert (((y[a:e] != 0) != keep) & (ref.abs() > 1e-4)).sum().item() == 0
        assert (got - ref).abs().max().item() < 1e-5 * max(ref.abs().max().item(), 1.0), s


@pytest.mark.parametrize("S,lo,hi,K,N", [(28, 180, 300, 256, 512), (28, 1, 64, 256, 512), (9, 100, 384, 128, 256), (12, 200, 320, 256, 512)])
@pytest.mark.parametrize("relu", [True, False])
def test_linear_bn_backward_input(S, lo, hi, K, N, relu):
    gen = torch.Generator().manual_seed(S * 77 + N)
    ptr, mx = _segments(gen, S, lo, hi, empty_at=2)
    rows = ptr[-1]
    assert ops.linear_bn_supported(S, mx, K, N)
    g_out = torch.randn(rows, K, generator=gen).to(DEV)
    w = (torch.randn(K, N, generator=gen) / K ** 0.5).to(DEV)           # the upper Linear's weight [out = K, in = N]
    x = torch.randn(rows, N, generator=gen).to(DEV)
    gam = (1 + 0.2 * torch.randn(N, generator=gen)).to(DEV)
    bet = (0.3 * torch.randn(N, generator=gen)).to(DEV)
    seg = torch.tensor(ptr, dtype=torch.int32, device=DEV)
    cfg = ops.make_bn_config(True, relu)
    _, sm, sr = ops.bn_fwd(x, None, seg, mx, gam, bet, None, None, cfg)
    g_x, sums = ops.linear_bn_bwd_input(g_out, w, x, seg, mx, gam, bet, sm, sr, cfg)
    # (a) separate launches
    g_y = ops.gemm(ops.NN, g_out, w)
    g_x2, gg, gb = ops.bn_bwd(g_y, x, None, seg, mx, gam, bet, None, None, sm, sr, cfg)
    scale = g_x2.abs().max().item()
    assert (g_x - g_x2).abs().max().item() < 3e-5 * scale
    assert _rel(sums[:, 0].sum(0), gb[0], 1e-2) < 2e-5 and _rel(sums[:, 1].sum(0), gg[0], 1e-2) < 2e-5
    # (b) fp64 autograd per segment
    for s in range(0, S, 3):
        a, e = ptr[s], ptr[s + 1]
        if e == a:
            assert sums[s].abs().max().item() == 0
            continue
        if e - a == 1:                      # one row: xhat = 0 and the gradient cancels (torch refuses the case)
            assert g_x[a:e].abs().max().item() < 1e-4 * scale
            continue
        xs = x[a:e].double().requires_grad_(True)
        yy = torch.nn.functional.batch_norm(xs, None, None, gam.double(), bet.double(), True, 0.1, 1e-5)
        if relu:
            yy = torch.relu(yy)
        (yy * (g_out[a:e].double() @ w.double())).sum().backward()
        assert (g_x[a:e].double() - xs.grad).abs().max().item() < 1e-4 * max(xs.grad.abs().max().item(), 1e-3 * scale), s


def test_linear_bn_rejects_what_it_does_not_cover():
    assert not ops.linear_bn_supported(28, 400, 256, 512) and not ops.linear_bn_supported(28, 321, 256, 512)
    assert ops.linear_bn_supported(28, 384, 512, 256) and not ops.linear_bn_supported(28, 200, 250, 512)
    x, w = torch.zeros(10, 256, device=DEV), torch.zeros(512, 256, device=DEV)
    seg = torch.tensor([0, 10], dtype=torch.int32, device=DEV)
    g = torch.ones(512, device=DEV)
    with pytest.raises(Exception):
        ops.linear_bn_fwd(x, w, None, None, seg, 10, g, g, ops.make_bn_config(False, True))          # eval mode: running statistics
