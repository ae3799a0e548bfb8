"""Operator-level parity: each C-ABI kernel vs the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): indices bit-exact; fp32 within 1e-4 relative.
Relative error is measured against the tensor's max magnitude (a per-element relative
test is meaningless next to exact zeros produced by ReLU).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gnn_pretraining_amd import ops, synthetic as S          # noqa: E402
from oracle import graph_ops as OG                            # noqa: E402

DEV = "cuda:0"
RTOL = 1e-4


def close(got, want, rtol=RTOL, what="", scale=None):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    scale = max(want.abs().max().item(), 1e-30) if scale is None else scale
    err = (got - want).abs().max().item() / scale
    assert err <= rtol, f"{what}: max rel err {err:.3e} > {rtol}"


def rand_edges(gen, n, e):
    return torch.randint(0, n, (2, e), generator=gen)


# ------------------------------------------------------------------ csr_build
@pytest.mark.parametrize("n,e", [(1, 0), (5, 0), (7, 3), (264, 992), (300, 5000), (16384, 60000),
                                 (16385, 70000), (50000, 400000), (2708, 10858)])
def test_csr_build_bit_exact(n, e):
    gen = torch.Generator().manual_seed(n * 7 + e)
    ei = rand_edges(gen, n, e) if e else torch.empty(2, 0, dtype=torch.long)
    csr = ops.csr_build(ei.to(DEV), n)
    torch.cuda.synchronize()
    assert int(csr.status.item()) == 0
    for by, (rp, col, perm) in (("dst", (csr.rowptr, csr.col, csr.perm)), ("src", (csr.rowptr_t, csr.col_t, csr.perm_t))):
        orp, ocol, operm = OG.coo_to_csr(ei, n, by)
        assert torch.equal(rp.cpu(), orp), f"rowptr {by}"
        assert torch.equal(perm.cpu(), operm), f"perm {by}"
        assert torch.equal(col.cpu(), ocol), f"col {by}"


def test_csr_build_high_degree_rows():
    # star graph: one hub with degree 5000 (exercises the wave-cooperative row ordering)
    n = 6000
    hub = torch.zeros(5000, dtype=torch.long)
    leaves = torch.arange(1, 5001)
    ei = torch.cat([torch.stack([hub, leaves]), torch.stack([leaves, hub])], dim=1)
    ei = ei[:, torch.randperm(ei.size(1), generator=torch.Generator().manual_seed(3))]
    csr = ops.csr_build(ei.to(DEV), n)
    for by, (rp, col, perm) in (("dst", (csr.rowptr, csr.col, csr.perm)), ("src", (csr.rowptr_t, csr.col_t, csr.perm_t))):
        orp, ocol, operm = OG.coo_to_csr(ei, n, by)
        assert torch.equal(rp.cpu(), orp) and torch.equal(perm.cpu(), operm) and torch.equal(col.cpu(), ocol)


def test_csr_build_reports_bad_endpoints():
    ei = torch.tensor([[0, 1, 9, 2], [1, -1, 0, 0]])
    csr = ops.csr_build(ei.to(DEV), 3)
    assert int(csr.status.item()) == 2
    assert csr.rowptr.cpu().tolist() == [0, 1, 2, 2]        # only edges 0->1 and 2->0 survive


# ------------------------------------------------------------ gin aggregation
@pytest.mark.parametrize("graphs,F", [(1, 256), (8, 256), (32, 256), (8, 128), (8, 512), (8, 64), (3, 1024)])
def test_gin_aggregate_fwd_bwd(graphs, F):
    gen = torch.Generator().manual_seed(graphs * 1000 + F)
    b = S.domain_batch(gen, 21, graphs)
    n = b.num_nodes
    x = torch.randn(n, F, generator=gen)
    eps = torch.tensor([0.37])
    g = torch.randn(n, F, generator=gen)
    xr, er = x.clone().requires_grad_(), eps.clone().requires_grad_()
    want = OG.gin_aggregate(xr, b.edge_index, er)
    want.backward(g)
    csr = ops.csr_build(b.edge_index.to(DEV), n)
    xd, ed, gd = x.to(DEV), eps.to(DEV), g.to(DEV)
    out = ops.gin_aggregate_fwd(xd, csr.rowptr, csr.col, ed)
    close(out, want, what="aggregate fwd")
    gx, ge = ops.gin_aggregate_bwd(gd, csr.rowptr_t, csr.col_t, ed, xd)
    close(gx, xr.grad, what="aggregate g_x")
    # a cancelling sum of N*F products: error is relative to the magnitude summed, not to the tiny result
    close(ge, er.grad, what="aggregate g_eps", scale=(g * x).abs().sum().item() / (n * F) ** 0.5)


@pytest.mark.parametrize("with_eps_grad", [True, False])
def test_gin_aggregate_streaming_backward(with_eps_grad):
    """N >= 65,536 rows: the backward takes the LDS-resident-tile kernel on the transposed CSR (the roofline rung's `roofline_bwd`),
    with the row products <g, x> of the eps gradient riding along.  Against autograd through the oracle's aggregation: ENZYMES-
    shaped graphs (~69 k rows), and a random graph whose neighbours mostly lie outside the LDS tile, with isolated rows."""
    gen = torch.Generator().manual_seed(78)
    b = S.domain_batch(gen, 4, 2100)
    for ei, n in ((b.edge_index, b.num_nodes), (None, 516 * 128)):
        if ei is None:
            ei = torch.randint(0, n, (2, 300_000), generator=gen)
            ei = ei[:, (ei[0] % 5) != 0]                        # every 5th row sends nothing: empty rows of the transposed CSR
        assert n >= 65536
        x = torch.randn(n, 256, generator=gen).requires_grad_(True)
        eps = torch.tensor([0.3], requires_grad=True)
        g = torch.randn(n, 256, generator=gen)
        OG.gin_aggregate(x, ei, eps).backward(g)
        csr = ops.csr_build(ei.to(DEV), n)
        gx, geps = ops.gin_aggregate_bwd(g.to(DEV), csr.rowptr_t, csr.col_t, eps.detach().to(DEV), x.detach().to(DEV) if with_eps_grad else None)
        close(gx, x.grad, what="streaming aggregate backward: g_x")
        if with_eps_grad:
            close(geps, eps.grad, rtol=2e-4, what="streaming aggregate backward: g_eps (one sum of n * 256 products)")


def test_gin_aggregate_streaming_kernel():
    """N >= 65,536 rows takes the LDS-resident-tile kernel (the roofline rung's path); a dense random graph
    forces its un-staged fallback (more neighbour ids per 128-row tile than the LDS stage holds)."""
    gen = torch.Generator().manual_seed(77)
    b = S.domain_batch(gen, 4, 2100)                        # ~69k rows of ENZYMES-shaped graphs
    n = b.num_nodes
    assert n >= 65536
    x = torch.randn(n, 256, generator=gen)
    eps = torch.tensor([0.2])
    csr = ops.csr_build(b.edge_index.to(DEV), n)
    out = ops.gin_aggregate_fwd(x.to(DEV), csr.rowptr, csr.col, eps.to(DEV))
    close(out, OG.gin_aggregate(x, b.edge_index, eps), what="streaming aggregate")
    n2 = 66000
    ei = torch.randint(0, n2, (2, 1_400_000), generator=gen)
    x2 = torch.randn(n2, 256, generator=gen)
    csr = ops.csr_build(ei.to(DEV), n2)
    out = ops.gin_aggregate_fwd(x2.to(DEV), csr.rowptr, csr.col, eps.to(DEV))
    close(out, OG.gin_aggregate(x2, ei, eps), what="streaming aggregate, un-staged tiles")
    # staged tiles whose neighbours mostly lie OUTSIDE the LDS tile (global fallback per neighbour), a row count that is an
    # exact multiple of the 128-row tile, isolated rows, eps = 0
    n3 = 516 * 128
    ei = torch.randint(0, n3, (2, 300_000), generator=gen)
    ei = ei[:, (ei[1] % 7) != 0]                            # every 7th row has no incoming edge
    x3 = torch.randn(n3, 256, generator=gen)
    csr = ops.csr_build(ei.to(DEV), n3)
    out = ops.gin_aggregate_fwd(x3.to(DEV), csr.rowptr, csr.col, torch.zeros(1, device=DEV))
    close(out, OG.gin_aggregate(x3, ei, torch.zeros(1)), what="streaming aggregate, cross-tile neighbours")
    assert torch.equal(out[::7].cpu(), x3[::7])


def test_gin_aggregate_isolated_and_empty():
    x = torch.randn(5, 256)
    ei = torch.tensor([[0, 1], [1, 0]])
    csr = ops.csr_build(ei.to(DEV), 5)
    out = ops.gin_aggregate_fwd(x.to(DEV), csr.rowptr, csr.col, torch.zeros(1, device=DEV))
    close(out, OG.gin_aggregate(x, ei, torch.zeros(1)))
    # no edges at all
    e0 = torch.empty(2, 0, dtype=torch.long)
    csr = ops.csr_build(e0.to(DEV), 5)
    out = ops.gin_aggregate_fwd(x.to(DEV), csr.rowptr, csr.col, torch.ones(1, device=DEV))
    close(out, 2 * x)


def test_gin_aggregate_cora_shape_and_linearity():
    gen = torch.Generator().manual_seed(5)
    c = S.cora_like(gen, dim=16)
    n = c.num_nodes
    x, y = torch.randn(n, 256, generator=gen), torch.randn(n, 256, generator=gen)
    csr = ops.csr_build(c.edge_index.to(DEV), n)
    eps = torch.tensor([0.1], device=DEV)
    f = lambda t: ops.gin_aggregate_fwd(t.to(DEV), csr.rowptr, csr.col, eps)
    close(f(x), OG.gin_aggregate(x, c.edge_index, torch.tensor([0.1])), what="cora aggregate")
    close(f(2 * x + y), 2 * f(x) + f(y), what="linearity")       # size-independent property


# ------------------------------------------------------------------- pooling
@pytest.mark.parametrize("graphs,F", [(1, 256), (8, 256), (16, 512)])
def test_mean_and_max_pool(graphs, F):
    gen = torch.Generator().manual_seed(graphs + F)
    b = S.domain_batch(gen, 21, graphs)
    x = torch.relu(torch.randn(b.num_nodes, F, generator=gen))     # post-ReLU: many tied zeros
    x[:, :7] = 0.0                                                 # columns where EVERY node ties at 0
    g = torch.randn(graphs, F, generator=gen)
    ptr = torch.tensor(b.ptr_host, dtype=torch.int32, device=DEV)
    xd, gd = x.to(DEV), g.to(DEV)
    # mean
    xr = x.clone().requires_grad_()
    want = OG.global_mean_pool(xr, b.batch); want.backward(g)
    close(ops.segment_sum(xd, ptr, None, mean=True), want, what="mean pool")
    close(ops.row_gather(gd, b.batch.to(DEV), ptr), xr.grad, what="mean pool bwd")
    # max (+ tie rule)
    xr = x.clone().requires_grad_()
    want = OG.global_max_pool(xr, b.batch); want.backward(g)
    out = ops.segment_max_fwd(xd, ptr)
    close(out, want, what="max pool")
    close(ops.segment_max_bwd(gd, xd, out, ptr), xr.grad, what="max pool bwd (even split between ties)")


def test_row_gather_and_scatter_add_backward():
    gen = torch.Generator().manual_seed(11)
    n, F, m = 300, 256, 700
    x = torch.randn(n, F, generator=gen)
    idx = torch.randint(0, n, (m,), generator=gen)                 # repeats: backward must accumulate
    g = torch.randn(m, F, generator=gen)
    xr = x.clone().requires_grad_()
    xr[idx].backward(g)
    close(ops.row_gather(x.to(DEV), idx.to(DEV)), x[idx], what="gather")
    # scatter-add backward = segmented sum over a CSR of (row -> gathered positions)
    ei = torch.stack([torch.arange(m), idx])                       # "edge" k -> idx[k]
    csr = ops.csr_build(ei.to(DEV), max(n, m))
    gx = ops.segment_sum(g.to(DEV), csr.rowptr[: n + 1].contiguous(), csr.col)
    close(gx, xr.grad, what="gather bwd")


# ----------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(264, 512, 256), (264, 256, 512), (1, 1, 1), (33, 7, 21), (300, 256, 37),
                                   (1000, 256, 768), (2708, 256, 1433), (6700, 512, 256), (40000, 512, 256),
                                   (65, 130, 67)])
def test_gemm_modes(M, N, K):
    gen = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen)
    bias = torch.randn(N, generator=gen)
    G = torch.randn(M, N, generator=gen)
    Ad, Wd, Gd = A.to(DEV), W.to(DEV), G.to(DEV)
    ref = lambda a, b: (a.double() @ b.double()).float()
    close(ops.gemm(ops.NT, Ad, Wd, bias.to(DEV)), ref(A, W.t()) + bias, what="NT + bias")
    close(ops.gemm(ops.NT, Ad, Wd, bias.to(DEV), relu=True), torch.relu(ref(A, W.t()) + bias), what="NT + bias + relu")
    close(ops.gemm(ops.NN, Gd, Wd), ref(G, W), what="NN (input grad)")
    close(ops.gemm(ops.TN, Gd, Ad), ref(G.t(), A), what="TN (weight grad, split-K)")
    acc = torch.randn(N, K, generator=gen)
    close(ops.gemm(ops.TN, Gd, Ad, out=acc.to(DEV).clone(), alpha=0.5, accumulate=True), acc + 0.5 * ref(G.t(), A),
          what="TN accumulate")
    close(ops.colsum(Gd), G.double().sum(0).float(), what="colsum")


def test_gemm_rejects_bad_shapes():
    from gnn_pretraining_amd._lib import GnnmpError
    a = torch.zeros(4, 8, device=DEV)
    with pytest.raises(GnnmpError):
        ops.gemm(ops.NT, a, torch.zeros(5, 9, device=DEV))
    with pytest.raises(GnnmpError):
        ops.gemm(ops.NT, a.cpu(), a.cpu())


# ------------------------------------------------------------------ BatchNorm
def _bn_oracle(x, res, seg, gamma, beta, rm, rv, training, relu):
    """Per-segment torch batch_norm, segments in order (== S separate module calls)."""
    import torch.nn.functional as F
    outs = []
    for a, b in zip(seg[:-1], seg[1:]):
        u = x[a:b] + (res[a:b] if res is not None else 0)
        y = F.batch_norm(u, rm, rv, gamma, beta, training=training, momentum=0.1, eps=1e-5)
        outs.append(torch.relu(y) if relu else y)
    return torch.cat(outs)


@pytest.mark.parametrize("seg,C,relu,with_res", [([0, 264], 256, True, False), ([0, 40, 300, 301 + 30, 600], 512, True, True),
                                                 ([0, 2708], 256, True, True), ([0, 1500, 1500 + 700], 64, False, False),
                                                 ([0, 2, 5], 128, True, False), ([0, 3500, 3500 + 1100], 128, True, True)])   # short / medium / chunked regimes
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("slabs", [False, True])     # True: rendezvous words given, segments of 1,025-4,096 rows run as 128-row slabs
def test_bn_fwd_bwd(seg, C, relu, with_res, training, slabs):
    gen = torch.Generator().manual_seed(sum(seg) + C)
    rows = seg[-1]
    x = torch.randn(rows, C, generator=gen) * 2 + 0.5
    res = torch.randn(rows, C, generator=gen) if with_res else None
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    rm0, rv0 = torch.randn(C, generator=gen) * 0.1, torch.rand(C, generator=gen) + 0.5
    gy = torch.randn(rows, C, generator=gen)
    # oracle
    xr = x.clone().requires_grad_()
    rr = res.clone().requires_grad_() if with_res else None
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rm, rv = rm0.clone(), rv0.clone()
    want = _bn_oracle(xr, rr, seg, gr, br, rm, rv, training, relu)
    want.backward(gy)
    # HIP
    d = lambda t: None if t is None else t.to(DEV)
    segd = torch.tensor(seg, dtype=torch.int32, device=DEV)
    mx = max(b - a for a, b in zip(seg[:-1], seg[1:]))
    rmd, rvd = d(rm0.clone()), d(rv0.clone())
    sync = torch.zeros(ops.bn_sync_words(C, len(seg) - 1), dtype=torch.int32, device=DEV) if slabs else None
    cfg = ops.make_bn_config(training, relu, sync=sync)
    y, sm, sr = ops.bn_fwd(d(x), d(res), segd, mx, d(gamma), d(beta), rmd, rvd, cfg)
    close(y, want, what="bn fwd")
    close(rmd, rm, what="running_mean"); close(rvd, rv, what="running_var")
    gu, gg, gb = ops.bn_bwd(d(gy), d(x), d(res), segd, mx, d(gamma), d(beta), rmd, rvd, sm, sr, cfg)
    close(gu, xr.grad, rtol=2e-4, what="bn g_x")
    if with_res:
        close(gu, rr.grad, rtol=2e-4, what="bn g_residual")
    close(gg[0], gr.grad, rtol=2e-4, what="bn g_gamma"); close(gb[0], br.grad, rtol=2e-4, what="bn g_beta")
    if slabs:
        assert sync[0].item() == 0 and sync[2].item() == 0, "sync[0] != 0: a wait timed out; sync[2] != 0: a workgroup never left"


def test_bn_slab_form_draws_the_masks_of_the_strip_form_and_repeats_bitwise():
    """The Cora-sized segment as 128-row slabs (gmp_bn_config.sync) against one workgroup per column strip: same dropout mask (keyed by
    element), statistics equal to rounding, and the slab form's own results bit-identical from call to call (fixed combination order)."""
    gen = torch.Generator().manual_seed(77)
    rows, C = 2708, 512
    x, gy = (torch.randn(rows, C, generator=gen) * 3 + 1).to(DEV), torch.randn(rows, C, generator=gen).to(DEV)
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(DEV), torch.randn(C, generator=gen).to(DEV)
    segd = torch.tensor([0, rows], dtype=torch.int32, device=DEV)
    sync = torch.zeros(ops.bn_sync_words(C, 1), dtype=torch.int32, device=DEV)
    outs = []
    for sy in (None, sync, sync):
        cfg = ops.make_bn_config(True, True, dropout_p=0.2, seed=99, stream_id=5, sync=sy)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        y, sm, sr = ops.bn_fwd(x, None, segd, rows, gamma, beta, rm, rv, cfg)
        gu, gg, gb = ops.bn_bwd(gy, x, None, segd, rows, gamma, beta, rm, rv, sm, sr, cfg)
        outs.append((y, sm, sr, rm, rv, gu, gg, gb))
    strip, slab, again = outs
    for a, b in zip(slab, again):
        assert torch.equal(a, b)
    assert ((strip[0] != 0) != (slab[0] != 0)).sum().item() <= 4          # same dropout mask; a ReLU edge may flip at rounding level
    for a, b, what in zip(strip, slab, ("y", "mean", "rstd", "running_mean", "running_var", "g_u", "g_gamma", "g_beta")):
        close(a, b, rtol=1e-5, what="slab vs strip " + what)
    assert sync[0].item() == 0 and sync[1].item() == 4 and sync[2].item() == 0     # no time-out; four launches = four generations; all left


def test_bn_param_grad_groups_and_dropout_consistency():
    gen = torch.Generator().manual_seed(9)
    seg = [0, 100, 220, 300, 450]
    C = 256
    x = torch.randn(seg[-1], C, generator=gen)
    gy = torch.randn(seg[-1], C, generator=gen)
    gamma, beta = torch.ones(C), torch.zeros(C)
    d = lambda t: t.to(DEV)
    segd = torch.tensor(seg, dtype=torch.int32, device=DEV)
    cfg = ops.make_bn_config(True, True, dropout_p=0.2, seed=1234, stream_id=3)
    y, sm, sr = ops.bn_fwd(d(x), None, segd, 150, d(gamma), d(beta), None, None, cfg)
    y2, _, _ = ops.bn_fwd(d(x), None, segd, 150, d(gamma), d(beta), None, None, cfg)
    assert torch.equal(y, y2)                                      # counter-based mask is reproducible
    cfg0 = ops.make_bn_config(True, True)
    y0, _, _ = ops.bn_fwd(d(x), None, segd, 150, d(gamma), d(beta), None, None, cfg0)
    kept = (y != 0)
    close(y[kept], (y0 / 0.8)[kept], what="kept elements are scaled by 1/(1-p)")
    frac = 1.0 - (y != 0).sum().item() / max((y0 != 0).sum().item(), 1)
    assert 0.17 < frac < 0.23, f"drop fraction {frac}"
    # backward regenerates the same mask: gradient is zero exactly where the output was dropped
    gu, gg2, gb2 = ops.bn_bwd(d(gy), d(x), None, segd, 150, d(gamma), d(beta), None, None, sm, sr, cfg, [0, 1, 4])
    gu1, gg1, gb1 = ops.bn_bwd(d(gy), d(x), None, segd, 150, d(gamma), d(beta), None, None, sm, sr, cfg)
    close(gg2.sum(0), gg1[0], what="group sums add up (gamma)")
    close(gb2.sum(0), gb1[0], what="group sums add up (beta)")
    assert torch.equal(gu, gu1)


# -------------------------------------------------------- LP edge features
def test_lp_edge_features():
    gen = torch.Generator().manual_seed(21)
    n, K, F = 264, 2000, 256
    h = torch.relu(torch.randn(n, F, generator=gen))               # zeros -> |hs-hd| hits its kink
    edges = torch.randint(0, n, (2, K), generator=gen)
    edges[:, :5] = torch.tensor([[3, 3, 3, 3, 3], [3, 3, 3, 3, 3]])  # self pairs: hs == hd exactly
    g = torch.randn(K, 3 * F, generator=gen)
    hr = h.clone().requires_grad_()
    hs, hd = hr[edges[0]], hr[edges[1]]
    want = torch.cat([hs + hd, hs * hd, (hs - hd).abs()], dim=1)
    want.backward(g)
    feat = ops.lp_edge_features_fwd(h.to(DEV), edges.to(DEV))
    close(feat, want, what="lp features")
    ghs, ghd = ops.lp_edge_features_bwd(g.to(DEV), h.to(DEV), edges.to(DEV))
    # reduce onto nodes through CSRs of the decoder edges
    csr = ops.csr_build(edges.to(DEV), n)
    gh = ops.segment_sum(ghs, csr.rowptr_t, csr.perm_t)            # by source
    gh = ops.segment_sum(ghd, csr.rowptr, csr.perm, out=gh, accumulate=True)   # by target
    close(gh, hr.grad, what="lp features bwd")


# ------------------------------------------------------------------ NT-Xent
@pytest.mark.parametrize("n,d,T", [(8, 128, 0.5), (170, 128, 0.5), (333, 128, 0.2), (1, 128, 0.5), (2, 4, 1.0)])
def test_nt_xent(n, d, T):
    from oracle.tasks import nt_xent
    gen = torch.Generator().manual_seed(n + d)
    z1, z2 = torch.randn(n, d, generator=gen), torch.randn(n, d, generator=gen)
    a, b = z1.clone().requires_grad_(), z2.clone().requires_grad_()
    if n == 1:
        pytest.skip("2x2 similarity: both rows have a single admissible column, loss is identically 0")
    want, size = nt_xent(a, b, T)
    (want * 0.25).backward()
    loss, ws = ops.nt_xent_fwd(z1.to(DEV), z2.to(DEV), T)
    close(loss, want.reshape(1), what="nt_xent loss")
    g1, g2 = ops.nt_xent_bwd(z1.to(DEV), z2.to(DEV), T, torch.tensor([0.25], device=DEV), ws)
    close(g1, a.grad, rtol=2e-4, what="nt_xent g_z1"); close(g2, b.grad, rtol=2e-4, what="nt_xent g_z2")


@pytest.mark.parametrize("ns", [[8, 8, 8, 8], [201, 0, 37, 150], [1, 5], [0, 0, 0]])
def test_nt_xent_grouped_equals_the_single_problem_calls(ns):
    """the per-domain problems of a contrastive task in one call: per group the loss sum and the gradient are those of
    gmp_nt_xent_fwd/_bwd on that group alone (zero padding adds nothing): gradients bit-exact, loss sums to summation order"""
    gen = torch.Generator().manual_seed(sum(ns) + len(ns))
    d, T = 128, 0.37
    offs, rows = [], 3                                   # leading rows that belong to no group
    for n in ns:
        offs.append(rows)
        rows += 2 * n + 1                                # and a stray row between groups
    z = torch.randn(rows, d, generator=gen).to(DEV)
    gs = torch.tensor([0.125], device=DEV)
    gz, sums, total = ops.nt_xent_grouped(z, ns, offs, T, gs)
    want_total = 0.0
    touched = torch.zeros(rows, dtype=torch.bool)
    for g, n in enumerate(ns):
        if n == 0:
            assert float(sums[g]) == 0.0
            continue
        z1, z2 = z[offs[g]:offs[g] + n].contiguous(), z[offs[g] + n:offs[g] + 2 * n].contiguous()
        loss, ws = ops.nt_xent_fwd(z1, z2, T)
        g1, g2 = ops.nt_xent_bwd(z1, z2, T, gs, ws)
        assert abs(float(sums[g]) - float(loss)) <= 1e-6 * abs(float(loss)), f"group {g} loss"      # same rows, another summation order
        assert torch.equal(gz[offs[g]:offs[g] + n], g1) and torch.equal(gz[offs[g] + n:offs[g] + 2 * n], g2), f"group {g} gradient"
        want_total += float(loss)
        touched[offs[g]:offs[g] + 2 * n] = True
    assert abs(float(total) - want_total) <= 1e-5 * max(abs(want_total), 1.0)
    assert float(gz[~touched.to(DEV)].abs().max() if (~touched).any() else 0.0) == 0.0


def test_fused_sigmoid_bce_equals_the_four_separate_kernels():
    import ctypes as C
    from gnn_pretraining_amd import _lib as L
    gen = torch.Generator().manual_seed(5)
    n = 7937
    x = (torch.randn(n, generator=gen) * 6).to(DEV)                  # saturating scores too
    x[:3] = torch.tensor([-120.0, 120.0, 0.0])
    y = (torch.rand(n, generator=gen) < 0.5).float().to(DEV)
    gs = torch.tensor([1.0 / n], device=DEV)
    l = L.lib()
    st = ops._stream(x)
    ws = torch.empty(l.gmp_loss_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    loss, p, gx = torch.empty(1, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    L.check(l.gmp_sigmoid_bce_sum_fwd_bwd(ops._ptr(x), ops._ptr(y), n, ops._ptr(gs), ops._ptr(loss), ops._ptr(p), ops._ptr(gx), ops._ptr(ws), ws.numel(), st), "fused")
    p2, gp, gx2, loss2 = torch.empty(n, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV), torch.empty(1, device=DEV)
    L.check(l.gmp_sigmoid_fwd(ops._ptr(x), ops._ptr(p2), n, st), "s")
    L.check(l.gmp_bce_sum_fwd(ops._ptr(p2), ops._ptr(y), n, ops._ptr(loss2), ops._ptr(ws), ws.numel(), st), "b")
    L.check(l.gmp_bce_sum_bwd(ops._ptr(p2), ops._ptr(y), ops._ptr(gs), ops._ptr(gp), n, st), "bb")
    L.check(l.gmp_sigmoid_bwd(ops._ptr(gp), ops._ptr(p2), ops._ptr(gx2), n, st), "sb")
    assert torch.equal(p, p2) and torch.equal(gx, gx2) and torch.equal(loss, loss2)
    assert bool(torch.isfinite(gx).all()) and bool(torch.isfinite(loss).all())
    want = torch.nn.functional.binary_cross_entropy(torch.sigmoid(x.cpu()), y.cpu(), reduction="sum")
    close(loss, want.reshape(1), what="sigmoid+bce loss")


def test_bn_split_entry_points_equal_the_inline_ones():
    """gmp_bn_fwd without running stats + gmp_bn_running_update(_batch), and gmp_bn_bwd with 0 groups + gmp_bn_param_grads,
    give bit-identical results to the all-in-one calls (the step executor uses the split forms off its critical path)."""
    import ctypes as C
    from gnn_pretraining_amd import _lib as L
    gen = torch.Generator().manual_seed(9)
    S_, Cc = 6, 512
    sizes = torch.tensor([40, 0, 300, 17, 64, 129])
    ptr = torch.zeros(S_ + 1, dtype=torch.int32); ptr[1:] = sizes.cumsum(0)
    N = int(ptr[-1]); mx = int(sizes.max()); ptr = ptr.to(DEV)
    x, g = torch.randn(N, Cc, generator=gen).to(DEV), torch.randn(N, Cc, generator=gen).to(DEV)
    gam, bet = (torch.rand(Cc, generator=gen) + 0.5).to(DEV), (torch.randn(Cc, generator=gen) * 0.1).to(DEV)
    rm0, rv0 = torch.randn(Cc, generator=gen).to(DEV), (torch.rand(Cc, generator=gen) + 0.5).to(DEV)
    cfg = ops.make_bn_config(True, True)
    rm_a, rv_a = rm0.clone(), rv0.clone()
    y_a, sm, sr = ops.bn_fwd(x, None, ptr, mx, gam, bet, rm_a, rv_a, cfg)                       # inline running update
    y_b, sm_b, sr_b = ops.bn_fwd(x, None, ptr, mx, gam, bet, None, None, cfg)                   # deferred
    l, st = L.lib(), ops._stream(x)
    rm_b, rv_b = rm0.clone(), rv0.clone()
    L.check(l.gmp_bn_running_update(ops._ptr(ptr), None, S_, Cc, ops._ptr(rm_b), ops._ptr(rv_b), ops._ptr(sm_b), ops._ptr(sr_b), C.byref(cfg), st), "ru")
    rm_c, rv_c = rm0.clone(), rv0.clone()
    P = C.c_void_p
    L.check(l.gmp_bn_running_update_batch(1, ops._ptr(ptr), S_, None, (C.c_int32 * 1)(Cc), (P * 1)(rm_c.data_ptr()), (P * 1)(rv_c.data_ptr()),
                                          (P * 1)(sm_b.data_ptr()), (P * 1)(sr_b.data_ptr()), C.byref(cfg), st), "rub")
    live = (sizes > 0).to(DEV)                                         # an empty segment leaves its statistics rows unwritten
    assert torch.equal(y_a, y_b) and torch.equal(sm[live], sm_b[live]) and torch.equal(sr[live], sr_b[live])
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b) and torch.equal(rm_a, rm_c) and torch.equal(rv_a, rv_c)
    assert not torch.equal(rm_a, rm0)
    groups = [0, 2, 3, 6]
    gu_a, gg_a, gb_a = ops.bn_bwd(g, x, None, ptr, mx, gam, bet, None, None, sm, sr, cfg, groups)
    G = len(groups) - 1
    ws = torch.empty(l.gmp_bn_workspace_bytes(N, Cc, S_, mx), dtype=torch.uint8, device=DEV)
    gu_b = torch.empty_like(x)
    gg_b, gb_b = torch.empty(G, Cc, device=DEV), torch.empty(G, Cc, device=DEV)
    arr = (C.c_int32 * (G + 1))(*groups)
    L.check(l.gmp_bn_bwd(ops._ptr(g), ops._ptr(x), None, ops._ptr(ptr), None, S_, mx, N, Cc, ops._ptr(gam), ops._ptr(bet), None, None, ops._ptr(sm), ops._ptr(sr),
                         ops._ptr(gu_b), None, None, None, None, None, 0, C.byref(cfg), ops._ptr(ws), ws.numel(), st), "bwd0")
    L.check(l.gmp_bn_param_grads(ops._ptr(ws), S_, Cc, ops._ptr(gg_b), ops._ptr(gb_b), C.cast(arr, C.c_void_p), None, None, G, st), "pg")
    assert torch.equal(gu_a, gu_b) and torch.equal(gg_a, gg_b) and torch.equal(gb_a, gb_b)


def test_gin_aggregate_full_size_properties():
    """The roofline rung (65,536 graphs: 2.1 M rows, x = 2.2 GB) is too large for the CPU oracle to finish in seconds; at that size
    the kernel is held to two size-independent identities of GINConv's sum aggregation (gnn.py:29):
      * column checksum: sum_r out[r,:] = sum_u x[u,:] * ((1 + eps) + #times u is a neighbour)   (float64 accumulation)
      * linearity:        agg(a*x + y) = a*agg(x) + agg(y)."""
    from gnn_pretraining_amd.graph import Batch
    gen = torch.Generator().manual_seed(7)
    base = Batch.from_data_list([S.random_graph(gen, 4) for _ in range(1024)])
    reps, n0, e0 = 64, base.num_nodes, base.num_edges
    N, E = n0 * reps, e0 * reps
    ei = base.edge_index.to(DEV)
    ei_big = (ei.view(2, 1, e0) + (torch.arange(reps, device=DEV) * n0).view(1, reps, 1)).reshape(2, E).contiguous()
    csr = ops.csr_build(ei_big, N)
    del ei_big
    eps = torch.tensor([0.3], device=DEV)
    x = torch.randn(N, 256, device=DEV)
    out = ops.gin_aggregate_fwd(x, csr.rowptr, csr.col, eps)
    w = (1.0 + 0.3) + torch.bincount(csr.col.long(), minlength=N).double()
    want = (x.double() * w[:, None]).sum(0)
    got = out.double().sum(0)
    scale = (x.double().abs() * w[:, None]).sum(0)                   # magnitude summed per column (the sums themselves cancel)
    assert float(((got - want).abs() / scale).max()) < 1e-6
    y = torch.randn(N, 256, device=DEV)
    lhs = ops.gin_aggregate_fwd(2.5 * x + y, csr.rowptr, csr.col, eps)
    rhs = 2.5 * out + ops.gin_aggregate_fwd(y, csr.rowptr, csr.col, eps)
    del x, y
    err = float((lhs - rhs).abs().max())
    assert err < 1e-3 * float(rhs.abs().max()), err
    assert bool(torch.isfinite(out).all())


def test_csr_build_segmented_equals_the_whole_batch_build():
    """Block-diagonal batches (the stacked step: one block per forward() call) get one workgroup per block; all six arrays
    must be those of gmp_csr_build on the whole batch -- including empty blocks (domains absent from a step) and blocks of
    isolated nodes."""
    import ctypes as C
    from gnn_pretraining_amd import _lib as L
    gen = torch.Generator().manual_seed(17)
    blocks = [S.domain_batch(gen, 4, k) for k in (8, 1, 3, 8, 32)]
    rows, eds, seg_row, seg_edge = 0, [], [0], [0]
    for i, b in enumerate(blocks):
        eds.append(b.edge_index + rows)
        rows += b.num_nodes
        seg_row.append(rows); seg_edge.append(seg_edge[-1] + b.num_edges)
        if i == 1:                                             # an empty block, and a block without edges (5 isolated nodes)
            seg_row.append(rows); seg_edge.append(seg_edge[-1])
            rows += 5
            seg_row.append(rows); seg_edge.append(seg_edge[-1])
    seg_row.append(rows); seg_edge.append(seg_edge[-1])        # trailing empty block
    ei = torch.cat(eds, dim=1).contiguous().to(DEV)
    N, E, Sg = rows, ei.size(1), len(seg_row) - 1
    want = ops.csr_build(ei, N)
    l, st = L.lib(), ops._stream(ei)
    i32 = lambda n: torch.full((n,), -7, dtype=torch.int32, device=DEV)
    out = [i32(N + 1), i32(E), i32(E), i32(N + 1), i32(E), i32(E)]
    status = torch.ones(1, dtype=torch.int32, device=DEV)
    sr, se = torch.tensor(seg_row, dtype=torch.int32, device=DEV), torch.tensor(seg_edge, dtype=torch.int32, device=DEV)
    mr, me = max(b - a for a, b in zip(seg_row[:-1], seg_row[1:])), max(b - a for a, b in zip(seg_edge[:-1], seg_edge[1:]))
    L.check(l.gmp_csr_build_segmented(ops._ptr(ei), N, E, ops._ptr(sr), ops._ptr(se), Sg, mr, me, *[ops._ptr(t) for t in out], ops._ptr(status), st), "seg")
    assert int(status) == 0
    for got, exp, name in zip(out, (want.rowptr, want.col, want.perm, want.rowptr_t, want.col_t, want.perm_t),
                              ("rowptr", "col", "perm", "rowptr_t", "col_t", "perm_t")):
        assert torch.equal(got, exp), name
    # an edge that leaves its block is dropped and counted, never dereferenced
    bad = ei.clone(); bad[0, 0] = N - 1
    L.check(l.gmp_csr_build_segmented(ops._ptr(bad), N, E, ops._ptr(sr), ops._ptr(se), Sg, mr, me, *[ops._ptr(t) for t in out], ops._ptr(status), st), "seg")
    assert int(status) == 1
    with pytest.raises(Exception):
        L.check(l.gmp_csr_build_segmented(ops._ptr(ei), N, E, ops._ptr(sr), ops._ptr(se), Sg, 100000, me, *[ops._ptr(t) for t in out], ops._ptr(status), st), "seg")


@pytest.mark.parametrize("rows", [0, 1, 255, 15431])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_linear_to_one_column_kernels(rows, p):
    """The scorer's Linear(hidden, 1) behind ReLU + dropout (src/models/heads.py:45-52) as row dot / outer product / weighted column
    sum: against torch in fp64 with the mask the library's own dropout kernel draws for the same (seed, site) -- 1e-5 of the largest
    entry (fp32 sums over 256 columns / 15 k rows); the dropped copy and the mask are bit-identical to gmp_dropout_fwd's."""
    gen = torch.Generator().manual_seed(rows + 7)
    F = 256
    x = torch.relu(torch.randn(rows, F, generator=gen)).to(DEV)
    w, b = torch.randn(F, generator=gen).to(DEV), torch.randn(1, generator=gen).to(DEV)
    g = torch.randn(rows, generator=gen).to(DEV)
    d, y = ops.dropout_rowdot_fwd(x, w, b, p, 99, 103)
    d_ref = ops.dropout_fwd(x, p, 99, 103) if p > 0 else x
    assert torch.equal(d, d_ref)
    if rows == 0:
        ow, ob = ops.weighted_colsum(g, d)
        assert ow.abs().max().item() == 0 and ob.item() == 0
        return
    y_ref = d_ref.double() @ w.double() + b.double()
    assert (y.double() - y_ref).abs().max().item() <= 1e-5 * max(y_ref.abs().max().item(), 1.0)
    gi = ops.outer_relu_dropout_bwd(g, w, x, p, 99, 103)
    keep = (d_ref != 0).double() / (1 - p) if p > 0 else (x > 0).double()
    gi_ref = g.double()[:, None] * w.double()[None, :] * keep
    assert (gi.double() - gi_ref).abs().max().item() <= 1e-6 * max(gi_ref.abs().max().item(), 1.0)
    ow, ob = ops.weighted_colsum(g, d)
    ow_ref, ob_ref = g.double() @ d_ref.double(), g.double().sum()
    assert (ow.double() - ow_ref).abs().max().item() <= 1e-5 * max(ow_ref.abs().max().item(), 1.0)
    assert abs(ob.item() - ob_ref.item()) <= 1e-5 * max(abs(ob_ref.item()), g.abs().sum().item() * 1e-2, 1.0)


@pytest.mark.parametrize("K", [0, 1, 300, 15431])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_merged_pair_rows_keep_a_dropout_mask_per_ordered_row(K, p):
    """gmp_lp_pair_* (one row per unordered pair, heads.py:57-61 being symmetric) against the UNMERGED kernels run over the reference's ordered
    list (tasks.py:111-120) with the same (seed, site): every ordered row must get the mask it would get there (heads.py:44-52 drops every row
    independently) -- scores bit for bit, loss / input gradient / weight gradient to 1e-6 (sums in another order)."""
    from gnn_pretraining_amd import _lib as L
    gen = torch.Generator().manual_seed(K + 3)
    F = 256
    two = torch.rand(K, generator=gen) < 0.7                                   # rows that stand for two ordered rows
    n_ord = K + int(two.sum())
    perm = torch.randperm(n_ord, generator=gen)                                # their positions in the ordered list, scattered
    pos = torch.full((2, K), -1, dtype=torch.int32)
    pos[0] = perm[:K].to(torch.int32)
    pos[1, two] = perm[K:].to(torch.int32)
    sign = torch.where(torch.rand(K, generator=gen) < 0.5, 1.0, -1.0) * (1.0 + two.float())
    y1 = torch.relu(torch.randn(K, F, generator=gen))
    w, b = torch.randn(F, generator=gen) * 0.1, torch.randn(1, generator=gen)
    gs = torch.tensor([1.0 / max(n_ord, 1)])
    y2, loss, g_y2, g_y1, g_w, g_b = ops.lp_pair_head(y1.to(DEV), w.to(DEV), b.to(DEV), pos.to(DEV), sign.to(DEV), gs.to(DEV), p, 4242, 101)
    if K == 0:
        assert loss.item() == 0 and g_w.abs().max().item() == 0 and g_b.item() == 0
        return
    # the ordered list, scored by the unmerged kernels
    row_of = torch.empty(n_ord, dtype=torch.long)
    row_of[pos[0].long()] = torch.arange(K)
    row_of[pos[1, two].long()] = torch.arange(K)[two]
    y1o = y1[row_of].contiguous().to(DEV)
    lab = (sign[row_of] > 0).float().to(DEV)
    d, yo = ops.dropout_rowdot_fwd(y1o, w.to(DEV), b.to(DEV), p, 4242, 101)
    l = L.lib()
    loss_o, g_yo, prob = torch.zeros(1, device=DEV), torch.empty(n_ord, device=DEV), torch.empty(n_ord, device=DEV)
    lws = ops._ws(l.gmp_loss_workspace_bytes(n_ord), DEV)
    L.check(l.gmp_sigmoid_bce_sum_fwd_bwd(ops._ptr(yo), ops._ptr(lab), n_ord, ops._ptr(gs.to(DEV)), ops._ptr(loss_o), ops._ptr(prob), ops._ptr(g_yo),
                                          ops._ptr(lws), lws.numel(), ops._stream(yo)), "bce")
    g_y1o = ops.outer_relu_dropout_bwd(g_yo, w.to(DEV), y1o, p, 4242, 101)
    g_wo, g_bo = ops.weighted_colsum(g_yo, d)
    yo, g_yo, g_y1o = yo.cpu(), g_yo.cpu(), g_y1o.cpu()
    y2, g_y2 = y2.cpu(), g_y2.cpu()
    assert torch.equal(y2[0], yo[pos[0].long()]) and torch.equal(y2[1][two], yo[pos[1, two].long()])          # scores: bit for bit
    assert torch.equal(g_y2[0], g_yo[pos[0].long()]) and torch.equal(g_y2[1][two], g_yo[pos[1, two].long()])
    assert (g_y2[1][~two] == 0).all()
    if p > 0 and two.any():                                                                                   # the two masks of a row do differ
        assert (y2[0][two] != y2[1][two]).any()
    rel = lambda a, b: (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)
    assert rel(loss.cpu(), loss_o.cpu()) <= 1e-6
    want_g_y1 = torch.zeros(K, F, dtype=torch.float64).index_add_(0, row_of, g_y1o.double())                  # both ordered rows meet in the merged one
    assert rel(g_y1.cpu(), want_g_y1) <= 1e-6
    assert rel(g_w.cpu(), g_wo.cpu()) <= 2e-6 and abs(g_b.item() - g_bo.item()) <= 2e-6 * max(abs(g_bo.item()), g_yo.abs().sum().item() * 1e-2)
