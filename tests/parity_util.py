"""Shared helpers for the module/step parity tests."""
import numpy as np
import torch
import torch.nn as nn

from gnn_pretraining_amd.graph import Batch
from oracle import graph_ops as OG


from oracle.harness import copy_state, set_dropout, to_oracle      # noqa: E402,F401  (shared with smoke() / bench.py)


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-30)


def assert_close(got, want, rtol, what="", floor=0.0):
    """max |got - want| <= rtol * max(max|want|, floor).  `floor` guards tensors that are analytically
    zero (e.g. the gradient of a bias that feeds a train-mode BatchNorm): both sides hold rounding noise."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    e = (got - want).abs().max().item() / max(want.abs().max().item(), floor, 1e-30)
    assert e <= rtol, f"{what}: max rel err {e:.3e} > {rtol}"


def assert_grad_close(got, want, gmax, what=""):
    """Gradient parity through ReLU networks.  An fp32 pre-activation within rounding of 0 gates its ReLU
    differently in two correct implementations (about one such element per backbone forward at these
    sizes: 1e6 pre-activations x density(0) x 2e-6); the gradient of the graph containing it then moves
    by ~1e-3.  scripts/diag_precision.py shows both implementations otherwise sit 2-7e-7 from fp64.
    scripts/diag_tasks_fp64.py shows the error is bimodal -- ~1e-6 without a flip, ~1e-3 with one -- and
    that the fp32 ORACLE shows the same flips against its own fp64 run (up to 1e-1 on a scalar eps
    gradient).  So: max-norm within 1e-1, L2 within 2e-2, measured against max(|want|, 1e-3 * largest
    gradient) (the floor covers analytically-zero gradients such as a bias feeding a train-mode BatchNorm).
    (L2 was 1e-2 until round 3: the fp32 oracle's own rounding depends on the host CPU's thread count, and on one GPU box
    three checks landed at 1.02e-2 .. 1.09e-2 -- among them an analytically-zero bias gradient where the ORACLE held 1.8e-6
    of noise and the HIP path 3.6e-7.)  Flip-free gradient checks at 1e-4 .. 2e-4 are the per-operator tests in
    test_gpu_ops.py, and the shared-gate step tests (oracle/gates.py) hold every gradient to 2.5e-4."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    d = got - want
    # a scalar gradient (GINConv.eps) is one cancelling sum of N*256 products: its own magnitude says nothing
    # about the size of the terms, so it is held against the largest gradient instead
    floor = gmax if want.numel() == 1 else 1e-3 * gmax
    e_max = d.abs().max().item() / max(want.abs().max().item(), floor, 1e-30)
    e_l2 = d.norm().item() / max(want.norm().item(), floor * want.numel() ** 0.5, 1e-30)
    assert e_max <= 1e-1 and e_l2 <= 2e-2, f"{what}: max-norm rel {e_max:.3e}, L2 rel {e_l2:.3e}"


def engine_gate_tapes(eng, plan, art):
    """The ReLU gates the HIP engine used in its last step, as one oracle.gates tape per task, in the oracle's call order
    (oracle/tasks.py: per task, per domain [per view]: encoder ReLU, then inner / outer ReLU of the five GIN layers, then the
    task head's ReLU(s) for that domain; link prediction: the signs of hs - hd first).  Read from the activations the step leaves behind: h[0] (encoder output; rows the
    node-feature-masking task overwrote with the mask token come from its saved targets), r1[l] / h[l + 1] per layer, the
    heads' y1 buffers.  Dropout must be off (a dropped unit reads as a closed gate)."""
    from oracle.gates import GateTape
    N = plan.N
    enc = (eng.h[0][:N] > 0).cpu()
    if "node_feat_mask" in eng.tasks and plan.nfm_rows[-1]:
        idx = torch.from_numpy(plan.a64["nfm_idx"]).long()
        enc[idx] = (eng.hd["nfm_tgt"][:idx.numel()] > 0).cpu()
    inner = [(eng.r1[l][:N] > 0).cpu() for l in range(len(eng.r1))]
    outer = [(eng.h[l + 1][:N] > 0).cpu() for l in range(len(eng.r1))]
    segs = {}
    for si, (ti, di) in enumerate(zip(plan.seg_task, plan.seg_dom)):
        segs.setdefault((ti, di), []).append(si)

    def backbone(si, with_layers=True):
        a, b = plan.seg_ptr[si], plan.seg_ptr[si + 1]
        out = [enc[a:b]]
        if with_layers:
            for l in range(len(inner)):
                out += [inner[l][a:b], outer[l][a:b]]
        return out

    tapes = {}
    for ti, t in enumerate(eng.tasks):
        m = []
        lp_off = 0
        for di, d in enumerate(eng.domains):
            ss = segs.get((ti, di), [])
            if t == "node_feat_mask":
                r0, r1 = plan.nfm_rows[di], plan.nfm_rows[di + 1]
                m += backbone(ss[0], with_layers=r1 > r0)             # no masked node: the oracle stops after the encoder
                if r1 > r0:
                    m.append((eng.hd["nfm_y1"][r0:r1] > 0).cpu())
            elif t == "link_pred":
                m += backbone(ss[0])
                # the oracle scores the reference's ordered list (positives, then negatives); the engine holds one row per unordered
                # pair (engine.merge_mirrored_pairs): look every ordered pair's row up by its (min, max) key
                hb, r0 = eng.last_inputs.host[d], plan.seg_ptr[ss[0]]
                ordered = np.concatenate([hb.edge_index.numpy(), np.asarray(art["link_pred"][d], dtype=np.int64)], axis=1) + r0
                a, b = int(plan.a32["lp_seg_eptr"][di]), int(plan.a32["lp_seg_eptr"][di + 1])
                rows, R = plan.a64["lp_edges"][:, a:b], plan.N + 1
                key_rows = np.minimum(rows[0], rows[1]) * R + np.maximum(rows[0], rows[1])
                order = np.argsort(key_rows, kind="stable")
                key_ord = np.minimum(ordered[0], ordered[1]) * R + np.maximum(ordered[0], ordered[1])
                at = order[np.searchsorted(key_rows[order], key_ord)] if b > a else np.zeros(0, dtype=np.int64)
                assert (key_rows[at] == key_ord).all()
                e = torch.from_numpy(ordered).long().to(eng.h[-1].device)
                m.append(torch.sign(eng.h[-1][e[0]] - eng.h[-1][e[1]]).to(torch.int8).cpu())      # sign(hs - hd): the |.| feature's kink
                m.append((eng.hd["lp_y1"][a:b] > 0).cpu()[torch.from_numpy(at)])
            elif t == "node_contrast":
                for si in ss:
                    m += backbone(si)
                n = plan.nc_n[di]
                if n:
                    r0 = plan.nc_rows[di]
                    m += [(eng.hd["nc_y1"][r0:r0 + n] > 0).cpu(), (eng.hd["nc_y1"][r0 + n:r0 + 2 * n] > 0).cpu()]
            elif t == "graph_contrast":
                if not ss:
                    continue
                for si in ss:
                    m += backbone(si)
                B, r0 = plan.gc_n[di], plan.gc_rows[di]
                m += [(eng.hd["gc_y1"][r0:r0 + B] > 0).cpu(), (eng.hd["gc_y1"][r0 + B:r0 + 2 * B] > 0).cpu()]
            elif t == "graph_prop":
                m += backbone(ss[0])
                m.append((eng.hd["gp_y1"][plan.gp_rows[di]:plan.gp_rows[di + 1]] > 0).cpu())
            elif t == "domain_adv":
                m += backbone(ss[0])
                a, b = plan.seg_ptr[ss[0]], plan.seg_ptr[ss[0] + 1]
                g0 = sum(1 for s in plan.a32["da_ptr"][:-1] if s < a)
                g1 = sum(1 for s in plan.a32["da_ptr"][:-1] if s < b)
                m.append((eng.hd["da_y1"][g0:g1] > 0).cpu())
        tapes[t] = GateTape(m)
    return tapes


def assert_grad_tight(got, want, gmax, what="", tol=2e-4):
    """Flip-free gradient parity (both sides used the same ReLU gates): max |got - want| <= tol * max(max|want|, 1e-3 * largest
    gradient of the task) -- the floor only covers analytically-zero gradients (a bias in front of a train-mode BatchNorm), where
    both sides hold rounding noise; a scalar (GINConv.eps: one cancelling sum of N * 256 products) is held against gmax."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    floor = gmax if want.numel() == 1 else 1e-3 * gmax
    e = (got - want).abs().max().item() / max(want.abs().max().item(), floor, 1e-30)
    assert e <= tol, f"{what}: max rel err {e:.3e} > {tol}"
    return e
