// Host half of a pre-training step in the reference's RNG order, as one native call per (task, domain batch).
//
// The reference draws masks / negatives / augmentations graph by graph from ONE CPU torch.Generator
// (src/models/pretrain_model.py:71-80, src/pretrain/tasks.py:107-111, src/pretrain/augmentations.py:17-111); the Python
// implementation of that order (models/pretrain_model.py draw_mask_indices, pretrain/tasks.py sample_negative_edges,
// pretrain/augmentations.py _augment_one + engine.StepEngine._draw_views) costs ~3 ms per step in interpreter and numpy
// call overhead and holds the GIL while it runs.  This file is the same arithmetic in C++ on the caller's generator --
// CPUGeneratorImpl::random() for randperm's Fisher-Yates swaps exactly as ATen's randperm_cpu does them, ATen's own
// uniform_real_distribution<float> for torch.rand(1) -- so the produced index arrays are bit-identical
// (tests/test_hostdraw.py), and it releases the GIL.  It is an accelerator of host code, not a second implementation of a
// GPU path: when the module is not built the Python code runs instead.
#include <torch/extension.h>

#include <ATen/CPUGeneratorImpl.h>
#include <ATen/core/DistributionsHelper.h>

#include <algorithm>
#include <mutex>
#include <vector>

namespace {

struct Rng {
    at::CPUGeneratorImpl* g;
    // torch.randperm(n, generator=g) for n < 2^32 / 20 (ATen/native/TensorFactories.cpp randperm_cpu)
    void randperm(int64_t n, std::vector<int64_t>& r) {
        r.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) r[(size_t)i] = i;
        for (int64_t i = 0; i < n - 1; ++i) {
            const int64_t z = (int64_t)(g->random() % (uint64_t)(n - i));
            std::swap(r[(size_t)i], r[(size_t)(z + i)]);
        }
    }
    // torch.rand(1, generator=g).item()
    float rand1() {
        at::uniform_real_distribution<float> u(0.0f, 1.0f);
        return u(g);
    }
};

at::Tensor to_tensor(const std::vector<int64_t>& v) {
    at::Tensor t = at::empty({(int64_t)v.size()}, at::kLong);
    std::copy(v.begin(), v.end(), t.data_ptr<int64_t>());
    return t;
}

at::Tensor to_tensor2(const std::vector<int64_t>& a, const std::vector<int64_t>& b) {
    at::Tensor t = at::empty({2, (int64_t)a.size()}, at::kLong);
    std::copy(a.begin(), a.end(), t.data_ptr<int64_t>());
    std::copy(b.begin(), b.end(), t.data_ptr<int64_t>() + a.size());
    return t;
}

void check_ptrs(const at::Tensor& ptr, const at::Tensor& eptr, const at::Tensor& ei) {
    TORCH_CHECK(ptr.dtype() == at::kLong && eptr.dtype() == at::kLong && ei.dtype() == at::kLong, "hostdraw: int64 tensors expected");
    TORCH_CHECK(ptr.is_contiguous() && eptr.is_contiguous() && ei.is_contiguous() && ptr.device().is_cpu() && ei.device().is_cpu(),
                "hostdraw: contiguous CPU tensors expected");
    TORCH_CHECK(ptr.dim() == 1 && eptr.dim() == 1 && ptr.numel() == eptr.numel() && ei.dim() == 2 && ei.size(0) == 2, "hostdraw: shapes");
}

}  // namespace

// models/pretrain_model.py draw_mask_indices
at::Tensor mask_indices(at::Tensor ptr, at::Generator gen) {
    TORCH_CHECK(ptr.dtype() == at::kLong && ptr.is_contiguous() && ptr.device().is_cpu(), "hostdraw: ptr");
    auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
    std::vector<int64_t> out, perm;
    {
        pybind11::gil_scoped_release nogil;
        std::lock_guard<std::mutex> lock(impl->mutex_);
        Rng rng{impl};
        const int64_t* p = ptr.data_ptr<int64_t>();
        for (int64_t gidx = 0; gidx + 1 < ptr.numel(); ++gidx) {
            const int64_t s = p[gidx], n = p[gidx + 1] - s;
            if (n >= 3) {
                const int64_t k = std::max<int64_t>(1, (int64_t)(n * 0.15));
                rng.randperm(n, perm);
                for (int64_t i = 0; i < k; ++i) out.push_back(perm[(size_t)i] + s);
            }
        }
    }
    return to_tensor(out);
}

// pretrain/tasks.py sample_negative_edges
at::Tensor negative_edges(at::Tensor ptr, at::Tensor eptr, at::Tensor edge_index, at::Generator gen) {
    check_ptrs(ptr, eptr, edge_index);
    auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
    std::vector<int64_t> os, od, perm, cand;
    std::vector<char> adj;
    {
        pybind11::gil_scoped_release nogil;
        std::lock_guard<std::mutex> lock(impl->mutex_);
        Rng rng{impl};
        const int64_t *p = ptr.data_ptr<int64_t>(), *ep = eptr.data_ptr<int64_t>(), *src = edge_index.data_ptr<int64_t>();
        const int64_t E = edge_index.size(1);
        const int64_t* dst = src + E;
        for (int64_t gidx = 0; gidx + 1 < ptr.numel(); ++gidx) {
            const int64_t s = p[gidx], n = p[gidx + 1] - s, es = ep[gidx], ee = ep[gidx + 1];
            adj.assign((size_t)(n * n), 0);
            for (int64_t e = es; e < ee; ++e) {
                const int64_t a = src[e] - s, b = dst[e] - s;
                adj[(size_t)(a * n + b)] = 1;
                adj[(size_t)(b * n + a)] = 1;
            }
            for (int64_t i = 0; i < n; ++i) adj[(size_t)(i * n + i)] = 1;
            cand.clear();
            for (int64_t f = 0; f < n * n; ++f)
                if (!adj[(size_t)f]) cand.push_back(f);
            const int64_t k = std::min<int64_t>(ee - es, (int64_t)cand.size());
            if (k == 0) continue;
            rng.randperm((int64_t)cand.size(), perm);
            for (int64_t i = 0; i < k; ++i) {
                const int64_t f = cand[(size_t)perm[(size_t)i]];
                os.push_back(f / n + s);
                od.push_back(f % n + s);
            }
        }
    }
    return to_tensor2(os, od);
}

// engine.StepEngine._draw_views over pretrain/augmentations.py _augment_one: two views of every graph of a domain batch.
// Returns, per view: rows (kept nodes, batch numbering), edges [2, e'] (view numbering), ptr [B+1], rowmask (bit c set = column c
// zeroed; empty tensor when no graph of the view drew an attribute mask), common (view-local ids kept in BOTH views).
std::vector<at::Tensor> draw_views(at::Tensor ptr, at::Tensor eptr, at::Tensor edge_index, int64_t num_features, at::Generator gen) {
    check_ptrs(ptr, eptr, edge_index);
    TORCH_CHECK(num_features >= 0 && num_features <= 64, "hostdraw: attribute masks are 64-bit column sets");
    auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
    struct Acc {
        std::vector<int64_t> rows, es, ed, ptr{0}, masks, common;
        bool any_mask = false;
    } acc[2];
    {
        pybind11::gil_scoped_release nogil;
        std::lock_guard<std::mutex> lock(impl->mutex_);
        Rng rng{impl};
        const int64_t *p = ptr.data_ptr<int64_t>(), *ep = eptr.data_ptr<int64_t>(), *src = edge_index.data_ptr<int64_t>();
        const int64_t E = edge_index.size(1);
        const int64_t* dst = src + E;
        std::vector<int64_t> perm, kept[2], ve_s[2], ve_d[2], relabel, ts, td;
        std::vector<char> flag[2];
        uint64_t mask[2];
        for (int64_t gidx = 0; gidx + 1 < ptr.numel(); ++gidx) {
            const int64_t s = p[gidx], n = p[gidx + 1] - s, es = ep[gidx], ee = ep[gidx + 1];
            for (int v = 0; v < 2; ++v) {
                // ---- _augment_one: node drop
                ve_s[v].clear(); ve_d[v].clear();
                if (n >= 3) {
                    const int64_t keep_n = n - std::max<int64_t>(1, (int64_t)(n * 0.2));
                    rng.randperm(n, perm);
                    kept[v].assign(perm.begin(), perm.begin() + keep_n);
                    std::sort(kept[v].begin(), kept[v].end());
                    relabel.assign((size_t)n, -1);
                    for (int64_t i = 0; i < keep_n; ++i) relabel[(size_t)kept[v][(size_t)i]] = i;
                    for (int64_t e = es; e < ee; ++e) {                       // subgraph(): edge order preserved
                        const int64_t a = relabel[(size_t)(src[e] - s)], b = relabel[(size_t)(dst[e] - s)];
                        if (a >= 0 && b >= 0) { ve_s[v].push_back(a); ve_d[v].push_back(b); }
                    }
                } else {
                    kept[v].resize((size_t)n);
                    for (int64_t i = 0; i < n; ++i) kept[v][(size_t)i] = i;
                    for (int64_t e = es; e < ee; ++e) { ve_s[v].push_back(src[e] - s); ve_d[v].push_back(dst[e] - s); }
                }
                // ---- edge drop
                if (rng.rand1() < 0.2f) {
                    const int64_t e = (int64_t)ve_s[v].size();
                    if (e >= 3) {
                        const int64_t keep_e = e - std::max<int64_t>(1, (int64_t)(e * 0.2));
                        rng.randperm(e, perm);
                        ts.resize((size_t)keep_e); td.resize((size_t)keep_e);
                        for (int64_t i = 0; i < keep_e; ++i) { ts[(size_t)i] = ve_s[v][(size_t)perm[(size_t)i]]; td[(size_t)i] = ve_d[v][(size_t)perm[(size_t)i]]; }
                        ve_s[v].swap(ts); ve_d[v].swap(td);
                    }
                }
                // ---- attribute mask
                mask[v] = 0;
                if (rng.rand1() < 0.2f) {
                    if (num_features >= 3) {
                        const int64_t m = std::max<int64_t>(1, (int64_t)(num_features * 0.2));
                        rng.randperm(num_features, perm);
                        for (int64_t i = 0; i < m; ++i) mask[v] |= (uint64_t)1 << perm[(size_t)i];
                    }
                }
                flag[v].assign((size_t)n, 0);
                for (int64_t k : kept[v]) flag[v][(size_t)k] = 1;
            }
            for (int v = 0; v < 2; ++v) {
                Acc& a = acc[v];
                const int64_t base = a.ptr.back();
                for (size_t i = 0; i < kept[v].size(); ++i) {
                    a.rows.push_back(kept[v][i] + s);
                    a.masks.push_back((int64_t)mask[v]);
                    if (flag[1 - v][(size_t)kept[v][i]]) a.common.push_back((int64_t)i + base);
                }
                if (mask[v]) a.any_mask = true;
                for (size_t i = 0; i < ve_s[v].size(); ++i) { a.es.push_back(ve_s[v][i] + base); a.ed.push_back(ve_d[v][i] + base); }
                a.ptr.push_back(base + (int64_t)kept[v].size());
            }
        }
    }
    std::vector<at::Tensor> out;
    for (int v = 0; v < 2; ++v) {
        out.push_back(to_tensor(acc[v].rows));
        out.push_back(to_tensor2(acc[v].es, acc[v].ed));
        out.push_back(to_tensor(acc[v].ptr));
        out.push_back(acc[v].any_mask ? to_tensor(acc[v].masks) : at::empty({0}, at::kLong));
        out.push_back(to_tensor(acc[v].common));
    }
    return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("mask_indices", &mask_indices, "per-graph randperm(n)[:max(1, int(.15 n))] + offset for n >= 3");
    m.def("negative_edges", &negative_edges, "per graph: as many uniform non-edges as it has COO entries");
    m.def("draw_views", &draw_views, "two augmented views of every graph of a batch, as index arrays");
}
