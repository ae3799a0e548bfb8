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
#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>
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

// models/pretrain_model.py draw_mask_indices.  The *_core functions neither touch the GIL nor lock the generator: their callers do
// (once per call for the single-artefact entry points, once per STEP for draw_step below).
at::Tensor mask_indices_core(const at::Tensor& ptr, at::CPUGeneratorImpl* impl) {
    static thread_local std::vector<int64_t> out, perm;
    out.clear();
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
    return to_tensor(out);
}
at::Tensor mask_indices(at::Tensor ptr, at::Generator gen) {
    TORCH_CHECK(ptr.dtype() == at::kLong && ptr.is_contiguous() && ptr.device().is_cpu(), "hostdraw: ptr");
    auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
    pybind11::gil_scoped_release nogil;
    std::lock_guard<std::mutex> lock(impl->mutex_);
    return mask_indices_core(ptr, impl);
}

// ---- pretrain/tasks.py sample_negative_edges: PyG's batched_negative_sampling, drawn from PYTHON's random ---------------------
// PyG's sampler (torch_geometric/utils/_negative_sampling.py `sample`) calls random.sample(range(population), k) on Python's
// global Mersenne Twister, so the native twin carries a CPython-compatible MT19937: same state words (random.getstate()[1]),
// getrandbits(k) = genrand_uint32() >> (32 - k), _randbelow_with_getrandbits and both branches of random.sample (pool for
// n <= 21 + 4 ** ceil(log(3k, 4)), rejection set otherwise) -- CPython 3.10 Lib/random.py.  Given an equal state it returns the
// negatives the Python code returns and leaves the equal state behind (tests/test_hostdraw.py).
struct PyRandom {
    uint32_t mt[624];
    int pos = 624;

    void setstate(const at::Tensor& st) {
        TORCH_CHECK(st.dtype() == at::kLong && st.numel() == 625 && st.is_contiguous(), "PyRandom: 625 int64 state words expected");
        const int64_t* v = st.data_ptr<int64_t>();
        for (int i = 0; i < 624; ++i) mt[i] = (uint32_t)v[i];
        TORCH_CHECK(v[624] >= 0 && v[624] <= 624, "PyRandom: bad state position");
        pos = (int)v[624];
    }
    at::Tensor getstate() const {
        at::Tensor t = at::empty({625}, at::kLong);
        int64_t* v = t.data_ptr<int64_t>();
        for (int i = 0; i < 624; ++i) v[i] = mt[i];
        v[624] = pos;
        return t;
    }
    uint32_t next32() {          // genrand_uint32 of _randommodule.c
        constexpr int N = 624, M = 397;
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
        if (pos >= N) {
            int kk;
            uint32_t y;
            for (kk = 0; kk < N - M; ++kk) {
                y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
                mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
            }
            for (; kk < N - 1; ++kk) {
                y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
                mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
            }
            y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
            pos = 0;
        }
        uint32_t y = mt[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // Random._randbelow_with_getrandbits(n), 0 < n < 2^32
    int64_t randbelow(int64_t n) {
        int k = 0;
        for (int64_t v = n; v; v >>= 1) ++k;            // n.bit_length()
        uint32_t r = next32() >> (32 - k);
        while ((int64_t)r >= n) r = next32() >> (32 - k);
        return (int64_t)r;
    }
    // random.sample(range(n), k)
    void sample_range(int64_t n, int64_t k, std::vector<int64_t>& out, std::vector<int64_t>& pool, std::vector<char>& seen) {
        out.resize((size_t)k);
        double setsize = 21;
        if (k > 5) setsize += std::pow(4.0, std::ceil(std::log((double)(k * 3)) / std::log(4.0)));
        if ((double)n <= setsize) {
            pool.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) pool[(size_t)i] = i;
            for (int64_t i = 0; i < k; ++i) {
                const int64_t j = randbelow(n - i);
                out[(size_t)i] = pool[(size_t)j];
                pool[(size_t)j] = pool[(size_t)(n - i - 1)];
            }
        } else {
            seen.assign((size_t)n, 0);
            for (int64_t i = 0; i < k; ++i) {
                int64_t j = randbelow(n);
                while (seen[(size_t)j]) j = randbelow(n);
                seen[(size_t)j] = 1;
                out[(size_t)i] = j;
            }
        }
    }

    // sample_negative_edges(batch, rng) for a whole domain batch
    at::Tensor negative_edges(at::Tensor ptr, at::Tensor eptr, at::Tensor edge_index) {
        check_ptrs(ptr, eptr, edge_index);
        pybind11::gil_scoped_release nogil;
        return negative_edges_core(ptr, eptr, edge_index);
    }
    at::Tensor negative_edges_core(const at::Tensor& ptr, const at::Tensor& eptr, const at::Tensor& edge_index) {
        // scratch that keeps its capacity across calls (a fresh 100 KB vector per call grows through malloc's mmap threshold: page faults)
        static thread_local std::vector<int64_t> os, od, rnd, pool, neg;
        static thread_local std::vector<char> is_edge, taken, seen;
        os.clear(); od.clear();
        {
            const int64_t *p = ptr.data_ptr<int64_t>(), *ep = eptr.data_ptr<int64_t>(), *src = edge_index.data_ptr<int64_t>();
            const int64_t E = edge_index.size(1), num_neg = E;
            const int64_t* dst = src + E;
            for (int64_t gidx = 0; gidx + 1 < ptr.numel(); ++gidx) {
                const int64_t s = p[gidx], n = p[gidx + 1] - s, es = ep[gidx], ee = ep[gidx + 1];
                if (n < 2) continue;
                const int64_t pop = n * n - n;
                TORCH_CHECK(pop < (int64_t(1) << 31), "hostdraw: graph too large for the negative sampler");
                // to_undirected(pos) without self loops, as slots of the n (n - 1) index space: slot(i, j) = i (n - 1) + j - (i < j)
                is_edge.assign((size_t)pop, 0);
                int64_t nidx = 0;
                auto mark = [&](int64_t a, int64_t b) {
                    if (a == b) return;
                    const size_t slot = (size_t)(a * (n - 1) + b - (a < b ? 1 : 0));
                    if (!is_edge[slot]) { is_edge[slot] = 1; ++nidx; }
                };
                for (int64_t e = es; e < ee; ++e) {
                    mark(src[e] - s, dst[e] - s);
                    mark(dst[e] - s, src[e] - s);
                }
                if (nidx >= pop) continue;
                const double prob = 1.0 - (double)nidx / (double)pop;
                const int64_t size = (int64_t)(1.1 * (double)num_neg / prob);
                neg.clear();
                if (pop <= size) {
                    // every try is arange(pop) and draws nothing: the first one takes every non-edge slot in order, the other two find
                    // them all taken -- so the result is the first num_neg non-edges, emitted row by row without the slot division
                    int64_t left = num_neg;
                    for (int64_t r = 0; r < n && left > 0; ++r) {
                        const char* row = is_edge.data() + (size_t)(r * (n - 1));
                        for (int64_t j = 0; j < n - 1 && left > 0; ++j)
                            if (!row[j]) {
                                os.push_back(r + s);
                                od.push_back(j + (r <= j ? 1 : 0) + s);
                                --left;
                            }
                    }
                    continue;
                }
                taken.assign((size_t)pop, 0);
                for (int attempt = 0; attempt < 3; ++attempt) {
                    sample_range(pop, size, rnd, pool, seen);
                    for (int64_t v : rnd)
                        if (!is_edge[(size_t)v] && !taken[(size_t)v]) neg.push_back(v);
                    for (int64_t v : neg) taken[(size_t)v] = 1;      // np.isin(rnd, neg_idx) of the NEXT try (duplicates within one
                    if ((int64_t)neg.size() >= num_neg) {              // try cannot occur: random.sample draws without replacement)
                        neg.resize((size_t)num_neg);
                        break;
                    }
                }
                for (int64_t v : neg) {
                    const int64_t r = v / (n - 1);
                    int64_t c = v % (n - 1);
                    if (r <= c) ++c;
                    os.push_back(r + s);
                    od.push_back(c + s);
                }
            }
        }
        return to_tensor2(os, od);
    }
};

// engine.StepEngine._draw_views over pretrain/augmentations.py _augment_one: two views of every graph of a domain batch.
// Returns, per view: rows (kept nodes, batch numbering), edges [2, e'] (view numbering), ptr [B+1], rowmask (bit c set = column c
// zeroed; empty tensor when no graph of the view drew an attribute mask), common (view-local ids kept in BOTH views).
std::vector<at::Tensor> draw_views_core(const at::Tensor& ptr, const at::Tensor& eptr, const at::Tensor& edge_index, int64_t num_features,
                                        at::CPUGeneratorImpl* impl) {
    struct Acc {
        std::vector<int64_t> rows, es, ed, ptr{0}, masks, common;
        bool any_mask = false;
    } acc[2];
    {
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

std::vector<at::Tensor> draw_views(at::Tensor ptr, at::Tensor eptr, at::Tensor edge_index, int64_t num_features, at::Generator gen) {
    check_ptrs(ptr, eptr, edge_index);
    TORCH_CHECK(num_features >= 0 && num_features <= 64, "hostdraw: attribute masks are 64-bit column sets");
    auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
    pybind11::gil_scoped_release nogil;
    std::lock_guard<std::mutex> lock(impl->mutex_);
    return draw_views_core(ptr, eptr, edge_index, num_features, impl);
}

// engine.StepEngine.draw (rng_mode 'reference') in ONE call: every artefact of a step, tasks in the given order, domains in the given
// order -- the order the reference consumes its generator in (pretrain.py:113-155 over tasks.py compute_loss) -- with the GIL
// released and the generator locked once.  kinds: 0 masks, 1 negatives, 2 node-contrast views, 3 graph-contrast views (None for a
// domain with fewer than two graphs, tasks.py:241).  Result [task][domain]: () for a domain without graphs, (tensor,) for masks /
// negatives, the ten tensors of draw_views for views, None where the reference skips the pair.
pybind11::list draw_step(std::vector<int> kinds, std::vector<std::tuple<at::Tensor, at::Tensor, at::Tensor, int64_t>> doms, at::Generator gen,
                         PyRandom& negatives) {
    for (auto& dm : doms) {
        check_ptrs(std::get<0>(dm), std::get<1>(dm), std::get<2>(dm));
        TORCH_CHECK(std::get<3>(dm) >= 0 && std::get<3>(dm) <= 64, "hostdraw: attribute masks are 64-bit column sets");
    }
    for (int k : kinds) TORCH_CHECK(k >= 0 && k <= 3, "hostdraw: unknown artefact kind");
    auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
    const size_t T = kinds.size(), D = doms.size();
    std::vector<std::vector<at::Tensor>> res(T * D);
    std::vector<char> none(T * D, 0);
    {
        pybind11::gil_scoped_release nogil;
        std::lock_guard<std::mutex> lock(impl->mutex_);
        for (size_t t = 0; t < T; ++t)
            for (size_t di = 0; di < D; ++di) {
                const at::Tensor &ptr = std::get<0>(doms[di]), &eptr = std::get<1>(doms[di]), &ei = std::get<2>(doms[di]);
                const int64_t B = ptr.numel() - 1;
                auto& r = res[t * D + di];
                if (B <= 0) continue;
                switch (kinds[t]) {
                    case 0: r.push_back(mask_indices_core(ptr, impl)); break;
                    case 1: r.push_back(negatives.negative_edges_core(ptr, eptr, ei)); break;
                    case 2: r = draw_views_core(ptr, eptr, ei, std::get<3>(doms[di]), impl); break;
                    default:
                        if (B >= 2) r = draw_views_core(ptr, eptr, ei, std::get<3>(doms[di]), impl);
                        else none[t * D + di] = 1;
                }
            }
    }
    pybind11::list out;
    for (size_t t = 0; t < T; ++t) {
        pybind11::list row;
        for (size_t di = 0; di < D; ++di) {
            if (none[t * D + di]) { row.append(pybind11::none()); continue; }
            pybind11::tuple tup(res[t * D + di].size());
            for (size_t i = 0; i < res[t * D + di].size(); ++i) tup[i] = pybind11::cast(res[t * D + di][i]);
            row.append(tup);
        }
        out.append(row);
    }
    return out;
}


// The link-prediction scorer's features [hs+hd, hs*hd, |hs-hd|] (src/models/heads.py:57-61) are symmetric in (src, dst), so the
// ordered pairs (i, j) and (j, i) of one label score identically: the engine scores each unordered pair once and carries its
// multiplicity as a weight.  pos / neg: [2, E] local node ids of ONE domain batch, each grouped by graph in batch order (PyG's
// collation and batched_negative_sampling both are); ptr: the batch's node offsets.  Returns pairs [2, K'] (min, max) + offset --
// all positives, then all negatives, each in first-occurrence order -- and signed multiplicities (+count positives, -count
// negatives).  Pairs never cross graphs, so the counting table is one graph's n x n (a few KB: it stays in L1).
std::vector<at::Tensor> merge_mirrored_pairs(at::Tensor pos, at::Tensor neg, at::Tensor ptr, int64_t offset) {
    for (const at::Tensor* t : {&pos, &neg})
        TORCH_CHECK(t->dim() == 2 && t->size(0) == 2 && t->scalar_type() == at::kLong && t->is_contiguous() && t->device().is_cpu(),
                    "hostdraw: pairs must be contiguous CPU int64 [2, E]");
    TORCH_CHECK(ptr.dim() == 1 && ptr.numel() >= 1 && ptr.scalar_type() == at::kLong && ptr.is_contiguous(), "hostdraw: ptr");
    const int64_t cap = pos.size(1) + neg.size(1);
    static thread_local std::vector<int64_t> va, vb;          // scratch that keeps its capacity across calls
    static thread_local std::vector<float> vw;
    static thread_local std::vector<uint16_t> count;
    va.resize((size_t)cap); vb.resize((size_t)cap); vw.resize((size_t)cap);
    int64_t out = 0;
    const char* err = nullptr;
    {
        pybind11::gil_scoped_release nogil;
        int64_t *oa = va.data(), *ob = vb.data();
        float* ow = vw.data();
        const int64_t* p = ptr.data_ptr<int64_t>();
        const int64_t G = ptr.numel() - 1;
        for (int grp = 0; grp < 2 && !err; ++grp) {
            const at::Tensor& t = grp ? neg : pos;
            const int64_t E = t.size(1);
            const int64_t *s = t.data_ptr<int64_t>(), *d = s + E;
            const float sign = grp ? -1.f : 1.f;
            int64_t e = 0;
            for (int64_t gi = 0; gi < G && e < E && !err; ++gi) {
                const int64_t lo = p[gi], hi = p[gi + 1], n = hi - lo;
                int64_t e1 = e;
                while (e1 < E && s[e1] >= lo && s[e1] < hi) ++e1;             // this graph's run of pairs
                if (e1 == e) continue;
                if (n > 4096) { err = "hostdraw: graph too large for the pair table"; break; }
                if ((int64_t)count.size() < n * n) count.assign((size_t)(n * n), 0);      // (every touched entry is reset below: all zero between graphs)
                for (int64_t k = e; k < e1; ++k) {
                    const int64_t x = s[k] - lo, y = d[k] - lo;
                    if (y < 0 || y >= n) { err = "hostdraw: pair crosses graphs"; break; }
                    uint16_t& c = count[(size_t)((x < y ? x : y) * n + (x < y ? y : x))];
                    if (c == 65535) { err = "hostdraw: pair multiplicity overflow"; break; }
                    ++c;
                }
                if (err) break;
                for (int64_t k = e; k < e1; ++k) {
                    const int64_t x = s[k] - lo, y = d[k] - lo, a = x < y ? x : y, b = x < y ? y : x;
                    uint16_t& c = count[(size_t)(a * n + b)];
                    if (c) {
                        oa[out] = a + lo + offset; ob[out] = b + lo + offset; ow[out] = sign * (float)c;
                        ++out;
                        c = 0;
                    }
                }
                e = e1;
            }
            if (!err && e != E) err = "hostdraw: pairs are not grouped by graph in batch order";
        }
    }
    if (err) {
        std::fill(count.begin(), count.end(), 0);              // (a failed call may have left counts behind)
        TORCH_CHECK(false, err);
    }
    at::Tensor pairs = at::empty({2, out}, at::kLong), w = at::empty({out}, at::kFloat);
    if (out) {
        std::memcpy(pairs.data_ptr<int64_t>(), va.data(), (size_t)out * sizeof(int64_t));
        std::memcpy(pairs.data_ptr<int64_t>() + out, vb.data(), (size_t)out * sizeof(int64_t));
        std::memcpy(w.data_ptr<float>(), vw.data(), (size_t)out * sizeof(float));
    }
    return {pairs, w};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("mask_indices", &mask_indices, "per-graph randperm(n)[:max(1, int(.15 n))] + offset for n >= 3");
    pybind11::class_<PyRandom>(m, "PyRandom", "CPython-compatible MT19937 stream (random.getstate()[1] words) for PyG's negative sampler")
        .def(pybind11::init<>())
        .def("setstate", &PyRandom::setstate)
        .def("getstate", &PyRandom::getstate)
        .def("negative_edges", &PyRandom::negative_edges, "batched_negative_sampling(to_undirected(pos), batch, num_neg_samples=E) of a domain batch");
    m.def("merge_mirrored_pairs", &merge_mirrored_pairs, "unordered pairs + signed multiplicities of a domain's positive and negative pairs");
    m.def("draw_step", &draw_step, "all index artefacts of one step in the reference's order, one call");
    m.def("draw_views", &draw_views, "two augmented views of every graph of a batch, as index arrays");
}
