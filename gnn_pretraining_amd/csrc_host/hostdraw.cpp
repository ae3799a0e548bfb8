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
#include <map>
#include <string>
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
                if (kinds[t] == 3 && B < 2) { none[t * D + di] = 1; continue; }      // no graph-contrast pair without two graphs (also: no graph at all)
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
// shared by merge_mirrored_pairs and plan_step: appends the merged pairs of ONE domain batch to (oa, ob, ow, opa, opb); returns an error text or null.
// A merged row stands for ONE or TWO ordered rows of the reference's list (heads.py:44-52 draws an independent dropout mask for every
// ordered row, so the scorer's tail needs to know which they were): opa[row] = position of the first occurrence in the reference's ordered
// list of the step, opb[row] = of the second (-1: none); positions count from ord_base (this domain's first positive; its negatives follow
// its positives, tasks.py:111-113).  ow = sign * (1 or 2).  A third occurrence of a pair (a duplicated edge) starts a row of its own.
static const char* merge_pairs_core(const int64_t* ps, const int64_t* pd, int64_t Ep, const int64_t* ns, const int64_t* nd, int64_t En,
                                    const int64_t* p, int64_t G, int64_t offset, int64_t ord_base, std::vector<int64_t>& oa, std::vector<int64_t>& ob,
                                    std::vector<float>& ow, std::vector<int32_t>& opa, std::vector<int32_t>& opb, std::vector<int32_t>& slot) {
    size_t out = oa.size();
    const size_t cap = out + (size_t)(Ep + En) + 1;           // (+1: the branch-free loop below writes one row past the last emitted one)
    oa.resize(cap); ob.resize(cap); ow.resize(cap); opa.resize(cap); opb.resize(cap);
    int64_t *wa = oa.data(), *wb = ob.data();
    float* ww = ow.data();
    int32_t *pa = opa.data(), *pb = opb.data();
    const char* err = nullptr;
    if (ord_base + Ep + En > INT32_MAX) err = "hostdraw: ordered pair list too long for int32 positions";
    for (int grp = 0; grp < 2 && !err; ++grp) {
        const int64_t E = grp ? En : Ep;
        const int64_t *s = grp ? ns : ps, *d = grp ? nd : pd;
        const float sign = grp ? -1.f : 1.f;
        const int64_t ord0 = ord_base + (grp ? Ep : 0);
        int64_t e = 0;
        for (int64_t gi = 0; gi < G && e < E && !err; ++gi) {
            const int64_t lo = p[gi], hi = p[gi + 1], n = hi - lo;
            int64_t e1 = e;
            while (e1 < E && s[e1] >= lo && s[e1] < hi) ++e1;             // this graph's run of pairs
            if (e1 == e) continue;
            if (n > 4096) { err = "hostdraw: graph too large for the pair table"; break; }
            if ((int64_t)slot.size() < n * n) slot.assign((size_t)(n * n), 0);      // (every touched entry is reset below: all zero between graphs)
            for (int64_t k = e; k < e1; ++k) {
                const int64_t y = d[k] - lo;
                if (y < 0 || y >= n) { err = "hostdraw: pair crosses graphs"; break; }
            }
            if (err) break;
            // slot[a * n + b] = 1 + the row that holds ONE occurrence of the pair so far (0: none, or its row is full).  Branch-free: half of
            // these tests go either way with no pattern (fresh pairs every step), a mispredict each
            for (int64_t k = e; k < e1; ++k) {
                const int64_t x = s[k] - lo, y = d[k] - lo, a = x < y ? x : y, b = x < y ? y : x;
                int32_t& sl = slot[(size_t)(a * n + b)];
                const bool fresh = sl == 0;
                const size_t row = fresh ? out : (size_t)(sl - 1);
                const int32_t ord = (int32_t)(ord0 + k);
                wa[row] = a + lo + offset; wb[row] = b + lo + offset;      // (the same values when the row exists already)
                pa[row] = fresh ? ord : pa[row];
                pb[row] = fresh ? -1 : ord;
                ww[row] = fresh ? sign : 2.f * sign;
                sl = fresh ? (int32_t)(out + 1) : 0;
                out += fresh;
            }
            for (int64_t k = e; k < e1; ++k) {
                const int64_t x = s[k] - lo, y = d[k] - lo;
                slot[(size_t)((x < y ? x : y) * n + (x < y ? y : x))] = 0;
            }
            e = e1;
        }
        if (!err && e != E) err = "hostdraw: pairs are not grouped by graph in batch order";
    }
    if (err) std::fill(slot.begin(), slot.end(), 0);            // (a failed call may have left entries behind)
    oa.resize(out); ob.resize(out); ow.resize(out); opa.resize(out); opb.resize(out);
    return err;
}

std::vector<at::Tensor> merge_mirrored_pairs(at::Tensor pos, at::Tensor neg, at::Tensor ptr, int64_t offset, int64_t ord_base) {
    for (const at::Tensor* t : {&pos, &neg})
        TORCH_CHECK(t->dim() == 2 && t->size(0) == 2 && t->scalar_type() == at::kLong && t->is_contiguous() && t->device().is_cpu(),
                    "hostdraw: pairs must be contiguous CPU int64 [2, E]");
    TORCH_CHECK(ptr.dim() == 1 && ptr.numel() >= 1 && ptr.scalar_type() == at::kLong && ptr.is_contiguous(), "hostdraw: ptr");
    static thread_local std::vector<int64_t> va, vb;          // scratch that keeps its capacity across calls
    static thread_local std::vector<float> vw;
    static thread_local std::vector<int32_t> vpa, vpb, slot;
    va.clear(); vb.clear(); vw.clear(); vpa.clear(); vpb.clear();
    const char* err;
    {
        pybind11::gil_scoped_release nogil;
        const int64_t Ep = pos.size(1), En = neg.size(1);
        err = merge_pairs_core(pos.data_ptr<int64_t>(), pos.data_ptr<int64_t>() + Ep, Ep, neg.data_ptr<int64_t>(), neg.data_ptr<int64_t>() + En, En,
                               ptr.data_ptr<int64_t>(), ptr.numel() - 1, offset, ord_base, va, vb, vw, vpa, vpb, slot);
    }
    TORCH_CHECK(!err, err);
    const int64_t out = (int64_t)va.size();
    at::Tensor pairs = at::empty({2, out}, at::kLong), w = at::empty({out}, at::kFloat), ord = at::empty({2, out}, at::kInt);
    if (out) {
        std::memcpy(pairs.data_ptr<int64_t>(), va.data(), (size_t)out * sizeof(int64_t));
        std::memcpy(pairs.data_ptr<int64_t>() + out, vb.data(), (size_t)out * sizeof(int64_t));
        std::memcpy(w.data_ptr<float>(), vw.data(), (size_t)out * sizeof(float));
        std::memcpy(ord.data_ptr<int32_t>(), vpa.data(), (size_t)out * sizeof(int32_t));
        std::memcpy(ord.data_ptr<int32_t>() + out, vpb.data(), (size_t)out * sizeof(int32_t));
    }
    return {pairs, w, ord};
}

// engine.StepEngine.plan in one call with the GIL released: the stacked layout of a step -- segments (one per reference forward()
// call), the per-task index arrays, the link-prediction pair list with mirrored pairs merged, the tiles, and the two upload images
// (int32 / int64, every array 16-byte aligned) -- from the raw result of draw_step.  Array for array what the Python plan builds
// (tests/test_hostdraw.py compares them); the Python version stays for the other index modes and as the reading copy.
// kinds: engine task order (0 node_feat_mask, 1 link_pred, 2 node_contrast, 3 graph_contrast, 4 graph_prop, 5 domain_adv);
// art[t][d] for kinds 0..3 in the same order: the tuples draw_step returns (None where the reference skips the pair).
namespace {
struct Img {                                   // one named array of an upload image
    std::string name;
    std::vector<int64_t> v;
};
struct PlanOut {
    std::vector<Img> a32, a64;
    std::vector<int64_t>& add(std::vector<Img>& img, const char* name) {
        img.push_back(Img{name, {}});
        return img.back().v;
    }
};
}  // namespace

pybind11::dict plan_step(std::vector<int> kinds, std::vector<std::tuple<at::Tensor, at::Tensor, at::Tensor, int64_t>> doms,
                         std::vector<int64_t> row_off, pybind11::list art, bool lp_merge, int64_t fwd_ranges, int64_t hidden, int64_t prop_dim) {
    const size_t T = kinds.size(), D = doms.size();
    TORCH_CHECK(row_off.size() == D, "plan_step: one row offset per domain");
    // ---- unpack the artefacts while the GIL is held (tensor handles only)
    struct Art { bool none = false; std::vector<at::Tensor> t; };
    std::vector<std::vector<Art>> arts(T, std::vector<Art>(D));
    {
        size_t ai = 0;
        for (size_t t = 0; t < T; ++t) {
            if (kinds[t] > 3) continue;
            TORCH_CHECK(ai < (size_t)pybind11::len(art), "plan_step: artefact list too short");
            pybind11::list row = art[ai++].cast<pybind11::list>();
            TORCH_CHECK((size_t)pybind11::len(row) == D, "plan_step: one artefact per domain");
            for (size_t d = 0; d < D; ++d) {
                pybind11::object o = row[d];
                if (o.is_none()) { arts[t][d].none = true; continue; }
                for (auto h : o.cast<pybind11::tuple>()) {
                    at::Tensor x = h.cast<at::Tensor>();
                    TORCH_CHECK(x.scalar_type() == at::kLong && x.is_contiguous() && x.device().is_cpu(), "plan_step: artefacts are contiguous CPU int64 tensors");
                    arts[t][d].t.push_back(x);
                }
            }
        }
    }
    for (auto& dm : doms) check_ptrs(std::get<0>(dm), std::get<1>(dm), std::get<2>(dm));

    PlanOut out;
    std::vector<int64_t> seg_ptr{0}, seg_dom, seg_task, task_row{0}, src_rows, e_src, e_dst, seg_edges;
    std::vector<std::pair<int64_t, const int64_t*>> rowmask_at;       // (first row, per-row masks) of the views that drew one
    std::vector<int64_t> rowmask_len;
    std::vector<std::pair<int64_t, int64_t>> skipped;                 // (task, domain)
    std::vector<float> lp_labels;
    std::map<std::string, int64_t> sc;                                // scalars
    std::map<std::string, std::vector<int64_t>> lists;
    std::vector<int64_t> sizes(T, 0);
    const char* err = nullptr;
    {
        pybind11::gil_scoped_release nogil;
        {
            int64_t nn = 0, ee = 0;
            for (auto& dm : doms) { nn += std::get<0>(dm).data_ptr<int64_t>()[std::get<0>(dm).numel() - 1]; ee += std::get<2>(dm).size(1); }
            src_rows.reserve((size_t)(nn * (T + 2))); e_src.reserve((size_t)(ee * (T + 2))); e_dst.reserve((size_t)(ee * (T + 2)));
        }
        auto n_of = [&](size_t d) { const at::Tensor& p = std::get<0>(doms[d]); return p.data_ptr<int64_t>()[p.numel() - 1]; };
        auto add_segment = [&](int64_t ti, int64_t di, const int64_t* rows, int64_t nrows, int64_t roff, const int64_t* es, const int64_t* ed, int64_t ne) {
            const int64_t r0 = seg_ptr.back();
            {   // bulk appends (resize + a plain loop the compiler vectorises; element-wise push_back made this call as slow as numpy)
                const size_t o = src_rows.size();
                src_rows.resize(o + (size_t)nrows);
                int64_t* w = src_rows.data() + o;
                if (rows) for (int64_t i = 0; i < nrows; ++i) w[i] = rows[i] + roff;
                else for (int64_t i = 0; i < nrows; ++i) w[i] = roff + i;
                const size_t oe = e_src.size();
                e_src.resize(oe + (size_t)ne); e_dst.resize(oe + (size_t)ne);
                int64_t *ws = e_src.data() + oe, *wd = e_dst.data() + oe;
                for (int64_t e = 0; e < ne; ++e) ws[e] = es[e] + r0;
                for (int64_t e = 0; e < ne; ++e) wd[e] = ed[e] + r0;
            }
            seg_edges.push_back(ne);
            seg_ptr.push_back(r0 + nrows); seg_dom.push_back(di); seg_task.push_back(ti);
            return r0;
        };
        std::vector<int64_t> r0s;
        for (size_t t = 0; t < T && !err; ++t) {
            const int kind = kinds[t];
            if (kind == 0 || kind == 1 || kind == 4 || kind == 5) {
                r0s.clear();
                for (size_t d = 0; d < D; ++d) {
                    const at::Tensor& ei = std::get<2>(doms[d]);
                    const int64_t E = ei.size(1);
                    r0s.push_back(add_segment((int64_t)t, (int64_t)d, nullptr, n_of(d), row_off[d], ei.data_ptr<int64_t>(), ei.data_ptr<int64_t>() + E, E));
                }
                if (kind == 0) {
                    auto& idx = out.add(out.a64, "nfm_idx");
                    auto& rows = lists["nfm_rows"];
                    rows.push_back(0);
                    for (size_t d = 0; d < D; ++d) {
                        int64_t m = 0;
                        if (!arts[t][d].t.empty()) {
                            const at::Tensor& x = arts[t][d].t[0];
                            m = x.numel();
                            for (int64_t i = 0; i < m; ++i) idx.push_back(x.data_ptr<int64_t>()[i] + r0s[d]);
                        }
                        if (m == 0) skipped.emplace_back((int64_t)t, (int64_t)d);
                        rows.push_back(rows.back() + m);
                    }
                    sizes[t] = rows.back() * hidden;
                } else if (kind == 1) {
                    std::vector<int64_t> pa, pb, seg_eptr{0};
                    std::vector<int32_t> ord_a, ord_b;               // merged rows: the one or two ordered positions each stands for
                    int64_t ordered = 0;
                    for (size_t d = 0; d < D && !err; ++d) {
                        const at::Tensor& ptr = std::get<0>(doms[d]);
                        const at::Tensor& ei = std::get<2>(doms[d]);
                        const int64_t Ep = ei.size(1);
                        const int64_t *ps = ei.data_ptr<int64_t>(), *pd = ps + Ep;
                        const int64_t *ns = nullptr, *nd = nullptr;
                        int64_t En = 0;
                        if (!arts[t][d].t.empty()) {
                            const at::Tensor& x = arts[t][d].t[0];
                            if (x.dim() != 2 || x.size(0) != 2) { err = "plan_step: negatives must be [2, K]"; break; }
                            En = x.size(1); ns = x.data_ptr<int64_t>(); nd = ns + En;
                        }
                        const int64_t ord_base = ordered;
                        ordered += Ep + En;
                        const int64_t before = (int64_t)pa.size();
                        if (!lp_merge) {
                            for (int64_t e = 0; e < Ep; ++e) { pa.push_back(ps[e] + r0s[d]); pb.push_back(pd[e] + r0s[d]); lp_labels.push_back(1.f); }
                            for (int64_t e = 0; e < En; ++e) { pa.push_back(ns[e] + r0s[d]); pb.push_back(nd[e] + r0s[d]); lp_labels.push_back(-1.f); }
                        } else {
                            static thread_local std::vector<int32_t> slot;
                            err = merge_pairs_core(ps, pd, Ep, ns, nd, En, ptr.data_ptr<int64_t>(), ptr.numel() - 1, r0s[d], ord_base, pa, pb, lp_labels,
                                                   ord_a, ord_b, slot);
                            if (err) break;
                        }
                        seg_eptr.push_back(seg_eptr.back() + (int64_t)pa.size() - before);
                    }
                    if (err) break;
                    auto& le = out.add(out.a64, "lp_edges");
                    le = pa; le.insert(le.end(), pb.begin(), pb.end());
                    sc["lp_K"] = (int64_t)pa.size();
                    sizes[t] = ordered;
                    auto& lsp = out.add(out.a32, "lp_seg_ptr");
                    lsp = r0s; lsp.push_back(seg_ptr.back());
                    out.add(out.a32, "lp_seg_eptr") = seg_eptr;
                    if (lp_merge) {
                        auto& lpos = out.add(out.a32, "lp_pos");
                        lpos.assign(ord_a.begin(), ord_a.end());
                        lpos.insert(lpos.end(), ord_b.begin(), ord_b.end());
                    }
                    sc["lp_S"] = (int64_t)D; sc["lp_rows_end"] = seg_ptr.back();
                    int64_t mr = 0, me = 0;
                    for (size_t d = 0; d < D; ++d) { mr = std::max(mr, lsp[d + 1] - lsp[d]); me = std::max(me, seg_eptr[d + 1] - seg_eptr[d]); }
                    sc["lp_max_rows"] = mr; sc["lp_max_edges"] = me;
                } else {
                    std::vector<int64_t> starts, rows{0}, labels;
                    for (size_t d = 0; d < D; ++d) {
                        const at::Tensor& ptr = std::get<0>(doms[d]);
                        const int64_t* p = ptr.data_ptr<int64_t>();
                        const int64_t B = ptr.numel() - 1;
                        for (int64_t g = 0; g < B; ++g) { starts.push_back(r0s[d] + p[g]); labels.push_back((int64_t)d); }
                        rows.push_back(rows.back() + B);
                    }
                    const int64_t end = seg_ptr.back(), B = (int64_t)starts.size();
                    auto& pv = out.add(out.a32, kind == 4 ? "gp_ptr" : "da_ptr");
                    pv = starts; pv.push_back(end);
                    auto& gid = out.add(out.a64, kind == 4 ? "gp_gid" : "da_gid");
                    gid.resize((size_t)(pv[B] - pv[0]));
                    for (int64_t g = 0; g < B; ++g) std::fill(gid.begin() + (pv[g] - pv[0]), gid.begin() + (pv[g + 1] - pv[0]), g);
                    if (kind == 4) {
                        lists["gp_rows"] = rows; sc["gp_B"] = B; sc["gp_r0"] = task_row.back(); sc["gp_M"] = end - task_row.back();
                        sizes[t] = rows.back() * prop_dim;
                    } else {
                        out.add(out.a64, "da_labels") = labels;
                        sc["da_B"] = B; sc["da_r0"] = task_row.back(); sc["da_M"] = end - task_row.back();
                        sizes[t] = B;
                    }
                }
            } else {
                std::vector<int64_t> idx, rows{0}, ns, starts;
                for (size_t d = 0; d < D; ++d) {
                    const Art& a = arts[t][d];
                    if (a.none) { ns.push_back(0); rows.push_back(rows.back()); skipped.emplace_back((int64_t)t, (int64_t)d); continue; }
                    int64_t r0v[2];
                    for (int v = 0; v < 2; ++v) {
                        if (a.t.empty()) { r0v[v] = add_segment((int64_t)t, (int64_t)d, nullptr, 0, row_off[d], nullptr, nullptr, 0); continue; }
                        const at::Tensor &vr = a.t[5 * v], &ve = a.t[5 * v + 1], &vm = a.t[5 * v + 3];
                        const int64_t ne = ve.size(1);
                        r0v[v] = add_segment((int64_t)t, (int64_t)d, vr.data_ptr<int64_t>(), vr.numel(), row_off[d], ve.data_ptr<int64_t>(), ve.data_ptr<int64_t>() + ne, ne);
                        if (vm.numel()) { rowmask_at.emplace_back(r0v[v], vm.data_ptr<int64_t>()); rowmask_len.push_back(vm.numel()); }
                    }
                    if (kind == 2) {
                        int64_t n = 0;
                        if (!a.t.empty()) {
                            const at::Tensor &c1 = a.t[4], &c2 = a.t[9];
                            n = (c1.numel() >= 2 && c2.numel() >= 2) ? c1.numel() : 0;
                            if (n) {
                                for (int64_t i = 0; i < c1.numel(); ++i) idx.push_back(c1.data_ptr<int64_t>()[i] + r0v[0]);
                                for (int64_t i = 0; i < c2.numel(); ++i) idx.push_back(c2.data_ptr<int64_t>()[i] + r0v[1]);
                            }
                        }
                        ns.push_back(n);
                        if (!n) skipped.emplace_back((int64_t)t, (int64_t)d);
                        rows.push_back(rows.back() + 2 * n);
                    } else {
                        int64_t B = 0;
                        for (int v = 0; v < 2 && !a.t.empty(); ++v) {
                            const at::Tensor& vp = a.t[5 * v + 2];
                            if (v == 0) B = vp.numel() - 1;
                            for (int64_t g = 0; g + 1 < vp.numel(); ++g) starts.push_back(vp.data_ptr<int64_t>()[g] + r0v[v]);
                        }
                        ns.push_back(B);
                        rows.push_back(rows.back() + 2 * B);
                    }
                }
                if (kind == 2) {
                    out.add(out.a64, "nc_idx") = idx;
                    lists["nc_rows"] = rows; lists["nc_n"] = ns;
                } else {
                    const int64_t B = (int64_t)starts.size();
                    auto& pv = out.add(out.a32, "gc_ptr");
                    pv = starts; pv.push_back(seg_ptr.back());
                    auto& gid = out.add(out.a64, "gc_gid");
                    if (B) gid.resize((size_t)(pv[B] - pv[0]));
                    for (int64_t g = 0; g < B; ++g) std::fill(gid.begin() + (pv[g] - pv[0]), gid.begin() + (pv[g + 1] - pv[0]), g);
                    lists["gc_rows"] = rows; lists["gc_n"] = ns;
                    sc["gc_B"] = B; sc["gc_r0"] = task_row.back(); sc["gc_M"] = seg_ptr.back() - task_row.back();
                }
                sizes[t] = rows.back();
            }
            task_row.push_back(seg_ptr.back());
        }
        if (!err) {
            const int64_t N = seg_ptr.back(), S = (int64_t)seg_dom.size();
            sc["N"] = N; sc["S"] = S; sc["E"] = (int64_t)e_src.size();
            int64_t max_seg = 0, max_seg_edges = 0;
            for (int64_t i = 0; i < S; ++i) { max_seg = std::max(max_seg, seg_ptr[i + 1] - seg_ptr[i]); max_seg_edges = std::max(max_seg_edges, seg_edges[i]); }
            sc["max_seg"] = max_seg; sc["max_seg_edges"] = max_seg_edges;
            // the stacked forward runs as `fwd_ranges` row ranges on as many streams (gnnmp_step.h fwd_cut_*): cut k at the segment boundary
            // nearest k N / R, kept only while the cuts ascend strictly inside (0, N)
            std::vector<int64_t> cut_seg, cut_row;
            for (int64_t k = 1; k < fwd_ranges && S > 1; ++k) {
                int64_t best = -1, cut = 0;
                for (int64_t i = 1; i < S; ++i) {
                    const int64_t dist = std::llabs(fwd_ranges * seg_ptr[i] - k * N);
                    if (best < 0 || dist < best) { best = dist; cut = i; }
                }
                const int64_t prev = cut_row.empty() ? 0 : cut_row.back();
                if (seg_ptr[cut] > prev && seg_ptr[cut] < N) { cut_seg.push_back(cut); cut_row.push_back(seg_ptr[cut]); }
            }
            lists["fwd_cut_seg"] = cut_seg; lists["fwd_cut_row"] = cut_row;
            out.add(out.a32, "seg_ptr") = seg_ptr;
            out.add(out.a32, "seg_dom") = seg_dom;
            out.add(out.a32, "src_row") = src_rows;
            auto& sep = out.add(out.a32, "seg_eptr");
            sep.push_back(0);
            for (int64_t i = 0; i < S; ++i) sep.push_back(sep.back() + seg_edges[i]);
            auto& tiles = out.add(out.a32, "tiles");
            int64_t ntiles = 0;
            for (int64_t i = 0; i < S; ++i)
                for (int64_t r = seg_ptr[i]; r < seg_ptr[i + 1]; r += 32) { tiles.push_back(i); tiles.push_back(r); ++ntiles; }
            sc["num_tiles"] = ntiles;
            auto& eall = out.add(out.a64, "edge_index");
            eall = e_src; eall.insert(eall.end(), e_dst.begin(), e_dst.end());
            if (!rowmask_at.empty()) {
                auto& rm = out.add(out.a64, "rowmask");
                rm.assign((size_t)N, 0);
                for (size_t i = 0; i < rowmask_at.size(); ++i)
                    std::memcpy(rm.data() + rowmask_at[i].first, rowmask_at[i].second, (size_t)rowmask_len[i] * sizeof(int64_t));
            }
        }
    }
    TORCH_CHECK(!err, err);
    // ---- the two upload images: every array starts 16-byte aligned (int32: 4 elements, int64: 2)
    pybind11::dict res;
    auto pack = [&](std::vector<Img>& img, bool narrow, const char* key_img, const char* key_lay) {
        const int64_t gran = narrow ? 4 : 2;
        int64_t total = 0;
        for (auto& a : img) total += ((int64_t)a.v.size() + gran - 1) / gran * gran;
        at::Tensor cat = at::zeros({total}, narrow ? at::kInt : at::kLong);
        pybind11::list lay;
        int64_t o = 0;
        for (auto& a : img) {
            if (narrow) { int32_t* p = cat.data_ptr<int32_t>() + o; for (size_t i = 0; i < a.v.size(); ++i) p[i] = (int32_t)a.v[i]; }
            else if (!a.v.empty()) std::memcpy(cat.data_ptr<int64_t>() + o, a.v.data(), a.v.size() * sizeof(int64_t));
            lay.append(pybind11::make_tuple(a.name, o, (int64_t)a.v.size()));
            o += ((int64_t)a.v.size() + gran - 1) / gran * gran;
        }
        res[key_img] = cat;
        res[key_lay] = lay;
    };
    pack(out.a32, true, "cat32", "lay32");
    pack(out.a64, false, "cat64", "lay64");
    at::Tensor lab = at::empty({(int64_t)lp_labels.size()}, at::kFloat);
    if (!lp_labels.empty()) std::memcpy(lab.data_ptr<float>(), lp_labels.data(), lp_labels.size() * sizeof(float));
    res["lp_labels"] = lab;
    res["seg_ptr"] = seg_ptr; res["seg_dom"] = seg_dom; res["seg_task"] = seg_task; res["task_row"] = task_row; res["sizes"] = sizes;
    pybind11::list sk;
    for (auto& x : skipped) sk.append(pybind11::make_tuple(x.first, x.second));
    res["skipped"] = sk;
    for (auto& kv : sc) res[kv.first.c_str()] = kv.second;
    for (auto& kv : lists) res[kv.first.c_str()] = kv.second;
    return res;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("mask_indices", &mask_indices, "per-graph randperm(n)[:max(1, int(.15 n))] + offset for n >= 3");
    pybind11::class_<PyRandom>(m, "PyRandom", "CPython-compatible MT19937 stream (random.getstate()[1] words) for PyG's negative sampler")
        .def(pybind11::init<>())
        .def("setstate", &PyRandom::setstate)
        .def("getstate", &PyRandom::getstate)
        .def("negative_edges", &PyRandom::negative_edges, "batched_negative_sampling(to_undirected(pos), batch, num_neg_samples=E) of a domain batch");
    m.def("merge_mirrored_pairs", &merge_mirrored_pairs, "unordered pairs + signed multiplicities of a domain's positive and negative pairs");
    m.def("plan_step", &plan_step, "the stacked layout and upload images of one step from draw_step's result, GIL released");
    m.def("draw_step", &draw_step, "all index artefacts of one step in the reference's order, one call");
    m.def("draw_views", &draw_views, "two augmented views of every graph of a batch, as index arrays");
}
