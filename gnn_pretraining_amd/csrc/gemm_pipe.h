// The pipelined fp32 MFMA GEMM core (included by gemm_f32.hip; shares its GemmArgs).
//
// What the first kernel (gemm_kernel, 64x64 tiles, register staging, one barrier per 32-deep K-step, 3-4 waves per SIMD)
// left on the table at the backbone's shapes (M ~ 7 k rows, 256 <-> 512; profiles/README.md): every workgroup of the single
// resident wave of blocks runs in lockstep, so the prologue's first loads, the K-step barriers and the 15 MB of epilogue
// stores are all exposed -- 7.3 us of fixed cost on a 27 us launch whose MFMA work is 12.3 us.
//
// This kernel is built the way MI355X wants an MFMA-bound loop at ~1 block per CU (cdna_hip_programming.md section 5):
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), into a ring
//     of STAGES 32-deep K-steps, and stay in flight ACROSS the K-step barrier: the only wait in the loop is a counted
//     s_waitcnt vmcnt(N) that leaves the younger stages outstanding, followed by a raw s_barrier;
//   * a wave owns a (TM*32) x (TN*32) output tile as TM x TN accumulators of v_mfma_f32_32x32x2_f32 (exact fp32);
//   * a k-CONTIGUOUS operand (row-major activations, Linear weights in the forward) is staged as [row][32 k] with the
//     eight 16-byte chunks of a row XOR-swizzled by (row >> 1) & 7 -- applied on the SOURCE address, the DMA writes LDS
//     lane-linear -- so that ONE conflict-free ds_read_b128 per lane yields the fragments of FOUR MFMAs: lanes 0-31 take
//     k-quad 2g, lanes 32-63 k-quad 2g+1 of their row, register j of both operands then holds k = 8g + 4*half + j, which
//     is all the instruction needs (any k pairing is a valid dot product as long as A and B agree);
//   * a k-MAJOR operand (weights in the input-gradient GEMM, both operands of the weight-gradient GEMM) is staged as
//     [32 k][R] and read with conflict-free ds_read_b32 at the same k assignment;
//   * the weight-gradient form reduces over an arbitrary row range: its last, partial K-step loads clamped rows and zeroes
//     the fragments of k >= range in registers (one extra copy of the step's code, nothing in the main loop).
// Per output element the accumulation order (K-steps ascending; inside a step g = 0..3, j = 0..3, half 0 before half 1 as the
// MFMA's own k order) is the same for every tile shape, so the grouped / sliced forms stay bitwise consistent with each other.
#pragma once

namespace g2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BK = 32;
constexpr int THREADS = 256;
#ifndef GMP_PIPE_SCHED
#define GMP_PIPE_SCHED 1
#endif


// One LDS-DMA instruction: 64 lanes x 16 bytes from each lane's own global address to LDS at `lds_addr` + 16 * lane.
// Inline asm on purpose: hipcc waits s_waitcnt vmcnt(0) before the next ds_read of an array a __builtin_amdgcn_global_load_lds
// may have written (it cannot see that the ring's stages are disjoint), which drains the whole pipeline every K-step; an asm
// load is outside its bookkeeping, and the waits below are counted by hand.  M0 (the LDS base of the DMA) is compiler-reserved:
// saved and restored inside the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const float* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_address(const char* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)p);
}

// ---- one operand tile of a stage: R rows (or columns) x 32 k ---------------------------------------------------------
template <int R, bool KC>
struct Operand {
    static constexpr int BYTES = R * BK * 4;
    static constexpr int INSTR = BYTES / 1024;        // LDS-DMA wave-instructions per stage (1 KiB each)
    static constexpr int PER_WAVE = INSTR / 4;        // issued by each of the 4 waves
    static constexpr int LPR = R / 4;                 // k-major: lanes per k-row (16-byte pieces of R floats)

    const float* p[PER_WAVE];                         // this lane's source pointer per instruction, at the current K-step
    int64_t step;                                     // floats to advance per K-step

    // r0: first row (column) of the tile, rtot: valid extent (rows beyond are clamped to the last valid one: their products
    // land in output rows / columns the epilogue never stores), k0: first k of the block's range
    // wrap: rows beyond the extent are spread over valid rows (r0 + row mod extent) instead of all reading the last one -- the
    // segment kernels pad up to half a tile, and a hundred lanes asking for one cache line serialise in the L2
    __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t rtot, int64_t k0, int lane, int wave,
                                         bool wrap = false) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int ii = wave + 4 * i;
            if (KC) {
                const int row = ii * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                int64_t gr = r0 + row;
                gr = gr < rtot ? gr : (wrap ? r0 + row % (int)(rtot - r0) : rtot - 1);
                p[i] = P + gr * ld + k0 + 4 * chunk;
            } else {
                const int k = ii * (64 / LPR) + lane / LPR, q = lane % LPR;
                int64_t gr = r0 + 4 * q;
                gr = gr + 3 < rtot ? gr : rtot - 4;
                p[i] = P + (k0 + k) * ld + gr;
            }
        }
        step = KC ? BK : BK * ld;
    }

    __device__ __forceinline__ void issue(unsigned stage_addr, int wave) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            glds16(p[i], stage_addr + (wave + 4 * i) * 1024);
            p[i] += step;
        }
    }

    // k-major only: the partial last K-step of a reduction range that ends at kend -- rows at or beyond it are read from the
    // last valid row (in bounds) and zeroed in the fragments
    __device__ __forceinline__ void issue_tail(unsigned stage_addr, int wave, int lane, int64_t klen, int64_t ld) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int k = (wave + 4 * i) * (64 / LPR) + lane / LPR;
            const int64_t back = k < klen ? 0 : (int64_t)(k - (klen - 1)) * ld;
            glds16(p[i] - back, stage_addr + (wave + 4 * i) * 1024);
        }
    }
};

// fragments of sub-step g (eight k) for one 32-row block: f[j] is the operand register of MFMA j
template <int R, bool KC>
__device__ __forceinline__ void read_frag(const char* tile, int rb, int g, int l31, int half, float (&f)[4]) {
    if (KC) {
        const int r = rb + l31;
        const float4 v = *reinterpret_cast<const float4*>(tile + r * 128 + ((((2 * g + half) ^ ((r >> 1) & 7))) << 4));
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = *reinterpret_cast<const float*>(tile + ((8 * g + 4 * half + j) * R + rb + l31) * 4);
    }
}

// STAGES is the SCHEDULE: K-step s + STAGES - 1 is issued at the hand-over of step s.  BUFS is how many LDS buffers back it: STAGES (the DMA
// of step s + STAGES - 1 goes to the buffer of step s - 1), or STAGES - 1 ("early free"): at the hand-over every fragment of stage s is in
// registers already (its last sub-step's reads were issued a sub-step earlier), so after the wave's own lgkmcnt(0) and the barrier the
// buffer of stage s ITSELF can take the new DMA.  One buffer less per block for the same prefetch distance: a 64 x 64 tile with a
// 3-stage schedule needs 32 KB instead of 48, and FOUR blocks share a CU -- the 928 tiles of a layer GEMM are resident at once, one
// round instead of 1.8, one exposed prologue / epilogue instead of two.
template <int TM, int TN, bool A_KC, bool B_KC, int STAGES, int BUFS = STAGES>
struct Cfg {
    static_assert(BUFS == STAGES || BUFS == STAGES - 1, "buffers: one per stage of the schedule, or one less (early free)");
    static_assert(BUFS >= 2, "at least two buffers");
    static constexpr int BM = 64 * TM, BN = 64 * TN;
    using OA = Operand<BM, A_KC>;
    using OB = Operand<BN, B_KC>;
    static constexpr int STAGE_BYTES = OA::BYTES + OB::BYTES;
    static constexpr int LDS_BYTES = BUFS * STAGE_BYTES;
    static constexpr int LOADS = OA::PER_WAVE + OB::PER_WAVE;      // LDS-DMA instructions per wave per stage
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Fragments of sub-step g of stage `st` for this wave's TM x TN blocks of 32 rows / columns.
template <int TM, int TN, bool A_KC, bool B_KC, int STAGES>
__device__ __forceinline__ void read_sub(const char* st, int g, int wm, int wn, int l31, int half, float (&fa)[TM][4], float (&fb)[TN][4]) {
    using C = Cfg<TM, TN, A_KC, B_KC, STAGES>;          // (only tile geometry is used here: the same for every BUFS)
#pragma unroll
    for (int i = 0; i < TM; ++i) read_frag<C::BM, A_KC>(st, wm * (C::BM / 2) + 32 * i, g, l31, half, fa[i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) read_frag<C::BN, B_KC>(st + C::OA::BYTES, wn * (C::BN / 2) + 32 * j, g, l31, half, fb[j]);
}

// The 4 * TM * TN MFMAs of sub-step g.  TAIL: zero the operands of k >= klen (last, partial K-step of the weight-gradient form).
template <int TM, int TN, bool TAIL>
__device__ __forceinline__ void mfma_sub(f32x16 (&acc)[TM][TN], const float (&fa)[TM][4], const float (&fb)[TN][4], int g, int half, int klen) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
        const bool dead = TAIL && (8 * g + 4 * half + j4 >= klen);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float a = dead ? 0.f : fa[i][j4];
                const float b = dead ? 0.f : fb[j][j4];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
            }
    }
}

// order hint for one sub-step: the first MFMA, then the LDS reads of the NEXT sub-step (they issue while the matrix pipe runs and
// have 15 MFMAs = ~960 cycles to land), then the rest -- left alone hipcc issues the reads behind the MFMAs they follow in the
// source and waits for them with the pipe idle (one wave per SIMD: nobody else fills it)
template <int NREADS, int NMFMA>
__device__ __forceinline__ void sched_sub() {
#if GMP_PIPE_SCHED == 1
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - 1, 0);
#endif
}

template <int TM, int TN, bool A_KC, bool B_KC, int STAGES, int BUFS = STAGES>
__global__ __launch_bounds__(THREADS) void gemm_pipe_kernel(const GemmArgs g, int tiles_m, int tiles_n) {
    using C = Cfg<TM, TN, A_KC, B_KC, STAGES, BUFS>;
    constexpr bool EARLY = BUFS < STAGES;
    constexpr bool IS_TN = !A_KC && !B_KC;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    if (g.sig_flag && t == 0 && blockIdx.x == 0 && blockIdx.z == 0)
        __hip_atomic_store(g.sig_flag, g.sig_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);

    // blockIdx.x -> (row tile, column tile): the column tiles of one row tile sit on one XCD (blocks b and b + 8 share one), so
    // a row panel of A enters one L2; bijective for any grid size
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    int64_t m0 = (int64_t)tile_m * C::BM;
    const int64_t n0 = (int64_t)tile_n * C::BN;

    const float* __restrict__ Bp = g.B;
    const float* __restrict__ biasp = g.bias;
    float* Cp = g.C;
    int64_t Mrows = g.M;
    int64_t kbeg = 0, kend = g.K;
    if (g.groups > 0) {
        const int grp = IS_TN ? blockIdx.z / g.gsplit : blockIdx.z;
        if (IS_TN) {                               // reduction range = the group's rows (or one slice of them)
            kbeg = g.grow[grp];
            kend = g.grow[grp + 1];
            if (g.gsplit > 1) {
                const int64_t steps = (kend - kbeg + BK - 1) / BK, per = (steps + g.gsplit - 1) / g.gsplit;
                kbeg += (int64_t)(blockIdx.z % g.gsplit) * per * BK;
                kend = kbeg + per * BK < kend ? kbeg + per * BK : kend;
                Cp = g.gpart + (int64_t)blockIdx.z * g.M * g.N;
            } else {
                Cp += g.coff[grp];
            }
        } else {                                    // NT / NN: row range of A and C, per-group B (and bias)
            m0 += g.grow[grp];
            Mrows = g.grow[grp + 1];
            if (m0 >= Mrows) return;
            Bp += g.boff[grp];
            if (biasp) biasp += g.biasoff[grp];
        }
    }
    // split-K of the NT / NN forms (few output tiles, long K: the fine-tune encoder's 2,708 x 1,440 -> 256): slice z takes a run of whole
    // K-steps and leaves its partial tile in g.partial[z] ([M, N] each); splitk_reduce_kernel adds the slices in order and applies the epilogue
    const bool ksplit = !IS_TN && g.groups == 0 && g.splitk > 1;
    if (ksplit) {
        const int64_t steps = g.K / BK, per = (steps + g.splitk - 1) / g.splitk;
        kbeg = (int64_t)blockIdx.z * per * BK;
        kend = kbeg + per * BK < g.K ? kbeg + per * BK : g.K;
        Cp = g.partial + (int64_t)blockIdx.z * g.M * g.N;
    }
    const int64_t klen_total = kend > kbeg ? kend - kbeg : 0;
    const int nfull = (int)(klen_total / BK);
    const int ktail = (int)(klen_total - (int64_t)nfull * BK);       // > 0 only in the weight-gradient form
    const int nsteps = nfull + (ktail ? 1 : 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename C::OA la;
    typename C::OB lb;
    if (nsteps > 0) {
        la.init(g.A, g.lda, m0, Mrows, kbeg, lane, wave);
        lb.init(Bp, g.ldb, n0, g.N, kbeg, lane, wave);
    }
    auto stage_ptr = [&](int s) -> char* { return smem + (s % BUFS) * C::STAGE_BYTES; };
    const unsigned lds0 = lds_address(smem);
    auto issue = [&](int s) {          // stage of K-step s; wave-uniform control flow
        const unsigned sp = lds0 + (s % BUFS) * C::STAGE_BYTES;
        if (IS_TN && ktail && s == nfull) {
            la.issue_tail(sp, wave, lane, ktail, g.lda);
            lb.issue_tail(sp + C::OA::BYTES, wave, lane, ktail, g.ldb);
        } else {
            la.issue(sp, wave);
            lb.issue(sp + C::OA::BYTES, wave);
        }
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) issue(s);

    // bias-gradient rider of the weight-gradient form: column sums of the A tile, read back out of LDS (blocks of column tile 0)
    const bool want_asum = IS_TN && g.groups > 0 && g.asum != nullptr && tile_n == 0;
    constexpr int QUADS = C::BM / 4, KSL = THREADS / QUADS, KPER = BK / KSL;
    float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- main loop, one K-step per iteration, rotated by a sub-step ---------------------------------------------------------
    // Stage s is consumed as four sub-steps of 8 k.  The hand-over to stage s + 1 -- counted wait for its DMA, the block's one
    // barrier per K-step, the issue of stage s + 3, the first fragment reads of stage s + 1 -- sits BEFORE the last sub-step's
    // MFMAs (whose operands are in registers by then), so the barrier skew, the DMA issue and the LDS latency of the hand-over
    // are covered by 4 * TM * TN MFMAs instead of standing between two K-steps.  Hazards: the DMA issued after the barrier of
    // step s fills the buffer of stage s - 1, which every wave finished reading before it reached that barrier; stage s + 1 is
    // read only after this wave's counted vmcnt (its own part has landed) AND the barrier (everyone's has).
    float fa0[TM][4], fa1[TM][4], fb0[TN][4], fb1[TN][4];
    constexpr int NREADS = (A_KC ? TM : 4 * TM) + (B_KC ? TN : 4 * TN), NMFMA = 4 * TM * TN;
    if (nsteps > 0) {          // stage 0: the prologue's younger stages (at most STAGES - 2 of them) may stay in flight
        if (STAGES >= 4 && nsteps >= 3) wait_vmcnt<2 * C::LOADS>();
        else if (STAGES >= 3 && nsteps >= 2) wait_vmcnt<C::LOADS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_sub<TM, TN, A_KC, B_KC, STAGES>(stage_ptr(0), 0, wm, wn, l31, half, fa0, fb0);
    }
    auto step = [&](int s, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value;
        const char* st = stage_ptr(s);
        const int klen = TAIL ? ktail : BK;
        if (want_asum) {
            const int mq = t % QUADS, ks = t / QUADS;
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk) {
                const int k = ks * KPER + kk;
                if (!TAIL || k < klen) {
                    const float4 v = *reinterpret_cast<const float4*>(st + (k * C::BM + 4 * mq) * 4);
                    colacc.x += v.x; colacc.y += v.y; colacc.z += v.z; colacc.w += v.w;
                }
            }
        }
        read_sub<TM, TN, A_KC, B_KC, STAGES>(st, 1, wm, wn, l31, half, fa1, fb1);
        mfma_sub<TM, TN, TAIL>(acc, fa0, fb0, 0, half, klen);
        sched_sub<NREADS, NMFMA>();
        read_sub<TM, TN, A_KC, B_KC, STAGES>(st, 2, wm, wn, l31, half, fa0, fb0);
        mfma_sub<TM, TN, TAIL>(acc, fa1, fb1, 1, half, klen);
        sched_sub<NREADS, NMFMA>();
        read_sub<TM, TN, A_KC, B_KC, STAGES>(st, 3, wm, wn, l31, half, fa1, fb1);
        mfma_sub<TM, TN, TAIL>(acc, fa0, fb0, 2, half, klen);
        sched_sub<NREADS, NMFMA>();
        if (s + 1 < nsteps) {
            if (STAGES >= 4 && s + 2 < nsteps) wait_vmcnt<C::LOADS>();          // 4-stage schedule: stage s + 2 may stay in flight
            else wait_vmcnt<0>();
            // early free: the DMA issued below lands in THIS stage's buffer -- this wave's reads of it (sub-step 3's were issued behind the
            // first MFMA of sub-step 2) must have returned before any wave may overwrite it; the barrier alone does not wait for them
            if (EARLY) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + STAGES - 1 < nsteps) issue(s + STAGES - 1);
            read_sub<TM, TN, A_KC, B_KC, STAGES>(stage_ptr(s + 1), 0, wm, wn, l31, half, fa0, fb0);
        }
        mfma_sub<TM, TN, TAIL>(acc, fa1, fb1, 3, half, klen);
    };
    for (int s = 0; s < nfull; ++s) step(s, std::false_type{});
    if (ktail) step(nfull, std::true_type{});

    if (want_asum) {       // combine the k-slices in slice order through LDS (all stages are consumed by now)
        __builtin_amdgcn_s_barrier();
        float4* red = reinterpret_cast<float4*>(smem);
        red[t] = colacc;                              // [ks][mq]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t < C::BM && m0 + t < g.M) {
            const float* rf = reinterpret_cast<const float*>(smem);
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) sum += rf[(ks * QUADS + (t >> 2)) * 4 + (t & 3)];
            if (g.gsplit > 1) g.gasum_part[(int64_t)blockIdx.z * g.M + m0 + t] = sum;
            else g.asum[g.asumoff[blockIdx.z] + m0 + t] = sum;
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    // The accumulators hold one COLUMN per lane (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) +
    // 4 * (lane >> 5)): stored as they stand that is 16 * TM * TN dword stores per lane, 256 bytes each, and with every block of
    // the grid reaching its epilogue together the store ISSUE was 9 us of a 30 us launch (K = 256, 128 x 128 tiles; measured with
    // the stores compiled out).  So each wave turns its own sub-tile through LDS (the ring is free by now) and writes whole
    // float4 row pieces: 4 * TM * TN dwordx4 stores per lane, 1 KiB each.  Same arithmetic per element, in the same order.
    float* out = Cp;
    int64_t ldo = g.ldc;
    const bool partial = (IS_TN && g.groups > 0 && g.gsplit > 1) || ksplit;
    if (partial) ldo = g.N;
    constexpr int WR = 32 * TM, WC = 32 * TN;                 // the wave's sub-tile
    const int64_t row0 = m0 + wm * WR, col0 = n0 + wn * WC;
    if (g.vecC || partial) {
        __builtin_amdgcn_s_barrier();                         // every wave has finished reading the last stages
        float* tile = reinterpret_cast<float*>(smem) + wave * (WR * WC);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * WC + j * 32 + l31] = acc[i][j][r];
        constexpr int LR = WC / 4, RPI = 64 / LR;              // lanes per row, rows per wave-instruction
        const int c4 = lane % LR, rsub = lane / LR;
        const int64_t col = col0 + 4 * c4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!partial && biasp) {
            if (col + 3 < g.N) bv = *reinterpret_cast<const float4*>(biasp + col);
            else {
                if (col < g.N) bv.x = biasp[col];
                if (col + 1 < g.N) bv.y = biasp[col + 1];
                if (col + 2 < g.N) bv.z = biasp[col + 2];
            }
        }
#pragma unroll
        for (int q = 0; q < WR / RPI; ++q) {
            const int rl = q * RPI + rsub;
            float4 v = *reinterpret_cast<const float4*>(tile + rl * WC + 4 * c4);
            const int64_t row = row0 + rl;
            if (row >= Mrows || col >= g.N) continue;
            float* o = out + row * ldo + col;
            const bool full = col + 3 < g.N;
            if (!partial) {
                v.x = g.alpha * v.x + bv.x; v.y = g.alpha * v.y + bv.y; v.z = g.alpha * v.z + bv.z; v.w = g.alpha * v.w + bv.w;
                if (g.accumulate) {
                    if (full) {
                        const float4 c = *reinterpret_cast<const float4*>(o);
                        v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
                    } else {
                        v.x += o[0];
                        if (col + 1 < g.N) v.y += o[1];
                        if (col + 2 < g.N) v.z += o[2];
                    }
                }
                if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (full) {
                // (plain stores: with non-temporal ones the kernel alone is 2.5 us shorter -- no end-of-kernel L2 write-back of the 15 MB a
                // layer GEMM writes -- but the consumer then misses L2 and the whole step got slower, 1.62 against 1.46 ms; removed in round 3)
                *reinterpret_cast<float4*>(o) = v;
            }
            else {
                o[0] = v.x;
                if (col + 1 < g.N) o[1] = v.y;
                if (col + 2 < g.N) o[2] = v.z;
            }
        }
        return;
    }
    // output not 16-byte addressable (odd leading dimension / offset): the accumulators as they stand
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = col0 + j * 32 + l31;
            if (col >= g.N) continue;
            const float bv = (!partial && biasp) ? biasp[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= Mrows) continue;
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.accumulate) v += out[row * ldo + col];
                if (g.relu) v = fmaxf(v, 0.f);
                out[row * ldo + col] = v;
            }
        }
}

}  // namespace g2
