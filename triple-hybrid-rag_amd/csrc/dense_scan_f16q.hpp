// Shortlist scan with the QUERIES in registers and the ROWS shared through LDS
// (thr_dense_topk_f16 over the fragment-major float16 copy; the default f16 scan since round 2).
// Two kernels share the pieces below (QAcc / qsx_steps / QEmit: the MFMA shape, the k-loop of a
// half tile, the per-lane segment emit): dense_scan_f16q, described first -- 4-wave blocks, the
// kernel of dim 1024 -- and dense_scan_f16qs further down, the staggered 8-wave block that is
// the default at dim <= 768.
//
// Why (profiles/README.md, round 2): the round-1 kernel (dense_scan_f16p) kept 96 queries in
// LDS and let every wave stream its own rows from L2 -- 8 KiB of row fragments per 24 MFMAs and
// wave, 42.7 B/clk/CU at the full matrix rate against an L1 path of 64 B/clk.  Its waves sat in
// s_waitcnt 44 % of the time and the matrix pipe was 52 % busy.  Here the operand roles are
// swapped, which takes the row loads off the waves' critical path altogether:
//
//   * a wave owns 32 queries for the whole launch and keeps their float16 image as the B operands
//     of all DIM/16 k-steps in registers (192 VGPRs at dim 768) -- loaded once, coalesced, from
//     the fragment-major query image pack_queries_f16 writes;
//   * rows reach LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no LDS store instructions) in
//     HALF tiles (32 rows x DIM/2: 24 KiB at dim 768): the copy is fragment-major, so each 1 KiB
//     piece IS the A operand of one k-step and lands lane-linear; one conflict-free ds_read_b128
//     per MFMA and wave;
//   * a block is 4 waves (one per SIMD, 128 queries) and TWO blocks share a CU (2 x 76 KiB of
//     LDS at dim 768).  The first version of this kernel ran ONE 8-wave block per CU and its phase
//     stamps (profiles/r2_scan_stamps_v1.json) showed why that is wrong: the older wave of a SIMD
//     takes the matrix pipe for its whole k-loop, the younger runs after it, and the block-wide
//     barrier then holds everybody until the last wave has emitted -- the pipe idled 40 % of
//     every tile.  Two independent blocks have independent barriers, so one block's DMA issue,
//     emit and barrier wait run under the other block's MFMAs;
//   * ring of NB half-tile buffers, ONE workgroup barrier per half tile: wait for the own pieces
//     of half-tile j (counted vmcnt: the next half-tile's pieces stay in flight), barrier
//     (everybody's pieces of j have landed, everybody is done reading j-1), then the pieces of
//     j+NB-1 are issued into j-1's buffer BETWEEN the MFMAs of j.
//
// The copy holds the rows NORMALISED (d/||d|| rounded to float16, thr_dense_quantize_f16) with
// NaN for rows without an embedding and for the padding rows of the last tile: an accumulator
// is then the scan score itself (no 1/||d|| gather) and `acc >= tau` is false for every row that
// must not be emitted -- the fast path of the epilogue is one compare + branch per accumulator
// register.  A lane that passes writes (score, row) straight to global memory at its OWN cursor:
// lane (query c, row group g) of the block working on row slice `slice` owns segment
// SEGS * slice + g of query c's candidate area (SEGS = 2 row halves with the 32x32x16 MFMA shape,
// 4 row groups with 16x16x32), so there is no staging, no atomic and no flush -- the 1400 cycles
// per tile the LDS-staged, atomically flushed emit of the first version took (the returning
// atomics also drained the DMA queue).  select_band reads the segments in place
// (cand_cnt[q * nseg + s] entries each).
#pragma once

namespace thr {

constexpr int Q_RING = 4;  // A fragments in flight per wave
constexpr int Q_NW = 4;    // waves per block, 32 queries each

template <int DIM>
struct QScan {
    static constexpr int KS = DIM / 16;             // k-steps = 1 KiB pieces per row tile
    static constexpr int HS = KS / 2;               // ... per half tile
    static constexpr int HALF_BYTES = HS * 1024;
    static constexpr int PER_CU = DIM <= 768 ? 2 : 1;   // blocks per CU (256 B-operand VGPRs at 1024)
    static constexpr int NB = (160 * 1024 / PER_CU) / HALF_BYTES > 4 ? 4 : (160 * 1024 / PER_CU) / HALF_BYTES;
    static constexpr int PER = HS / Q_NW;           // pieces a wave issues per half tile
#ifndef THR_Q_RING_1024
#define THR_Q_RING_1024 8
#endif
    // A fragments in flight per wave: a block that is alone on its CU (dim 1024) has one wave per
    // SIMD, nobody else's MFMAs cover an LDS read that returns late
    static constexpr int RING = PER_CU == 1 ? THR_Q_RING_1024 : Q_RING;
#ifdef Q_PROBE_EXTRA
    static constexpr int VM_PER_PIECE = 2, LDS_BYTES = NB * HALF_BYTES + 32 * 1024;   // (diagnostic: dma())
#else
    static constexpr int VM_PER_PIECE = 1, LDS_BYTES = NB * HALF_BYTES;
#endif
    static_assert(HS % Q_NW == 0 && NB >= 3, "half tile must split evenly over the waves; >= 3 buffers");
};

// queries float32 [n, DIM] -> fragment-major float16 image [qpad/32][KS][64 lanes][8 halves]
// (lane (c, h) of k-step s holds dims 16 s + 8 h .. + 8 of query 32 tile + c: the B operand of
// v_mfma_f32_32x32x16_f16), zeros for padding queries, plus
// qerr[q] = ||fp16(q) - q|| / ||q|| rounded up (the query-side term of the f16 certificate).
// One wave per 32 queries.
template <int DIM, int SHAPE>
__global__ __launch_bounds__(256) void pack_queries_f16(const float* __restrict__ queries,
                                                        int n_queries, f32x4* __restrict__ qfrag,
                                                        float* __restrict__ qerr) {
    constexpr int KS = DIM / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double red[4][4][32];   // [wave][e, n of query half 0 | e, n of half 1][c]
    double e = 0.0, nn = 0.0, e_hi = 0.0, n_hi = 0.0;
    // SHAPE 32: lane (c, h) of fragment s holds dims 16 s + 8 h .. + 8 of query 32 tile + c.
    // SHAPE 16: fragment qb * KS/2 + k32, lane (c, g) holds dims 32 k32 + 8 g .. + 8 of query
    //           32 tile + 16 qb + c (the B operand of v_mfma_f32_16x16x32_f16).
    // The 4 waves of the block take every fourth fragment.
#pragma unroll 4
    for (int s = wave; s < KS; s += 4) {
        int q, d0;
        if constexpr (SHAPE == 32) {
            q = blockIdx.x * 32 + (lane & 31);
            d0 = 16 * s + 8 * (lane >> 5);
        } else {
            q = blockIdx.x * 32 + 16 * (s / (KS / 2)) + (lane & 15);
            d0 = 32 * (s % (KS / 2)) + 8 * (lane >> 4);
        }
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (q < n_queries) {
            const f32x4* src = reinterpret_cast<const f32x4*>(queries + (int64_t)q * DIM + d0);
            lo = src[0];
            hi = src[1];
        }
        const f32x4 packed = pack_f16x8(lo, hi);
        qfrag[((int64_t)blockIdx.x * KS + s) * 64 + lane] = packed;
        const half8 hv = __builtin_bit_cast(half8, packed);
        double e1 = 0.0, n1 = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? lo[j] : hi[j - 4];
            const double d = (double)v - (double)(float)hv[j];
            e1 += d * d;
            n1 += (double)v * (double)v;
        }
        if (SHAPE == 32 || s < KS / 2) {
            e += e1;
            nn += n1;
        } else {   // SHAPE 16, second query half (16 + c)
            e_hi += e1;
            n_hi += n1;
        }
    }
    // eq = ||fp16(q) - q|| / ||q||, rounded up: lanes of one query first, then the 4 waves
    if constexpr (SHAPE == 32) {
        e += __shfl_xor(e, 32, WAVE);
        nn += __shfl_xor(nn, 32, WAVE);
        if (lane < 32) {
            red[wave][0][lane] = e;
            red[wave][1][lane] = nn;
        }
    } else {
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            e += __shfl_xor(e, m, WAVE);
            nn += __shfl_xor(nn, m, WAVE);
            e_hi += __shfl_xor(e_hi, m, WAVE);
            n_hi += __shfl_xor(n_hi, m, WAVE);
        }
        if (lane < 16) {
            red[wave][0][lane] = e;
            red[wave][1][lane] = nn;
            red[wave][0][16 + lane] = e_hi;
            red[wave][1][16 + lane] = n_hi;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double es = 0.0, ns = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            es += red[w][0][threadIdx.x];
            ns += red[w][1][threadIdx.x];
        }
        const float rel = ns > 0.0 ? (float)sqrt(es / ns) : 0.f;
        qerr[blockIdx.x * 32 + threadIdx.x] = __uint_as_float(__float_as_uint(rel) + 1u);
    }
}

// The k-loop of one half tile, fully unrolled: wait for the A fragment of step S (the Q_RING - 1
// younger reads stay in flight), multiply, refill the ring slot with step S + Q_RING, and after
// every fourth MFMA issue one DMA piece of the half tile NB-1 ahead.  The LDS reads are inline
// asm (hipcc would wait vmcnt(0) before an LDS read that may alias an LDS-DMA in flight); the
// "+v" on the waited fragment orders the MFMA behind the wait.  (Plain functions, not lambdas:
// asm operands do not capture.)
#define QS_RD(dst, ks) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(abase), "n"((ks) * 1024) : "memory")
template <int S, int HS, int RING>
__device__ __forceinline__ void qs_fill(f32x4 (&a)[RING], uint32_t abase) {
    if constexpr (S < RING && S < HS) {
        QS_RD(a[S], S);
        qs_fill<S + 1, HS>(a, abase);
    }
}
// SHAPE = 32: v_mfma_f32_32x32x16_f16, one MFMA per 1 KiB piece (piece = 32 rows x 16 dims).
// SHAPE = 16: v_mfma_f32_16x16x32_f16, two MFMAs per piece (piece = 16 rows x 32 dims, against the
//             wave's two 16-query halves).  Same bytes, same cadence (a piece per 32 matrix-pipe
//             cycles); the micro-architecture guide measures the 16x16x32 shape at ~1.15x the
//             FLOP/s of 32x32x16 once the chip is power-limited, which this kernel is.  The
//             fragment-major images differ (quantize_f16_norm / pack_queries_f16 take the shape).
// The accumulators are 16 registers either way; QAcc maps register x to (row, query half).
template <int SHAPE>
struct QAcc;
// SHAPE = 48 (round 4, dim 1024): the 16x16x32 MFMA against THREE 16-query blocks -- 48 queries
//             per wave, 192 per CU: three MFMAs per LDS fragment, and every row tile that is
//             brought into LDS serves half again as many queries (the query image is a sequence of
//             16-query blocks either way: block B's fragments sit at [B * KS/2 + k32]).
// QW = queries per wave, NQB = 16-query blocks, NREG = accumulator registers.
template <>
struct QAcc<32> {
    static constexpr int NQL = 1, SEGS = 2;   // queries per lane; candidate segments per row slice
    static constexpr int QW = 32, NQB = 2, NREG = 16;
    f32x16 v;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int x = 0; x < 16; ++x) v[x] = 0.f;
    }
    template <int X> __device__ __forceinline__ float get() const { return v[X]; }
    // lane (c = lane & 31, h = lane >> 5): register x is row (x&3) + 8 (x>>2) + 4 h of query c
    template <int X> static __device__ __forceinline__ int row(int lane) { return (X & 3) + 8 * (X >> 2) + 4 * (lane >> 5); }
    template <int X> static constexpr int qsel() { return 0; }
    static __device__ __forceinline__ int query(int lane, int) { return lane & 31; }
    static __device__ __forceinline__ int seg(int lane) { return lane >> 5; }
};
template <>
struct QAcc<48> {
    static constexpr int NQL = 3, SEGS = 4;
    static constexpr int QW = 48, NQB = 3, NREG = 24;
    f32x4 t[6];   // tile [row half ra][query block qb] at 3 ra + qb
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int x = 0; x < 6; ++x) t[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    template <int X> __device__ __forceinline__ float get() const { return t[X >> 2][X & 3]; }
    // lane (c = lane & 15, g = lane >> 4): register x = 4 (3 ra + qb) + j is row 16 ra + 4 g + j
    // of query 16 qb + c
    template <int X> static __device__ __forceinline__ int row(int lane) { return 16 * ((X >> 2) / 3) + 4 * (lane >> 4) + (X & 3); }
    template <int X> static constexpr int qsel() { return (X >> 2) % 3; }
    static __device__ __forceinline__ int query(int lane, int qb) { return 16 * qb + (lane & 15); }
    static __device__ __forceinline__ int seg(int lane) { return lane >> 4; }
};
template <>
struct QAcc<16> {
    static constexpr int NQL = 2, SEGS = 4;
    static constexpr int QW = 32, NQB = 2, NREG = 16;
    f32x4 t[4];   // tile [row half ra][query half qb] at 2 ra + qb
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int x = 0; x < 4; ++x) t[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    template <int X> __device__ __forceinline__ float get() const { return t[X >> 2][X & 3]; }
    // lane (c = lane & 15, g = lane >> 4): register x = 4 (2 ra + qb) + j is row 16 ra + 4 g + j
    // of query 16 qb + c
    template <int X> static __device__ __forceinline__ int row(int lane) { return 16 * (X >> 3) + 4 * (lane >> 4) + (X & 3); }
    template <int X> static constexpr int qsel() { return (X >> 2) & 1; }
    static __device__ __forceinline__ int query(int lane, int qb) { return 16 * qb + (lane & 15); }
    static __device__ __forceinline__ int seg(int lane) { return lane >> 4; }
};

// piece S of a half tile (HS pieces): SHAPE 32 -> k-step K0 + S; SHAPE 16 -> row half S & 1 of
// k32-step (K0 + S) / 2, B operands bq[qb * KS/2 + k32]
// the MFMAs of piece S: SHAPE 32 one, else one per 16-query block of the wave
template <int S, int K0, int KS, int SHAPE, int NBQ>
__device__ __forceinline__ void qsx_mfma(const f32x4& a, const f32x4 (&bq)[NBQ], QAcc<SHAPE>& acc) {
    if constexpr (SHAPE == 32) {
        acc.v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a),
                                                       __builtin_bit_cast(half8, bq[K0 + S]), acc.v, 0, 0, 0);
    } else {
        constexpr int k32 = (K0 + S) / 2, ra = S & 1, NQB = QAcc<SHAPE>::NQB;
        static_assert(NBQ == NQB * KS / 2, "one B fragment per 16-query block and 32-dim step");
        static_for<0, NQB>([&](auto qb) {
            acc.t[NQB * ra + qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                __builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bq[qb * (KS / 2) + k32]),
                acc.t[NQB * ra + qb], 0, 0, 0);
        });
    }
}
template <int S, int HS, int K0, int KS, int PER, int SHAPE, int RING, int NBQ, typename Issue>
__device__ __forceinline__ void qsx_steps(f32x4 (&a)[RING], const f32x4 (&bq)[NBQ], QAcc<SHAPE>& acc,
                                          uint32_t abase, Issue& issue_piece) {
    if constexpr (S < HS) {
        constexpr int left = HS - S - 1 < RING - 1 ? HS - S - 1 : RING - 1;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[S % RING]) : "n"(left) : "memory");
        qsx_mfma<S, K0, KS, SHAPE>(a[S % RING], bq, acc);
        if constexpr (S + RING < HS) QS_RD(a[S % RING], S + RING);
        constexpr int every = HS / PER;
        if constexpr (S % every == 1 && S / every < PER) issue_piece(std::integral_constant<int, S / every>{});
        qsx_steps<S + 1, HS, K0, KS, PER, SHAPE, RING, NBQ>(a, bq, acc, abase, issue_piece);
    }
}

// the emit of one accumulator register (a struct member, not a lambda: its store is inline asm).
// A lane's candidate segment is addressed as a 32-bit byte offset from `cand` (one SGPR pair for
// the base instead of a 64-bit pointer per lane): slot = next free entry, end = one past the last.
template <int SHAPE>
struct QEmit {
    static constexpr int NQL = QAcc<SHAPE>::NQL;
    const Cand* cand;
    const int32_t* doc_coll;
    float tau[NQL];
    int qc[NQL];
    uint32_t slot[NQL], end[NQL];
    template <int X>
    __device__ __forceinline__ void one(const QAcc<SHAPE>& acc, uint32_t row0, int lane) {
        constexpr int qs = QAcc<SHAPE>::template qsel<X>();
        const float v = acc.template get<X>();
        if (v >= tau[qs]) {   // false for NaN
            const uint32_t row = row0 + (uint32_t)QAcc<SHAPE>::template row<X>(lane);
            // (a filtered query pays a dependent gather here, and its wait drains the DMA queue)
            if (qc[qs] != -1 && doc_coll[row] != qc[qs]) return;
            if (slot[qs] < end[qs]) {
                const uint64_t word = (uint64_t)__float_as_uint(v) | ((uint64_t)row << 32);
                // (inline asm: a store hipcc can see would make it wait vmcnt(0) -- DMA included --
                // at the loop's back edge)
                asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(slot[qs]), "v"(word), "s"(cand) : "memory");
            }
            slot[qs] += 8;
        }
    }
    template <int X>
    __device__ __forceinline__ void all(const QAcc<SHAPE>& acc, uint32_t row0, int lane) {
        if constexpr (X < QAcc<SHAPE>::NREG) {
            one<X>(acc, row0, lane);
            all<X + 1>(acc, row0, lane);
        }
    }
};

// qsx_steps with the emit of the PREVIOUS row tile's accumulators threaded through it (the 4-wave
// kernel below, MODE_FILTER): a block of that kernel has ONE wave per SIMD, so while a wave runs
// its epilogue nothing feeds the SIMD's matrix pipe -- the stamps put that at a quarter of a half
// tile.  Here the 16 accumulator registers of tile i - 1 are compared / stored one at a time
// between the MFMAs of tile i (8 per half tile, evenly spaced): an MFMA occupies the pipe for
// 16 cycles after it issues, which is what one register's compare-and-branch takes to issue.
template <int S, int HS, int K0, int KS, int PER, int SHAPE, int HF, int RING, int NBQ, typename Issue>
__device__ __forceinline__ void qsx_steps_pe(f32x4 (&a)[RING], const f32x4 (&bq)[NBQ], QAcc<SHAPE>& acc,
                                             uint32_t abase, Issue& issue_piece, QEmit<SHAPE>& em,
                                             const QAcc<SHAPE>& prv, uint32_t row_prv, int lane) {
    if constexpr (S < HS) {
        constexpr int left = HS - S - 1 < RING - 1 ? HS - S - 1 : RING - 1;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[S % RING]) : "n"(left) : "memory");
        qsx_mfma<S, K0, KS, SHAPE>(a[S % RING], bq, acc);
        if constexpr (S + RING < HS) QS_RD(a[S % RING], S + RING);
        constexpr int every = HS / PER;
        if constexpr (S % every == 1 && S / every < PER) issue_piece(std::integral_constant<int, S / every>{});
        // half the registers of the previous tile per half tile, evenly spaced: register j after the
        // step at which j = floor(S * NRH / HS) is about to change
        constexpr int NRH = QAcc<SHAPE>::NREG / 2, j = S * NRH / HS;
        if constexpr ((S + 1) * NRH / HS > j) em.template one<NRH * HF + j>(prv, row_prv, lane);
        qsx_steps_pe<S + 1, HS, K0, KS, PER, SHAPE, HF, RING, NBQ>(a, bq, acc, abase, issue_piece, em, prv, row_prv, lane);
    }
}
#undef QS_RD

// PROF: diagnostic build (thr_dense_scan_stamps_f16): s_memtime stamps around the phases of the
// half-tile loop, summed per wave into stamps[(block * 4 + wave) * 8 + {0: wait for the own DMA
// pieces, 1: barrier, 2: ring fill, 3: k-loop (with the DMA issue), 4: emit, 5: half tiles,
// 6: whole loop, 7: HW_ID}].  Each stamp drains the wave's LDS/SMEM queue, so the build is slower
// than the real one; it only says where the time goes.
template <int DIM, int MODE, bool PROF = false, int SHAPE = 32>
__global__ __launch_bounds__(Q_NW * 64, QScan<DIM>::PER_CU) void dense_scan_f16q(
    const f32x4* __restrict__ packed, const f32x4* __restrict__ qfrag, int n_qtiles,
    int64_t n_tiles, int64_t tile_stride, const float* __restrict__ tau,
    int* __restrict__ seg_cnt, Cand* __restrict__ cand, int seg_cap,
    float* __restrict__ sample_scores, int64_t sample_ld,
    const int32_t* __restrict__ doc_coll, const int32_t* __restrict__ query_coll, int n_queries,
    unsigned long long* __restrict__ stamps = nullptr) {
    using C = QScan<DIM>;
    using A = QAcc<SHAPE>;
    constexpr int KS = C::KS, HS = C::HS, NB = C::NB, PER = C::PER, NQL = A::NQL, QW = A::QW;
    constexpr int NBQ = SHAPE == 32 ? KS : A::NQB * KS / 2;   // B fragments of the wave's queries
    extern __shared__ f32x4 lds_rows[];  // NB half-tile buffers

    const ScanSlot slot = scan_slot(n_qtiles);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q32 = slot.qtile * Q_NW + wave;  // this wave's tile of QW (32, or 48) queries
    const bool idle = (int64_t)q32 * QW >= n_queries;   // all padding: only moves row pieces

    // B operands: the wave's 32 queries, all k-steps, in registers for the whole launch
    f32x4 bq[NBQ];
    if (!idle)
        static_for<0, NBQ>([&](auto s) {
            bq[s] = qfrag[((int64_t)q32 * NBQ + s) * 64 + lane];
        });
    // the lane's private candidate segments (MODE_FILTER): query q, segment SEGS * slice + seg(lane)
    const int nseg = A::SEGS * slot.nslices;
    const int my_seg = A::SEGS * slot.slice + A::seg(lane);
    QEmit<SHAPE> em;
    em.cand = cand;
    em.doc_coll = doc_coll;
    uint32_t start[NQL];
#pragma unroll
    for (int u = 0; u < NQL; ++u) {
        const int q = q32 * QW + A::query(lane, u);
        em.tau[u] = MODE == MODE_FILTER ? tau[q] : 0.f;
        // collection filter of this lane's query (-1: none): checked only for rows that pass tau
        em.qc[u] = (MODE == MODE_FILTER && query_coll && q < n_queries) ? query_coll[q] : -1;
        start[u] = (uint32_t)(((int64_t)q * CAND_CAP + (int64_t)my_seg * seg_cap) * sizeof(Cand));
        em.slot[u] = start[u];
        em.end[u] = start[u] + (uint32_t)(seg_cap * sizeof(Cand));
    }
    // Retire these loads HERE, visibly to hipcc: left pending, their first use (the first MFMA
    // of the tile loop) gets an s_waitcnt vmcnt(0) on every trip, which would drain the DMA
    // of the next half tiles each time.
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // this block's row tiles: slice, slice + nslices, ...; half tile j = (tile j/2, dims half j&1)
    const int64_t first = slot.slice, step = slot.nslices;
    const int64_t n_mine = first < n_tiles ? (n_tiles - first + step - 1) / step : 0;
    const int64_t n_half = 2 * n_mine;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds_rows;
    // piece p (wave + 4 p) of the block's j-th half tile -> buffer buf; past the end the last
    // half tile is requested again (never read): the count of pieces in flight stays what the
    // vmcnt waits assume
    auto piece_src = [&](int64_t j) -> const f32x4* {
        const int64_t jc = j < n_half ? j : n_half - 1;
        const int64_t t = first + (jc >> 1) * step;
        return packed + ((t * tile_stride * KS + (jc & 1) * HS + wave) * 64 + lane);
    };
#ifdef Q_PROBE_EXTRA
    f32x4 probe_reg = {0.f, 0.f, 0.f, 0.f};   // (diagnostic builds only: see below)
#endif
    auto dma = [&](const f32x4* src, int buf, int p) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + p * Q_NW * 64),
            (__attribute__((address_space(3))) void*)(size_t)(lds_base + buf * C::HALF_BYTES +
                                                              (wave + p * Q_NW) * 1024),
            16, 0, 0);
#ifdef Q_PROBE_EXTRA
        // Diagnostic builds (_build.build_variant(name, ["Q_PROBE_EXTRA=1|2"]), dim 1024): is the
        // wave's instruction issue what bounds the k-loop?  Every row piece is issued a SECOND time
        // into 32 KiB of LDS nobody reads -- 1: as another LDS-DMA piece, 2: as a plain load into a
        // register plus a ds_write_b128 of that register (stale data: only the issue slots
        // matter).  Results are unchanged.  Measured (THR_DENSE_QW=32 scripts/probe_dim1024.py
        // 1000000 2048, one box): 4.201 ms plain, 4.180 with the pieces doubled, 4.160 with the load
        // + write added -- the extra instructions are free: not issue-bound.
        const uint32_t scratch = lds_base + NB * C::HALF_BYTES + (wave + (p & 7) * Q_NW) * 1024;
        if constexpr (Q_PROBE_EXTRA == 1) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + p * Q_NW * 64),
                (__attribute__((address_space(3))) void*)(size_t)scratch, 16, 0, 0);
        } else {
            asm volatile("ds_write_b128 %0, %1" ::"v"(scratch + lane * 16), "v"(probe_reg) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(probe_reg) : "v"(src + p * Q_NW * 64) : "memory");
        }
#endif
    };
    if (n_half > 0) {
#pragma unroll
        for (int b = 0; b < NB - 1; ++b) {
            const f32x4* src = piece_src(b);
#pragma unroll
            for (int p = 0; p < PER; ++p) dma(src, b, p);
        }
    }

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, t_loop = 0, t_prev = 0;
    auto stamp = [&](int j) {
        if constexpr (PROF) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (j >= 0) ph[j] += t - t_prev;
            t_prev = t;
        }
    };
    if constexpr (PROF) t_loop = __builtin_amdgcn_s_memtime();

    A acc;
    int buf = 0;
    if (idle) {
        // same barriers, same DMA order and wait counts as a working wave; no LDS reads, no
        // MFMAs, nothing to emit (see dense_scan_f16qs below)
#pragma unroll 1
        for (int64_t j = 0; j < n_half; ++j) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PER * C::VM_PER_PIECE) : "memory");
            __builtin_amdgcn_s_barrier();
            const int nbuf = buf == 0 ? NB - 1 : buf - 1;
            const f32x4* src = piece_src(j + NB - 1);
#pragma unroll
            for (int p = 0; p < PER; ++p) dma(src, nbuf, p);
            buf = buf + 1 == NB ? 0 : buf + 1;
        }
    }
    // MODE_FILTER: the emit of a row tile runs between the MFMAs of the NEXT one (qsx_steps_pe), out
    // of the other of two accumulator sets; the first tile "emits" a set of -inf (a compare and a
    // branch per register), the last tile's set is emitted after the loop.  Same barriers, same
    // DMA order and wait counts per half tile as below.
    // (Only where a block is alone on its CU -- dim 1024, one wave per SIMD: with two blocks per CU
    // the other block's MFMAs already cover an epilogue, and the second accumulator set does not fit
    // the 256-register budget there.)
    // (And not with 48 queries per wave: 384 B-operand registers leave no room for a second set.)
#ifdef Q_NO_PIPE_EMIT   // (A/B builds: the round-3 loop, the emit after its own tile)
    constexpr bool PIPE_EMIT = false;
#else
    constexpr bool PIPE_EMIT = true;
#endif
    if constexpr (PIPE_EMIT && MODE == MODE_FILTER && !PROF && C::PER_CU == 1 && SHAPE != 48) {
        A acc2[2];
#pragma unroll
        for (int x = 0; x < A::NREG; ++x) {
            if constexpr (SHAPE == 32) acc2[1].v[x] = -INFINITY;
            else acc2[1].t[x >> 2][x & 3] = -INFINITY;
        }
        uint32_t row_of[2] = {0u, 0u};
#define QS_HALF_PE(hf, CUR, PRV)                                                                   \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PER * C::VM_PER_PIECE) : "memory");                      \
        __builtin_amdgcn_s_barrier();                                                              \
        const int nbuf = buf == 0 ? NB - 1 : buf - 1; /* half tile j-1's buffer */                 \
        const f32x4* src = piece_src(2 * i + (hf) + NB - 1);                                       \
        auto issue_piece = [&](auto P) { dma(src, nbuf, decltype(P)::value); };                    \
        const uint32_t abase = lds_base + buf * C::HALF_BYTES + lane * 16;                         \
        f32x4 a[C::RING];                                                                           \
        qs_fill<0, HS>(a, abase);                                                                  \
        qsx_steps_pe<0, HS, (hf) * HS, KS, PER, SHAPE, (hf)>(a, bq, acc2[CUR], abase, issue_piece, \
                                                            em, acc2[PRV], row_of[PRV], lane);     \
        buf = buf + 1 == NB ? 0 : buf + 1;                                                         \
    }
        const int64_t n_it = idle ? 0 : n_mine;
        int64_t i = 0;
#pragma unroll 1
        for (; i + 1 < n_it; i += 2) {
            acc2[0].zero();
            QS_HALF_PE(0, 0, 1)
            QS_HALF_PE(1, 0, 1)
            row_of[0] = (uint32_t)((first + i * step) * tile_stride * 32);
            ++i;
            acc2[1].zero();
            QS_HALF_PE(0, 1, 0)
            QS_HALF_PE(1, 1, 0)
            row_of[1] = (uint32_t)((first + i * step) * tile_stride * 32);
            --i;
        }
        if (i < n_it) {   // an odd tile left: into set 0, emitting set 1; then set 0 itself
            acc2[0].zero();
            QS_HALF_PE(0, 0, 1)
            QS_HALF_PE(1, 0, 1)
            em.template all<0>(acc2[0], (uint32_t)((first + i * step) * tile_stride * 32), lane);
        } else if (n_it > 0) {
            em.template all<0>(acc2[1], row_of[1], lane);
        }
#undef QS_HALF_PE
    } else {
    // one trip = one row tile = two half tiles (the accumulators run through both)
#pragma unroll 1
    for (int64_t i = 0; i < (idle ? 0 : n_mine); ++i) {
        // (a macro, not a lambda: asm operands do not capture)
#define QS_HALF(hf)                                                                                \
    {                                                                                              \
        stamp(-1);                                                                                 \
        /* own pieces of half tile 2i+hf done (the NB-2 younger half tiles' pieces -- and the      \
           emit's few stores among them, which are over-waited for -- may stay in flight) */       \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PER * C::VM_PER_PIECE) : "memory");                      \
        stamp(0);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                              \
        stamp(1);                                                                                  \
        const int nbuf = buf == 0 ? NB - 1 : buf - 1; /* half tile j-1's buffer */                 \
        const f32x4* src = piece_src(2 * i + (hf) + NB - 1);                                       \
        auto issue_piece = [&](auto P) { dma(src, nbuf, decltype(P)::value); };                    \
        const uint32_t abase = lds_base + buf * C::HALF_BYTES + lane * 16;                         \
        f32x4 a[C::RING];                                                                           \
        qs_fill<0, HS>(a, abase);                                                                  \
        stamp(2);                                                                                  \
        qsx_steps<0, HS, (hf) * HS, KS, PER, SHAPE>(a, bq, acc, abase, issue_piece);                \
        if constexpr (PROF) {                                                                      \
            if constexpr (SHAPE == 32) asm volatile("" : "+v"(acc.v));                               \
            else {                                                                                 \
                asm volatile("" : "+v"(acc.t[0]), "+v"(acc.t[1]), "+v"(acc.t[2]), "+v"(acc.t[3]));   \
                if constexpr (A::NQB == 3) asm volatile("" : "+v"(acc.t[4]), "+v"(acc.t[5]));       \
            }                                                                                      \
        }                                                                                          \
        stamp(3);                                                                                  \
        buf = buf + 1 == NB ? 0 : buf + 1;                                                         \
    }
        acc.zero();
        QS_HALF(0)
        QS_HALF(1)
#undef QS_HALF

        const int64_t t = first + i * step;
        if constexpr (MODE == MODE_ALL) {
            if constexpr (SHAPE == 32) {
                // accumulator registers 4g..4g+3 are 4 consecutive rows: one 16-byte store each
                float* dst = sample_scores + (int64_t)(q32 * 32 + (lane & 31)) * sample_ld + t * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sv = acc.v[4 * g + j];
                        v[j] = sv == sv ? sv : -INFINITY;  // NaN: no such row / no embedding
                    }
                    *reinterpret_cast<f32x4*>(dst + 8 * g) = v;
                }
            } else {
#pragma unroll
                for (int x = 0; x < 2 * A::NQB; ++x) {   // tile (ra, qb): rows 16 ra + 4 g .. + 4 of query 16 qb + c
                    f32x4 v = acc.t[x];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] == v[j] ? v[j] : -INFINITY;
                    float* dst = sample_scores + (int64_t)(q32 * QW + 16 * (x % A::NQB) + (lane & 15)) * sample_ld +
                                 t * 32 + 16 * (x / A::NQB) + 4 * (lane >> 4);
                    *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
        } else {
            em.template all<0>(acc, (uint32_t)(t * tile_stride * 32), lane);
        }
        stamp(4);
    }
    }
    // nothing may still be landing in LDS when the block retires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MODE == MODE_FILTER) {
#pragma unroll
        for (int u = 0; u < NQL; ++u)
            seg_cnt[(int64_t)(q32 * QW + A::query(lane, u)) * nseg + my_seg] = (int)((em.slot[u] - start[u]) / sizeof(Cand));
    }
    if constexpr (PROF) {
        if (lane == 0) {
            unsigned long long* o = stamps + ((int64_t)blockIdx.x * Q_NW + wave) * 8;
            for (int j = 0; j < 5; ++j) o[j] = ph[j];
            o[5] = (unsigned long long)n_half;
            o[6] = __builtin_amdgcn_s_memtime() - t_loop;
            o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dense_scan_f16qs: the same scan with ONE 8-wave block per CU (256 queries) whose two wave groups
// run STAGGERED by half a tile -- the default at dim <= 768.
//
// What the counters of the two-blocks-per-CU kernel above said (profiles/r2_scan_v2_*.json): pipe
// 64 % busy; the older block of a CU wins the matrix pipe and retires when 3/4 of the launch is
// over, so the pipe is half empty for the last quarter; every CU pulls each tile twice; and the
// chip drops its clock to 1.5-1.7 GHz under the load.  Here the 8 waves share each half tile
// (half the DMA pieces and L2 bytes per flop) and are kept in step by one barrier per half tile,
// but group B (waves 4-7, the younger wave of each SIMD) works one half tile behind group A:
//
//     interval k       A (waves 0-3)                        B (waves 4-7)
//     k = 2i           MFMA half 2i   (k-steps 0..HS)       MFMA half 2i-1 (second half of tile i-1)
//     k = 2i+1         MFMA half 2i+1, then EMIT tile i     EMIT tile i-1, then MFMA half 2i
//
// so an emit always runs under the other group's MFMAs (A is the older wave of the SIMD and gets
// the pipe first: its MFMAs cover B's emit at the start of the interval, B's cover A's at the
// end), and B -- whose half tile was published one barrier earlier -- has its first fragments in
// registers before the barrier opens, which covers A's LDS latency after it.
// Barrier k: every wave has waited for its own pieces of half tile k (3 younger half tiles'
// pieces stay in flight); afterwards A is done with half k-1 and B with half k-2, so half k-2's
// buffer takes the pieces of half k+4: ring of 6 half-tile buffers (144 KiB at dim 768).
// ---------------------------------------------------------------------------------------------
constexpr int QS_NW = 8;
constexpr int QS_NBUF = 6;

template <int DIM>
struct QStag {
    static constexpr int KS = DIM / 16, HS = KS / 2, HALF_BYTES = HS * 1024;
    static constexpr int PER = HS / QS_NW;               // pieces a wave issues per half tile
    static constexpr int LDS_BYTES = QS_NBUF * HALF_BYTES;
    static_assert(HS % QS_NW == 0 && LDS_BYTES <= 160 * 1024, "dim 512 / 768 only");
};

template <int DIM, int MODE, int SHAPE = 32>
__global__ __launch_bounds__(QS_NW * 64) void dense_scan_f16qs(
    const f32x4* __restrict__ packed, const f32x4* __restrict__ qfrag, int n_qtiles,
    int64_t n_tiles, int64_t tile_stride, const float* __restrict__ tau,
    int* __restrict__ seg_cnt, Cand* __restrict__ cand, int seg_cap,
    float* __restrict__ sample_scores, int64_t sample_ld,
    const int32_t* __restrict__ doc_coll, const int32_t* __restrict__ query_coll, int n_queries) {
    using C = QStag<DIM>;
    using A = QAcc<SHAPE>;
    constexpr int KS = C::KS, HS = C::HS, PER = C::PER, NBUF = QS_NBUF, NQL = A::NQL;
    extern __shared__ f32x4 lds_rows[];  // 6 half-tile buffers

    const ScanSlot slot = scan_slot(n_qtiles);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q32 = slot.qtile * QS_NW + wave;   // this wave's 32 queries

    // B operands: the wave's 32 queries, every k-step, in registers for the whole launch
    // (both shapes: KS fragments of 1 KiB, image [q32 tile][KS][64 lanes])
    f32x4 bq[KS];
    if ((int64_t)q32 * 32 < n_queries)
        static_for<0, KS>([&](auto s) {
            bq[s] = qfrag[((int64_t)q32 * KS + s) * 64 + lane];
        });
    const int nseg = A::SEGS * slot.nslices;
    const int my_seg = A::SEGS * slot.slice + A::seg(lane);
    QEmit<SHAPE> em;
    em.cand = cand;
    em.doc_coll = doc_coll;
    uint32_t start[NQL];
#pragma unroll
    for (int u = 0; u < NQL; ++u) {
        const int q = q32 * 32 + A::query(lane, u);
        em.tau[u] = MODE == MODE_FILTER ? tau[q] : 0.f;
        // collection filter of this lane's query (-1: none): checked only for rows that pass tau
        em.qc[u] = (MODE == MODE_FILTER && query_coll && q < n_queries) ? query_coll[q] : -1;
        start[u] = (uint32_t)(((int64_t)q * CAND_CAP + (int64_t)my_seg * seg_cap) * sizeof(Cand));
        em.slot[u] = start[u];
        em.end[u] = start[u] + (uint32_t)(seg_cap * sizeof(Cand));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // retire these loads visibly to hipcc (see dense_scan_f16q)

    const int64_t first = slot.slice, step = slot.nslices;
    const int64_t n_mine = first < n_tiles ? (n_tiles - first + step - 1) / step : 0;
    const int64_t n_half = 2 * n_mine;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds_rows;
    auto piece_src = [&](int64_t j) -> const f32x4* {
        const int64_t jc = j < n_half ? j : n_half - 1;  // past the end: the last half again (never read)
        const int64_t t = first + (jc >> 1) * step;
        return packed + ((t * tile_stride * KS + (jc & 1) * HS + wave) * 64 + lane);
    };
    auto dma = [&](const f32x4* src, int buf, int p) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + p * QS_NW * 64),
            (__attribute__((address_space(3))) void*)(size_t)(lds_base + buf * C::HALF_BYTES +
                                                              (wave + p * QS_NW) * 1024),
            16, 0, 0);
    };
    auto emit_all = [&](const A& acc, int64_t i) {   // MODE_ALL: the sample scores of tile i
        const int64_t t = first + i * step;
        if constexpr (SHAPE == 32) {
            float* dst = sample_scores + (int64_t)(q32 * 32 + (lane & 31)) * sample_ld + t * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // registers 4g..4g+3 are 4 consecutive rows
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sv = acc.v[4 * g + j];
                    v[j] = sv == sv ? sv : -INFINITY;   // NaN: no such row / no embedding
                }
                *reinterpret_cast<f32x4*>(dst + 8 * g) = v;
            }
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) {   // tile (ra, qb): rows 16 ra + 4 g .. + 4 of query 16 qb + c
                f32x4 v = acc.t[x];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] == v[j] ? v[j] : -INFINITY;
                float* dst = sample_scores + (int64_t)(q32 * 32 + 16 * (x & 1) + (lane & 15)) * sample_ld +
                             t * 32 + 16 * (x >> 1) + 4 * (lane >> 4);
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    };
#define QS_EMIT(i_)                                                                                \
    if constexpr (MODE == MODE_ALL) {                                                              \
        emit_all(acc, (i_));                                                                       \
    } else {                                                                                       \
        em.template all<0>(acc, (uint32_t)((first + (i_) * step) * tile_stride * 32), lane);       \
    }
    // barrier k: own pieces of half k have landed (3 younger halves' pieces may stay in flight)
#define QS_SYNC()                                                      \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");     \
    __builtin_amdgcn_s_barrier();

    if (n_half > 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4* src = piece_src(b);
#pragma unroll
            for (int p = 0; p < PER; ++p) dma(src, b, p);
        }
    }
    A acc;
    f32x4 a[Q_RING];
    if ((int64_t)q32 * 32 >= n_queries) {
        // ---- a wave whose 32 queries are all padding (small batches: one query keeps 7 of the 8
        // waves here): it only moves its share of the row pieces -- same barriers, same DMA order
        // and wait counts as a working wave, no LDS reads, no MFMAs, nothing to emit.  A launch
        // with one live wave per CU is bound by the row stream alone. ----
        int buf = wave < 4 ? 4 : 5;   // buffer of the next half this wave issues
        int64_t j = wave < 4 ? 4 : 5;
        if (wave >= 4) {
            QS_SYNC()  // barrier 0
            if (n_half > 0) {
                const f32x4* src = piece_src(4);
#pragma unroll
                for (int p = 0; p < PER; ++p) dma(src, 4, p);
            }
        }
#pragma unroll 1
        for (int64_t k = 0; k < n_half; ++k, ++j) {
            QS_SYNC()   // A: barrier k, then half k + 4;  B: barrier k + 1, then half k + 5
            const f32x4* src = piece_src(j);
#pragma unroll
            for (int p = 0; p < PER; ++p) dma(src, buf, p);
            buf = buf + 1 == NBUF ? 0 : buf + 1;
        }
        if (wave < 4) {
            QS_SYNC()  // barrier 2n
        }
    } else if (wave < 4) {
        // ---- group A: half k in interval k ----
        int buf = 0;
#pragma unroll 1
        for (int64_t i = 0; i < n_mine; ++i) {
            {
                QS_SYNC()  // barrier 2i
                const int nbuf = buf + 4 >= NBUF ? buf + 4 - NBUF : buf + 4;
                const f32x4* src = piece_src(2 * i + 4);
                auto issue_piece = [&](auto P) { dma(src, nbuf, decltype(P)::value); };
                const uint32_t abase = lds_base + buf * C::HALF_BYTES + lane * 16;
                qs_fill<0, HS>(a, abase);
                acc.zero();
                qsx_steps<0, HS, 0, KS, PER, SHAPE>(a, bq, acc, abase, issue_piece);
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
            {
                QS_SYNC()  // barrier 2i+1
                const int nbuf = buf + 4 >= NBUF ? buf + 4 - NBUF : buf + 4;
                const f32x4* src = piece_src(2 * i + 5);
                auto issue_piece = [&](auto P) { dma(src, nbuf, decltype(P)::value); };
                const uint32_t abase = lds_base + buf * C::HALF_BYTES + lane * 16;
                qs_fill<0, HS>(a, abase);
                qsx_steps<0, HS, HS, KS, PER, SHAPE>(a, bq, acc, abase, issue_piece);
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
            QS_EMIT(i)
        }
        QS_SYNC()  // barrier 2n: B's last interval
    } else {
        // ---- group B: half k-1 in interval k ----
        QS_SYNC()  // barrier 0: nothing to multiply yet
        if (n_half > 0) {
            const f32x4* src = piece_src(4);
#pragma unroll
            for (int p = 0; p < PER; ++p) dma(src, 4, p);
        }
        int buf = 0;  // buffer of half 2i
#pragma unroll 1
        for (int64_t i = 0; i < n_mine; ++i) {
            const int buf1 = buf + 1 == NBUF ? 0 : buf + 1;       // half 2i+1
            const int buf2 = buf1 + 1 == NBUF ? 0 : buf1 + 1;     // half 2i+2
            {
                QS_SYNC()  // barrier 2i+1
                if (i > 0) {
                    QS_EMIT(i - 1)
                }
                const int nbuf = buf + 5 >= NBUF ? buf + 5 - NBUF : buf + 5;   // half 2i+5
                const f32x4* src = piece_src(2 * i + 5);
                auto issue_piece = [&](auto P) { dma(src, nbuf, decltype(P)::value); };
                const uint32_t abase = lds_base + buf * C::HALF_BYTES + lane * 16;
                if (i == 0) qs_fill<0, HS>(a, abase);   // later trips: filled before the barrier
                acc.zero();
                qsx_steps<0, HS, 0, KS, PER, SHAPE>(a, bq, acc, abase, issue_piece);
                qs_fill<0, HS>(a, lds_base + buf1 * C::HALF_BYTES + lane * 16);   // half 2i+1: published at this barrier
            }
            {
                QS_SYNC()  // barrier 2i+2
                const f32x4* src = piece_src(2 * i + 6);   // into half 2i's buffer, which this group just left
                auto issue_piece = [&](auto P) { dma(src, buf, decltype(P)::value); };
                const uint32_t abase = lds_base + buf1 * C::HALF_BYTES + lane * 16;
                qsx_steps<0, HS, HS, KS, PER, SHAPE>(a, bq, acc, abase, issue_piece);
                if (i + 1 < n_mine)
                    qs_fill<0, HS>(a, lds_base + buf2 * C::HALF_BYTES + lane * 16);   // half 2i+2: published at this barrier
            }
            buf = buf2;
        }
        if (n_mine > 0) {
            QS_EMIT(n_mine - 1)
        }
    }
#undef QS_SYNC
#undef QS_EMIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be landing in LDS
    if constexpr (MODE == MODE_FILTER) {
#pragma unroll
        for (int u = 0; u < NQL; ++u)
            seg_cnt[(int64_t)(q32 * 32 + A::query(lane, u)) * nseg + my_seg] = (int)((em.slot[u] - start[u]) / sizeof(Cand));
    }
}

// float32 corpus -> fragment-major float16 copy of the NORMALISED rows (d * (1/||d||), round to
// nearest even); rows with ||d|| = 0 and the padding rows of the last tile are NaN.  Also
// max over rows of ||d16 - d/||d|| || (float64, against the exactly normalised row), rounded up,
// via atomicMax on the float bits.  packed == nullptr: measure only.  One wave per row (rows of
// the padded tail included).
__global__ __launch_bounds__(256) void quantize_f16_norm(const float* __restrict__ docs,
                                                         int64_t n_docs, int dim, int shape,
                                                         _Float16* __restrict__ packed,
                                                         unsigned int* __restrict__ max_err_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6);
    const int64_t n_pad = (n_docs + 31) / 32 * 32;
    if (row >= n_pad) return;
    const int64_t tile = row >> 5;
    const int r = (int)(row & 31), ks = dim / 16;
    const float* x = docs + row * dim;
    double nn = 0.0;
    if (row < n_docs)
        for (int i = lane; i < dim; i += WAVE) nn += (double)x[i] * (double)x[i];
    for (int m = 32; m >= 1; m >>= 1) nn += __shfl_xor(nn, m, WAVE);
    const double dn = sqrt(nn);
    const bool live = dn > 0.0;  // same rows thr_doc_norms gives inv_norm = 0 ("no embedding")
    double err = 0.0;
    for (int i = lane; i < dim; i += WAVE) {
        const float v = live ? x[i] : 0.f;
        const _Float16 hv = live ? (_Float16)(float)((double)v / dn) : (_Float16)__builtin_nanf("");
        if (packed) {
            if (shape == 32) {   // piece s = 32 rows x dims [16 s, +16): lane r + 32 hh, hh = dim half
                const int s = i >> 4, hh = (i >> 3) & 1, e = i & 7;
                packed[(((tile * ks + s) * 64) + r + 32 * hh) * 8 + e] = hv;
            } else {             // piece 2 k32 + ra = rows [16 ra, +16) x dims [32 k32, +32): lane (r & 15) + 16 g
                const int k32 = i >> 5, g = (i >> 3) & 3, e = i & 7, ra = r >> 4;
                packed[(((tile * ks + 2 * k32 + ra) * 64) + (r & 15) + 16 * g) * 8 + e] = hv;
            }
        }
        if (live) {
            const double d = (double)v / dn - (double)(float)hv;
            err += d * d;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) err += __shfl_xor(err, m, WAVE);
    if (lane == 0 && live) {
        float rel = (float)sqrt(err);
        rel = __uint_as_float(__float_as_uint(rel) + 1u);
        atomicMax(max_err_bits, __float_as_uint(rel));
    }
}

// The in-flight-rounding scan (dense_scan_f16) rounds the float32 rows AS THEY ARE: its row error
// term is max_d ||fp16(d) - d|| / ||d|| (+inf when a value leaves the float16 range), rounded up,
// accumulated as ordered float bits with atomicMax.  One wave per row.
__global__ __launch_bounds__(256) void measure_f16_error(const float* __restrict__ docs, int64_t n_docs,
                                                         int dim, unsigned int* __restrict__ max_rel_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6);
    if (row >= n_docs) return;
    const float* x = docs + row * dim;
    double err = 0.0, nrm = 0.0;
    for (int i = lane; i < dim; i += WAVE) {
        const float v = x[i];
        const double d = (double)v - (double)(float)(_Float16)v;
        err += d * d;
        nrm += (double)v * (double)v;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        err += __shfl_xor(err, m, WAVE);
        nrm += __shfl_xor(nrm, m, WAVE);
    }
    if (lane == 0 && nrm > 0.0) {
        float rel = (float)sqrt(err / nrm);
        rel = __uint_as_float(__float_as_uint(rel) + 1u);
        atomicMax(max_rel_bits, __float_as_uint(rel));  // positive floats order like their bits
    }
}

}  // namespace thr
