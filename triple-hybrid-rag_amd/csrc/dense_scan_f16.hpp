// Shortlist scan on the f16 matrix cores over the float32 rows themselves
// (thr_dense_topk_f16 with docs16 == NULL): the rows are rounded to float16 in registers on the
// way into the LDS transpose tile -- no second copy of the corpus.  (The flavour that streams a
// float16 copy is dense_scan_f16p.hpp: fragment-major copy, no LDS transpose.)
//
// The float32 corpus stays the source of truth: every returned score is the float64
// rescoring of float32 rows, and the top-k is certified with an error bound that also covers
// the quantisation of rows and queries (see select_rescore / DESIGN.md 4.1):
//     |fp16-scan score - true cosine| <= ea*(1+eq) + eq + eps32        (relative to 1)
//   ea = max over rows of ||d16 - d|| / ||d||   (measured at index build, thr_dense_quantize_f16)
//   eq = ||q16 - q|| / ||q||                    (measured per query in kth_select)
//   eps32 = fp32 accumulation bound of the MFMA chain
//
// Kernel structure = dense_scan_mfma2 (coalesced loads -> register ring -> per-wave LDS
// transpose tile -> fragment reads by inline asm with hand-counted lgkmcnt), with a stage =
// 64 dims and ONE v_mfma_f32_32x32x16_f16 per 16-byte fragment pair; 64 queries per pass (96 KiB
// of LDS as f16) at dim <= 768, 32 at dim 1024.
#pragma once

namespace thr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int H_WAVES = 8;
constexpr int H_THREADS = H_WAVES * WAVE;

// round 8 floats to nearest-even float16 (same rounding as quantize_f16, whose error bound covers
// both flavours)
__device__ __forceinline__ f32x4 pack_f16x8(f32x4 lo, f32x4 hi) {
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    const half4 a = __builtin_convertvector(lo, half4), b = __builtin_convertvector(hi, half4);
    half8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f32x4, v);
}

// NQ = query sub-tiles of 32 (1 or 2).  A stage (64 dims of 32 rows) is 8 KiB of float32: each
// lane loads float4 #c and #(8+c) of its row's 16 (two fully coalesced 128-byte row segments
// per 8 lanes) and packs them into ONE 16-byte f16 chunk, i.e. chunk c of a stage holds dims
// {4c..4c+3, 32+4c..32+4c+3}.  The query tile is laid out with the same permutation; k is only a
// summation index, so the dot products are unchanged.  The register ring holds 2 stages (16 KiB
// in flight per wave).
template <int DIM, int MODE, bool nt_loads, int NQ>
__global__ __launch_bounds__(H_THREADS) void dense_scan_f16(
    const void* __restrict__ docs16, const float* __restrict__ inv_norm, int64_t n_docs,
    const float* __restrict__ queries, int n_queries, int64_t n_tiles, int64_t tile_stride,
    const float* __restrict__ tau, int* __restrict__ tile_cnt, Cand* __restrict__ tile_list,
    int tile_cap, float* __restrict__ sample_scores, int64_t sample_ld,
    const int32_t* __restrict__ doc_coll = nullptr, const int32_t* __restrict__ query_coll = nullptr) {
    constexpr int QT = 32 * NQ;
    constexpr int CPR = DIM / 8;   // 16-byte chunks (8 halves) per row
    constexpr int GPR = DIM / 4;   // 16-byte chunks per float32 row in global memory
    constexpr int GSTEP = 16;      // ... per stage
    constexpr int NG = DIM / 256;  // groups of 4 stages of 64 dims
    constexpr int QBITS = 32 - ROW_BITS_F16;
    static_assert(DIM % 256 == 0 && NG >= 2, "f16 scan needs dim % 256 == 0 and dim >= 512");
    static_assert(QT <= (1 << QBITS), "query-in-tile index must fit the packed candidate word");
    extern __shared__ float4 lds_q[];  // [QT][CPR] f16 queries | H_WAVES stage tiles | H_WAVES wbufs

    const ScanSlot slot = scan_slot((n_queries + QT - 1) / QT);
    const int qtile = slot.qtile;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    f32x4* lds_v = reinterpret_cast<f32x4*>(lds_q);
    f32x4* stage = lds_v + QT * CPR + wave * MF2_STAGE_F4;
    Cand* wbuf = reinterpret_cast<Cand*>(lds_q + QT * CPR + H_WAVES * MF2_STAGE_F4) + wave * WBUF;
    int wcnt = 0;
    auto flush = [&]() {
        int base = 0;
        if (lane == 0) base = atomicAdd(&tile_cnt[qtile], wcnt);
        base = __shfl(base, 0, WAVE);
        for (int i = lane; i < wcnt; i += WAVE)
            if (base + i < tile_cap) tile_list[(int64_t)qtile * tile_cap + base + i] = wbuf[i];
        __builtin_amdgcn_s_waitcnt(0x0F70);
        wcnt = 0;
    };

    // query tile: float32 -> float16 (round to nearest even), swizzled like the f32 kernels
    for (int i = threadIdx.x; i < QT * CPR; i += H_THREADS) {
        const int q = i / CPR, c = i % CPR;
        const int qg = qtile * QT + q;
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
        if (qg < n_queries) {
            // dims of chunk c: the two float4 the row loader pairs up
            const int d_lo = 64 * (c >> 3) + 4 * (c & 7);
            const int d_hi = d_lo + 32;
            const float* qsrc = queries + (int64_t)qg * DIM;
            const float4 lo = *reinterpret_cast<const float4*>(qsrc + d_lo);
            const float4 hi = *reinterpret_cast<const float4*>(qsrc + d_hi);
            v[0] = (_Float16)lo.x; v[1] = (_Float16)lo.y; v[2] = (_Float16)lo.z; v[3] = (_Float16)lo.w;
            v[4] = (_Float16)hi.x; v[5] = (_Float16)hi.y; v[6] = (_Float16)hi.z; v[7] = (_Float16)hi.w;
        }
        lds_v[mf_qslot(q, c, CPR)] = __builtin_bit_cast(f32x4, v);
    }
    __syncthreads();

    float my_tau[NQ];
    int my_qc[NQ];   // collection filter of the lane's queries (-1: none), applied as rows pass tau
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
        my_tau[s] = MODE == MODE_FILTER ? tau[qtile * QT + 32 * s + r] : 0.f;
        my_qc[s] = (MODE == MODE_FILTER && query_coll && qtile * QT + 32 * s + r < n_queries)
                       ? query_coll[qtile * QT + 32 * s + r] : -1;
    }
    const int64_t wave_id = (int64_t)slot.slice * H_WAVES + wave;
    const int64_t wave_stride = (int64_t)slot.nslices * H_WAVES;
    const f32x4* docs4 = reinterpret_cast<const f32x4*>(docs16);

    const int lrow = lane >> 3, lchunk = lane & 7;
    auto load_off = [&](int64_t t, int i) -> int64_t {
        int64_t row = t * tile_stride * MF_ROWS + lrow + 8 * i;
        row = row < n_docs ? row : n_docs - 1;
        return row * GPR + lchunk;
    };
    int wslot0 = mf2_slot(lrow, lchunk), wslot1 = mf2_slot(lrow + 8, lchunk);
    int wslot2 = mf2_slot(lrow + 16, lchunk), wslot3 = mf2_slot(lrow + 24, lchunk);
    const uint32_t q_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds_v;
    const uint32_t st_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)stage;
    // byte addresses: A fragment of quad j in the stage tile; query row bases of the sub-tiles
    uint32_t ra[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = st_lds + (uint32_t)mf2_slot(r, 2 * j + h) * 16u;
    uint32_t qrow[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) qrow[s] = q_lds + (uint32_t)((32 * s + r) * CPR) * 16u;
    int qlow[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) qlow[x] = (((x >> 2) * 8 + 2 * (x & 3) + h) ^ r) & 15;

#define THR_PIN(x) asm volatile("" : "+v"(x))
#define HS_LD(ptr) (nt_loads ? __builtin_nontemporal_load(&docs4[ptr]) : docs4[ptr])
#define HS_LOAD1(dst, i, p)                                             \
    THR_PIN(p); dst[i] = HS_LD(p);                                      \
    dst[4 + i] = HS_LD(p + 8);                                          \
    p += GSTEP;
#define HS_LOAD(dst) HS_LOAD1(dst, 0, p0) HS_LOAD1(dst, 1, p1) HS_LOAD1(dst, 2, p2) HS_LOAD1(dst, 3, p3)
#define HS_STORE1(src, i, ws)                                           \
    THR_PIN(ws);                                                        \
    stage[ws] = pack_f16x8(src[i], src[4 + i]);
#define HS_STORE(src) HS_STORE1(src, 0, wslot0) HS_STORE1(src, 1, wslot1) HS_STORE1(src, 2, wslot2) HS_STORE1(src, 3, wslot3)
    // 1 + NQ LDS reads per quad: the row fragment and one query fragment per sub-tile.
    // qoff = chunk offset (in 16-byte units) of the stage's first chunk group (multiple of 16).
#define HS_READ(F, quadslot, qoff, par, quad)                                                  \
    {                                                                                          \
        const uint32_t cb = (uint32_t)((qoff) + qlow[(par) * 4 + (quad)]) * 16u;               \
        asm volatile("ds_read_b128 %0, %1" : "=v"(F.a) : "v"(ra[quadslot]) : "memory");        \
        asm volatile("ds_read_b128 %0, %1" : "=v"(F.b[0]) : "v"(qrow[0] + cb) : "memory");     \
        if constexpr (NQ > 1)                                                                  \
            asm volatile("ds_read_b128 %0, %1" : "=v"(F.b[NQ - 1]) : "v"(qrow[NQ - 1] + cb) : "memory"); \
    }
#define HS_WAIT2(n, F)                                                                          \
    if constexpr (NQ > 1)                                                                       \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(F.a), "+v"(F.b[0]), "+v"(F.b[NQ - 1]) : : "memory"); \
    else                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(F.a), "+v"(F.b[0]) : : "memory");
#define HS_MMA(F)                                                                               \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, F.a),             \
                                                    __builtin_bit_cast(half8, F.b[0]), acc[0], 0, 0, 0); \
    if constexpr (NQ > 1)                                                                       \
        acc[NQ - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                   \
            __builtin_bit_cast(half8, F.a), __builtin_bit_cast(half8, F.b[NQ - 1]), acc[NQ - 1], 0, 0, 0); \
    asm volatile("" : "+v"(acc[0]));
    // lgkmcnt bookkeeping per stage (R = 1 + NQ reads per fragment set):
    //   before MMA q0 / q1: younger = the other pending set           -> R
    //   before MMA q2 / q3: younger = one set + 4 stage-tile writes   -> R + 4
#define HS_STAGE(u, ringn, qcur, qnxt)                                       \
    HS_WAITR(f0) HS_MMA(f0) HS_READ(f0, 2, qcur, (u) & 1, 2)                 \
    HS_WAITR(f1) HS_MMA(f1) HS_READ(f1, 3, qcur, (u) & 1, 3)                 \
    HS_STORE(ringn)                                                          \
    HS_LOAD(ringn)                                                           \
    HS_WAITR4(f0) HS_MMA(f0) HS_READ(f0, 0, qnxt, ((u) + 1) & 1, 0)          \
    HS_WAITR4(f1) HS_MMA(f1) HS_READ(f1, 1, qnxt, ((u) + 1) & 1, 1)
#define HS_WAITR(F)  if constexpr (NQ > 1) { HS_WAIT2(3, F) } else { HS_WAIT2(2, F) }
#define HS_WAITR4(F) if constexpr (NQ > 1) { HS_WAIT2(7, F) } else { HS_WAIT2(6, F) }

    struct Frag {
        f32x4 a;
        f32x4 b[NQ];
    };
    Frag f0, f1;
    f0.a = f1.a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NQ; ++s) f0.b[s] = f1.b[s] = f0.a;
    f32x4 ring0[8], ring1[8];  // 2 ring slots: 2 float4 per loader row of a stage
    int64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    int64_t t = wave_id;
    if (t < n_tiles) {
        p0 = load_off(t, 0); p1 = load_off(t, 1); p2 = load_off(t, 2); p3 = load_off(t, 3);
        HS_LOAD(ring0)
        HS_LOAD(ring1)
        HS_STORE(ring0)
        HS_LOAD(ring0)
        HS_READ(f0, 0, 0, 0, 0)
        HS_READ(f1, 1, 0, 0, 1)
    }
    for (; t < n_tiles; t += wave_stride) {
        const int64_t row0 = t * tile_stride * MF_ROWS;
        int idx = r;
        if (row0 + idx >= n_docs) idx = (int)(n_docs - 1 - row0);
        THR_PIN(idx);
        const float my_inv = inv_norm[row0 + idx];
        const int64_t tn = t + wave_stride < n_tiles ? t + wave_stride : t;
        const int64_t on0 = load_off(tn, 0), on1 = load_off(tn, 1), on2 = load_off(tn, 2),
                      on3 = load_off(tn, 3);
        f32x16 acc[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s][i] = 0.f;
        // kept rolled: unrolled, every (stage, quad) LDS address becomes its own hoisted VGPR
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int qb = 32 * g;                      // chunk offset of stages 4g, 4g+1
            const int qn = g + 1 < NG ? qb + 32 : 0;    // first chunks of the next group / tile
            // ring of 2: the refill issued in stage u is stage 4g+u+3 -- from u = 1 of the
            // last group on, that is the head of the wave's next row tile
            HS_STAGE(0, ring1, qb, qb)
            if (g == NG - 1) { p0 = on0; p1 = on1; p2 = on2; p3 = on3; }
            HS_STAGE(1, ring0, qb, qb + 16)
            HS_STAGE(2, ring1, qb + 16, qb + 16)
            HS_STAGE(3, ring0, qb + 16, qn)
        }
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            if constexpr (MODE == MODE_ALL) {
                // accumulator registers 4g..4g+3 are 4 consecutive rows: one 16-byte store each
                const int qg = qtile * QT + 32 * s + r;
                float* dst = sample_scores + (int64_t)qg * sample_ld + t * MF_ROWS + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = j + 8 * g + 4 * h;
                        const float inv = __shfl(my_inv, row, WAVE);
                        v[j] = (row0 + row < n_docs && inv > 0.f) ? acc[s][4 * g + j] * inv : -INFINITY;
                    }
                    *reinterpret_cast<f32x4*>(dst + 8 * g) = v;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                const float inv = __shfl(my_inv, row, WAVE);
                const bool ok = row0 + row < n_docs;
                const float sc = acc[s][i] * inv;
                if constexpr (MODE == MODE_ALL) {
                    (void)ok; (void)sc;
                } else {
                    bool pass = ok && inv > 0.f && sc >= my_tau[s];
                    if (pass && my_qc[s] != -1 && doc_coll[row0 + row] != my_qc[s]) pass = false;
                    const uint64_t m = __ballot(pass);
                    if (m) {
                        const int pos = wcnt + __popcll(m & ((1ull << lane) - 1ull));
                        if (pass)
                            wbuf[pos] = Cand{sc, ((uint32_t)(32 * s + r) << ROW_BITS_F16) |
                                                     (uint32_t)(row0 + row)};
                        wcnt += __popcll(m);
                        if (wcnt > WBUF - WAVE) flush();
                    }
                }
            }
        }
    }
#undef HS_WAITR4
#undef HS_WAITR
#undef HS_STAGE
#undef HS_MMA
#undef HS_WAIT2
#undef HS_READ
#undef HS_STORE
#undef HS_STORE1
#undef HS_LOAD
#undef HS_LOAD1
#undef HS_LD
#undef THR_PIN
    if constexpr (MODE == MODE_FILTER) {
        if (wcnt > 0) flush();
    }
}

}  // namespace thr
