// Shortlist scan over the FRAGMENT-MAJOR float16 copy of the corpus (thr_dense_topk_f16 with
// docs16 != NULL; the copy is written by thr_dense_quantize_f16).
//
// PMC on the LDS-transpose kernels (dense_scan_f16<>) showed the LDS instruction path saturated:
// per 8 MFMAs a wave issued 12 ds_read_b128 and 4 ds_write_b128, ~170 LDS-pipe cycles per stage
// and wave, 8 waves per CU -> ~1350 cycles per stage round against 512 of MFMA time.  The row
// operand does not have to pass through LDS at all if the copy is stored the way the matrix core
// wants it:
//     packed[row tile of 32][stage of 64 dims][quad j of 16 dims][lane = r + 32 h][8 halves]
//         = d16[32*tile + r][64*stage + 16*j + 8*h .. + 8)
// i.e. every 1 KiB block is the register image of one v_mfma_f32_32x32x16_f16 A operand.  A wave
// then loads its fragments with fully coalesced 1 KiB global loads straight into the register
// ring it multiplies from; LDS only serves the query fragments (NQ reads per quad, no writes, no
// stage tile -- so 64 queries per pass fit at dim 1024 as well).  Rows past n_docs inside the last
// tile are zeros in the copy and masked in the epilogue as before.  NQ = 3 (96 queries per pass,
// 144 KiB of LDS at dim 768, exactly 256 VGPRs) cuts the row bytes per MFMA by a third: the
// host picks it when the batch's tile count then fills the CUs better (f16_pick_nq).
//
// One wave works on TWO row tiles at a time (64 rows x 64 queries, 4 accumulator tiles): a query
// fragment read from LDS then feeds 2 MFMAs per sub-tile instead of 1.  With one row tile per
// wave the 8 waves of a CU issued one ds_read_b128 per MFMA, which is exactly the LDS array's
// rate (256 B/clk/CU) when the matrix pipe is full -- both ran at ~50 %.  The register ring holds
// 2 stages of both tiles (the same 16 KiB in flight per wave; a deeper ring measured slower).
#pragma once

namespace thr {

template <int DIM, int MODE, bool nt_loads, int NQ>
__global__ __launch_bounds__(H_THREADS) void dense_scan_f16p(
    const f32x4* __restrict__ packed, const float* __restrict__ inv_norm, int64_t n_docs,
    const float* __restrict__ queries, int n_queries, int64_t n_tiles, int64_t tile_stride,
    const float* __restrict__ tau, int* __restrict__ tile_cnt, Cand* __restrict__ tile_list,
    int tile_cap, float* __restrict__ sample_scores, int64_t sample_ld) {
    constexpr int QT = 32 * NQ;
    constexpr int CPR = DIM / 8;   // 16-byte chunks (8 halves) per query row
    constexpr int NS = DIM / 64;   // stages per row tile
    constexpr int NG = NS / 4;     // groups of 4 stages; ring slots alternate 0,1,0,1 inside a group
    constexpr int QBITS = 32 - ROW_BITS_F16;
    static_assert(DIM % 256 == 0 && NG >= 2, "f16 scan needs dim % 256 == 0 and dim >= 512");
    static_assert(QT <= (1 << QBITS), "query-in-tile index must fit the packed candidate word");
    extern __shared__ float4 lds_q[];  // [QT][CPR] f16 queries | H_WAVES wbufs

    const ScanSlot slot = scan_slot((n_queries + QT - 1) / QT);
    const int qtile = slot.qtile;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    f32x4* lds_v = reinterpret_cast<f32x4*>(lds_q);
    Cand* wbuf = reinterpret_cast<Cand*>(lds_q + QT * CPR) + wave * WBUF;
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

    // query tile: float32 -> float16 (round to nearest even), swizzled like the other scans
    for (int i = threadIdx.x; i < QT * CPR; i += H_THREADS) {
        const int q = i / CPR, c = i % CPR;
        const int qg = qtile * QT + q;
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
        if (qg < n_queries) {
            const float4* src = reinterpret_cast<const float4*>(queries + (int64_t)qg * DIM + 8 * c);
            const float4 lo = src[0], hi = src[1];
            v[0] = (_Float16)lo.x; v[1] = (_Float16)lo.y; v[2] = (_Float16)lo.z; v[3] = (_Float16)lo.w;
            v[4] = (_Float16)hi.x; v[5] = (_Float16)hi.y; v[6] = (_Float16)hi.z; v[7] = (_Float16)hi.w;
        }
        lds_v[mf_qslot(q, c, CPR)] = __builtin_bit_cast(f32x4, v);
    }
    __syncthreads();

    float my_tau[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s)
        my_tau[s] = MODE == MODE_FILTER ? tau[qtile * QT + 32 * s + r] : 0.f;
    const int64_t wave_id = (int64_t)slot.slice * H_WAVES + wave;
    const int64_t wave_stride = (int64_t)slot.nslices * H_WAVES;

    // 16-byte index of this lane's fragment of (visited tile t, stage 0, quad 0); +64 per quad.
    // A wave's unit of work is the PAIR of visited tiles (2P, 2P+1); a missing second tile (odd
    // tile count) is loaded as a duplicate of the first and never emitted.
    auto frag_off = [&](int64_t t) -> int64_t { return t * tile_stride * (NS * 256) + lane; };
    const int64_t n_pairs = (n_tiles + 1) / 2;
    const uint32_t q_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds_v;
    uint32_t qrow[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) qrow[s] = q_lds + (uint32_t)((32 * s + r) * CPR) * 16u;
    int qlow[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) qlow[x] = (((x >> 2) * 8 + 2 * (x & 3) + h) ^ r) & 15;

#define THR_PIN(x) asm volatile("" : "+v"(x))
#define PS_LD(ptr) (nt_loads ? __builtin_nontemporal_load(&packed[ptr]) : packed[ptr])
    // the four row fragments of one stage of both tiles, then on to the next stage
#define PS_LOAD(dst)                                                   \
    THR_PIN(pa);                                                       \
    dst[0] = PS_LD(pa); dst[1] = PS_LD(pa + 64); dst[2] = PS_LD(pa + 128); dst[3] = PS_LD(pa + 192); \
    pa += 256;                                                         \
    THR_PIN(pb);                                                       \
    dst[4] = PS_LD(pb); dst[5] = PS_LD(pb + 64); dst[6] = PS_LD(pb + 128); dst[7] = PS_LD(pb + 192); \
    pb += 256;
    // query fragments of one quad (one per sub-tile); qoff = chunk offset of the stage's group
#define PS_READ(B, qoff, par, quad)                                                            \
    {                                                                                          \
        const uint32_t cb = (uint32_t)((qoff) + qlow[(par) * 4 + (quad)]) * 16u;               \
        asm volatile("ds_read_b128 %0, %1" : "=v"(B[0]) : "v"(qrow[0] + cb) : "memory");       \
        if constexpr (NQ > 1)                                                                  \
            asm volatile("ds_read_b128 %0, %1" : "=v"(B[1]) : "v"(qrow[1] + cb) : "memory");   \
        if constexpr (NQ > 2)                                                                  \
            asm volatile("ds_read_b128 %0, %1" : "=v"(B[2]) : "v"(qrow[2] + cb) : "memory");   \
    }
    // the other pending set (NQ reads) is the only thing younger than the set waited for
#define PS_WAIT(B)                                                                              \
    if constexpr (NQ == 3)                                                                      \
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]) : : "memory"); \
    else if constexpr (NQ == 2)                                                                 \
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(B[0]), "+v"(B[1]) : : "memory");             \
    else                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(B[0]) : : "memory");
#define PS_MMA1(ACC, A, B)                                                                      \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, A),               \
                                                    __builtin_bit_cast(half8, B[0]), ACC[0], 0, 0, 0); \
    if constexpr (NQ > 1)                                                                       \
        ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                        \
            __builtin_bit_cast(half8, A), __builtin_bit_cast(half8, B[1]), ACC[1], 0, 0, 0);    \
    if constexpr (NQ > 2)                                                                       \
        ACC[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                        \
            __builtin_bit_cast(half8, A), __builtin_bit_cast(half8, B[2]), ACC[2], 0, 0, 0);
#define PS_MMA(A0, A1, B)                                                                       \
    PS_MMA1(acc_a, A0, B) PS_MMA1(acc_b, A1, B)                                                 \
    asm volatile("" : "+v"(acc_a[0]));
    // a stage multiplies from its ring slot (tile a: [0..3], tile b: [4..7]) and refills the slot
    // with the stage 2 ahead
#define PS_STAGE(u, ringc, qcur, qnxt)                                                  \
    PS_WAIT(b0) PS_MMA(ringc[0], ringc[4], b0) PS_READ(b0, qcur, (u) & 1, 2)            \
    PS_WAIT(b1) PS_MMA(ringc[1], ringc[5], b1) PS_READ(b1, qcur, (u) & 1, 3)            \
    PS_WAIT(b0) PS_MMA(ringc[2], ringc[6], b0) PS_READ(b0, qnxt, ((u) + 1) & 1, 0)      \
    PS_WAIT(b1) PS_MMA(ringc[3], ringc[7], b1) PS_READ(b1, qnxt, ((u) + 1) & 1, 1)      \
    PS_LOAD(ringc)

    f32x4 b0[NQ], b1[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) b0[s] = b1[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ring0[8], ring1[8];
    int64_t pa = 0, pb = 0;
    int64_t P = wave_id;
    auto tile_b = [&](int64_t pr) -> int64_t { return 2 * pr + 1 < n_tiles ? 2 * pr + 1 : 2 * pr; };
    if (P < n_pairs) {
        pa = frag_off(2 * P);
        pb = frag_off(tile_b(P));
        PS_LOAD(ring0)
        PS_LOAD(ring1)
        PS_READ(b0, 0, 0, 0)
        PS_READ(b1, 0, 0, 1)
    }
    for (; P < n_pairs; P += wave_stride) {
        const int64_t ta = 2 * P, tb = 2 * P + 1;
        const bool has_b = tb < n_tiles;
        const int64_t row0a = ta * tile_stride * MF_ROWS, row0b = (has_b ? tb : ta) * tile_stride * MF_ROWS;
        int ia = r, ib = r;
        if (row0a + ia >= n_docs) ia = (int)(n_docs - 1 - row0a);
        if (row0b + ib >= n_docs) ib = (int)(n_docs - 1 - row0b);
        THR_PIN(ia);
        THR_PIN(ib);
        const float inv_a = inv_norm[row0a + ia], inv_b = inv_norm[row0b + ib];
        const int64_t Pn = P + wave_stride < n_pairs ? P + wave_stride : P;
        const int64_t pna = frag_off(2 * Pn), pnb = frag_off(tile_b(Pn));
        f32x16 acc_a[NQ], acc_b[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_a[s][i] = acc_b[s][i] = 0.f;
        // kept rolled: unrolled, every (stage, quad) LDS address becomes its own hoisted VGPR
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int qb = 32 * g;                      // chunk offset of stages 4g, 4g+1
            const int qn = g + 1 < NG ? qb + 32 : 0;    // first chunks of the next group / tile
            PS_STAGE(0, ring0, qb, qb)
            PS_STAGE(1, ring1, qb, qb + 16)
            // the refill issued in stage u is stage 4g+u+2: from u = 2 of the last group on, that
            // is the head of the wave's next pair of row tiles
            if (g == NG - 1) { pa = pna; pb = pnb; }
            PS_STAGE(2, ring0, qb + 16, qb + 16)
            PS_STAGE(3, ring1, qb + 16, qn)
        }
        auto emit = [&](const f32x16* acc, int64_t t, int64_t row0, float my_inv) {
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
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                        const float inv = __shfl(my_inv, row, WAVE);
                        const bool ok = row0 + row < n_docs;
                        const float sc = acc[s][i] * inv;
                        const bool pass = ok && inv > 0.f && sc >= my_tau[s];
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
        };
        emit(acc_a, ta, row0a, inv_a);
        if (has_b) emit(acc_b, tb, row0b, inv_b);
    }
#undef PS_STAGE
#undef PS_MMA
#undef PS_MMA1
#undef PS_WAIT
#undef PS_READ
#undef PS_LOAD
#undef PS_LD
#undef THR_PIN
    if constexpr (MODE == MODE_FILTER) {
        if (wcnt > 0) flush();
    }
}

// float32 corpus -> fragment-major float16 copy (round to nearest even) + the largest relative
// row error max_d ||d16 - d|| / ||d||, accumulated as ordered float bits with atomicMax.
// packed == nullptr: measure only.  One wave per row; the copy must have been zeroed (tail rows).
__global__ __launch_bounds__(256) void quantize_f16(const float* __restrict__ docs, int64_t n_docs,
                                                    int dim, _Float16* __restrict__ packed,
                                                    unsigned int* __restrict__ max_rel_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6);
    if (row >= n_docs) return;
    const float* x = docs + row * dim;
    const int64_t tile = row >> 5;
    const int r = (int)(row & 31), ns = dim / 64;
    double err = 0.0, nrm = 0.0;
    for (int i = lane; i < dim; i += WAVE) {
        const float v = x[i];
        const _Float16 hv = (_Float16)v;
        if (packed) {
            const int s = i >> 6, j = (i >> 4) & 3, hh = (i >> 3) & 1, e = i & 7;
            packed[((((tile * ns + s) * 4 + j) * 64) + r + 32 * hh) * 8 + e] = hv;
        }
        const double d = (double)v - (double)(float)hv;
        err += d * d;
        nrm += (double)v * (double)v;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        err += __shfl_xor(err, m, WAVE);
        nrm += __shfl_xor(nrm, m, WAVE);
    }
    if (lane == 0 && nrm > 0.0) {
        // round the ratio UP to float so the stored bound is never below the true one
        float rel = (float)sqrt(err / nrm);
        rel = __uint_as_float(__float_as_uint(rel) + 1u);
        atomicMax(max_rel_bits, __float_as_uint(rel));  // positive floats order like their bits
    }
}

}  // namespace thr

