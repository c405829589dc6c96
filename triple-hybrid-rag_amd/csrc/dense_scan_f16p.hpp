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
// tile are zeros in the copy and masked in the epilogue as before.
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
    // register ring depth = stages per unrolled group (16 KiB per wave in flight).  A ring of 6
    // (possible at dim 768) measured slower, 1.90 vs 1.81 ms: in-flight depth is not the limit.
    constexpr int RING = 4;
    constexpr int NG = NS / RING;
    constexpr int QBITS = 32 - ROW_BITS_F16;
    static_assert(DIM % 256 == 0 && NG >= 2, "f16 scan needs dim % 256 == 0 and dim >= 512");
    static_assert(NS % RING == 0, "whole groups per row tile");
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

    // 16-byte index of this lane's fragment of (tile t, stage 0, quad 0); +64 per quad
    auto frag_off = [&](int64_t t) -> int64_t { return t * tile_stride * (NS * 256) + lane; };
    const uint32_t q_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds_v;
    uint32_t qrow[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) qrow[s] = q_lds + (uint32_t)((32 * s + r) * CPR) * 16u;
    int qlow[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) qlow[x] = (((x >> 2) * 8 + 2 * (x & 3) + h) ^ r) & 15;

#define THR_PIN(x) asm volatile("" : "+v"(x))
#define PS_LD(ptr) (nt_loads ? __builtin_nontemporal_load(&packed[ptr]) : packed[ptr])
    // the four row fragments of one stage, then on to the next stage
#define PS_LOAD(dst)                                                   \
    THR_PIN(p);                                                        \
    dst[0] = PS_LD(p); dst[1] = PS_LD(p + 64); dst[2] = PS_LD(p + 128); dst[3] = PS_LD(p + 192); \
    p += 256;
    // query fragments of one quad (one per sub-tile); qoff = chunk offset of the stage's group
#define PS_READ(B, qoff, par, quad)                                                            \
    {                                                                                          \
        const uint32_t cb = (uint32_t)((qoff) + qlow[(par) * 4 + (quad)]) * 16u;               \
        asm volatile("ds_read_b128 %0, %1" : "=v"(B[0]) : "v"(qrow[0] + cb) : "memory");       \
        if constexpr (NQ > 1)                                                                  \
            asm volatile("ds_read_b128 %0, %1" : "=v"(B[NQ - 1]) : "v"(qrow[NQ - 1] + cb) : "memory"); \
    }
    // the other pending set (NQ reads) is the only thing younger than the set waited for
#define PS_WAIT(B)                                                                              \
    if constexpr (NQ > 1)                                                                       \
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(B[0]), "+v"(B[NQ - 1]) : : "memory");        \
    else                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(B[0]) : : "memory");
#define PS_MMA(A, B)                                                                            \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, A),               \
                                                    __builtin_bit_cast(half8, B[0]), acc[0], 0, 0, 0); \
    if constexpr (NQ > 1)                                                                       \
        acc[NQ - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                   \
            __builtin_bit_cast(half8, A), __builtin_bit_cast(half8, B[NQ - 1]), acc[NQ - 1], 0, 0, 0); \
    asm volatile("" : "+v"(acc[0]));
    // stage u of a group multiplies from ring slot u and refills it with the stage 4 ahead
#define PS_STAGE(u, ringc, qcur, qnxt)                                        \
    PS_WAIT(b0) PS_MMA(ringc[0], b0) PS_READ(b0, qcur, (u) & 1, 2)            \
    PS_WAIT(b1) PS_MMA(ringc[1], b1) PS_READ(b1, qcur, (u) & 1, 3)            \
    PS_WAIT(b0) PS_MMA(ringc[2], b0) PS_READ(b0, qnxt, ((u) + 1) & 1, 0)      \
    PS_WAIT(b1) PS_MMA(ringc[3], b1) PS_READ(b1, qnxt, ((u) + 1) & 1, 1)      \
    PS_LOAD(ringc)

    f32x4 b0[NQ], b1[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) b0[s] = b1[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ring0[4], ring1[4], ring2[4], ring3[4], ring4[4], ring5[4];
    int64_t p = 0;
    int64_t t = wave_id;
    if (t < n_tiles) {
        p = frag_off(t);
        PS_LOAD(ring0)
        PS_LOAD(ring1)
        PS_LOAD(ring2)
        PS_LOAD(ring3)
        if constexpr (RING == 6) {
            PS_LOAD(ring4)
            PS_LOAD(ring5)
        }
        PS_READ(b0, 0, 0, 0)
        PS_READ(b1, 0, 0, 1)
    }
    for (; t < n_tiles; t += wave_stride) {
        const int64_t row0 = t * tile_stride * MF_ROWS;
        int idx = r;
        if (row0 + idx >= n_docs) idx = (int)(n_docs - 1 - row0);
        THR_PIN(idx);
        const float my_inv = inv_norm[row0 + idx];
        const int64_t tn = t + wave_stride < n_tiles ? t + wave_stride : t;
        const int64_t pn = frag_off(tn);
        f32x16 acc[NQ];
#pragma unroll
        for (int s = 0; s < NQ; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s][i] = 0.f;
        // kept rolled: unrolled, every (stage, quad) LDS address becomes its own hoisted VGPR
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            // chunk offset of stage RING*g + u is qb + 16 * (u >> 1) (RING is even)
            const int qb = 8 * RING * g;
            const int qn = g + 1 < NG ? qb + 8 * RING : 0;  // first chunks of the next group / tile
            if (g == NG - 1) p = pn;                    // the last group refills for the next tile
            PS_STAGE(0, ring0, qb, qb)
            PS_STAGE(1, ring1, qb, qb + 16)
            PS_STAGE(2, ring2, qb + 16, qb + 16)
            if constexpr (RING == 4) {
                PS_STAGE(3, ring3, qb + 16, qn)
            } else {
                PS_STAGE(3, ring3, qb + 16, qb + 32)
                PS_STAGE(4, ring4, qb + 32, qb + 32)
                PS_STAGE(5, ring5, qb + 32, qn)
            }
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
    }
#undef PS_STAGE
#undef PS_MMA
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
