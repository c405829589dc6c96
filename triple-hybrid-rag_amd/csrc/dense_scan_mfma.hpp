// Streaming scan on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// Why MFMA for an HBM-bound scan: with 32 queries per corpus pass the scan needs
// 2*32*768 flop per 3 KiB row = 16 flop/B, i.e. ~98 TFLOP/s at the HBM rate.  The
// VALU version of this kernel (dense_scan<> in dense.hip) issues one v_fmac per 128
// flop and tops out at ~51 TFLOP/s (3.2 TB/s, profiles/r1_bench_valu_qt32_*): it is
// VALU-issue-bound, not bandwidth-bound.  v_mfma_f32_32x32x2_f32 has the same nominal
// rate (64 flop/clk/SIMD) but is ONE instruction per 4096 flop, and its result is an
// exact k-ordered f32 fma chain (MI355X guide, "FP32-input MFMA"), so the numerics
// contract (fp32 shortlist + float64 rescoring + error-bound certificate) is unchanged.
// This is still a row-streaming kernel: each doc row is read exactly once per pass.
//
// Tile: one wave owns 32 consecutive rows x the tile's 32 queries (16 accumulator
// VGPRs: query on the lane, rows in the registers).  k is only a summation index, so
// the dims are dealt to the two k-slots of the MFMA such that lane (r, h) loads the
// 16 bytes [8j + 4h, +4) of row r: every global_load_dwordx4 covers 32 rows x 32
// contiguous bytes and four consecutive j's consume each 128-B line completely.
// The query tile sits in LDS, XOR-swizzled by (query & 15) on the 16-byte chunk index
// so that the 32 lanes of a ds_read_b128 (one query row each, stride 3 KiB) hit
// distinct bank groups.
#pragma once
#include <type_traits>

namespace thr {

// compile-time loop: the body sees its index as a constant expression, so register arrays
// indexed with it stay in registers (a "#pragma unroll" the compiler declines would demote
// them to scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_THREADS = 512;
constexpr int MF_WAVES = MF_THREADS / WAVE;
constexpr int MF_ROWS = 32;   // rows per wave tile
constexpr int MF_QT = 32;     // queries per tile pass
constexpr int MF_NPF = 8;     // float4 row chunks kept in flight per lane (8 KiB per wave)

// swizzled float4 index of 16-byte chunk `cidx` of query row q (row length D8*2 chunks)
__device__ __forceinline__ int mf_qslot(int q, int cidx, int chunks_per_row) {
    return q * chunks_per_row + ((cidx & ~15) | ((cidx ^ q) & 15));
}

template <int D8, int MODE>  // D8 = dim / 8
__global__ __launch_bounds__(MF_THREADS) void dense_scan_mfma(
    const float* __restrict__ docs, const float* __restrict__ inv_norm, int64_t n_docs,
    const float* __restrict__ queries, int n_queries,
    int64_t n_tiles,      // row tiles (of 32 rows) this launch visits
    int64_t tile_stride,  // actual row tile = visited index * tile_stride
    const float* __restrict__ tau, int* __restrict__ tile_cnt, Cand* __restrict__ tile_list,
    int tile_cap, float* __restrict__ sample_scores, int64_t sample_ld,
    const int32_t* __restrict__ doc_coll = nullptr, const int32_t* __restrict__ query_coll = nullptr) {
    constexpr int D = D8 * 8;
    constexpr int CPR = D / 4;  // 16-byte chunks per row
    extern __shared__ float4 lds_q[];  // [MF_QT][CPR] swizzled, then MF_WAVES * WBUF staging slots

    const ScanSlot slot = scan_slot((n_queries + MF_QT - 1) / MF_QT);
    const int qtile = slot.qtile;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    Cand* wbuf = reinterpret_cast<Cand*>(lds_q + MF_QT * CPR) + wave * WBUF;
    int wcnt = 0;
    auto flush = [&]() {
        int base = 0;
        if (lane == 0) base = atomicAdd(&tile_cnt[qtile], wcnt);
        base = __shfl(base, 0, WAVE);
        for (int i = lane; i < wcnt; i += WAVE)
            if (base + i < tile_cap) tile_list[(int64_t)qtile * tile_cap + base + i] = wbuf[i];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): only loads pending afterwards
        wcnt = 0;
    };

    for (int i = threadIdx.x; i < MF_QT * CPR; i += MF_THREADS) {
        const int q = i / CPR, c = i % CPR;
        const int qg = qtile * MF_QT + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qg < n_queries) v = reinterpret_cast<const float4*>(queries)[(int64_t)qg * CPR + c];
        lds_q[mf_qslot(q, c, CPR)] = v;
    }
    __syncthreads();

    const float my_tau = MODE == MODE_FILTER ? tau[qtile * MF_QT + r] : 0.f;
    // collection filter of this lane's query (-1: none), applied as rows pass tau: tau was taken
    // from the sample rows of THAT collection, so without the filter here a collection of 1/c of
    // the corpus lets c times the aimed candidates through and overflows the tile list
    const int my_qc = (MODE == MODE_FILTER && query_coll && qtile * MF_QT + r < n_queries)
                          ? query_coll[qtile * MF_QT + r] : -1;
    const int64_t wave_id = (int64_t)slot.slice * MF_WAVES + wave;
    const int64_t wave_stride = (int64_t)slot.nslices * MF_WAVES;
    const float4* docs4 = reinterpret_cast<const float4*>(docs);

#define THR_PIN(x) asm volatile("" : "+v"(x))
    // float4 offset of this lane's 16-byte column in the tile's rows (row clamped at the tail)
    auto lane_off = [&](int64_t t) -> int64_t {
        int64_t row = t * tile_stride * MF_ROWS + r;
        row = row < n_docs ? row : n_docs - 1;
        return row * CPR + h;
    };

    float4 a[MF_NPF];
    int64_t t = wave_id;
    int64_t off = 0;
    if (t < n_tiles) {
        off = lane_off(t);
#pragma unroll
        for (int j = 0; j < MF_NPF; ++j) {
            THR_PIN(off);
            a[j] = docs4[off + 2 * j];
        }
    }
    for (; t < n_tiles; t += wave_stride) {
        const int64_t row0 = t * tile_stride * MF_ROWS;
        int idx = r;
        if (row0 + idx >= n_docs) idx = (int)(n_docs - 1 - row0);
        THR_PIN(idx);
        const float my_inv = inv_norm[row0 + idx];  // oldest load in the queue at emission time
        const int64_t tn = t + wave_stride < n_tiles ? t + wave_stride : t;
        const int64_t off_next = lane_off(tn);

        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        int qb = r * CPR;  // LDS row base of this lane's query
        THR_PIN(qb);
        float4 b0 = lds_q[qb + (((0 * 2 + h) ^ r) & 15)];
#pragma unroll
        for (int j = 0; j < D8; ++j) {
            const float4 bv = b0;
            if (j + 1 < D8) {
                const int c = 2 * (j + 1) + h;
                THR_PIN(qb);
                b0 = lds_q[qb + ((c & ~15) | ((c ^ r) & 15))];
            }
            const float4 av = a[j % MF_NPF];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            // refill the slot just consumed: this tile's chunk j+NPF, or the next tile's head.
            // Tying the address to the accumulator keeps the load below the MFMAs that read
            // the slot (otherwise hipcc renames the slot and hoists every load of the tile).
            if (j + MF_NPF < D8) {
                asm volatile("" : "+v"(off), "+v"(acc));
                a[j % MF_NPF] = docs4[off + 2 * (j + MF_NPF)];
            } else {
                int64_t o2 = off_next;
                asm volatile("" : "+v"(o2), "+v"(acc));
                a[j % MF_NPF] = docs4[o2 + 2 * (j + MF_NPF - D8)];
            }
        }
        off = off_next;

        // accumulator layout: column (query) = lane & 31, row = (i&3) + 8*(i>>2) + 4*h
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            const float inv = __shfl(my_inv, row, WAVE);
            const bool ok = row0 + row < n_docs;
            const float sc = acc[i] * inv;
            if constexpr (MODE == MODE_ALL) {
                const int qg = qtile * MF_QT + r;
                sample_scores[(int64_t)qg * sample_ld + t * MF_ROWS + row] =
                    (ok && inv > 0.f) ? sc : -INFINITY;
            } else {
                bool pass = ok && inv > 0.f && sc >= my_tau;
                if (pass && my_qc != -1 && doc_coll[row0 + row] != my_qc) pass = false;
                const uint64_t m = __ballot(pass);
                if (m) {
                    const int pos = wcnt + __popcll(m & ((1ull << lane) - 1ull));
                    if (pass)
                        wbuf[pos] = Cand{sc, ((uint32_t)r << ROW_BITS) | (uint32_t)(row0 + row)};
                    wcnt += __popcll(m);
                    if (wcnt > WBUF - WAVE) flush();  // the next ballot can add 64 more
                }
            }
        }
    }
#undef THR_PIN
    if constexpr (MODE == MODE_FILTER) {
        if (wcnt > 0) flush();
    }
}

// ---------------------------------------------------------------------------
// v2: same tile and the same dim->k-slot assignment (identical fp32 results), but rows
// reach the matrix cores through a per-wave LDS transpose tile:
//   global_load_dwordx4, 8 lanes per row = one full 128-B line per 8 lanes (v1's
//   fragment-shaped loads touch 16 lines per quarter-wave and are L1-tag-bound)
//   -> register ring, 4 stages (16 KiB per wave) in flight -> ds_write_b128 into the
//   wave's own [32 rows][32 dims] tile (XOR-swizzled) -> ds_read_b128 A fragments -> 16 MFMAs.
// The tile is wave-private: no workgroup barrier anywhere in the loop.  The loop is a
// software pipeline over stages (32 dims each), continuous across row tiles:
//     stage s:  MMA q0 | read q2 | MMA q1 | read q3 | write stage s+1 to LDS, refill its
//               ring slot with stage s+5 | MMA q2 | read (s+1).q0 | MMA q3 | read (s+1).q1
// LDS operations of one wave execute in order, so writing stage s+1 over the tile after the
// last fragment reads of stage s have been ISSUED is safe, and every fragment is requested a
// full MFMA quad (256 cycles) before it is used.  Stages run in groups of 4 (= ring depth)
// so every register-array index is a constant.
// ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: plain SSA loads/stores
constexpr int MF2_AHEAD = 4;               // ring depth: stages (4 KiB each) in flight per wave
constexpr int MF2_STAGE_F4 = MF_ROWS * 8;  // float4 slots per stage tile (32 rows x 8 chunks)

// float4 slot of (row, chunk) inside a stage tile: chunk ^ ((row >> 1) & 7)
__device__ __forceinline__ int mf2_slot(int row, int chunk) {
    return row * 8 + (chunk ^ ((row >> 1) & 7));
}

template <int D8, int MODE, bool nt_loads, int NW>  // NW = waves per workgroup
__global__ __launch_bounds__(NW * WAVE) void dense_scan_mfma2(
    const float* __restrict__ docs, const float* __restrict__ inv_norm, int64_t n_docs,
    const float* __restrict__ queries, int n_queries, int64_t n_tiles, int64_t tile_stride,
    const float* __restrict__ tau, int* __restrict__ tile_cnt, Cand* __restrict__ tile_list,
    int tile_cap, float* __restrict__ sample_scores, int64_t sample_ld,
    const int32_t* __restrict__ doc_coll = nullptr, const int32_t* __restrict__ query_coll = nullptr) {
    constexpr int D = D8 * 8;
    constexpr int CPR = D / 4;
    constexpr int NG = D / 128;  // groups of 4 stages of 32 dims
    static_assert(D % 128 == 0 && NG >= 2, "dim must be a multiple of 128, >= 256");
    extern __shared__ float4 lds_q[];  // [32][CPR] queries | NW stage tiles | NW wbufs

    const ScanSlot slot = scan_slot((n_queries + MF_QT - 1) / MF_QT);
    const int qtile = slot.qtile;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    // (HIP's float4 is a class: copying one from a local array into LDS is emitted as a
    // memcpy out of a stack slot, which pins the whole register ring in scratch memory)
    f32x4* lds_v = reinterpret_cast<f32x4*>(lds_q);
    f32x4* stage = lds_v + MF_QT * CPR + wave * MF2_STAGE_F4;
    Cand* wbuf = reinterpret_cast<Cand*>(lds_q + MF_QT * CPR + NW * MF2_STAGE_F4) + wave * WBUF;
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

    for (int i = threadIdx.x; i < MF_QT * CPR; i += (NW * WAVE)) {
        const int q = i / CPR, c = i % CPR;
        const int qg = qtile * MF_QT + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qg < n_queries) v = reinterpret_cast<const float4*>(queries)[(int64_t)qg * CPR + c];
        lds_q[mf_qslot(q, c, CPR)] = v;
    }
    __syncthreads();

    const float my_tau = MODE == MODE_FILTER ? tau[qtile * MF_QT + r] : 0.f;
    // collection filter of this lane's query (-1: none), applied as rows pass tau: tau was taken
    // from the sample rows of THAT collection, so without the filter here a collection of 1/c of
    // the corpus lets c times the aimed candidates through and overflows the tile list
    const int my_qc = (MODE == MODE_FILTER && query_coll && qtile * MF_QT + r < n_queries)
                          ? query_coll[qtile * MF_QT + r] : -1;
    const int64_t wave_id = (int64_t)slot.slice * NW + wave;
    const int64_t wave_stride = (int64_t)slot.nslices * NW;
    const f32x4* docs4 = reinterpret_cast<const f32x4*>(docs);

    // loader role of this lane: row (lane >> 3) + 8*i of the tile, 16-byte chunk (lane & 7)
    const int lrow = lane >> 3, lchunk = lane & 7;
    auto load_off = [&](int64_t t, int i) -> int64_t {
        int64_t row = t * tile_stride * MF_ROWS + lrow + 8 * i;
        row = row < n_docs ? row : n_docs - 1;
        return row * CPR + lchunk;
    };
    // LDS slots this lane writes (one per i) / reads as A fragment (one per quad); and the
    // swizzled low part of the query-chunk index for (stage parity, quad)
    int wslot0 = mf2_slot(lrow, lchunk), wslot1 = mf2_slot(lrow + 8, lchunk);
    int wslot2 = mf2_slot(lrow + 16, lchunk), wslot3 = mf2_slot(lrow + 24, lchunk);
    int rslot0 = mf2_slot(r, h), rslot1 = mf2_slot(r, 2 + h), rslot2 = mf2_slot(r, 4 + h),
        rslot3 = mf2_slot(r, 6 + h);
    int qlow[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) qlow[x] = (((x >> 2) * 8 + 2 * (x & 3) + h) ^ r) & 15;

#define THR_PIN(x) asm volatile("" : "+v"(x))
    // request chunk-stage data of the 4 loader rows at the running pointers, then advance them
    // (rows are read once per pass and never again by this CU: non-temporal loads)
#define MF2_LD(ptr) (nt_loads ? __builtin_nontemporal_load(&docs4[ptr]) : docs4[ptr])
#define MF2_LOAD(dst)                                       \
    THR_PIN(p0); dst[0] = MF2_LD(p0); p0 += 8;              \
    THR_PIN(p1); dst[1] = MF2_LD(p1); p1 += 8;              \
    THR_PIN(p2); dst[2] = MF2_LD(p2); p2 += 8;              \
    THR_PIN(p3); dst[3] = MF2_LD(p3); p3 += 8;
#define MF2_STORE(src)                              \
    THR_PIN(wslot0); stage[wslot0] = src[0];        \
    THR_PIN(wslot1); stage[wslot1] = src[1];        \
    THR_PIN(wslot2); stage[wslot2] = src[2];        \
    THR_PIN(wslot3); stage[wslot3] = src[3];
    // Fragment reads are issued by inline asm and retired by hand-counted s_waitcnt: left to
    // hipcc, each ds_read sinks down to its first use and the wave stalls a full LDS round trip
    // before every MFMA quad.  "memory" clobbers keep hipcc's own LDS stores (the stage tile
    // writes) and these reads in program order; the wait takes the fragments as in/out operands
    // so the MFMAs that consume them cannot be hoisted above it.  LDS operations of one wave
    // retire in order, so lgkmcnt(N) with N = LDS ops issued AFTER the wanted read is exact
    // (anything else the compiler has in flight only makes the wait longer, never shorter).
#define MF2_READ(fa, fb, rs, qb16, par, quad)                                             \
    {                                                                                     \
        const uint32_t qaddr = q_lds + (uint32_t)((qb16) + qlow[(par) * 4 + (quad)]) * 16u; \
        const uint32_t aaddr = st_lds + (uint32_t)(rs) * 16u;                              \
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb) : "v"(qaddr) : "memory");            \
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa) : "v"(aaddr) : "memory");            \
    }
#define MF2_WAIT(n, fa, fb) \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(fa), "+v"(fb) : : "memory");
#define MF2_MMA(fa, fb)                                                           \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);         \
    asm volatile("" : "+v"(acc));
    // stage u of a group: this stage's query chunks at qcur (parity u&1), the next stage's at
    // qnxt (parity (u+1)&1); ringn = ring slot of stage s+1
#define MF2_STAGE(u, ringn, qcur, qnxt)                                     \
    MF2_WAIT(2, a0, b0) /* younger: the other pending fragment pair */      \
    MF2_MMA(a0, b0)                                                         \
    MF2_READ(a0, b0, rslot2, qcur, (u) & 1, 2)                              \
    MF2_WAIT(2, a1, b1)                                                     \
    MF2_MMA(a1, b1)                                                         \
    MF2_READ(a1, b1, rslot3, qcur, (u) & 1, 3)                              \
    MF2_STORE(ringn)                                                        \
    MF2_LOAD(ringn)                                                         \
    MF2_WAIT(6, a0, b0) /* younger: q3 pair + 4 stage-tile writes */        \
    MF2_MMA(a0, b0)                                                         \
    MF2_READ(a0, b0, rslot0, qnxt, ((u) + 1) & 1, 0)                        \
    MF2_WAIT(6, a1, b1) /* younger: 4 stage-tile writes + next q0 pair */   \
    MF2_MMA(a1, b1)                                                         \
    MF2_READ(a1, b1, rslot1, qnxt, ((u) + 1) & 1, 1)

    // LDS byte addresses for the asm reads
    const uint32_t q_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds_v;
    const uint32_t st_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)stage;
    f32x4 ring0[4], ring1[4], ring2[4], ring3[4];
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
    int64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    int64_t t = wave_id;
    const int qrow = r * CPR;
    if (t < n_tiles) {
        p0 = load_off(t, 0); p1 = load_off(t, 1); p2 = load_off(t, 2); p3 = load_off(t, 3);
        MF2_LOAD(ring0)
        MF2_LOAD(ring1)
        MF2_LOAD(ring2)
        MF2_LOAD(ring3)
        MF2_STORE(ring0)   // stage 0 of the first tile
        MF2_LOAD(ring0)    // <- stage 4
        MF2_READ(a0, b0, rslot0, qrow, 0, 0)
        MF2_READ(a1, b1, rslot1, qrow, 0, 1)
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

        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // kept rolled: unrolled, every (stage, quad) LDS address becomes its own hoisted VGPR
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int qb = qrow + 32 * g;                           // chunks of stages 4g, 4g+1
            const int qn = g + 1 < NG ? qb + 32 : qrow;             // first chunks of the next group
            MF2_STAGE(0, ring1, qb, qb)
            MF2_STAGE(1, ring2, qb, qb + 16)
            MF2_STAGE(2, ring3, qb + 16, qb + 16)
            // the refill issued in the 4th stage is stage 4g+8: at g == NG-2 that is the head of
            // the wave's NEXT row tile
            if (g == NG - 2) { p0 = on0; p1 = on1; p2 = on2; p3 = on3; }
            MF2_STAGE(3, ring0, qb + 16, qn)
        }

        if constexpr (MODE == MODE_ALL) {
            // accumulator registers 4g..4g+3 are 4 consecutive rows: one 16-byte store each
            const int qg = qtile * MF_QT + r;
            float* dst = sample_scores + (int64_t)qg * sample_ld + t * MF_ROWS + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = j + 8 * g + 4 * h;
                    const float inv = __shfl(my_inv, row, WAVE);
                    v[j] = (row0 + row < n_docs && inv > 0.f) ? acc[4 * g + j] * inv : -INFINITY;
                }
                *reinterpret_cast<f32x4*>(dst + 8 * g) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                const float inv = __shfl(my_inv, row, WAVE);
                const bool ok = row0 + row < n_docs;
                const float sc = acc[i] * inv;
                bool pass = ok && inv > 0.f && sc >= my_tau;
                if (pass && my_qc != -1 && doc_coll[row0 + row] != my_qc) pass = false;
                const uint64_t m = __ballot(pass);
                if (m) {
                    const int pos = wcnt + __popcll(m & ((1ull << lane) - 1ull));
                    if (pass)
                        wbuf[pos] = Cand{sc, ((uint32_t)r << ROW_BITS) | (uint32_t)(row0 + row)};
                    wcnt += __popcll(m);
                    if (wcnt > WBUF - WAVE) flush();
                }
            }
        }
    }
#undef MF2_STAGE
#undef MF2_MMA
#undef MF2_READ
#undef MF2_WAIT
#undef MF2_STORE
#undef MF2_LOAD
#undef MF2_LD
#undef THR_PIN
    if constexpr (MODE == MODE_FILTER) {
        if (wcnt > 0) flush();
    }
}

}  // namespace thr
