// Dense channel: exact brute-force cosine top-k on gfx950 (MI355X).
//
// Stands where the reference calls SQL rag2_semantic_search
// (database/migrations/20260114_rag2_schema.sql:377-410, from
// src/voice_agent/rag2/retrieval.py:304-312): `1 - (embedding <=> q)`
// ORDER BY distance LIMIT k.  The reference answers it from an approximate
// HNSW index inside PostgreSQL; this is the exact scan that index approximates.
//
// Pipeline per batch of queries (all kernels on one stream, no host sync):
//   K0 pack_queries_f16         (default scan) queries -> fragment-major f16 register image
//   K1 scan<MODE_ALL>           score a strided SAMPLE of row groups for every query tile
//   K2 kth_select               tau[q] ~ the ks-th largest sample score: about `aim` rows of
//                               the corpus will pass it
//   K3 scan<MODE_FILTER>        THE dominant kernel: stream the corpus once per query tile,
//                               emit (score, row) >= tau[q].  Default: dense_scan_f16qs (f16
//                               MFMA over the normalised f16 copy, queries in registers);
//                               dense_scan_f16 / dense_scan_mfma2 (dense_scan_mfma at dim 1024)
//                               are the other flavours
//   K4a select_band             per query: the band of candidates that can still reach the
//                               top-k -> a shortlist of rows
//   K4b rescore_rank            shortlist re-scored in float64 with sequential accumulation
//                               (the oracle's contract), sorted (score desc, row asc),
//                               certified with the scan's error bound
//   K5 thr_dense_rescue         uncertified queries redone exhaustively
// Algorithmic HBM bytes of K3 = n_docs * dim * 4 per tile pass.
#include <stdlib.h>

#include "thr_common.hpp"

namespace thr {

constexpr int CHUNK = 256;                 // floats per wave-wide float4 load (1 KiB)
constexpr int MODE_ALL = 0, MODE_FILTER = 1;
constexpr int CAND_CAP = 16384;            // candidates kept per query between K3 and K4
constexpr int SAMPLE_MAX = 1 << 20;        // upper bound of the sample (rows) for the tau estimate
constexpr int WBUF = 256;                  // per-wave LDS staging slots for passing rows
constexpr int ROW_BITS = 27;               // tile-list entries pack (query-in-tile << 27 | row)
constexpr uint32_t ROW_MASK = (1u << ROW_BITS) - 1;
constexpr int ROW_BITS_F16 = 25;           // f16 shortlist scans: up to 96 queries per tile -> 7 bits

struct Cand {
    float score;
    uint32_t doc;
};

}  // namespace thr
// ---------------------------------------------------------------------------
// Block -> (row slice, query tile) for the MFMA scans.
// Every query tile streams the same rows, so the launch is laid out for the 8 private L2s:
// workgroups are dealt round-robin over the XCDs (b and b+8 share one), and a 1-D grid of
// 8 * m * n_qtiles blocks is decoded so that the blocks resident together on one XCD are the
// n_qtiles query tiles of the SAME row slice.  They walk identical addresses in step: the
// first one to ask for a line pulls it from HBM, the others hit it in that XCD's L2.
// Placement is a speed matter only: any dispatch order gives the same result.
// ---------------------------------------------------------------------------
struct ScanSlot {
    int qtile, slice, nslices;
};
__device__ __forceinline__ ScanSlot scan_slot(int n_qtiles) {
    ScanSlot s;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    s.qtile = j % n_qtiles;
    s.slice = xcd + 8 * (j / n_qtiles);
    s.nslices = gridDim.x / n_qtiles;
    return s;
}

#include "dense_scan_mfma.hpp"
#include "dense_scan_f16.hpp"
#include "dense_scan_f16q.hpp"
namespace thr {

// K3b: split a tile's mixed candidate list into the per-query lists K4 reads.  Each block
// owns a contiguous slice of the list and reserves its output ranges with ONE global atomic
// per query (counts first, in LDS), instead of one returning global atomic per entry.
constexpr int BUCKET_BLOCKS = 32;
__global__ __launch_bounds__(256) void bucket_candidates(const int* __restrict__ tile_cnt,
                                                         const Cand* __restrict__ tile_list,
                                                         int tile_cap, int qtile, int row_bits,
                                                         int* __restrict__ cand_cnt,
                                                         Cand* __restrict__ cand) {
    __shared__ int count[128], base[128], fill[128];
    const uint32_t row_mask = (1u << row_bits) - 1u;
    const int tile = blockIdx.y;
    int n = tile_cnt[tile];
    n = n < tile_cap ? n : tile_cap;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const Cand* list = tile_list + (int64_t)tile * tile_cap;
    if (threadIdx.x < 128) count[threadIdx.x] = fill[threadIdx.x] = 0;
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x)
        atomicAdd(&count[list[i].doc >> row_bits], 1);
    __syncthreads();
    if (threadIdx.x < qtile && count[threadIdx.x] > 0)
        base[threadIdx.x] = atomicAdd(&cand_cnt[tile * qtile + threadIdx.x], count[threadIdx.x]);
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const Cand e = list[i];
        const int ql = (int)(e.doc >> row_bits);
        const int p = base[ql] + atomicAdd(&fill[ql], 1);
        if (p < CAND_CAP)
            cand[(int64_t)(tile * qtile + ql) * CAND_CAP + p] = Cand{e.score, e.doc & row_mask};
    }
}

// ---------------------------------------------------------------------------
// 8-bit-digit radix select of the kk-th largest key among n (block-wide).
// keyfn(i) -> uint32 order-preserving key.  Returns the key; *n_greater gets the
// number of keys strictly greater.  hist = 256 ints of LDS, bc = 4 ints of LDS.
// ---------------------------------------------------------------------------
template <typename KeyFn>
__device__ uint32_t block_radix_select(KeyFn keyfn, int n, int kk, int* hist, int* bc, int* n_greater) {
    uint32_t prefix = 0, mask = 0;
    int remaining = kk, greater = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        // 8 keys per thread per trip, loads issued together: one dependent load per
        // trip would make every pass a chain of memory round trips
        for (int base = threadIdx.x; base < n; base += 8 * blockDim.x) {
            uint32_t key[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * blockDim.x;
                key[u] = i < n ? keyfn(i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * blockDim.x;
                if (i < n && (key[u] & mask) == prefix) atomicAdd(&hist[(key[u] >> shift) & 255], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int cum = 0, b = 255;
            for (; b > 0; --b) {
                if (cum + hist[b] >= remaining) break;
                cum += hist[b];
            }
            bc[0] = b;
            bc[1] = cum;
        }
        __syncthreads();
        int b = bc[0];
        remaining -= bc[1];
        greater += bc[1];
        prefix |= (uint32_t)b << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    *n_greater = greater;
    return prefix;
}

// Same select over items each thread enumerates itself: keyfn(u), u in [0, my_n) (the
// candidate lists of select_band: a thread's items are my_ptr[u * my_stride]).
template <typename KeyFn>
__device__ uint32_t block_radix_select_local(KeyFn keyfn, int my_n, int kk, int* hist, int* bc) {
    uint32_t prefix = 0, mask = 0;
    int remaining = kk;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int u0 = 0; u0 < my_n; u0 += 8) {
            uint32_t key[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) key[u] = u0 + u < my_n ? keyfn(u0 + u) : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u0 + u < my_n && (key[u] & mask) == prefix) atomicAdd(&hist[(key[u] >> shift) & 255], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int cum = 0, b = 255;
            for (; b > 0; --b) {
                if (cum + hist[b] >= remaining) break;
                cum += hist[b];
            }
            bc[0] = b;
            bc[1] = cum;
        }
        __syncthreads();
        remaining -= bc[1];
        prefix |= (uint32_t)bc[0] << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    return prefix;
}

// ---------------------------------------------------------------------------
// Two-pass LOWER BOUND of the kk-th largest key: 12-bit digits over the top 24 key bits, the
// low 8 bits of the result are zero.  For a float key that is a value at most 2^-15 (relative)
// below the true kk-th -- all a threshold needs (it only has to let the top kk through), at
// half the passes of the exact select.  key(i) is evaluated for i in [0, n); the bins of a pass
// are searched by all threads (per-thread partial sums + one wave scan).
// hist = 4096 ints of LDS, aux = 8 ints of LDS.  blockDim.x must be 256.
// ---------------------------------------------------------------------------
constexpr int CS_BINS = 4096;
__device__ __forceinline__ void coarse_find_bin(const int* hist, int remaining, int* aux) {
    // thread t owns the 16 bins [4096 - 16(t+1), 4096 - 16t): t = 0 holds the largest keys
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int top = CS_BINS - 16 * t;
    int mine = 0;
#pragma unroll
    for (int b = 1; b <= 16; ++b) mine += hist[top - b];
    int incl = mine;  // inclusive scan over t within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += v;
    }
    if (lane == 63) aux[4 + w] = incl;
    if (t == 0) aux[0] = 0, aux[1] = -1;
    __syncthreads();
    int before = 0;
    for (int x = 0; x < w; ++x) before += aux[4 + x];
    incl += before;
    const int excl = incl - mine;
    if (excl < remaining && remaining <= incl) {
        int cum = excl, b = top - 1;
        for (; b > top - 16; --b) {
            if (cum + hist[b] >= remaining) break;
            cum += hist[b];
        }
        aux[0] = b;
        aux[1] = cum;
    }
    __syncthreads();
    if (aux[1] < 0) {  // fewer than `remaining` keys in all: bin 0 (cannot happen for kk <= n)
        if (t == 255) aux[0] = 0, aux[1] = incl - hist[0];
        __syncthreads();
    }
}

template <typename KeyFn>
__device__ uint32_t block_coarse_select(KeyFn keyfn, int n, int kk, int* hist, int* aux) {
    uint32_t prefix = 0;
    int remaining = kk;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = threadIdx.x; i < CS_BINS; i += 256) hist[i] = 0;
        __syncthreads();
        for (int base = threadIdx.x; base < n; base += 8 * 256) {
            uint32_t key[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256;
                key[u] = i < n ? keyfn(i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256;
                if (i >= n) continue;
                if (pass == 0) atomicAdd(&hist[key[u] >> 20], 1);
                else if ((key[u] >> 20) == (prefix >> 20)) atomicAdd(&hist[(key[u] >> 8) & 4095], 1);
            }
        }
        __syncthreads();
        coarse_find_bin(hist, remaining, aux);
        remaining -= aux[1];
        prefix |= (uint32_t)aux[0] << (pass == 0 ? 20 : 8);
        __syncthreads();
    }
    return prefix;
}
// float value of a truncated key; a truncated -inf key decodes to NaN: map it back
__device__ __forceinline__ float coarse_value(uint32_t key) {
    const float v = fkey_inv(key);
    return v == v ? v : -INFINITY;
}

// true when the block's query is a padding row of its tile or an all-zero vector: such a query
// must never emit (every row ties at 0 and would flood the tile's candidate list); a real zero
// query therefore ends up uncertified and is answered by the exhaustive path.
__device__ bool query_is_void(const float* __restrict__ queries, int n_queries, int dim, int q,
                              int* flag) {
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (q < n_queries) {
        int nz = 0;
        for (int i = threadIdx.x; i < dim; i += blockDim.x) nz |= queries[(int64_t)q * dim + i] != 0.f;
        if (nz) *flag = 1;
    }
    __syncthreads();
    return *flag == 0;
}

// K2: tau[q] = kk-th largest of sample_scores[q][0..n_sample)  (+inf for void queries;
// n_sample == 0 means "no sample pass": tau = -inf, every row is a candidate)
// With a collection filter (query_coll[q] != -1) the sample rows of other collections count as
// -inf: tau becomes the kk-th best SAMPLED ROW OF THAT COLLECTION, so the scan lets through about
// as many rows of the collection as it would unfiltered rows (sample entry i is row
// (i / unit) * stride * unit + i % unit); fewer than kk such rows in the sample -> tau = -inf.
__global__ __launch_bounds__(256) void kth_select(const float* __restrict__ sample_scores,
                                                  int64_t sample_ld, int n_sample, int kk,
                                                  const float* __restrict__ queries, int n_queries,
                                                  int dim, float* __restrict__ tau,
                                                  float* __restrict__ qerr,
                                                  const int32_t* __restrict__ doc_coll,
                                                  const int32_t* __restrict__ query_coll, int unit,
                                                  int64_t stride, int64_t n_docs) {
    __shared__ int hist[CS_BINS];
    __shared__ int aux[8];
    __shared__ int flag;
    __shared__ double red[2][256];
    const int q = blockIdx.x;
    if (qerr) {
        // eq = ||fp16(q) - q|| / ||q||, rounded up: the query-side term of the f16 certificate
        double e = 0.0, nn = 0.0;
        if (q < n_queries)
            for (int i = threadIdx.x; i < dim; i += blockDim.x) {
                const float v = queries[(int64_t)q * dim + i];
                const double dd = (double)v - (double)(float)(_Float16)v;
                e += dd * dd;
                nn += (double)v * (double)v;
            }
        red[0][threadIdx.x] = e;
        red[1][threadIdx.x] = nn;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) {
                red[0][threadIdx.x] += red[0][threadIdx.x + o];
                red[1][threadIdx.x] += red[1][threadIdx.x + o];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float rel = red[1][0] > 0.0 ? (float)sqrt(red[0][0] / red[1][0]) : 0.f;
            qerr[q] = __uint_as_float(__float_as_uint(rel) + 1u);
        }
    }
    if (query_is_void(queries, n_queries, dim, q, &flag)) {
        if (threadIdx.x == 0) tau[q] = INFINITY;
        return;
    }
    if (n_sample < kk) {
        if (threadIdx.x == 0) tau[q] = -INFINITY;
        return;
    }
    const float* s = sample_scores + (int64_t)q * sample_ld;
    const int qc = (query_coll && q < n_queries) ? query_coll[q] : -1;
    auto val = [&](int i) {
        float v = s[i];
        if (qc != -1) {
            const int64_t row = (int64_t)(i / unit) * stride * unit + i % unit;
            if (row >= n_docs || doc_coll[row] != qc) v = -INFINITY;
        }
        return v;
    };
    // Any threshold near the kk-th sample score serves (the certificate only needs "the scan
    // emitted every row >= tau"), and kk is a fraction of a percent of the sample.  Fast path:
    // with M the largest sample score, only the values in [M/2, M] are binned (4096 linear
    // bins: a handful of LDS atomics instead of one per sample, most of which would collide on
    // the two or three exponent bins around zero); when at least kk of them sit there, tau is
    // the lower edge of the bin that holds the kk-th.  Otherwise (M <= 0, or a sample that is
    // not bell-shaped) the two-pass key select below decides.
    __shared__ float kred[4];
    __shared__ int kcnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto sweep = [&](auto&& fn) {   // 8 loads in flight per thread
        for (int base = threadIdx.x; base < n_sample; base += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base + u * 256 < n_sample ? val(base + u * 256) : -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u) fn(v[u]);
        }
    };
    float m = -INFINITY;
    sweep([&](float v) { m = fmaxf(m, v); });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
    if (lane == 0) kred[wave] = m;
    if (threadIdx.x == 0) kcnt = 0;
    for (int i = threadIdx.x; i < CS_BINS; i += 256) hist[i] = 0;
    __syncthreads();
    m = fmaxf(fmaxf(kred[0], kred[1]), fmaxf(kred[2], kred[3]));
    if (m > 0.f && m < INFINITY) {
        const float thr = 0.5f * m, scale = 4095.f / (m - thr);
        int c = 0;
        sweep([&](float v) {
            if (v >= thr) {
                const int bn = (int)((v - thr) * scale);
                atomicAdd(&hist[bn > CS_BINS - 1 ? CS_BINS - 1 : bn], 1);
                ++c;
            }
        });
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, WAVE);
        if (lane == 0 && c) atomicAdd(&kcnt, c);
        __syncthreads();
        if (kcnt >= kk) {   // (block-uniform)
            coarse_find_bin(hist, kk, aux);
            if (threadIdx.x == 0) tau[q] = thr + (float)aux[0] / scale;
            return;
        }
        __syncthreads();
    }
    // a lower bound of the kk-th sample score is as good a threshold as the score itself
    const uint32_t key = block_coarse_select([&](int i) { return fkey(val(i)); }, n_sample, kk, hist, aux);
    if (threadIdx.x == 0) tau[q] = coarse_value(key);
}

// sequential float64 accumulation of float32 products: the oracle's contract
// (oracle/thr_oracle.py seq_dot_f64).  Products are exact in float64.
__device__ __forceinline__ double seq_dot_f64(const float* __restrict__ a, const float* b, int d) {
    double s = 0.0;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    for (int i = 0; i < d / 4; ++i) {
        float4 x = a4[i];
        s = __dadd_rn(s, __dmul_rn((double)x.x, (double)b[4 * i + 0]));
        s = __dadd_rn(s, __dmul_rn((double)x.y, (double)b[4 * i + 1]));
        s = __dadd_rn(s, __dmul_rn((double)x.z, (double)b[4 * i + 2]));
        s = __dadd_rn(s, __dmul_rn((double)x.w, (double)b[4 * i + 3]));
    }
    return s;
}

// K4: shortlist (select_band), then float64 rescoring, ordering, certificate (rescore_rank).
// One block (4 waves) per query in each.  They were one kernel until the counters showed its two
// halves wanting different things: the selection is a chain of dependent memory round trips that
// only occupancy hides, the rescoring is float64-ALU and LDS bound and heavy on registers.
//
//  band    the candidates that can still reach the top-k: with a_k the k-th largest scan score,
//          k rows have true cosine >= a_k/||q|| - eps, so a row whose scan score is below
//          a_k - 2*eps*||q|| cannot beat them.  a_k is replaced by a lower bound from one
//          histogram pass (a slightly wider band, never a narrower one); the first 16
//          candidates per thread stay in registers across the passes.
//  rescore float64 SEQUENTIAL sums (the oracle's contract), one lane per row, rows dealt
//          round-robin to the 4 waves.  Each wave stages its rows through its own LDS tile, 32
//          dims at a time, with coalesced loads (8 lanes per 128-byte line) and the next chunks
//          already in flight in registers -- no block barrier inside the loop, the waves run
//          free.  Lane 63 of wave 3 accumulates ||q||^2 in the same instruction stream.
//  order   rank sort of the rescored rows under (score desc, id asc).
constexpr int SEL_THREADS = 256;
constexpr int RS_STRIDE = 9;       // float4 slots per staged row: 8 + 1 pad (conflict-free b128)
constexpr int SEL_REG = 16;        // candidates per thread kept in registers (4096 per query; the scan aims at ~2900)
constexpr int SEL_BIG_BAND = 1024; // band capacity of the second-chance launch
constexpr int SEL_FLAT = 8192;     // candidates of the per-lane segments addressed through a flat LDS index
static size_t band_lds_bytes(int dim) { return sizeof(float) * dim + sizeof(int) * CS_BINS; }
// K4a: the shortlist of one query -- which candidate rows get a float64 score.  Light on
// registers and LDS (four workgroups per CU): its phases are chains of dependent memory round
// trips (segment counts -> candidates -> histogram -> band), which only occupancy hides.
// Writes sel_rows[q][0..ns), sel_meta[q] = {ns, floor (float bits), overflow}.
constexpr int CAPB = SEL_BIG_BAND;   // rows the band may hold
//
// Document shards (thr_dense_shortlist_f16 / thr_dense_floor / thr_dense_finish_f16): TOPM = true is
// the pass BEFORE the exchange -- the query's top_m largest scan scores, each lowered by the scan's
// error bound to a lower bound of ||q|| x (true cosine) of its row, written to top_lb[q][0..top_m)
// (-inf padded) and nothing else.  The k-th largest of the shards' values together, gfloor[q], is
// then a lower bound of ||q|| x (the GLOBAL k-th best cosine): in the pass after the exchange a
// row whose scan score is below gfloor - 1.5 eps ||q|| cannot be one of the global k best and is
// not rescored -- a shard of G rescores about k / G rows instead of k.
template <bool TOPM>
__global__ __launch_bounds__(SEL_THREADS, 4) void select_band(
    int dim, const float* __restrict__ queries, const float* __restrict__ tau,
    const int* __restrict__ cand_cnt, const Cand* __restrict__ cand,
    const int* __restrict__ tile_cnt, int tile_cap, int qtile, int k, int kprime, double eps32,
    double doc_relerr, const float* __restrict__ qerr, int nseg, int seg_cap,
    const int32_t* __restrict__ doc_coll, const int32_t* __restrict__ query_coll,
    int32_t* __restrict__ sel_rows, int32_t* __restrict__ sel_meta,
    const float* __restrict__ gfloor, const float* __restrict__ lb_all, int n_shards, int lb_m,
    float* __restrict__ top_lb, int top_m) {
    extern __shared__ float4 lds_sel[];  // [dim/4] query | hist
    __shared__ int aux[8];
    __shared__ int bc[4];
    __shared__ int32_t s_id[CAPB];
    __shared__ int n_sel;
    __shared__ double wsum[4];
    float* lds_qv = reinterpret_cast<float*>(lds_sel);
    int* hist = reinterpret_cast<int*>(lds_sel + dim / 4);

    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // scan error bound relative to ||q||*||d||: fp32 accumulation, plus -- for the f16 matrix
    // core scans -- row and query quantisation: ea*(1+eq) + eq
    const double eq = qerr ? (double)qerr[q] : 0.0;
    const double eps = eps32 + doc_relerr * (1.0 + eq) + eq;
    const Cand* c = cand + (int64_t)q * CAND_CAP;
    // A thread's candidates: cand_at(u), u in [0, my_n).
    //   nseg == 0  one flat list of cand_cnt[q] entries (K3b's output): thread t takes t, t+256, ..
    //   nseg  > 0  dense_scan_f16q's layout: nseg segments of seg_cap slots, segment s filled by
    //              ONE lane of the scan with cand_cnt[q * nseg + s] entries (a count above seg_cap
    //              means entries were dropped).  Up to SEL_FLAT candidates are addressed through
    //              src_off, a flat index of the filled slots: thread t takes items t, t+256, ..
    const Cand* my_ptr = c + threadIdx.x;
    int my_n, my_c[4] = {0, 0, 0, 0};
    bool overflow, flat = true;
    int n;
    __shared__ unsigned short src_off[SEL_FLAT];
    if (nseg == 0) {
        const int cnt = cand_cnt[q];
        overflow = cnt > CAND_CAP || tile_cnt[q / qtile] > tile_cap;
        // Only slots [0, min(cnt, CAND_CAP)) were written.  (Round 1 read all CAND_CAP slots
        // whenever the TILE list had overflowed, even for a query of that tile with few
        // candidates of its own: stale workspace words became row indices -> out-of-bounds
        // gathers, the rc 134 abort of gpurun_out/t1.log.  An overflowed query is never
        // certified; thr_dense_rescue redoes it.)
        n = cnt < CAND_CAP ? cnt : CAND_CAP;
        my_n = n > (int)threadIdx.x ? (n - (int)threadIdx.x + SEL_THREADS - 1) / SEL_THREADS : 0;
    } else {
        // thread t owns segments t, t + 256, t + 512, t + 768 (host keeps nseg <= 4 * SEL_THREADS);
        // the flat order is thread-major: an exclusive scan of the per-thread totals places them
        __shared__ int wtot[4];
        bool over = false;
        int tot = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int sg = (int)threadIdx.x + x * SEL_THREADS;
            int sc = sg < nseg ? cand_cnt[(int64_t)q * nseg + sg] : 0;
            over |= sc > seg_cap;
            my_c[x] = sc < seg_cap ? sc : seg_cap;
            tot += my_c[x];
        }
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wtot[wave] = incl;
        overflow = __syncthreads_or(over) != 0;
        int base = incl - tot;
        for (int x = 0; x < wave; ++x) base += wtot[x];
        n = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        flat = n <= SEL_FLAT;
        if (flat) {
            // src_off[i] = slot of flat candidate i: the reads below are then coalesced (lane l
            // of a wave takes flat item l + 64 * ..., i.e. neighbouring slots of a segment)
            // instead of one segment per lane, which cost a cache line per lane and load
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int first = ((int)threadIdx.x + x * SEL_THREADS) * seg_cap;
                for (int j = 0; j < my_c[x]; ++j) src_off[base + j] = (unsigned short)(first + j);
                base += my_c[x];
            }
            __syncthreads();
            my_n = n > (int)threadIdx.x ? (n - (int)threadIdx.x + SEL_THREADS - 1) / SEL_THREADS : 0;
        } else {
            my_n = tot;   // (rare: a threshold far too low) each thread walks its own segments
        }
    }
    auto cand_at = [&](int u) -> Cand {
        if (nseg == 0) return my_ptr[(int64_t)u * SEL_THREADS];
        if (flat) return c[src_off[(int)threadIdx.x + u * SEL_THREADS]];
        int sg = threadIdx.x;
#pragma unroll
        for (int x = 0; x < 3; ++x)
            if (u >= my_c[x]) {
                u -= my_c[x];
                sg += SEL_THREADS;
            } else {
                break;
            }
        return c[(int64_t)sg * seg_cap + u];
    };

    // Collection filter (rag2_schema.sql:404-408): a candidate of another collection is read as
    // score -inf and skipped everywhere below (a row that passed the scan never scores -inf
    // itself).  The floor of the certificate still bounds every row of the RIGHT collection
    // outside the shortlist.
    const int qc = query_coll ? query_coll[q] : -1;
    auto load_cand = [&](int u) -> Cand {
        Cand e = cand_at(u);
        if (qc != -1 && doc_coll[e.doc] != qc) e.score = -INFINITY;
        return e;
    };
    // candidates this thread keeps in registers (loads in flight while the query is staged)
    Cand mine[SEL_REG];
#pragma unroll
    for (int u = 0; u < SEL_REG; ++u) mine[u] = u < my_n ? load_cand(u) : Cand{-INFINITY, 0u};
    if (qc != -1) {   // n = the candidates that pass the filter
        __shared__ int n_pass;
        if (threadIdx.x == 0) n_pass = 0;
        __syncthreads();
        int mine_ok = 0;
#pragma unroll
        for (int u = 0; u < SEL_REG; ++u) mine_ok += (u < my_n && mine[u].score > -INFINITY) ? 1 : 0;
        for (int u = SEL_REG; u < my_n; ++u) mine_ok += load_cand(u).score > -INFINITY ? 1 : 0;
        if (mine_ok) atomicAdd(&n_pass, mine_ok);
        __syncthreads();
        n = n_pass;
    }
    for (int i = threadIdx.x; i < dim / 4; i += SEL_THREADS)
        lds_sel[i] = reinterpret_cast<const float4*>(queries + (int64_t)q * dim)[i];
    if (threadIdx.x == 0) n_sel = 0;
    __syncthreads();

    // ||q|| upper bound (parallel float64 sum, inflated) -- only used to size the band
    {
        double part = 0.0;
        for (int i = threadIdx.x; i < dim; i += SEL_THREADS) part += (double)lds_qv[i] * (double)lds_qv[i];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
        if (lane == 0) wsum[wave] = part;
        __syncthreads();
    }
    const double qn_hi = sqrt(wsum[0] + wsum[1] + wsum[2] + wsum[3]) * (1.0 + 1e-6);

    // the shards' common floor: rows below it cannot be among the k best of all the shards.  Given
    // as gfloor[q], or as the shards' gathered lower bounds lb_all [n_shards, nq, lb_m]: the k-th
    // largest of this query's n_shards * lb_m values, found here by rank counting (<= 4096 values
    // in the LDS words of the histogram, which is not in use yet).
    float gF = -INFINITY;
    if (!TOPM) {
        if (gfloor) {
            gF = gfloor[q];
        } else if (lb_all && n_shards * lb_m >= k) {
            __shared__ float s_gF;
            float* fv = reinterpret_cast<float*>(hist);
            const int nv = n_shards * lb_m, nv4 = (nv + 3) & ~3;   // (-inf padding never counts)
            for (int i = threadIdx.x; i < nv4; i += SEL_THREADS)
                fv[i] = i < nv ? lb_all[((int64_t)(i / lb_m) * gridDim.x + q) * lb_m + i % lb_m] : -INFINITY;
            if (threadIdx.x == 0) s_gF = -INFINITY;
            __syncthreads();
            const f32x4* fv4 = reinterpret_cast<const f32x4*>(fv);
            for (int i = threadIdx.x; i < nv; i += SEL_THREADS) {
                const float v = fv[i];
                if (!(v > -INFINITY)) continue;
                int rank = 0;   // values ahead of v: larger ones, equal ones of a lower index
#pragma unroll 4
                for (int j = 0; j < nv4; j += 4) {   // (the same addresses in every lane: broadcast reads)
                    const f32x4 w = fv4[j >> 2];
                    rank += (w.x > v || (w.x == v && j < i)) ? 1 : 0;
                    rank += (w.y > v || (w.y == v && j + 1 < i)) ? 1 : 0;
                    rank += (w.z > v || (w.z == v && j + 2 < i)) ? 1 : 0;
                    rank += (w.w > v || (w.w == v && j + 3 < i)) ? 1 : 0;
                }
                if (rank == k - 1) s_gF = v;
            }
            __syncthreads();
            gF = s_gF;
            __syncthreads();   // (hist is zeroed below)
        }
    }
    float band_lo = -INFINITY;
    if (gF > -INFINITY) band_lo = nextafterf((float)((double)gF - 1.5 * eps * qn_hi), -INFINITY);
    float floor32 = tau[q];
    bool band_done = false;
    const int kk = TOPM ? top_m : k;   // the rank the histogram pass looks for
    float a_kk = -INFINITY;            // TOPM: a lower bound of the top_m-th largest scan score
    if (n > kk) {
        // a_k, a lower bound of the k-th largest scan score: ONE histogram pass over 4096 LINEAR
        // bins between the smallest and the largest live candidate (the scores all sit just above
        // tau: binned by float exponent, as the sample select does, they fall into two or three
        // bins and the LDS atomics of a wave serialise on one address), then the smallest score
        // of the bins that hold the k largest.  bin_of is monotone in the score (IEEE subtract,
        // multiply by a positive constant, truncate), so those bins hold every score >= a_k.
        __shared__ float fred[3][4];
        float lo = INFINITY, hi = -INFINITY;
#pragma unroll
        for (int u = 0; u < SEL_REG; ++u)
            if (u < my_n && mine[u].score > -INFINITY) {
                lo = fminf(lo, mine[u].score);
                hi = fmaxf(hi, mine[u].score);
            }
        for (int u = SEL_REG; u < my_n; ++u) {
            const float sc = load_cand(u).score;
            if (sc > -INFINITY) {
                lo = fminf(lo, sc);
                hi = fmaxf(hi, sc);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, WAVE));
            hi = fmaxf(hi, __shfl_xor(hi, o, WAVE));
        }
        if (lane == 0) fred[0][wave] = lo, fred[1][wave] = hi;
        for (int i = threadIdx.x; i < CS_BINS; i += SEL_THREADS) hist[i] = 0;
        __syncthreads();
        lo = fminf(fminf(fred[0][0], fred[0][1]), fminf(fred[0][2], fred[0][3]));
        hi = fmaxf(fmaxf(fred[1][0], fred[1][1]), fmaxf(fred[1][2], fred[1][3]));
        const float scale = hi - lo > 1e-30f ? 4095.f / (hi - lo) : 0.f;
        auto bin_of = [&](float sc) {
            const int bn = (int)((sc - lo) * scale);
            return bn > CS_BINS - 1 ? CS_BINS - 1 : bn;
        };
#pragma unroll
        for (int u = 0; u < SEL_REG; ++u)
            if (u < my_n && mine[u].score > -INFINITY) atomicAdd(&hist[bin_of(mine[u].score)], 1);
        for (int u = SEL_REG; u < my_n; ++u) {
            const float sc = load_cand(u).score;
            if (sc > -INFINITY) atomicAdd(&hist[bin_of(sc)], 1);
        }
        __syncthreads();
        coarse_find_bin(hist, kk, aux);
        const int kbin = aux[0];
        float a_k = INFINITY;
#pragma unroll
        for (int u = 0; u < SEL_REG; ++u)
            if (u < my_n && mine[u].score > -INFINITY && bin_of(mine[u].score) >= kbin)
                a_k = fminf(a_k, mine[u].score);
        for (int u = SEL_REG; u < my_n; ++u) {
            const float sc = load_cand(u).score;
            if (sc > -INFINITY && bin_of(sc) >= kbin) a_k = fminf(a_k, sc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a_k = fminf(a_k, __shfl_xor(a_k, o, WAVE));
        if (lane == 0) fred[2][wave] = a_k;
        __syncthreads();
        a_k = fminf(fminf(fred[2][0], fred[2][1]), fminf(fred[2][2], fred[2][3]));
        a_kk = a_k;
        const float band = (float)((double)a_k - 2.5 * eps * qn_hi);
        // (the float conversion may have rounded up); either bound rules a row out: the higher one
        band_lo = fmaxf(band_lo, nextafterf(band, -INFINITY));
    }
    if (TOPM) {
        // the top_m largest scan scores, each as a lower bound of ||q|| x cosine of its row (the
        // float conversion may round up: one step down).  The scores >= a_kk are the top_m and the
        // few more that share the last histogram bin: collected in LDS, ranked by counting.
        float* o = top_lb + (int64_t)q * top_m;
        float* vals = reinterpret_cast<float*>(s_id);   // CAPB values
        const double drop = eps * qn_hi;
        auto lowered = [&](float sc) { return nextafterf((float)((double)sc - drop), -INFINITY); };
        for (int i = threadIdx.x; i < top_m; i += SEL_THREADS) o[i] = -INFINITY;
        for (int u = 0; u < my_n; ++u) {
            const float sc = load_cand(u).score;
            if (sc > -INFINITY && sc >= a_kk) {
                const int p = atomicAdd(&n_sel, 1);
                if (p < CAPB) vals[p] = sc;
            }
        }
        __syncthreads();
        const int c = n_sel;
        if (c <= CAPB) {
            for (int i = threadIdx.x; i < c; i += SEL_THREADS) {
                const float v = vals[i];
                int rank = 0;
                for (int j = 0; j < c; ++j) {
                    const float w = vals[j];
                    rank += (w > v || (w == v && j < i)) ? 1 : 0;
                }
                if (rank < top_m) o[rank] = lowered(v);
            }
            return;
        }
        // (a tie wider than the LDS list at the top: the exact select, four passes)
        __syncthreads();
        if (threadIdx.x == 0) n_sel = 0;
        __syncthreads();
        const uint32_t tkey = block_radix_select_local(
            [&](int u) { return fkey(load_cand(u).score); }, my_n, top_m, hist, bc);
        for (int u = 0; u < my_n; ++u) {
            const Cand e = load_cand(u);
            if (fkey(e.score) > tkey) o[atomicAdd(&n_sel, 1)] = lowered(e.score);
        }
        __syncthreads();
        for (int u = 0; u < my_n; ++u) {
            const Cand e = load_cand(u);
            if (fkey(e.score) == tkey) {
                const int p = atomicAdd(&n_sel, 1);
                if (p < top_m) o[p] = lowered(e.score);
            }
        }
        return;
    }
    if (band_lo > -INFINITY) {
        // count and collect in one sweep; past CAPB rows only the count matters
#pragma unroll
        for (int u = 0; u < SEL_REG; ++u) {   // (one LDS atomic per wave and register slot)
            const bool in = u < my_n && mine[u].score >= band_lo;
            const unsigned long long m = __ballot(in);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&n_sel, __popcll(m));
                base = __shfl(base, 0, WAVE);
                const int p = base + __popcll(m & ((1ull << lane) - 1ull));
                if (in && p < CAPB) s_id[p] = mine[u].doc;
            }
        }
        for (int u = SEL_REG; u < my_n; ++u) {
            const Cand e = load_cand(u);
            if (e.score >= band_lo) {   // (band_lo > -inf: filtered candidates never pass)
                const int p = atomicAdd(&n_sel, 1);
                if (p < CAPB) s_id[p] = e.doc;
            }
        }
        __syncthreads();
        if (n_sel <= CAPB) {
            // rows outside the band: uncollected ones are below tau, collected ones below band_lo
            floor32 = fmaxf(floor32, band_lo);
            band_done = true;
        } else {
            __syncthreads();
            if (threadIdx.x == 0) n_sel = 0;
            __syncthreads();
        }
    }
    if (!band_done) {
        // the band does not fit the block (or the list is short): the kprime best, exactly
        if (n > kprime) {
            // (n > kprime candidates pass the filter, so the kprime-th largest key is a real score)
            const uint32_t tkey = block_radix_select_local(
                [&](int u) { return fkey(load_cand(u).score); }, my_n, kprime, hist, bc);
            floor32 = fkey_inv(tkey);
            for (int u = 0; u < my_n; ++u) {
                const Cand e = load_cand(u);
                if (fkey(e.score) > tkey) {
                    const int p = atomicAdd(&n_sel, 1);
                    s_id[p] = e.doc;
                }
            }
            __syncthreads();
            for (int u = 0; u < my_n; ++u) {
                const Cand e = load_cand(u);
                if (fkey(e.score) == tkey) {
                    const int p = atomicAdd(&n_sel, 1);
                    if (p < kprime) s_id[p] = e.doc;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0 && n_sel > kprime) n_sel = kprime;
        } else {
            for (int u = 0; u < my_n; ++u) {
                const Cand e = load_cand(u);
                if (e.score > -INFINITY) {
                    const int p = atomicAdd(&n_sel, 1);
                    s_id[p] = e.doc;
                }
            }
        }
    }
    __syncthreads();
    const int ns = n_sel;
    for (int i = threadIdx.x; i < ns; i += SEL_THREADS) sel_rows[(int64_t)q * CAPB + i] = s_id[i];
    if (threadIdx.x == 0) {
        sel_meta[4 * q + 0] = ns;
        sel_meta[4 * q + 1] = (int32_t)__float_as_uint(floor32);
        sel_meta[4 * q + 2] = overflow ? 1 : 0;
        sel_meta[4 * q + 3] = (int32_t)__float_as_uint(gF);
    }
}

// K4b: float64 scores of the shortlist, rank sort, certificate.  NB = the rows it can take: the
// first launch (256) serves every query whose shortlist fits, the second (1024) the few whose
// band was wider (score distributions squeezed into a narrow range: anisotropic embeddings put
// hundreds of rows within the f16 error band of the k-th); each exits at once on the others.
//
// Two waves per query, 64 rows per wave (every lane holds a row).  What bounds it is the gather:
// 2048 queries x ~105 rows x 3 KB = 645 MB read as scattered 128-byte lines, 4.3 TB/s at 150 us --
// four waves of 32 rows, two of 64, two or four chunks in flight, three to six workgroups per CU
// all land within 5 % of each other; 256-byte steps per row (half the occupancy) are 17 % slower.  The query is one more row of the shortlist: its dot
// product with itself, in the same sequential order, is ||q||^2.
constexpr int RR_WAVES = 2, RR_THREADS = 64 * RR_WAVES, RR_ROWS = 64;
static size_t rescore_lds_bytes(int dim) {   // the query as float64 | the waves' stage tiles
    return sizeof(double) * dim + sizeof(float4) * RR_WAVES * RR_ROWS * RS_STRIDE;
}
// One pass of a wave over its (up to) 8 U staged rows: U row groups of 8 per 32-dim chunk, 16 / U
// (at most 8) chunks of them in flight in registers.  Lanes of the groups that are not staged
// compute on stale LDS words; their slots are beyond the list and nothing reads the result.
// dim / 32 is a multiple of 8 for every row length the scans are built for.
template <int U>
//
// dot += x * y as ONE v_fma_f64 per element: the product of two float32 values is exact in
// float64 (48 significant bits), so fma(x, y, dot) rounds the same real number as the oracle's
// separate multiply and add -- the same bits at half the float64 instructions; the query is
// converted once per workgroup (q64), the rows as they are read.
__device__ __forceinline__ double rescore_pass(const f32x4* (&rp)[8], f32x4* stage, const double* q64,
                                               int nchunk, int lane, int lrow, int lch) {
    constexpr int D = U >= 2 ? 16 / U : 8;
    f32x4 nxt[D][U];
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[dd][u] = rp[u][8 * dd];   // (nchunk >= D)
    double dot = 0.0;
#pragma unroll 1
    for (int ck0 = 0; ck0 < nchunk; ck0 += D) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const int ck = ck0 + dd;
#pragma unroll
            for (int u = 0; u < U; ++u) stage[(lrow + 8 * u) * RS_STRIDE + lch] = nxt[dd][u];
            // (the last trips re-request the last chunk)
            const int cn = ck + D < nchunk ? ck + D : nchunk - 1;
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[dd][u] = rp[u][8 * cn];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const f32x4* src = stage + lane * RS_STRIDE;
            const double* qv = q64 + 32 * ck;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const f32x4 x = src[ch];
                dot = __fma_rn((double)x.x, qv[4 * ch + 0], dot);
                dot = __fma_rn((double)x.y, qv[4 * ch + 1], dot);
                dot = __fma_rn((double)x.z, qv[4 * ch + 2], dot);
                dot = __fma_rn((double)x.w, qv[4 * ch + 3], dot);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    return dot;
}

template <int NB>
__global__ __launch_bounds__(RR_THREADS, 3) void rescore_rank(
    const float* __restrict__ docs, const double* __restrict__ dnorm, int dim, int64_t id_base,
    const float* __restrict__ queries, int k, double eps32, double doc_relerr,
    const float* __restrict__ qerr, const int32_t* __restrict__ sel_rows,
    const int32_t* __restrict__ sel_meta, double* __restrict__ out_scores,
    int64_t* __restrict__ out_ids, int32_t* __restrict__ out_counts, uint32_t* __restrict__ out_flags) {
    const int q = blockIdx.x;
    const int ns = sel_meta[4 * q + 0];
    if (NB == THR_DENSE_MAX_K ? ns > THR_DENSE_MAX_K : ns <= THR_DENSE_MAX_K) return;
    const float floor32 = __uint_as_float((uint32_t)sel_meta[4 * q + 1]);
    const bool overflow = sel_meta[4 * q + 2] != 0;
    extern __shared__ float4 lds_sel[];  // [dim/2] the query as float64 | RR_WAVES stage tiles
    __shared__ double s_s[NB], o_s[NB];
    __shared__ int64_t s_id[NB], o_id[NB];
    __shared__ double s_qn;
    __shared__ int n_valid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double eq = qerr ? (double)qerr[q] : 0.0;
    const double eps = eps32 + doc_relerr * (1.0 + eq) + eq;
    double* q64 = reinterpret_cast<double*>(lds_sel);
    for (int i = threadIdx.x; i < dim; i += RR_THREADS) q64[i] = (double)queries[(int64_t)q * dim + i];
    for (int i = threadIdx.x; i < NB; i += RR_THREADS) {
        s_s[i] = o_s[i] = -INFINITY;
        s_id[i] = i < ns ? (int64_t)sel_rows[(int64_t)q * SEL_BIG_BAND + i] : INT64_MAX;
        o_id[i] = INT64_MAX;
    }
    if (threadIdx.x == 0) n_valid = 0;
    __syncthreads();

    // ---- float64 rescoring: SEQUENTIAL sums (the oracle's contract), one lane per row ----
    // (native vectors, not HIP's float4 class: see dense_scan_mfma2 -- a float4 array that is
    // copied into LDS is demoted to scratch memory)
    f32x4* stage = reinterpret_cast<f32x4*>(lds_sel + dim / 2) + wave * (RR_ROWS * RS_STRIDE);
    const f32x4* docs4 = reinterpret_cast<const f32x4*>(docs);
    const int lrow = lane >> 3, lch = lane & 7;
    const int cpr = dim / 4, nchunk = dim / 32;
    const f32x4* q4 = reinterpret_cast<const f32x4*>(queries + (int64_t)q * dim);
    // (the row norm of this thread's first shortlist slot: requested now, used after the loop)
    const double dn_first = (int)threadIdx.x < ns ? dnorm[s_id[threadIdx.x]] : 0.0;
    for (int b0 = 0; b0 <= ns; b0 += RR_WAVES * RR_ROWS) {
        // slot of (wave, staged row r) is b0 + wave + RR_WAVES r; slot ns is the query itself;
        // a lane loads 16 bytes of rows lrow + 8 u (8 lanes per 128-byte line)
        const int rem = ns - b0 - wave;          // this wave's slots of the pass: r <= rem / RR_WAVES
        if (rem < 0) continue;                   // (wave-uniform)
        const f32x4* rp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int j = b0 + wave + RR_WAVES * (lrow + 8 * u);
            j = j < ns ? j : ns;
            rp[u] = (j < ns ? docs4 + s_id[j] * cpr : q4) + lch;
        }
        const int jm = b0 + wave + RR_WAVES * lane;  // this lane's own slot
        // A short list (a shard under the common floor rescores ~k / G rows, a top-10 search ~12)
        // fills only the first row groups of the wave: it stages those alone and keeps more
        // chunks of them in flight instead -- the pass is a chain of memory round trips.
        const int groups = rem / RR_WAVES / 8 + 1;
        double dot;
        if (groups <= 1) dot = rescore_pass<1>(rp, stage, q64, nchunk, lane, lrow, lch);
        else if (groups <= 2) dot = rescore_pass<2>(rp, stage, q64, nchunk, lane, lrow, lch);
        else if (groups <= 4) dot = rescore_pass<4>(rp, stage, q64, nchunk, lane, lrow, lch);
        else dot = rescore_pass<8>(rp, stage, q64, nchunk, lane, lrow, lch);
        if (jm < ns) s_s[jm] = dot;               // the raw dot product for now
        else if (jm == ns) s_qn = __dsqrt_rn(dot);  // ||q||
    }
    __syncthreads();
    for (int p = threadIdx.x; p < ns; p += RR_THREADS) {
        const int64_t row = s_id[p];
        const double qn = s_qn, dn = p == (int)threadIdx.x ? dn_first : dnorm[row], dot = s_s[p];
        double sim = -INFINITY;
        if (dn > 0.0) sim = qn > 0.0 ? __ddiv_rn(dot, __dmul_rn(qn, dn)) : 0.0;
        s_s[p] = sim;
        s_id[p] = sim == -INFINITY ? INT64_MAX : row + id_base;
    }
    __syncthreads();
    // rank sort: ids are distinct, so (score desc, id asc) is a strict order on the valid rows;
    // rows without an embedding all carry (-inf, INT64_MAX), which is what o_s/o_id hold already
    for (int p = threadIdx.x; p < ns; p += RR_THREADS) {
        const double ms = s_s[p];
        const int64_t mi = s_id[p];
        if (mi == INT64_MAX) continue;
        int rank = 0;
        for (int i = 0; i < ns; ++i) rank += better(s_s[i], s_id[i], ms, mi) ? 1 : 0;
        o_s[rank] = ms;
        o_id[rank] = mi;
    }
    __syncthreads();

    // results + certificate
    int mine_valid = 0;
    for (int i = threadIdx.x; i < k; i += RR_THREADS) {
        const bool ok = o_s[i] > -INFINITY;
        mine_valid += ok ? 1 : 0;
        out_scores[(int64_t)q * k + i] = ok ? o_s[i] : -INFINITY;
        out_ids[(int64_t)q * k + i] = ok ? o_id[i] : -1;
    }
    if (mine_valid) atomicAdd(&n_valid, mine_valid);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int valid = n_valid;
        uint32_t flag = overflow ? THR_FLAG_OVERFLOW : 0u;
        bool cert;
        if (overflow) {
            cert = false;
        } else if (floor32 == -INFINITY) {
            cert = true;  // every row with an embedding was rescored
        } else {
            // rows outside the shortlist have scan score <= floor32, hence true
            // cosine <= floor32/||q|| + eps; the k-th best must clear that strictly -- this
            // shard's own k-th best, or the k-th best of all the shards, of which gF / ||q|| is a
            // lower bound (then the list may be shorter than k: the rest is on other shards).
            const float gF = __uint_as_float((uint32_t)sel_meta[4 * q + 3]);
            const bool own = valid >= k && s_qn > 0.0 && (o_s[k - 1] - (double)floor32 / s_qn) > eps;
            const bool all = gF > -INFINITY && s_qn > 0.0 &&
                             ((double)gF / s_qn - (double)floor32 / s_qn) > eps;
            cert = own || all;
        }
        out_flags[q] = flag | (cert ? THR_FLAG_CERTIFIED : 0u);
        out_counts[q] = valid;
    }
}

// ---------------------------------------------------------------------------
// Exhaustive float64 path: every row scored with the oracle's arithmetic, then an
// exact block top-k per (query, slab); slabs merged by a second kernel.
// ---------------------------------------------------------------------------
constexpr int EX_THREADS = 256;
constexpr int EX_CAP = 1024;
constexpr int EX_SLABS = 64;

__global__ __launch_bounds__(EX_THREADS) void exact_slab_topk(
    const float* __restrict__ docs, const double* __restrict__ dnorm, int64_t n_docs, int dim,
    const float* __restrict__ queries, int n_queries, int k, double* __restrict__ slab_s,
    int64_t* __restrict__ slab_id, const uint32_t* __restrict__ skip_certified,
    const int32_t* __restrict__ doc_coll, const int32_t* __restrict__ query_coll) {
    extern __shared__ float lds_qv[];
    __shared__ double b_s[EX_CAP];
    __shared__ int64_t b_id[EX_CAP];
    __shared__ int b_cnt;
    __shared__ double t_s;
    __shared__ int64_t t_id;
    __shared__ double s_qn;
    __shared__ unsigned long long s_todo;
    const int slab = blockIdx.x;
    // queries strided over gridDim.y, at most 64 per block: in rescue mode (skip_certified) the
    // grid is small and one ballot tells the block which of its queries still need the work
    // (instead of one mostly-empty block per query)
    {
        const int q = blockIdx.y + (int)threadIdx.x * (int)gridDim.y;
        const bool todo = threadIdx.x < 64 && q < n_queries &&
                          !(skip_certified && (skip_certified[q] & THR_FLAG_CERTIFIED));
        const unsigned long long m = __ballot(todo);
        if (threadIdx.x == 0) s_todo = m;
        __syncthreads();
    }
    for (unsigned long long todo = s_todo; todo; todo &= todo - 1) {
        const int q = blockIdx.y + (__ffsll((long long)todo) - 1) * (int)gridDim.y;
        __syncthreads();
        for (int i = threadIdx.x; i < dim; i += EX_THREADS) lds_qv[i] = queries[(int64_t)q * dim + i];
        __syncthreads();
        if (threadIdx.x == 0) s_qn = __dsqrt_rn(seq_dot_f64(lds_qv, lds_qv, dim));
        BlockTopK<EX_CAP, EX_THREADS> tk;
        tk.init(b_s, b_id, &b_cnt, &t_s, &t_id, k);
        const double qn = s_qn;
        const int qc = query_coll ? query_coll[q] : -1;
        const int64_t per = (n_docs + EX_SLABS - 1) / EX_SLABS;
        const int64_t lo = slab * per, hi = (lo + per < n_docs) ? lo + per : n_docs;
        for (int64_t base = lo; base < hi; base += EX_THREADS) {
            int64_t row = base + threadIdx.x;
            bool ok = row < hi;
            double sim = -INFINITY;
            if (ok) {
                double dn = dnorm[row];
                if (qc != -1 && doc_coll[row] != qc) dn = 0.0;   // another collection: not a row of this search
                if (dn > 0.0) {
                    double dot = seq_dot_f64(docs + row * dim, lds_qv, dim);
                    sim = qn > 0.0 ? __ddiv_rn(dot, __dmul_rn(qn, dn)) : 0.0;
                }
            }
            tk.push(ok && sim > -INFINITY, sim, row);
        }
        int n = tk.finish();
        for (int i = threadIdx.x; i < k; i += EX_THREADS) {
            int64_t o = ((int64_t)q * EX_SLABS + slab) * k + i;
            slab_s[o] = i < n ? b_s[i] : -INFINITY;
            slab_id[o] = i < n ? b_id[i] : INT64_MAX;
        }
    }
}

// merges n_lists ranked lists of k_in per query (layout [n_lists? no: q-major]) -> top k_out
__global__ __launch_bounds__(256) void merge_lists(const double* __restrict__ in_s,
                                                   const int64_t* __restrict__ in_id,
                                                   int64_t q_stride, int64_t list_stride,
                                                   int n_lists, int k_in, int k_out,
                                                   int64_t id_add, uint32_t flag_value,
                                                   double* __restrict__ out_s,
                                                   int64_t* __restrict__ out_id,
                                                   int32_t* __restrict__ out_counts,
                                                   uint32_t* __restrict__ out_flags,
                                                   const uint32_t* __restrict__ skip_certified = nullptr,
                                                   int32_t* __restrict__ n_done = nullptr) {
    if (skip_certified && (skip_certified[blockIdx.x] & THR_FLAG_CERTIFIED)) return;
    if (n_done && threadIdx.x == 0) atomicAdd(n_done, 1);
    __shared__ double b_s[EX_CAP];
    __shared__ int64_t b_id[EX_CAP];
    __shared__ int b_cnt;
    __shared__ double t_s;
    __shared__ int64_t t_id;
    const int q = blockIdx.x;
    BlockTopK<EX_CAP, EX_THREADS> tk;
    tk.init(b_s, b_id, &b_cnt, &t_s, &t_id, k_out);
    const int total = n_lists * k_in;
    for (int base = 0; base < total; base += blockDim.x) {
        int i = base + threadIdx.x;
        bool ok = i < total;
        double s = -INFINITY;
        int64_t id = INT64_MAX;
        if (ok) {
            int64_t o = (int64_t)q * q_stride + (int64_t)(i / k_in) * list_stride + (i % k_in);
            s = in_s[o];
            id = in_id[o];
        }
        tk.push(ok && s > -INFINITY && id >= 0 && id != INT64_MAX, s, id);
    }
    int n = tk.finish();
    for (int i = threadIdx.x; i < k_out; i += blockDim.x) {
        out_s[(int64_t)q * k_out + i] = i < n ? b_s[i] : -INFINITY;
        out_id[(int64_t)q * k_out + i] = i < n ? b_id[i] + id_add : -1;
    }
    if (threadIdx.x == 0) {
        if (out_counts) out_counts[q] = n;
        if (out_flags) out_flags[q] = flag_value;
    }
}

// Merge of at most EX_CAP candidates per query held entirely in LDS: the per-shard lists of the
// multi-GPU path (8 x <= 128).  The lists arrive ranked under (score desc, id asc) with disjoint
// ids, so an entry's place in the merged order is its own position plus, per other list, the
// number of entries ahead of it there (one binary search each) -- no sort.  A list that is NOT
// ranked makes the block fall back to a bitonic sort of everything (same result, slower).
__global__ __launch_bounds__(256) void merge_ranked_lists(const double* __restrict__ in_s,
                                                          const int64_t* __restrict__ in_id,
                                                          int64_t q_stride, int64_t list_stride,
                                                          int n_lists, int k_in, int k_out,
                                                          double* __restrict__ out_s,
                                                          int64_t* __restrict__ out_id,
                                                          int32_t* __restrict__ out_counts) {
    __shared__ double b_s[EX_CAP];
    __shared__ int64_t b_id[EX_CAP];
    __shared__ int unsorted, n_valid;
    const int q = blockIdx.x;
    const int total = n_lists * k_in;
    if (threadIdx.x == 0) unsorted = 0, n_valid = 0;
    for (int i = threadIdx.x; i < EX_CAP; i += blockDim.x) {
        double sc = -INFINITY;
        int64_t id = INT64_MAX;
        if (i < total) {
            const int64_t o = (int64_t)q * q_stride + (int64_t)(i / k_in) * list_stride + (i % k_in);
            sc = in_s[o];
            id = in_id[o];
            if (!(sc > -INFINITY) || id < 0) sc = -INFINITY, id = INT64_MAX;
        }
        b_s[i] = sc;
        b_id[i] = id;
    }
    for (int i = threadIdx.x; i < k_out; i += blockDim.x) {
        out_s[(int64_t)q * k_out + i] = -INFINITY;
        out_id[(int64_t)q * k_out + i] = -1;
    }
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        mine += b_id[i] != INT64_MAX ? 1 : 0;
        if (i % k_in + 1 < k_in && better(b_s[i + 1], b_id[i + 1], b_s[i], b_id[i])) unsorted = 1;
    }
    if (mine) atomicAdd(&n_valid, mine);
    __syncthreads();
    if (!unsorted) {
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const double ms = b_s[i];
            const int64_t mi = b_id[i];
            if (mi == INT64_MAX) continue;
            const int a = i / k_in;
            int rank = i % k_in;
            for (int b = 0; b < n_lists && rank < k_out; ++b) {
                if (b == a) continue;
                int lo = 0, hi = k_in;  // first position of list b that is not ahead of (ms, mi)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (better(b_s[b * k_in + mid], b_id[b * k_in + mid], ms, mi)) lo = mid + 1;
                    else hi = mid;
                }
                rank += lo;
            }
            if (rank < k_out) {
                out_s[(int64_t)q * k_out + rank] = ms;
                out_id[(int64_t)q * k_out + rank] = mi;
            }
        }
    } else {
        bitonic_sort_desc<EX_CAP>(b_s, b_id);
        for (int i = threadIdx.x; i < k_out && i < n_valid; i += blockDim.x) {
            out_s[(int64_t)q * k_out + i] = b_s[i];
            out_id[(int64_t)q * k_out + i] = b_id[i];
        }
    }
    if (threadIdx.x == 0 && out_counts) out_counts[q] = n_valid < k_out ? n_valid : k_out;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct DensePlan {
    int qtile, ntiles, qpad, unit, kind, row_bits, ksample, nq;
    bool packed;  // KIND_F16 only: scan of the fragment-major copy (else float32 rows, rounded in flight)
    bool qreg;    // = packed: dense_scan_f16q[s] (queries in registers, rows through LDS); the
                  // candidate area is written in per-lane segments
    int64_t groups;
    int64_t sample_groups, sample_stride, sample_docs;
    size_t off_selrows, off_selmeta;
    bool sampled;
    int tile_cap;
    size_t off_tau, off_qerr, off_cnt, off_tcnt, off_cand, off_tlist, off_sample, off_qfrag, total;
};

constexpr int KIND_F32 = 0, KIND_F16 = 1;
// in-flight-rounding f16 scan (dense_scan_f16): query sub-tiles of 32 per pass -- 2 (64 queries,
// 96 KiB of LDS at dim 768) when the tile fits next to the transpose tiles, else 1
static size_t f16_lds_bytes(int dim, int nq) {
    return sizeof(_Float16) * 32 * nq * (size_t)dim +
           (sizeof(Cand) * WBUF + sizeof(float4) * MF2_STAGE_F4) * H_WAVES;
}
static int f16_pick_nq(int dim) { return f16_lds_bytes(dim, 2) <= 160 * 1024 ? 2 : 1; }

// The scan over the float16 copy (queries in registers, rows through LDS): dense_scan_f16qs
// (staggered 8-wave block) where 8 x 32 queries' B operands fit the registers of two waves per
// SIMD, else dense_scan_f16q (4-wave blocks); THR_DENSE_F16=q forces the latter (it has the
// stamped diagnostic build).  Read once.
static bool qreg_staggered(int dim) {
    static int forced_q = -1;
    if (forced_q < 0) {
        const char* e = getenv("THR_DENSE_F16");
        forced_q = (e && e[0] == 'q') ? 1 : 0;
    }
    return !forced_q && dim <= 768;
}
static int qreg_waves(int dim) { return qreg_staggered(dim) ? 8 : 4; }   // qreg_qw queries per wave
// MFMA shape of the staggered scan and therefore of the copy / query images: 16 (16x16x32, the
// default: 2.50 ms against 2.67 ms per 2048 x 1M x 768 launch) or 32 (32x32x16,
// THR_DENSE_MFMA=32).  Read once: the copy's layout depends on it.
static int qreg_shape(int dim) {
    static int v = 0;
    if (!v) {
        const char* e = getenv("THR_DENSE_MFMA");
        v = (e && atoi(e) == 32) ? 32 : 16;
    }
    (void)dim;   // (both register-resident kernels take either shape)
    return v;
}
// queries per wave: 32, or 48 at dim 1024 with the 16x16x32 shape (dense_scan_f16q<1024, .., 48>:
// three 16-query blocks per wave, 192 queries per CU; THR_DENSE_QW=32 keeps the 32-query kernel for A/B)
static int qreg_qw(int dim) {
    static int forced32 = -1;
    if (forced32 < 0) {
        const char* e = getenv("THR_DENSE_QW");
        forced32 = (e && atoi(e) == 32) ? 1 : 0;
    }
    return (dim == 1024 && !qreg_staggered(dim) && qreg_shape(dim) == 16 && !forced32) ? 48 : 32;
}
constexpr int QREG_MAX_SEG = 1024;
// The register-resident scans address a lane's candidate segment with a 32-bit byte offset from
// the start of the candidate area ((q * CAND_CAP + segment start) * sizeof(Cand)): a batch may
// hold as many (padded) queries as keep every offset below 2^32.
static int qreg_max_queries(int dim) {
    const int qt = qreg_qw(dim) * qreg_waves(dim);
    const int64_t m = (int64_t)UINT32_MAX / ((int64_t)CAND_CAP * (int64_t)sizeof(Cand));
    return (int)(m / qt * qt);
}

static DensePlan make_plan(int64_t n_docs, int n_queries, int kprime, int kind = KIND_F32,
                           int dim = 0, bool packed = false) {
    DensePlan p;
    p.kind = kind;
    p.packed = kind == KIND_F16 && packed;
    p.qreg = p.packed;
    p.nq = (kind == KIND_F16 && !p.packed) ? f16_pick_nq(dim) : 1;
    p.row_bits = (kind == KIND_F16 && !p.qreg) ? ROW_BITS_F16 : ROW_BITS;
    p.qtile = p.qreg ? qreg_qw(dim) * qreg_waves(dim)
              : kind == KIND_F16 ? 32 * p.nq : MF_QT;
    p.unit = MF_ROWS;
    p.ntiles = (n_queries + p.qtile - 1) / p.qtile;
    p.qpad = p.ntiles * p.qtile;
    const int64_t groups = (n_docs + p.unit - 1) / p.unit;
    // sample only when the corpus is larger than what the candidate list can hold anyway
    p.sampled = n_docs > CAND_CAP / 2;
    // tau = the ks-th best score of a sample of S rows lets (n / S) * ks rows per query through
    // on average; aim at 4096 (a quarter of CAND_CAP, half of a tile list's share).  ks is capped
    // at 64: the count of passing rows then spreads by ~1/8 of its mean (the tile share is 8
    // sigma away), and the sample pass + select cost a third of what ks = k' = 192 did.
    // The register-resident scan keeps one candidate segment per lane (no shared tile list), so
    // only the cost matters there: ks = 32 (spread ~1/6) halves the sample pass and the select.
    const int ks_cap = p.qreg ? 32 : 64;
    p.ksample = kprime < ks_cap ? kprime : ks_cap;
    // (the sample must grow with the corpus: a capped sample lets n / S * ks rows through, which
    // overflows the candidate lists of every query on a 10M-row shard)
    // The sample pass costs ~ n * ks / aim, the scan's emit + K4's candidate read ~ aim: at 1M rows
    // the two meet at aim = 4096 (profiles/r2_scan_tau_experiment.json), so aim follows sqrt(n)
    // below that; never under 8 k' (k' = 128: 1024 = k' + 7 sigma of the passing count at ks = 64).
    double aim = (p.qreg ? 2896.0 : 4096.0) * sqrt((double)n_docs / 1.0e6);   // (ks / 64 under the root)
    const double aim_lo = 8.0 * kprime < 4096.0 ? 8.0 * kprime : 4096.0;
    aim = aim < aim_lo ? aim_lo : aim > 4096.0 ? 4096.0 : aim;
    int64_t target = (int64_t)((double)n_docs * (double)p.ksample / aim);
    if (target > SAMPLE_MAX) target = SAMPLE_MAX;
    if (target < 4 * (int64_t)p.ksample) target = 4 * (int64_t)p.ksample;
    int64_t sg = (target + p.unit - 1) / p.unit;
    if (sg > groups) sg = groups;
    p.sample_stride = sg > 0 ? groups / sg : 1;
    if (p.sample_stride < 1) p.sample_stride = 1;
    p.sample_groups = p.sampled ? sg : 0;
    p.sample_docs = p.sample_groups * p.unit;
    p.groups = groups;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    p.off_tau = take(sizeof(float) * p.qpad);
    p.off_qerr = take(sizeof(float) * p.qpad);
    p.tile_cap = p.qreg ? 1 : p.qtile * (CAND_CAP / 2);   // (no tile lists in the qreg scan)
    // (off_cnt and off_tcnt are zeroed by one memset; the qreg scan keeps one count per segment)
    p.off_cnt = take(sizeof(int) * p.qpad * (p.qreg ? QREG_MAX_SEG : 1));
    p.off_tcnt = take(sizeof(int) * p.ntiles);
    p.off_cand = take(sizeof(Cand) * (size_t)p.qpad * CAND_CAP);
    p.off_tlist = take(sizeof(Cand) * (size_t)p.ntiles * p.tile_cap);
    p.off_sample = take(sizeof(float) * (size_t)p.qpad * (size_t)p.sample_docs);
    p.off_qfrag = take(p.qreg ? sizeof(_Float16) * (size_t)p.qpad * (size_t)dim : 0);
    p.off_selrows = take(sizeof(int32_t) * (size_t)p.qpad * SEL_BIG_BAND);   // K4a -> K4b shortlists
    p.off_selmeta = take(sizeof(int32_t) * 4 * (size_t)p.qpad);
    p.total = off;
    return p;
}

// The shards' common floor: the k-th largest of the n_shards * m lower bounds of a query
// (select_band<true> of every shard, gathered shard-major), -inf when fewer than k are finite.
// One workgroup per query; rank counting in LDS (n_shards * m is a few hundred values).
__global__ __launch_bounds__(256) void dense_floor_kernel(const float* __restrict__ lb, int n_shards,
                                                          int n_queries, int m, int k,
                                                          float* __restrict__ gfloor) {
    extern __shared__ float fl_v[];
    const int q = blockIdx.x, n = n_shards * m;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        fl_v[i] = lb[((int64_t)(i / m) * n_queries + q) * m + i % m];
    if (threadIdx.x == 0) gfloor[q] = -INFINITY;
    __syncthreads();
    if (n < k) return;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = fl_v[i];
        if (!(v > -INFINITY)) continue;
        int rank = 0;   // values ahead of v: larger ones, equal ones of a lower index
        for (int j = 0; j < n; ++j) {
            const float w = fl_v[j];
            rank += (w > v || (w == v && j < i)) ? 1 : 0;
        }
        if (rank == k - 1) gfloor[q] = v;
    }
}

static int g_num_cus = 0;
static int num_cus() {
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

// Grid of an MFMA scan (see scan_slot): 1-D, 8 * m * n_qtiles blocks, one block per CU.  m is
// chosen for the fullest last round of blocks, the smallest such m first (fewer, longer row
// slices; at 32 query tiles m = 1 and the whole launch is a single round).
static dim3 scan_grid(int ntiles, int64_t n_row_tiles, int waves, bool* shared_rows,
                      int blocks_per_cu = 1, int m_cap = 64) {
    const int cus = num_cus() * blocks_per_cu;   // block slots
    int64_t m_max = n_row_tiles / (8 * (int64_t)waves);  // every wave gets at least one row tile
    if (m_max < 1) m_max = 1;
    if (m_max > m_cap) m_max = m_cap;
    int best_m = 1;
    double best_eff = 0.0;
    for (int m = 1; m <= (int)m_max; ++m) {
        const int64_t g = 8 * (int64_t)m * ntiles;
        const int64_t rounds = (g + cus - 1) / cus;
        const double eff = (double)g / (double)(rounds * cus);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best_m = m;
        }
    }
    *shared_rows = ntiles > 1;
    return dim3((unsigned)(8 * best_m * ntiles), 1u);
}

// row loads: non-temporal when no other query tile will ask for the same lines, plain when the
// query tiles of a slice share them through L2
static bool scan_nt(bool shared_rows) { return !shared_rows; }

template <int MODE>
static int launch_scan_mfma(int dim, const float* docs, const float* inv_norm, int64_t n_docs,
                            const float* queries, int n_queries, int ntiles, int64_t n_row_tiles,
                            int64_t tile_stride, const float* tau, int* tile_cnt, Cand* tile_list,
                            int tile_cap, float* sample, int64_t sample_ld, hipStream_t st,
                            const int32_t* doc_coll = nullptr, const int32_t* query_coll = nullptr) {
    const size_t lds1 = sizeof(float) * MF_QT * (size_t)dim + sizeof(Cand) * MF_WAVES * WBUF;
    auto lds2_for = [&](int nw) {
        return sizeof(float) * MF_QT * (size_t)dim + (sizeof(Cand) * WBUF + sizeof(float4) * MF2_STAGE_F4) * nw;
    };
    // v2 adds a 4 KiB transpose tile per wave on top of the query tile: 8 waves fit up to dim
    // 768.  At dim 1024 only 4 waves (one per SIMD) would fit, and that measured slower than
    // the fragment-load variant with 8 waves (4.30 vs 4.93 TB/s), which therefore runs there.
    int nw = 0;
    if (dim % 128 == 0 && dim >= 256)
        nw = lds2_for(8) <= 160 * 1024 ? 8 : 0;
    const bool v2 = nw != 0;
    const size_t lds = v2 ? lds2_for(nw) : lds1;
    const int waves = v2 ? nw : MF_WAVES;
    bool shared_rows = false;
    const dim3 grid = scan_grid(ntiles, n_row_tiles, waves, &shared_rows);
    const bool nt = scan_nt(shared_rows);
#define THR_MF_LAUNCH(KERN, THREADS)                                                              \
    {                                                                                             \
        auto kern = KERN;                                                                         \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, st, docs, inv_norm, n_docs, queries,   \
                           n_queries, n_row_tiles, tile_stride, tau, tile_cnt, tile_list,         \
                           tile_cap, sample, sample_ld, doc_coll, query_coll);                    \
    }
#define THR_MF_CASE(D8)                                                                           \
    case D8:                                                                                      \
        if (!v2) THR_MF_LAUNCH((dense_scan_mfma<D8, MODE>), MF_THREADS)                           \
        else if (nt) THR_MF_LAUNCH((dense_scan_mfma2<D8, MODE, true, 8>), 512)                    \
        else THR_MF_LAUNCH((dense_scan_mfma2<D8, MODE, false, 8>), 512)                           \
        break;
    switch (dim / 8) {
        THR_MF_CASE(32)
        THR_MF_CASE(64)
        THR_MF_CASE(96)
        THR_MF_CASE(128)
        default:
            return THR_ERR_UNSUPPORTED;
    }
#undef THR_MF_CASE
#undef THR_MF_LAUNCH
    return launch_status();
}

// the in-flight-rounding f16 scan: streams the float32 rows and rounds them in registers
template <int MODE>
static int launch_scan_f16(int dim, int nq, const float* rows32, const float* inv_norm,
                           int64_t n_docs, const float* queries, int n_queries, int ntiles,
                           int64_t n_row_tiles, int64_t tile_stride, const float* tau, int* tile_cnt,
                           Cand* tile_list, int tile_cap, float* sample, int64_t sample_ld,
                           hipStream_t st, const int32_t* doc_coll = nullptr,
                           const int32_t* query_coll = nullptr) {
    const size_t lds = f16_lds_bytes(dim, nq);
    THR_RETURN_IF(lds > 160 * 1024, THR_ERR_UNSUPPORTED);
    bool shared_rows = false;
    const dim3 grid = scan_grid(ntiles, n_row_tiles, H_WAVES, &shared_rows);
    const bool nt = scan_nt(shared_rows);
#define THR_H_LAUNCH(KERN)                                                                        \
    {                                                                                             \
        auto kern = KERN;                                                                         \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(H_THREADS), lds, st, rows32, inv_norm, n_docs,        \
                           queries, n_queries, n_row_tiles, tile_stride, tau, tile_cnt, tile_list, \
                           tile_cap, sample, sample_ld, doc_coll, query_coll);                    \
    }
#define THR_H_INLINE(DIM, NQV)                                                                    \
    {                                                                                             \
        if (nt) THR_H_LAUNCH((dense_scan_f16<DIM, MODE, true, NQV>))                              \
        else THR_H_LAUNCH((dense_scan_f16<DIM, MODE, false, NQV>))                                \
    }
    switch (dim * 10 + nq) {
        case 5122: THR_H_INLINE(512, 2) break;
        case 7682: THR_H_INLINE(768, 2) break;
        case 10241: THR_H_INLINE(1024, 1) break;
        default: return THR_ERR_UNSUPPORTED;
    }
#undef THR_H_INLINE
#undef THR_H_LAUNCH
    return launch_status();
}

template <int MODE, bool PROF = false>
static int launch_scan_f16q(int dim, const _Float16* rows16, const _Float16* qfrag, int n_qtiles,
                            int64_t n_row_tiles, int64_t tile_stride, const float* tau, int* seg_cnt,
                            Cand* cand, float* sample, int64_t sample_ld, hipStream_t st,
                            int* nseg_out = nullptr, const int32_t* doc_coll = nullptr,
                            const int32_t* query_coll = nullptr, int n_queries = 1 << 30,
                            unsigned long long* stamps = nullptr, int* n_blocks = nullptr) {
    bool shared_rows = false;
    const bool stag = qreg_staggered(dim);
    // a lane's candidate segment is (row slice, row half): at most 256 slices (512 segments, two
    // per thread of select_rescore) -- enough for one block per CU when the batch is a single
    // workgroup tile of queries
    const dim3 grid = scan_grid(n_qtiles, n_row_tiles, 1, &shared_rows, (!stag && dim <= 768) ? 2 : 1, 32);
    const int shape = qreg_shape(dim);
    const int nseg = (shape == 16 ? 4 : 2) * (int)(grid.x / n_qtiles);
    if (nseg_out) *nseg_out = nseg;
    if (n_blocks) *n_blocks = (int)grid.x;
    if (PROF && !stamps) return THR_OK;   // size query
#define THR_QS_LAUNCH(DIM, SHAPE)                                                                 \
    {                                                                                             \
        auto kern = dense_scan_f16qs<DIM, MODE, SHAPE>;                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                           QStag<DIM>::LDS_BYTES);                                \
        if (e != hipSuccess) return (int)e;                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(QS_NW * 64), QStag<DIM>::LDS_BYTES, st,               \
                           (const f32x4*)rows16, (const f32x4*)qfrag, n_qtiles, n_row_tiles,      \
                           tile_stride, tau, seg_cnt, cand, CAND_CAP / nseg, sample, sample_ld,   \
                           doc_coll, query_coll, n_queries);                                      \
    }
    if (stag) {
        if (dim == 512 && shape == 32) THR_QS_LAUNCH(512, 32)
        else if (dim == 512) THR_QS_LAUNCH(512, 16)
        else if (shape == 32) THR_QS_LAUNCH(768, 32)
        else THR_QS_LAUNCH(768, 16)
        return launch_status();
    }
#undef THR_QS_LAUNCH
#define THR_Q_LAUNCH(DIM, SHAPE)                                                                  \
    {                                                                                             \
        auto kern = dense_scan_f16q<DIM, MODE, PROF, SHAPE>;                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                           QScan<DIM>::LDS_BYTES);                                \
        if (e != hipSuccess) return (int)e;                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(Q_NW * 64), QScan<DIM>::LDS_BYTES, st,                \
                           (const f32x4*)rows16, (const f32x4*)qfrag, n_qtiles, n_row_tiles,      \
                           tile_stride, tau, seg_cnt, cand, CAND_CAP / nseg, sample, sample_ld,   \
                           doc_coll, query_coll, n_queries, stamps);                              \
    }
    switch (dim) {
        case 512: if (shape == 16) THR_Q_LAUNCH(512, 16) else THR_Q_LAUNCH(512, 32) break;
        case 768: if (shape == 16) THR_Q_LAUNCH(768, 16) else THR_Q_LAUNCH(768, 32) break;
        case 1024:
            if (shape == 16 && qreg_qw(dim) == 48) THR_Q_LAUNCH(1024, 48)
            else if (shape == 16) THR_Q_LAUNCH(1024, 16)
            else THR_Q_LAUNCH(1024, 32)
            break;
        default: return THR_ERR_UNSUPPORTED;
    }
#undef THR_Q_LAUNCH
    return launch_status();
}

static int launch_pack_queries(int dim, const float* queries, int n_queries, int qpad,
                               _Float16* qfrag, float* qerr, hipStream_t st) {
    const dim3 grid((unsigned)(qpad / 32));
    const bool s16 = qreg_shape(dim) == 16;
#define THR_PACK(DIM, SHAPE) \
    hipLaunchKernelGGL((pack_queries_f16<DIM, SHAPE>), grid, dim3(256), 0, st, queries, n_queries, (f32x4*)qfrag, qerr)
    switch (dim) {
        case 512: if (s16) THR_PACK(512, 16); else THR_PACK(512, 32); break;
        case 768: if (s16) THR_PACK(768, 16); else THR_PACK(768, 32); break;
        case 1024: if (s16) THR_PACK(1024, 16); else THR_PACK(1024, 32); break;
        default: return THR_ERR_UNSUPPORTED;
    }
#undef THR_PACK
    return launch_status();
}

// fp32 error bound of the MFMA scan, relative to ||q||*||d||, in units of 2^-24: a dim-long fma chain
static double scan_eps(int dim) {
    const double u = 5.9604644775390625e-08;
    return ((double)dim + 16.0) * u;
}

}  // namespace thr

using namespace thr;

extern "C" size_t thr_dense_workspace_bytes(int64_t n_docs, int dim, int n_queries, int kprime) {
    (void)dim;
    if (n_docs <= 0 || n_queries <= 0) return 0;
    return make_plan(n_docs, n_queries, kprime).total;
}

// K1..K4 for every scan flavour.  p.kind == KIND_F32: float32 MFMA scan; KIND_F16: f16 MFMA scan
// over docs16, or over the float32 rows rounded in flight when docs16 == nullptr.
// phase: PIPE_ALL = one call; PIPE_SHORTLIST = K1..K3 + the top_m lower bounds (the candidate
// lists stay in the workspace); PIPE_FINISH = K4 on those lists with the shards' common floor.
enum { PIPE_ALL = 0, PIPE_SHORTLIST = 1, PIPE_FINISH = 2 };
static int dense_pipeline(const DensePlan& p, const float* docs, const _Float16* docs16,
                          double doc_relerr, const double* dnorm, const float* inv_norm,
                          int64_t n_docs, int dim, int64_t id_base, const float* queries,
                          int n_queries, int k, int kprime, double* out_scores, int64_t* out_ids,
                          int32_t* out_counts, uint32_t* out_flags, char* ws, hipStream_t st,
                          const int32_t* doc_coll, const int32_t* query_coll, int phase = PIPE_ALL,
                          const float* gfloor = nullptr, float* top_lb = nullptr, int top_m = 0,
                          const float* lb_all = nullptr, int n_shards = 0) {
    float* tau = (float*)(ws + p.off_tau);
    const bool h = p.kind == KIND_F16;
    float* qerr = h ? (float*)(ws + p.off_qerr) : nullptr;
    int* cnt = (int*)(ws + p.off_cnt);
    int* tcnt = (int*)(ws + p.off_tcnt);
    Cand* cand = (Cand*)(ws + p.off_cand);
    Cand* tlist = (Cand*)(ws + p.off_tlist);
    float* sample = (float*)(ws + p.off_sample);
    _Float16* qfrag = (_Float16*)(ws + p.off_qfrag);
    int nseg = 0;   // qreg scan: segments per query of the candidate area (else one flat list)
    auto scan = [&](bool all, int64_t units, int64_t stride, float* smp, int64_t ld) -> int {
        if (p.qreg)
            return all ? launch_scan_f16q<MODE_ALL>(dim, docs16, qfrag, p.ntiles, units, stride,
                                                    nullptr, nullptr, nullptr, smp, ld, st, nullptr,
                                                    nullptr, nullptr, n_queries)
                       : launch_scan_f16q<MODE_FILTER>(dim, docs16, qfrag, p.ntiles, units, stride,
                                                       tau, cnt, cand, nullptr, 0, st, &nseg, doc_coll,
                                                       query_coll, n_queries);
        if (h)
            return all ? launch_scan_f16<MODE_ALL>(dim, p.nq, docs, inv_norm, n_docs, queries,
                                                   n_queries, p.ntiles, units, stride, nullptr,
                                                   nullptr, nullptr, 0, smp, ld, st)
                       : launch_scan_f16<MODE_FILTER>(dim, p.nq, docs, inv_norm, n_docs, queries,
                                                      n_queries, p.ntiles, units, stride, tau, tcnt,
                                                      tlist, p.tile_cap, nullptr, 0, st, doc_coll, query_coll);
        return all ? launch_scan_mfma<MODE_ALL>(dim, docs, inv_norm, n_docs, queries, n_queries,
                                               p.ntiles, units, stride, nullptr, nullptr, nullptr,
                                               0, smp, ld, st)
                   : launch_scan_mfma<MODE_FILTER>(dim, docs, inv_norm, n_docs, queries, n_queries,
                                                  p.ntiles, units, stride, tau, tcnt, tlist,
                                                  p.tile_cap, nullptr, 0, st, doc_coll, query_coll);
    };
    int rc;
    const double u = 5.9604644775390625e-08;
    const double eps32 = h ? ((double)dim + 16.0) * u : scan_eps(dim);
    int32_t* sel_rows = (int32_t*)(ws + p.off_selrows);
    int32_t* sel_meta = (int32_t*)(ws + p.off_selmeta);
    if (phase == PIPE_FINISH) {
        // (the candidate area's layout as the filter scan of the shortlist call left it)
        if (p.qreg && (rc = launch_scan_f16q<MODE_FILTER, true>(dim, docs16, qfrag, p.ntiles, p.groups, 1,
                                                                 nullptr, nullptr, nullptr, nullptr, 0, st,
                                                                 &nseg)))
            return rc;
    } else {
    hipError_t e = hipMemsetAsync(cnt, 0, p.off_cand - p.off_cnt, st);  // cnt + tcnt
    if (e != hipSuccess) return (int)e;
    // (the qreg scan's query image comes with the query-side error term; kth_select then skips it)
    if (p.qreg && (rc = launch_pack_queries(dim, queries, n_queries, p.qpad, qfrag, qerr, st))) return rc;
    float* qerr_k2 = p.qreg ? nullptr : qerr;
    if (p.sampled) {
        if ((rc = scan(true, p.sample_groups, p.sample_stride, sample, p.sample_docs))) return rc;
        hipLaunchKernelGGL(kth_select, dim3(p.qpad), dim3(256), 0, st, sample, p.sample_docs,
                           (int)p.sample_docs, p.ksample, queries, n_queries, dim, tau, qerr_k2,
                           doc_coll, query_coll, p.unit, p.sample_stride, n_docs);
    } else {
        hipLaunchKernelGGL(kth_select, dim3(p.qpad), dim3(256), 0, st, (const float*)nullptr,
                           (int64_t)0, 0, p.ksample, queries, n_queries, dim, tau, qerr_k2,
                           doc_coll, query_coll, p.unit, (int64_t)1, n_docs);
    }
    if ((rc = launch_status())) return rc;
    if ((rc = scan(false, p.groups, 1, nullptr, 0))) return rc;
    if (!p.qreg) {  // the qreg scan writes the per-query lists itself
        hipLaunchKernelGGL(bucket_candidates, dim3(BUCKET_BLOCKS, p.ntiles), dim3(256), 0, st, tcnt,
                           tlist, p.tile_cap, p.qtile, p.row_bits, cnt, cand);
        if ((rc = launch_status())) return rc;
    }
    }
    if (phase == PIPE_SHORTLIST) {
        hipLaunchKernelGGL(select_band<true>, dim3(n_queries), dim3(SEL_THREADS), band_lds_bytes(dim), st,
                           dim, queries, tau, cnt, cand, tcnt, p.tile_cap, p.qtile, k, kprime, eps32,
                           doc_relerr, qerr, nseg, nseg ? CAND_CAP / nseg : 0, doc_coll, query_coll,
                           sel_rows, sel_meta, (const float*)nullptr, (const float*)nullptr, 0, 0, top_lb,
                           top_m);
        return launch_status();
    }
    hipLaunchKernelGGL(select_band<false>, dim3(n_queries), dim3(SEL_THREADS), band_lds_bytes(dim), st,
                       dim, queries, tau, cnt, cand, tcnt, p.tile_cap, p.qtile, k, kprime, eps32,
                       doc_relerr, qerr, nseg, nseg ? CAND_CAP / nseg : 0, doc_coll, query_coll,
                       sel_rows, sel_meta, gfloor, lb_all, n_shards, top_m, (float*)nullptr, 0);
    if ((rc = launch_status())) return rc;
    hipLaunchKernelGGL(rescore_rank<THR_DENSE_MAX_K>, dim3(n_queries), dim3(RR_THREADS),
                       rescore_lds_bytes(dim), st, docs, dnorm, dim, id_base, queries, k, eps32,
                       doc_relerr, qerr, sel_rows, sel_meta, out_scores, out_ids, out_counts, out_flags);
    if ((rc = launch_status())) return rc;
    // the queries whose band did not fit 256 rows
    hipLaunchKernelGGL(rescore_rank<SEL_BIG_BAND>, dim3(n_queries), dim3(RR_THREADS),
                       rescore_lds_bytes(dim), st, docs, dnorm, dim, id_base, queries, k, eps32,
                       doc_relerr, qerr, sel_rows, sel_meta, out_scores, out_ids, out_counts, out_flags);
    return launch_status();
}

static int dense_args_ok(const void* docs, const void* dnorm, const void* inv_norm,
                         const void* queries, const void* a, const void* b, const void* c,
                         const void* d, const void* ws, int64_t n_docs, int n_queries, int k,
                         int kprime) {
    THR_RETURN_IF(!docs || !dnorm || !inv_norm || !queries || !a || !b || !c || !d || !ws,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_docs <= 0 || n_queries <= 0 || k <= 0 || kprime < k ||
                      kprime > THR_DENSE_MAX_K,
                  THR_ERR_INVALID);
    return THR_OK;
}

extern "C" int thr_dense_topk(const float* docs, const double* dnorm, const float* inv_norm,
                              int64_t n_docs, int dim, int64_t id_base, const float* queries,
                              int n_queries, int k, int kprime, const int32_t* doc_coll,
                              const int32_t* query_coll, double* out_scores,
                              int64_t* out_ids, int32_t* out_counts, uint32_t* out_flags,
                              void* workspace, size_t workspace_bytes, thr_stream_t stream) {
    clear_status();
    int rc = dense_args_ok(docs, dnorm, inv_norm, queries, out_scores, out_ids, out_counts,
                           out_flags, workspace, n_docs, n_queries, k, kprime);
    if (rc) return rc;
    THR_RETURN_IF((query_coll != nullptr) != (doc_coll != nullptr), THR_ERR_INVALID);
    THR_RETURN_IF(dim <= 0 || dim % CHUNK != 0 || dim / CHUNK > 4, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(n_docs >= (int64_t)1 << ROW_BITS, THR_ERR_UNSUPPORTED);
    const DensePlan p = make_plan(n_docs, n_queries, kprime);
    THR_RETURN_IF(workspace_bytes < p.total, THR_ERR_WORKSPACE);
    return dense_pipeline(p, docs, nullptr, 0.0, dnorm, inv_norm, n_docs, dim, id_base, queries,
                          n_queries, k, kprime, out_scores, out_ids, out_counts, out_flags,
                          (char*)workspace, (hipStream_t)stream, doc_coll, query_coll);
}

extern "C" size_t thr_dense_f16_workspace_bytes(int64_t n_docs, int dim, int n_queries,
                                                int kprime) {
    if (n_docs <= 0 || n_queries <= 0) return 0;
    const size_t a = make_plan(n_docs, n_queries, kprime, KIND_F16, dim, false).total;
    const size_t b = make_plan(n_docs, n_queries, kprime, KIND_F16, dim, true).total;
    return a > b ? a : b;
}

extern "C" int thr_dense_f16_max_queries(int dim, int packed) {
    if (dim != 512 && dim != 768 && dim != 1024) return 0;
    return packed ? qreg_max_queries(dim) : INT32_MAX;
}

extern "C" int thr_dense_f16_query_tile(int dim, int packed, int n_queries) {
    if (dim != 512 && dim != 768 && dim != 1024) return 0;
    (void)n_queries;
    return packed ? qreg_qw(dim) * qreg_waves(dim) : 32 * f16_pick_nq(dim);
}

extern "C" size_t thr_dense_f16_copy_bytes(int64_t n_docs, int dim) {
    if (n_docs <= 0 || dim <= 0) return 0;
    return sizeof(_Float16) * (size_t)((n_docs + 31) / 32 * 32) * (size_t)dim;
}

extern "C" int thr_dense_quantize_f16(const float* docs, int64_t n_docs, int dim, uint16_t* docs16,
                                      float* max_rel_err, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!docs || !max_rel_err || n_docs <= 0 || dim <= 0, THR_ERR_INVALID);
    THR_RETURN_IF(dim % 64 != 0, THR_ERR_UNSUPPORTED);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(max_rel_err, 0, sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (docs16) {
        // normalised rows, NaN for rows without an embedding and for the padding of the last tile
        THR_RETURN_IF(dim % 16 != 0, THR_ERR_UNSUPPORTED);
        const int64_t n_pad = (n_docs + 31) / 32 * 32;
        hipLaunchKernelGGL(quantize_f16_norm, dim3((unsigned)((n_pad + 3) / 4)), dim3(256), 0, st,
                           docs, n_docs, dim, qreg_shape(dim), reinterpret_cast<_Float16*>(docs16),
                           reinterpret_cast<unsigned int*>(max_rel_err));
        return launch_status();
    }
    // measure only: the in-flight-rounding scan rounds the rows as they are
    hipLaunchKernelGGL(measure_f16_error, dim3((unsigned)((n_docs + 3) / 4)), dim3(256), 0, st, docs,
                       n_docs, dim, reinterpret_cast<unsigned int*>(max_rel_err));
    return launch_status();
}

extern "C" int thr_dense_topk_f16(const float* docs, const uint16_t* docs16, double doc_rel_err,
                                  const double* dnorm, const float* inv_norm, int64_t n_docs,
                                  int dim, int64_t id_base, const float* queries, int n_queries,
                                  int k, int kprime, const int32_t* doc_coll,
                                  const int32_t* query_coll, double* out_scores, int64_t* out_ids,
                                  int32_t* out_counts, uint32_t* out_flags, void* workspace,
                                  size_t workspace_bytes, thr_stream_t stream) {
    clear_status();
    int rc = dense_args_ok(docs, dnorm, inv_norm, queries, out_scores, out_ids, out_counts,
                           out_flags, workspace, n_docs, n_queries, k, kprime);
    if (rc) return rc;
    THR_RETURN_IF((query_coll != nullptr) != (doc_coll != nullptr), THR_ERR_INVALID);
    THR_RETURN_IF(!(doc_rel_err >= 0.0) || !(doc_rel_err < 1.0), THR_ERR_INVALID);
    THR_RETURN_IF(dim != 512 && dim != 768 && dim != 1024, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(n_docs >= (int64_t)1 << ROW_BITS_F16, THR_ERR_UNSUPPORTED);
    // (32-bit candidate-segment offsets: thr_dense_f16_max_queries; the caller splits the batch)
    THR_RETURN_IF(docs16 != nullptr && n_queries > qreg_max_queries(dim), THR_ERR_UNSUPPORTED);
    const DensePlan p = make_plan(n_docs, n_queries, kprime, KIND_F16, dim, docs16 != nullptr);
    THR_RETURN_IF(workspace_bytes < p.total, THR_ERR_WORKSPACE);
    return dense_pipeline(p, docs, reinterpret_cast<const _Float16*>(docs16), doc_rel_err, dnorm,
                          inv_norm, n_docs, dim, id_base, queries, n_queries, k, kprime, out_scores,
                          out_ids, out_counts, out_flags, (char*)workspace, (hipStream_t)stream,
                          doc_coll, query_coll);
}

static int f16_args_ok(const float* docs, const uint16_t* docs16, double doc_rel_err, int64_t n_docs,
                       int dim, int n_queries, const int32_t* doc_coll, const int32_t* query_coll) {
    (void)docs;
    THR_RETURN_IF((query_coll != nullptr) != (doc_coll != nullptr), THR_ERR_INVALID);
    THR_RETURN_IF(!(doc_rel_err >= 0.0) || !(doc_rel_err < 1.0), THR_ERR_INVALID);
    THR_RETURN_IF(dim != 512 && dim != 768 && dim != 1024, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(n_docs >= (int64_t)1 << ROW_BITS_F16, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(docs16 != nullptr && n_queries > qreg_max_queries(dim), THR_ERR_UNSUPPORTED);
    return THR_OK;
}

extern "C" int thr_dense_shortlist_f16(const float* docs, const uint16_t* docs16, double doc_rel_err,
                                       const float* inv_norm, int64_t n_docs, int dim,
                                       const float* queries, int n_queries, int kprime,
                                       const int32_t* doc_coll, const int32_t* query_coll, int m,
                                       float* top_lb, void* workspace, size_t workspace_bytes,
                                       thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!docs || !inv_norm || !queries || !top_lb || !workspace, THR_ERR_INVALID);
    THR_RETURN_IF(n_docs <= 0 || n_queries <= 0 || kprime <= 0 || kprime > THR_DENSE_MAX_K || m <= 0 ||
                      m > THR_DENSE_MAX_K,
                  THR_ERR_INVALID);
    int rc = f16_args_ok(docs, docs16, doc_rel_err, n_docs, dim, n_queries, doc_coll, query_coll);
    if (rc) return rc;
    const DensePlan p = make_plan(n_docs, n_queries, kprime, KIND_F16, dim, docs16 != nullptr);
    THR_RETURN_IF(workspace_bytes < p.total, THR_ERR_WORKSPACE);
    return dense_pipeline(p, docs, reinterpret_cast<const _Float16*>(docs16), doc_rel_err, nullptr,
                          inv_norm, n_docs, dim, 0, queries, n_queries, 0, kprime, nullptr, nullptr,
                          nullptr, nullptr, (char*)workspace, (hipStream_t)stream, doc_coll, query_coll,
                          PIPE_SHORTLIST, nullptr, top_lb, m);
}

extern "C" int thr_dense_floor(const float* top_lb, int n_shards, int n_queries, int m, int k,
                               float* gfloor, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!top_lb || !gfloor || n_shards <= 0 || n_queries <= 0 || m <= 0 || k <= 0, THR_ERR_INVALID);
    THR_RETURN_IF((int64_t)n_shards * m > 8192, THR_ERR_CAPACITY);
    hipLaunchKernelGGL(dense_floor_kernel, dim3(n_queries), dim3(256), sizeof(float) * n_shards * m,
                       (hipStream_t)stream, top_lb, n_shards, n_queries, m, k, gfloor);
    return launch_status();
}

extern "C" int thr_dense_finish_f16(const float* docs, const uint16_t* docs16, double doc_rel_err,
                                    const double* dnorm, const float* inv_norm, int64_t n_docs,
                                    int dim, int64_t id_base, const float* queries, int n_queries,
                                    int k, int kprime, const int32_t* doc_coll,
                                    const int32_t* query_coll, const float* gfloor,
                                    const float* top_lb_all, int n_shards, int m,
                                    double* out_scores, int64_t* out_ids, int32_t* out_counts,
                                    uint32_t* out_flags, void* workspace, size_t workspace_bytes,
                                    thr_stream_t stream) {
    clear_status();
    int rc = dense_args_ok(docs, dnorm, inv_norm, queries, out_scores, out_ids, out_counts,
                           out_flags, workspace, n_docs, n_queries, k, kprime);
    if (rc) return rc;
    if ((rc = f16_args_ok(docs, docs16, doc_rel_err, n_docs, dim, n_queries, doc_coll, query_coll))) return rc;
    THR_RETURN_IF(gfloor && top_lb_all, THR_ERR_INVALID);
    THR_RETURN_IF(top_lb_all && (n_shards <= 0 || m <= 0), THR_ERR_INVALID);
    THR_RETURN_IF(top_lb_all && (int64_t)n_shards * m > CS_BINS, THR_ERR_CAPACITY);
    const DensePlan p = make_plan(n_docs, n_queries, kprime, KIND_F16, dim, docs16 != nullptr);
    THR_RETURN_IF(workspace_bytes < p.total, THR_ERR_WORKSPACE);
    return dense_pipeline(p, docs, reinterpret_cast<const _Float16*>(docs16), doc_rel_err, dnorm,
                          inv_norm, n_docs, dim, id_base, queries, n_queries, k, kprime, out_scores,
                          out_ids, out_counts, out_flags, (char*)workspace, (hipStream_t)stream,
                          doc_coll, query_coll, PIPE_FINISH, gfloor, nullptr, m, top_lb_all, n_shards);
}

extern "C" int thr_dense_scan_probe(const float* docs, const float* inv_norm, int64_t n_docs,
                                    int dim, const float* queries, int n_queries, void* workspace,
                                    size_t workspace_bytes, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!docs || !inv_norm || !queries || !workspace, THR_ERR_INVALID);
    THR_RETURN_IF(dim <= 0 || dim % CHUNK != 0 || dim / CHUNK > 4, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(n_docs <= 0 || n_docs >= (int64_t)1 << ROW_BITS || n_queries <= 0,
                  THR_ERR_INVALID);
    const DensePlan p = make_plan(n_docs, n_queries, 128);
    THR_RETURN_IF(workspace_bytes < p.total, THR_ERR_WORKSPACE);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    // tau is whatever the last thr_dense_topk on this workspace left (a realistic filter
    // rate); the tile counters are reset so the lists never overflow across repeats
    hipError_t e = hipMemsetAsync(ws + p.off_tcnt, 0, sizeof(int) * p.ntiles, st);
    if (e != hipSuccess) return (int)e;
    return launch_scan_mfma<MODE_FILTER>(dim, docs, inv_norm, n_docs, queries, n_queries, p.ntiles,
                                        p.groups, 1, (const float*)(ws + p.off_tau),
                                        (int*)(ws + p.off_tcnt), (Cand*)(ws + p.off_tlist),
                                        p.tile_cap, nullptr, 0, st);
}

extern "C" int thr_dense_scan_probe_f16(const float* docs, const uint16_t* docs16,
                                        const float* inv_norm,
                                        int64_t n_docs, int dim, const float* queries,
                                        int n_queries, void* workspace, size_t workspace_bytes,
                                        thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF((!docs16 && !docs) || !inv_norm || !queries || !workspace, THR_ERR_INVALID);
    THR_RETURN_IF(dim != 512 && dim != 768 && dim != 1024, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(n_docs <= 0 || n_docs >= (int64_t)1 << ROW_BITS_F16 || n_queries <= 0,
                  THR_ERR_INVALID);
    THR_RETURN_IF(docs16 != nullptr && n_queries > qreg_max_queries(dim), THR_ERR_UNSUPPORTED);
    const DensePlan p = make_plan(n_docs, n_queries, 128, KIND_F16, dim, docs16 != nullptr);
    THR_RETURN_IF(workspace_bytes < p.total, THR_ERR_WORKSPACE);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (p.qreg) {
        // the query image and tau are the last thr_dense_topk_f16's
        return launch_scan_f16q<MODE_FILTER>(dim, reinterpret_cast<const _Float16*>(docs16),
                                             (const _Float16*)(ws + p.off_qfrag), p.ntiles, p.groups,
                                             1, (const float*)(ws + p.off_tau), (int*)(ws + p.off_cnt),
                                             (Cand*)(ws + p.off_cand), nullptr, 0, st, nullptr,
                                             nullptr, nullptr, n_queries);
    }
    hipError_t e = hipMemsetAsync(ws + p.off_tcnt, 0, sizeof(int) * p.ntiles, st);
    if (e != hipSuccess) return (int)e;
    return launch_scan_f16<MODE_FILTER>(dim, p.nq, docs, inv_norm,
                                        n_docs, queries, n_queries, p.ntiles, p.groups, 1,
                                        (const float*)(ws + p.off_tau), (int*)(ws + p.off_tcnt),
                                        (Cand*)(ws + p.off_tlist), p.tile_cap, nullptr, 0, st);
}

extern "C" int thr_dense_scan_stamps_f16(const uint16_t* docs16, int64_t n_docs, int dim,
                                         int n_queries, void* workspace, size_t workspace_bytes,
                                         unsigned long long* stamps, int* h_n_waves,
                                         thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!docs16 || !workspace || !h_n_waves, THR_ERR_INVALID);
    THR_RETURN_IF(dim != 512 && dim != 768 && dim != 1024, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(n_docs <= 0 || n_queries <= 0, THR_ERR_INVALID);
    // (only the 4-wave-block kernel has a stamped build: THR_DENSE_F16=q, or dim 1024)
    THR_RETURN_IF(qreg_staggered(dim), THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(n_queries > qreg_max_queries(dim), THR_ERR_UNSUPPORTED);
    const DensePlan p = make_plan(n_docs, n_queries, 128, KIND_F16, dim, true);
    THR_RETURN_IF(workspace_bytes < p.total, THR_ERR_WORKSPACE);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    int blocks = 0;
    int rc = launch_scan_f16q<MODE_FILTER, true>(
        dim, reinterpret_cast<const _Float16*>(docs16), (const _Float16*)(ws + p.off_qfrag), p.ntiles,
        p.groups, 1, (const float*)(ws + p.off_tau), (int*)(ws + p.off_cnt), (Cand*)(ws + p.off_cand),
        nullptr, 0, st, nullptr, nullptr, nullptr, n_queries, stamps, &blocks);
    *h_n_waves = blocks * qreg_waves(dim);
    return rc;
}

extern "C" size_t thr_dense_exact_workspace_bytes(int64_t n_docs, int n_queries) {
    (void)n_docs;
    return (size_t)n_queries * EX_SLABS * THR_DENSE_MAX_K * (sizeof(double) + sizeof(int64_t));
}

extern "C" int thr_dense_topk_exact(const float* docs, const double* dnorm, int64_t n_docs, int dim,
                                    int64_t id_base, const float* queries, int n_queries, int k,
                                    const int32_t* doc_coll, const int32_t* query_coll,
                                    double* out_scores, int64_t* out_ids, int32_t* out_counts,
                                    uint32_t* out_flags, void* workspace, size_t workspace_bytes,
                                    thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!docs || !dnorm || !queries || !out_scores || !out_ids || !out_counts ||
                      !out_flags || !workspace,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_docs <= 0 || n_queries <= 0 || k <= 0 || k > THR_DENSE_MAX_K, THR_ERR_INVALID);
    THR_RETURN_IF(dim <= 0 || dim % 4 != 0, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(workspace_bytes < thr_dense_exact_workspace_bytes(n_docs, n_queries),
                  THR_ERR_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    double* slab_s = (double*)workspace;
    int64_t* slab_id = (int64_t*)(slab_s + (size_t)n_queries * EX_SLABS * k);
    hipLaunchKernelGGL(exact_slab_topk, dim3(EX_SLABS, n_queries), dim3(EX_THREADS),
                       sizeof(float) * dim, st, docs, dnorm, n_docs, dim, queries, n_queries, k,
                       slab_s, slab_id, (const uint32_t*)nullptr, doc_coll, query_coll);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(merge_lists, dim3(n_queries), dim3(256), 0, st, slab_s, slab_id,
                       (int64_t)EX_SLABS * k, (int64_t)k, EX_SLABS, k, k, id_base,
                       THR_FLAG_CERTIFIED | THR_FLAG_EXACT, out_scores, out_ids, out_counts,
                       out_flags);
    return launch_status();
}

extern "C" size_t thr_dense_rescue_workspace_bytes(int n_queries, int k) {
    if (n_queries <= 0 || k <= 0) return 0;
    return (size_t)n_queries * EX_SLABS * (size_t)k * (sizeof(double) + sizeof(int64_t));
}

// Device-side completion of thr_dense_topk[_f16]: the queries whose flags lack
// THR_FLAG_CERTIFIED are redone on the exhaustive float64 path, in place, with no host read-back
// (workgroups of certified queries exit at once).  *n_rescued (device int32) is incremented
// once per redone query.
extern "C" int thr_dense_rescue(const float* docs, const double* dnorm, int64_t n_docs, int dim,
                                int64_t id_base, const float* queries, int n_queries, int k,
                                const int32_t* doc_coll, const int32_t* query_coll,
                                double* io_scores, int64_t* io_ids, int32_t* io_counts,
                                uint32_t* io_flags, int32_t* n_rescued, void* workspace,
                                size_t workspace_bytes, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!docs || !dnorm || !queries || !io_scores || !io_ids || !io_counts ||
                      !io_flags || !workspace,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_docs <= 0 || n_queries <= 0 || k <= 0 || k > THR_DENSE_MAX_K, THR_ERR_INVALID);
    THR_RETURN_IF(dim <= 0 || dim % 4 != 0, THR_ERR_UNSUPPORTED);
    THR_RETURN_IF(workspace_bytes < thr_dense_rescue_workspace_bytes(n_queries, k),
                  THR_ERR_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    double* slab_s = (double*)workspace;
    int64_t* slab_id = (int64_t*)(slab_s + (size_t)n_queries * EX_SLABS * k);
    const int rows = (n_queries + 63) / 64 > 16 ? (n_queries + 63) / 64 : (n_queries < 16 ? n_queries : 16);
    hipLaunchKernelGGL(exact_slab_topk, dim3(EX_SLABS, rows), dim3(EX_THREADS),
                       sizeof(float) * dim, st, docs, dnorm, n_docs, dim, queries, n_queries, k,
                       slab_s, slab_id, (const uint32_t*)io_flags, doc_coll, query_coll);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(merge_lists, dim3(n_queries), dim3(256), 0, st, slab_s, slab_id,
                       (int64_t)EX_SLABS * k, (int64_t)k, EX_SLABS, k, k, id_base,
                       THR_FLAG_CERTIFIED | THR_FLAG_EXACT, io_scores, io_ids, io_counts, io_flags,
                       (const uint32_t*)io_flags, n_rescued);
    return launch_status();
}

extern "C" int thr_merge_topk(const double* in_scores, const int64_t* in_ids, int n_queries,
                              int n_lists, int k_in, int64_t list_stride, int k_out,
                              double* out_scores, int64_t* out_ids, int32_t* out_counts,
                              thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!in_scores || !in_ids || !out_scores || !out_ids, THR_ERR_INVALID);
    THR_RETURN_IF(n_queries <= 0 || n_lists <= 0 || k_in <= 0 || k_out <= 0 || k_out > EX_CAP / 2,
                  THR_ERR_INVALID);
    if (list_stride == 0) list_stride = (int64_t)n_queries * k_in;  // [n_lists, n_queries, k_in]
    THR_RETURN_IF(list_stride < (int64_t)n_queries * k_in, THR_ERR_INVALID);
    if ((int64_t)n_lists * k_in <= EX_CAP) {
        hipLaunchKernelGGL(merge_ranked_lists, dim3(n_queries), dim3(256), 0, (hipStream_t)stream,
                           in_scores, in_ids, (int64_t)k_in, list_stride, n_lists, k_in, k_out,
                           out_scores, out_ids, out_counts);
        return launch_status();
    }
    hipLaunchKernelGGL(merge_lists, dim3(n_queries), dim3(256), 0, (hipStream_t)stream, in_scores,
                       in_ids, (int64_t)k_in, list_stride, n_lists, k_in, k_out,
                       (int64_t)0, 0u, out_scores, out_ids, out_counts, (uint32_t*)nullptr);
    return launch_status();
}
