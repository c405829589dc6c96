// Late-interaction rerank: ColBERT MaxSim on the CDNA4 matrix cores (gfx950).
//
// Stands where the reference calls Qwen3VLReranker._rerank_batch_native
// (src/voice_agent/rag2/retrieval.py:427 ->
// src/voice_agent/retrieval/reranker.py:287-354), an HTTP cross-encoder; the
// reference has no token-level arithmetic, so the score form is the oracle's
// (oracle/thr_oracle.py maxsim_scores):
//     score(q, c) = sum_i max_j <qtok[q,i,:], dtok[c,j,:]>      (float16 inputs)
//
// One wave per (query, candidate).  The product is formed TRANSPOSED,
// S^T = D * Q^T, with v_mfma_f32_32x32x16_f16: A = 32 doc tokens x 16 dims,
// B = 16 dims x 32 query tokens.  The accumulator then has the query token on
// the lane (col = lane & 31) and doc tokens in its 16 registers, so max over
// doc tokens is a per-lane register max, one lane-pair exchange joins the two
// row halves, and the sum over query tokens is a 5-step shuffle.  Both
// operands are 16-byte fragment loads straight from HBM/L2 (each doc-token row
// is streamed once and not shared between waves: no LDS round trip).
//
// Token layout.  Row-major [doc][token][dim] makes a fragment load touch 32 token rows x 32
// bytes, and every 128-byte line is touched by four different load instructions of the wave.
// thr_maxsim_pack re-lays the store FRAGMENT-MAJOR at index build -- [doc][tile of 32
// tokens][k-step][lane][8 halves], the exact register image of the A operand -- so a load
// instruction reads 1 KiB contiguous and each line is touched once (PACKED = true).
// fp16*fp16 products are exact in the fp32 accumulator; only the 128-term
// accumulation rounds (tests: 1e-4 absolute on scores <= 32).
// Algorithmic bytes per (q, c): d_tokens*tok_dim*2 (32 KiB at 128x128); flops 2*q_tokens*d_tokens*tok_dim.
#include "thr_common.hpp"

namespace thr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MS_THREADS = 256;
constexpr int MS_WAVES = MS_THREADS / WAVE;
constexpr int MS_MAX_KSTEPS = 16;  // tok_dim <= 256

// row-major tokens -> fragment-major: one thread per 16-byte chunk
__global__ __launch_bounds__(256) void maxsim_pack_kernel(const _Float16* __restrict__ dtok,
                                                          int64_t n_chunks, int d_tokens,
                                                          int ksteps, _Float16* __restrict__ packed) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // output chunk index
    if (o >= n_chunks) return;
    const int lane = (int)(o & 63);
    const int64_t rest = o >> 6;
    const int ks = (int)(rest % ksteps);
    const int64_t tile_g = rest / ksteps;                               // global tile index
    const int tiles = d_tokens / 32;
    const int64_t doc = tile_g / tiles;
    const int t = (int)(tile_g % tiles);
    const int r = lane & 31, h = lane >> 5;
    const int64_t src = ((doc * d_tokens + 32 * t + r) * ksteps + ks) * 2 + h;  // in 16-B chunks
    reinterpret_cast<half8*>(packed)[o] = reinterpret_cast<const half8*>(dtok)[src];
}

template <int KSTEPS, bool PACKED>
__global__ __launch_bounds__(MS_THREADS) void maxsim_kernel(
    const _Float16* __restrict__ qtok, int q_tokens, const _Float16* __restrict__ dtok,
    int64_t n_docs, int d_tokens, const int32_t* __restrict__ cand,
    const int64_t* __restrict__ cand_ids, int64_t id_base, int n_cand, float* __restrict__ out) {
    constexpr int TD = KSTEPS * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.y;
    const int c = blockIdx.x * MS_WAVES + wave;
    if (c >= n_cand) return;
    // local doc index: given directly, or as a global id of a shard that starts at id_base
    int64_t doc;
    if (cand_ids) {
        const int64_t g = cand_ids[(int64_t)q * n_cand + c];
        doc = g >= 0 ? g - id_base : -1;
    } else {
        doc = cand[(int64_t)q * n_cand + c];
    }
    if (doc < 0 || doc >= n_docs) {
        if (lane == 0) out[(int64_t)q * n_cand + c] = -INFINITY;
        return;
    }
    const int r = lane & 31, h = lane >> 5;
    const _Float16* D = dtok + (int64_t)doc * d_tokens * TD;
    float total = 0.f;
    for (int n0 = 0; n0 < q_tokens; n0 += 32) {
        // B fragments of this 32-query-token tile: lane holds Q[n0 + r][16*ks + 8*h + j]
        half8 bq[KSTEPS];
        const _Float16* Q = qtok + ((int64_t)q * q_tokens + n0 + r) * TD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) bq[ks] = *reinterpret_cast<const half8*>(Q + 16 * ks);
        float mx = -INFINITY;
        for (int m0 = 0; m0 < d_tokens; m0 += 32) {
            half8 a[KSTEPS];
            if constexpr (PACKED) {
                const half8* tile = reinterpret_cast<const half8*>(D) + (int64_t)(m0 / 32) * KSTEPS * 64 + lane;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = __builtin_nontemporal_load(tile + ks * 64);
            } else {
                const _Float16* Drow = D + (int64_t)(m0 + r) * TD + 8 * h;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = *reinterpret_cast<const half8*>(Drow + 16 * ks);
            }
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], bq[ks], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, acc[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));  // the two 16-row halves of every tile
        float s = mx;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, WAVE);
        total += s;
    }
    if (lane == 0) out[(int64_t)q * n_cand + c] = total;
}

}  // namespace thr

using namespace thr;

extern "C" int thr_maxsim_pack(const uint16_t* dtok, int64_t n_docs, int d_tokens, int tok_dim,
                               uint16_t* packed, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!dtok || !packed || dtok == packed || n_docs <= 0, THR_ERR_INVALID);
    THR_RETURN_IF(d_tokens <= 0 || d_tokens % 32 || tok_dim <= 0 || tok_dim % 16 ||
                      tok_dim / 16 > MS_MAX_KSTEPS,
                  THR_ERR_UNSUPPORTED);
    const int64_t n_chunks = n_docs * d_tokens * (tok_dim / 8);
    THR_RETURN_IF((n_chunks + 255) / 256 > 0x7fffffffll, THR_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(maxsim_pack_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const _Float16*>(dtok), n_chunks,
                       d_tokens, tok_dim / 16, reinterpret_cast<_Float16*>(packed));
    return launch_status();
}

static int launch_maxsim(const uint16_t* qtok, int n_queries, int q_tokens, const uint16_t* dtok,
                         int64_t n_docs, int d_tokens, int tok_dim, const int32_t* cand,
                         const int64_t* cand_ids, int64_t id_base, int n_cand, float* out_scores,
                         int dtok_packed, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!qtok || !dtok || (!cand && !cand_ids) || !out_scores, THR_ERR_INVALID);
    THR_RETURN_IF(n_queries <= 0 || n_docs <= 0 || n_cand <= 0, THR_ERR_INVALID);
    THR_RETURN_IF(q_tokens <= 0 || q_tokens % 32 || d_tokens <= 0 || d_tokens % 32 ||
                      tok_dim <= 0 || tok_dim % 16 || tok_dim / 16 > MS_MAX_KSTEPS,
                  THR_ERR_UNSUPPORTED);
    dim3 grid((n_cand + MS_WAVES - 1) / MS_WAVES, n_queries);
    const _Float16* Q = reinterpret_cast<const _Float16*>(qtok);
    const _Float16* Dk = reinterpret_cast<const _Float16*>(dtok);
    hipStream_t st = (hipStream_t)stream;
#define THR_MS_CASE(KS)                                                                          \
    case KS:                                                                                     \
        if (dtok_packed)                                                                         \
            hipLaunchKernelGGL((maxsim_kernel<KS, true>), grid, dim3(MS_THREADS), 0, st, Q,      \
                               q_tokens, Dk, n_docs, d_tokens, cand, cand_ids, id_base, n_cand,  \
                               out_scores);                                                      \
        else                                                                                     \
            hipLaunchKernelGGL((maxsim_kernel<KS, false>), grid, dim3(MS_THREADS), 0, st, Q,     \
                               q_tokens, Dk, n_docs, d_tokens, cand, cand_ids, id_base, n_cand,  \
                               out_scores);                                                      \
        break;
    switch (tok_dim / 16) {
        THR_MS_CASE(1) THR_MS_CASE(2) THR_MS_CASE(4) THR_MS_CASE(6) THR_MS_CASE(8)
        THR_MS_CASE(12) THR_MS_CASE(16)
        default:
            return THR_ERR_UNSUPPORTED;
    }
#undef THR_MS_CASE
    return launch_status();
}

extern "C" int thr_maxsim(const uint16_t* qtok, int n_queries, int q_tokens, const uint16_t* dtok,
                          int64_t n_docs, int d_tokens, int tok_dim, const int32_t* cand, int n_cand,
                          float* out_scores, int dtok_packed, thr_stream_t stream) {
    return launch_maxsim(qtok, n_queries, q_tokens, dtok, n_docs, d_tokens, tok_dim, cand, nullptr, 0,
                         n_cand, out_scores, dtok_packed, stream);
}

extern "C" int thr_maxsim_ids(const uint16_t* qtok, int n_queries, int q_tokens, const uint16_t* dtok,
                              int64_t n_docs, int d_tokens, int tok_dim, const int64_t* cand_ids,
                              int64_t id_base, int n_cand, float* out_scores, int dtok_packed,
                              thr_stream_t stream) {
    THR_RETURN_IF(!cand_ids, THR_ERR_INVALID);
    return launch_maxsim(qtok, n_queries, q_tokens, dtok, n_docs, d_tokens, tok_dim, nullptr, cand_ids,
                         id_base, n_cand, out_scores, dtok_packed, stream);
}
