// Candidate merge + weighted Reciprocal Rank Fusion on the device (gfx950).
//
// Bit-for-bit the arithmetic and ordering of the reference's Python:
//   RAG2Retriever._retrieve_candidates  src/voice_agent/rag2/retrieval.py:203-271
//     1-based rank per channel; candidates in first-sighting order (lexical rows, then new
//     semantic ids, then new graph ids); a duplicate id inside one channel keeps the LAST rank
//   RAG2Retriever._fuse_rrf             src/voice_agent/rag2/retrieval.py:358-376
//     score = 0.0 (+ w_l/(k+lr)) (+ w_s/(k+sr)) (+ w_g/(k+gr)) in float64, that add order;
//     sorted(..., reverse=True) is stable, so ties keep the sighting order.
// One workgroup per query; <= 3*128 candidates, all in LDS.  Host Python does the same for a
// single query (rag2/retrieval.py of this package); this kernel exists because at ~1e5
// queries/s a per-query Python loop would be the bottleneck of the batch path.
#include "thr_common.hpp"

namespace thr {

constexpr int RRF_THREADS = 128;
constexpr int RRF_MAXC = THR_RRF_MAX_PER_CHANNEL;  // per channel
constexpr int RRF_SLOTS = 512;                      // >= 3 * RRF_MAXC, power of two

// last 1-based position of id in l[0..n), 0 if absent
__device__ __forceinline__ int last_rank(const int64_t* l, int n, int64_t id) {
    int r = 0;
    for (int j = 0; j < n; ++j)
        if (l[j] == id) r = j + 1;
    return r;
}
__device__ __forceinline__ bool seen_before(const int64_t* l, int i, int64_t id) {
    for (int j = 0; j < i; ++j)
        if (l[j] == id) return true;
    return false;
}

// MODE 0: RAG2Retriever._fuse_rrf (above).  MODE 1 / 2: the standalone package's RRFFusion
// (triple-hybrid-rag/src/triple_hybrid_rag/core/fusion.py): a channel's table value is
// ``weight * (1.0 / (RRF_K + rank))`` of the id's LAST rank (:167-185); ``fuse`` (:52-165, MODE 1)
// adds it once per OCCURRENCE of the id in the channel, channels in the order lexical, semantic,
// graph; ``fuse_two_channels`` (:249-292, MODE 2; a = the lexical slot, b = the semantic slot)
// assigns a's value once and adds b's per occurrence.  Same sighting order, same stable sort.
constexpr int RRF_RAG2 = 0, RRF_STANDALONE = 1, RRF_TWO = 2;
__device__ __forceinline__ int occurrences(const int64_t* l, int n, int64_t id) {
    int c = 0;
    for (int j = 0; j < n; ++j) c += l[j] == id ? 1 : 0;
    return c;
}

template <int MODE>
__global__ __launch_bounds__(RRF_THREADS) void rrf_fuse_kernel(
    const int64_t* __restrict__ lex, int n_lex, const int64_t* __restrict__ sem, int n_sem,
    const int64_t* __restrict__ gra, int n_gra, double w_lex, double w_sem, double w_gra, int rrf_k,
    int top_k, int64_t* __restrict__ out_ids, double* __restrict__ out_scores,
    int32_t* __restrict__ out_ranks, int32_t* __restrict__ out_counts) {
    // one thread per list position: RRF_THREADS == RRF_MAXC
    static_assert(RRF_THREADS == RRF_MAXC && RRF_THREADS == 2 * WAVE, "one thread per position");
    __shared__ int64_t L[3][RRF_MAXC];
    __shared__ int len[3];
    __shared__ int wave_new[3][2];
    __shared__ double s_s[RRF_SLOTS];
    __shared__ int64_t c_id[RRF_SLOTS];
    __shared__ int c_rank[RRF_SLOTS][3];
    __shared__ int order[RRF_SLOTS];  // sorted position -> sighting position

    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t* src[3] = {lex, sem, gra};
    const int width[3] = {lex ? n_lex : 0, sem ? n_sem : 0, gra ? n_gra : 0};
    if (t < 3) len[t] = width[t];
    __syncthreads();
    // lists end at the first negative id
    for (int ch = 0; ch < 3; ++ch) {
        const int64_t id = t < width[ch] ? src[ch][(int64_t)q * width[ch] + t] : -1;
        L[ch][t] = id;
        if (t < width[ch] && id < 0) atomicMin(&len[ch], t);
    }
    __syncthreads();

    // first sighting flags, channel by channel, and their running count = sighting position
    bool nw[3];
    int within[3];
    for (int ch = 0; ch < 3; ++ch) {
        nw[ch] = false;
        if (t < len[ch]) {
            const int64_t id = L[ch][t];
            nw[ch] = !seen_before(L[ch], t, id);
            for (int e = 0; e < ch && nw[ch]; ++e)
                if (last_rank(L[e], len[e], id)) nw[ch] = false;
        }
        const uint64_t m = __ballot(nw[ch]);
        within[ch] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_new[ch][wave] = __popcll(m);
    }
    __syncthreads();
    int base = 0, n = 0;
    for (int ch = 0; ch < 3; ++ch) {
        const int here = base + (wave ? wave_new[ch][0] : 0) + within[ch];
        base += wave_new[ch][0] + wave_new[ch][1];
        if (!nw[ch]) continue;
        const int64_t id = L[ch][t];
        const int lr = last_rank(L[0], len[0], id);
        const int sr = last_rank(L[1], len[1], id);
        const int gr = last_rank(L[2], len[2], id);
        double score = 0.0;
        if (MODE == RRF_RAG2) {
            if (lr) score = __dadd_rn(score, __ddiv_rn(w_lex, (double)(rrf_k + lr)));
            if (sr) score = __dadd_rn(score, __ddiv_rn(w_sem, (double)(rrf_k + sr)));
            if (gr) score = __dadd_rn(score, __ddiv_rn(w_gra, (double)(rrf_k + gr)));
        } else {
            const double w[3] = {w_lex, w_sem, w_gra};
            const int rk[3] = {lr, sr, gr};
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (rk[c]) {
                    const double v = __dmul_rn(w[c], __ddiv_rn(1.0, (double)(rrf_k + rk[c])));
                    const int reps = (MODE == RRF_TWO && c == 0) ? 1 : occurrences(L[c], len[c], id);
                    for (int r = 0; r < reps; ++r) score = __dadd_rn(score, v);
                }
        }
        c_id[here] = id;
        c_rank[here][0] = lr;
        c_rank[here][1] = sr;
        c_rank[here][2] = gr;
        s_s[here] = score;
    }
    n = base;
    __syncthreads();
    // (score desc, sighting position asc) is exactly Python's stable descending sort; positions
    // are distinct, so each candidate's rank is the number of candidates ahead of it
    for (int p = t; p < n; p += RRF_THREADS) {
        const double ms = s_s[p];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double o = s_s[j];
            rank += (o > ms || (o == ms && j < p)) ? 1 : 0;
        }
        order[rank] = p;
    }
    __syncthreads();
    for (int i = t; i < top_k; i += RRF_THREADS) {
        const bool ok = i < n;
        const int p = ok ? order[i] : 0;
        out_ids[(int64_t)q * top_k + i] = ok ? c_id[p] : -1;
        out_scores[(int64_t)q * top_k + i] = ok ? s_s[p] : -INFINITY;
        if (out_ranks)
            for (int c = 0; c < 3; ++c)
                out_ranks[((int64_t)q * top_k + i) * 3 + c] = ok ? c_rank[p][c] : 0;
    }
    if (t == 0) out_counts[q] = n < top_k ? n : top_k;
}

// Rerank ordering (retrieval.py:449-455): every candidate carries ONE rerank score -- the maximum
// over n_lists score lists, of which exactly one holds a finite value when the lists are the
// per-shard MaxSim outputs of a document-sharded index (the owning shard scored the candidate,
// the others wrote -inf) -- and the candidates are stably sorted by ``rerank_score or 0``
// descending: a candidate nobody scored counts as 0.0, ties keep the fused order.  One workgroup
// per query, rank sort in LDS (n <= RRF_SLOTS).
__global__ __launch_bounds__(RRF_THREADS) void rerank_order_kernel(
    const float* __restrict__ scores, int n_lists, int64_t list_stride,
    const int64_t* __restrict__ ids, const int32_t* __restrict__ counts, int n, int top_k,
    int64_t* __restrict__ out_ids, double* __restrict__ out_scores, int32_t* __restrict__ out_counts) {
    __shared__ double s_s[RRF_SLOTS];
    __shared__ int order[RRF_SLOTS];
    const int q = blockIdx.x, t = threadIdx.x;
    int cnt = counts ? counts[q] : n;
    cnt = cnt < n ? cnt : n;
    for (int p = t; p < cnt; p += RRF_THREADS) {
        float best = -INFINITY;
        for (int l = 0; l < n_lists; ++l) best = fmaxf(best, scores[(int64_t)l * list_stride + (int64_t)q * n + p]);
        s_s[p] = best > -INFINITY ? (double)best : 0.0;   // ``rerank_score or 0``
    }
    __syncthreads();
    for (int p = t; p < cnt; p += RRF_THREADS) {
        const double ms = s_s[p];
        int rank = 0;
        for (int j = 0; j < cnt; ++j) {
            const double o = s_s[j];
            rank += (o > ms || (o == ms && j < p)) ? 1 : 0;
        }
        order[rank] = p;
    }
    __syncthreads();
    for (int i = t; i < top_k; i += RRF_THREADS) {
        const bool ok = i < cnt;
        const int p = ok ? order[i] : 0;
        out_ids[(int64_t)q * top_k + i] = ok ? ids[(int64_t)q * n + p] : -1;
        out_scores[(int64_t)q * top_k + i] = ok ? s_s[p] : -INFINITY;
    }
    if (t == 0) out_counts[q] = cnt < top_k ? cnt : top_k;
}

// What RRFFusion.fuse does AFTER the sort (fusion.py:187-247) and normalize_scores (:294-318), for
// a batch of fused lists (one workgroup per query, <= RRF_SLOTS rows, best first):
//   safety      keep rows whose best channel score -- max(semantic or 0, lexical or 0, graph or 0),
//               the score at the id's last rank in each channel -- is >= safety_threshold (> 0);
//   denoise     with >= 3 rows left: cut = numpy.percentile(rrf scores, (1 - alpha) * 100), linear
//               interpolation with numpy's own lerp (a + (b - a) t, or b - (b - a)(1 - t) from
//               t = 0.5 on; virtual index (n - 1) * quantile), keep rrf >= cut;
//   top_k       the first top_k of what is left (0 = all);
//   normalize   (score - min) / (max - min) over what is left, 1.0 when they are all equal.
__global__ __launch_bounds__(RRF_THREADS) void fuse_post_kernel(
    const int64_t* __restrict__ ids, const double* __restrict__ scores,
    const int32_t* __restrict__ ranks, const int32_t* __restrict__ counts, int n,
    const double* __restrict__ lex_s, int n_lex, const double* __restrict__ sem_s, int n_sem,
    const double* __restrict__ gra_s, int n_gra, double safety_threshold, int denoise,
    double quantile, int normalize, int top_k, int64_t* __restrict__ out_ids,
    double* __restrict__ out_scores, int32_t* __restrict__ out_counts) {
    __shared__ double s_s[2][RRF_SLOTS];
    __shared__ int64_t s_id[2][RRF_SLOTS];
    __shared__ int wave_cnt[2];
    __shared__ double red[2][2];
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int cnt = counts ? counts[q] : n;
    cnt = cnt < n ? cnt : n;
    // stable compaction of the rows for which keep(i) holds, from buffer ``from`` to the other
    auto compact = [&](int from, int m, auto keep) -> int {
        int base = 0;
        for (int b0 = 0; b0 < m; b0 += RRF_THREADS) {
            const int i = b0 + t;
            const bool k = i < m && keep(i);
            const uint64_t bal = __ballot(k);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            const int at = base + (wave ? wave_cnt[0] : 0) + __popcll(bal & ((1ull << lane) - 1ull));
            if (k) {
                s_s[from ^ 1][at] = s_s[from][i];
                s_id[from ^ 1][at] = s_id[from][i];
            }
            base += wave_cnt[0] + wave_cnt[1];
            __syncthreads();
        }
        return base;
    };
    // load + safety threshold
    const double* chs[3] = {lex_s, sem_s, gra_s};
    const int chw[3] = {n_lex, n_sem, n_gra};
    for (int i = t; i < cnt; i += RRF_THREADS) {
        s_s[0][i] = scores[(int64_t)q * n + i];
        s_id[0][i] = ids[(int64_t)q * n + i];
    }
    __syncthreads();
    int m = cnt, cur = 0;
    if (safety_threshold > 0.0 && ranks) {
        m = compact(0, cnt, [&](int i) {
            double best = 0.0;   // ``score or 0.0`` of an absent channel
            for (int c = 0; c < 3; ++c) {
                const int r = ranks[((int64_t)q * n + i) * 3 + c];
                if (r > 0 && chs[c]) {
                    const double v = chs[c][(int64_t)q * chw[c] + r - 1];
                    best = v > best ? v : best;
                }
            }
            return best >= safety_threshold;
        });
        cur = 1;
    }
    if (denoise && m >= 3) {
        // rows are sorted best first: ascending index a <-> row m - 1 - a
        const double v = __dmul_rn((double)(m - 1), quantile);
        int prev = (int)floor(v);
        prev = prev < 0 ? 0 : prev > m - 1 ? m - 1 : prev;
        const int next = prev + 1 > m - 1 ? m - 1 : prev + 1;
        const double g = __dsub_rn(v, (double)prev);
        const double a = s_s[cur][m - 1 - prev], bb = s_s[cur][m - 1 - next];
        const double diff = __dsub_rn(bb, a);
        const double cut = g >= 0.5 ? __dsub_rn(bb, __dmul_rn(diff, __dsub_rn(1.0, g)))
                                    : __dadd_rn(a, __dmul_rn(diff, g));
        __syncthreads();
        const int from = cur;
        m = compact(from, m, [&](int i) { return s_s[from][i] >= cut; });
        cur ^= 1;
    }
    if (top_k > 0 && m > top_k) m = top_k;
    double lo = 0.0, hi = 0.0;
    if (normalize && m > 0) {
        double mn = INFINITY, mx = -INFINITY;
        for (int i = t; i < m; i += RRF_THREADS) {
            const double v = s_s[cur][i];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double a = __shfl_xor(mn, o, WAVE), b2 = __shfl_xor(mx, o, WAVE);
            mn = a < mn ? a : mn;
            mx = b2 > mx ? b2 : mx;
        }
        if (lane == 0) {
            red[0][wave] = mn;
            red[1][wave] = mx;
        }
        __syncthreads();
        lo = red[0][0] < red[0][1] ? red[0][0] : red[0][1];
        hi = red[1][0] > red[1][1] ? red[1][0] : red[1][1];
    }
    for (int i = t; i < n; i += RRF_THREADS) {
        const bool ok = i < m;
        double v = ok ? s_s[cur][i] : -INFINITY;
        if (ok && normalize) v = hi == lo ? 1.0 : __ddiv_rn(__dsub_rn(v, lo), __dsub_rn(hi, lo));
        out_scores[(int64_t)q * n + i] = v;
        out_ids[(int64_t)q * n + i] = ok ? s_id[cur][i] : -1;
    }
    if (t == 0) out_counts[q] = m;
}

}  // namespace thr

using namespace thr;

extern "C" int thr_rerank_order(const float* scores, int n_lists, int64_t list_stride,
                                const int64_t* ids, const int32_t* counts, int n_queries, int n,
                                int top_k, int64_t* out_ids, double* out_scores,
                                int32_t* out_counts, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!scores || !ids || !out_ids || !out_scores || !out_counts, THR_ERR_INVALID);
    THR_RETURN_IF(n_queries <= 0 || n_lists <= 0 || n <= 0 || n > RRF_SLOTS || top_k <= 0 || top_k > n,
                  THR_ERR_INVALID);
    if (list_stride == 0) list_stride = (int64_t)n_queries * n;
    THR_RETURN_IF(list_stride < (int64_t)n_queries * n, THR_ERR_INVALID);
    hipLaunchKernelGGL(rerank_order_kernel, dim3(n_queries), dim3(RRF_THREADS), 0, (hipStream_t)stream,
                       scores, n_lists, list_stride, ids, counts, n, top_k, out_ids, out_scores,
                       out_counts);
    return launch_status();
}

extern "C" int thr_rrf_fuse(const int64_t* lex_ids, int n_lex, const int64_t* sem_ids, int n_sem,
                            const int64_t* graph_ids, int n_graph, int n_queries, double w_lex,
                            double w_sem, double w_graph, int rrf_k, int top_k, int64_t* out_ids,
                            double* out_scores, int32_t* out_ranks, int32_t* out_counts,
                            thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!out_ids || !out_scores || !out_counts, THR_ERR_INVALID);
    THR_RETURN_IF(n_queries <= 0 || top_k <= 0 || top_k > RRF_SLOTS || rrf_k < 0, THR_ERR_INVALID);
    THR_RETURN_IF(n_lex < 0 || n_sem < 0 || n_graph < 0 || n_lex > RRF_MAXC || n_sem > RRF_MAXC ||
                      n_graph > RRF_MAXC,
                  THR_ERR_INVALID);
    THR_RETURN_IF((!lex_ids || !n_lex) && (!sem_ids || !n_sem) && (!graph_ids || !n_graph),
                  THR_ERR_INVALID);
    hipLaunchKernelGGL(rrf_fuse_kernel<RRF_RAG2>, dim3(n_queries), dim3(RRF_THREADS), 0, (hipStream_t)stream,
                       n_lex ? lex_ids : nullptr, n_lex, n_sem ? sem_ids : nullptr, n_sem,
                       n_graph ? graph_ids : nullptr, n_graph, w_lex, w_sem, w_graph, rrf_k, top_k,
                       out_ids, out_scores, out_ranks, out_counts);
    return launch_status();
}

extern "C" int thr_rrf_fuse_standalone(const int64_t* lex_ids, int n_lex, const int64_t* sem_ids,
                                       int n_sem, const int64_t* graph_ids, int n_graph,
                                       int n_queries, double w_lex, double w_sem, double w_graph,
                                       int two_channels, int top_k, int64_t* out_ids,
                                       double* out_scores, int32_t* out_ranks, int32_t* out_counts,
                                       thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!out_ids || !out_scores || !out_counts, THR_ERR_INVALID);
    THR_RETURN_IF(n_queries <= 0 || top_k <= 0 || top_k > RRF_SLOTS, THR_ERR_INVALID);
    THR_RETURN_IF(n_lex < 0 || n_sem < 0 || n_graph < 0 || n_lex > RRF_MAXC || n_sem > RRF_MAXC ||
                      n_graph > RRF_MAXC,
                  THR_ERR_INVALID);
    THR_RETURN_IF((!lex_ids || !n_lex) && (!sem_ids || !n_sem) && (!graph_ids || !n_graph),
                  THR_ERR_INVALID);
    THR_RETURN_IF(two_channels && graph_ids && n_graph, THR_ERR_INVALID);
    const int rrf_k = 60;   // RRF_K of the standalone package (fusion.py:27)
#define THR_RRF_LAUNCH(MODE)                                                                        \
    hipLaunchKernelGGL(rrf_fuse_kernel<MODE>, dim3(n_queries), dim3(RRF_THREADS), 0, (hipStream_t)stream, \
                       n_lex ? lex_ids : nullptr, n_lex, n_sem ? sem_ids : nullptr, n_sem,          \
                       n_graph ? graph_ids : nullptr, n_graph, w_lex, w_sem, w_graph, rrf_k, top_k, \
                       out_ids, out_scores, out_ranks, out_counts)
    if (two_channels) THR_RRF_LAUNCH(RRF_TWO);
    else THR_RRF_LAUNCH(RRF_STANDALONE);
#undef THR_RRF_LAUNCH
    return launch_status();
}

extern "C" int thr_fuse_post(const int64_t* ids, const double* scores, const int32_t* ranks,
                             const int32_t* counts, int n_queries, int n, const double* lex_scores,
                             int n_lex, const double* sem_scores, int n_sem,
                             const double* graph_scores, int n_graph, double safety_threshold,
                             int denoise, double quantile, int normalize, int top_k,
                             int64_t* out_ids, double* out_scores, int32_t* out_counts,
                             thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!ids || !scores || !out_ids || !out_scores || !out_counts, THR_ERR_INVALID);
    THR_RETURN_IF(n_queries <= 0 || n <= 0 || n > RRF_SLOTS || top_k < 0, THR_ERR_INVALID);
    THR_RETURN_IF(safety_threshold > 0.0 && !ranks, THR_ERR_INVALID);
    THR_RETURN_IF(denoise && !(quantile >= 0.0 && quantile <= 1.0), THR_ERR_INVALID);
    THR_RETURN_IF(n_lex < 0 || n_sem < 0 || n_graph < 0, THR_ERR_INVALID);
    hipLaunchKernelGGL(fuse_post_kernel, dim3(n_queries), dim3(RRF_THREADS), 0, (hipStream_t)stream, ids,
                       scores, ranks, counts, n, lex_scores, n_lex, sem_scores, n_sem, graph_scores,
                       n_graph, safety_threshold, denoise, quantile, normalize, top_k, out_ids,
                       out_scores, out_counts);
    return launch_status();
}
