// Lexical channel: Okapi BM25 top-k over a CSR inverted index (gfx950).
//
// Stands where the reference calls SQL rag2_lexical_search
// (database/migrations/20260114_rag2_schema.sql:341-374, from
// src/voice_agent/rag2/retrieval.py:282-290).  The reference ranks with
// PostgreSQL's ts_rank_cd; the north-star mandates BM25, whose exact form is
// the oracle's (oracle/thr_oracle.py bm25_scores): OR semantics, float64,
// contributions added in query-term order, every operation one IEEE rounding.
//
// One workgroup per query.  Posting lists are doc-sorted, so a doc's score is
// assembled by its OWNER posting -- the posting of the first query term that
// contains the doc -- which binary-searches the later terms' lists.  That
// gives the fixed summation order with no atomics and no hash table.  When a
// query's lists fit, they are staged in LDS first (posting-block staging) and
// the searches run there; longer lists are searched in HBM/L2.
// Algorithmic bytes per query: sum_t df_t * (4 doc + 4 tf + 4 doclen) + T * 16.
#include "thr_common.hpp"

namespace thr {

constexpr int BM_THREADS = 256;
constexpr int BM_CAP = 1024;       // BlockTopK buffer
constexpr int BM_STAGE = 4096;     // postings staged in LDS (32 KiB) when the query fits

struct TermRange {
    int64_t lo;
    int len;
    int lds_off;  // offset into the staged arrays
};

// lower_bound on a doc-sorted posting list; returns index or -1
template <typename Ptr>
__device__ __forceinline__ int find_doc(Ptr docs, int len, int32_t d) {
    int lo = 0, hi = len;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (docs[mid] < d) lo = mid + 1; else hi = mid;
    }
    return (lo < len && docs[lo] == d) ? lo : -1;
}

__device__ __forceinline__ double bm25_contrib(double idf, double tf, double dl, double avgdl,
                                               double k1, double b) {
    // nrm = k1*((1-b) + b*(dl/avgdl)); contrib = idf*((tf*(k1+1))/(tf+nrm))
    const double nrm = __dmul_rn(k1, __dadd_rn(__dsub_rn(1.0, b), __dmul_rn(b, __ddiv_rn(dl, avgdl))));
    return __dmul_rn(idf, __ddiv_rn(__dmul_rn(tf, __dadd_rn(k1, 1.0)), __dadd_rn(tf, nrm)));
}

__global__ __launch_bounds__(BM_THREADS) void bm25_topk_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ post_tf, const float* __restrict__ doclen,
    const double* __restrict__ idf, double avgdl, double k1, double b, int64_t id_base,
    const int32_t* __restrict__ query_terms, int max_terms, int k, double* __restrict__ out_s,
    int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt) {
    __shared__ TermRange tr[THR_BM25_MAX_TERMS];
    __shared__ double t_idf[THR_BM25_MAX_TERMS];
    __shared__ int t_prefix[THR_BM25_MAX_TERMS + 1];
    __shared__ int n_terms;
    __shared__ double b_s[BM_CAP];
    __shared__ int64_t b_id[BM_CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    __shared__ int32_t st_doc[BM_STAGE];
    __shared__ int32_t st_tf[BM_STAGE];

    const int q = blockIdx.x;
    if (threadIdx.x == 0) {
        int nt = 0, total = 0;
        for (int t = 0; t < max_terms; ++t) {
            int term = query_terms[(int64_t)q * max_terms + t];
            if (term < 0) continue;
            int64_t lo = rowptr[term], hi = rowptr[term + 1];
            tr[nt].lo = lo;
            tr[nt].len = (int)(hi - lo);
            tr[nt].lds_off = total;
            t_idf[nt] = idf[term];
            t_prefix[nt] = total;
            total += (int)(hi - lo);
            ++nt;
        }
        t_prefix[nt] = total;
        n_terms = nt;
    }
    BlockTopK<BM_CAP> tk;
    tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);  // includes a barrier
    const int nt = n_terms;
    const int total = t_prefix[nt];
    const bool staged = total <= BM_STAGE;
    if (staged) {
        for (int i = threadIdx.x; i < total; i += BM_THREADS) {
            int t = 0;
            while (i >= t_prefix[t + 1]) ++t;
            int64_t p = tr[t].lo + (i - t_prefix[t]);
            st_doc[i] = post_doc[p];
            st_tf[i] = post_tf[p];
        }
        __syncthreads();
    }

    for (int base = 0; base < total; base += BM_THREADS) {
        const int i = base + threadIdx.x;
        bool owner = false;
        double score = 0.0;
        int32_t d = 0;
        if (i < total) {
            int t = 0;
            while (i >= t_prefix[t + 1]) ++t;
            const int off = i - t_prefix[t];
            int32_t tf0;
            if (staged) {
                d = st_doc[i];
                tf0 = st_tf[i];
            } else {
                d = post_doc[tr[t].lo + off];
                tf0 = post_tf[tr[t].lo + off];
            }
            owner = true;
            for (int e = 0; e < t && owner; ++e) {
                int f = staged ? find_doc(st_doc + tr[e].lds_off, tr[e].len, d)
                               : find_doc(post_doc + tr[e].lo, tr[e].len, d);
                if (f >= 0) owner = false;
            }
            if (owner) {
                const double dl = (double)doclen[d];
                score = __dadd_rn(0.0, bm25_contrib(t_idf[t], (double)tf0, dl, avgdl, k1, b));
                for (int e = t + 1; e < nt; ++e) {
                    int f;
                    int32_t tf;
                    if (staged) {
                        f = find_doc(st_doc + tr[e].lds_off, tr[e].len, d);
                        tf = f >= 0 ? st_tf[tr[e].lds_off + f] : 0;
                    } else {
                        f = find_doc(post_doc + tr[e].lo, tr[e].len, d);
                        tf = f >= 0 ? post_tf[tr[e].lo + f] : 0;
                    }
                    if (f >= 0)
                        score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)tf, dl, avgdl, k1, b));
                }
            }
        }
        tk.push(owner, score, (int64_t)d);
    }
    const int n = tk.finish();
    for (int i = threadIdx.x; i < k; i += BM_THREADS) {
        out_s[(int64_t)q * k + i] = i < n ? b_s[i] : -INFINITY;
        out_id[(int64_t)q * k + i] = i < n ? b_id[i] + id_base : -1;
    }
    if (threadIdx.x == 0) out_cnt[q] = n;
}

}  // namespace thr

using namespace thr;

extern "C" int thr_bm25_topk(const int64_t* rowptr, const int32_t* post_doc, const int32_t* post_tf,
                             const float* doclen, const double* idf, double avgdl, double k1,
                             double b, int64_t n_docs, int64_t id_base, const int32_t* query_terms,
                             int n_queries, int max_terms, int k, double* out_scores,
                             int64_t* out_ids, int32_t* out_counts, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!rowptr || !post_doc || !post_tf || !doclen || !idf || !query_terms ||
                      !out_scores || !out_ids || !out_counts,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_docs <= 0 || n_queries <= 0 || k <= 0 || k > THR_TOPK_MAX || max_terms <= 0 ||
                      max_terms > THR_BM25_MAX_TERMS || !(avgdl > 0.0),
                  THR_ERR_INVALID);
    hipLaunchKernelGGL(bm25_topk_kernel, dim3(n_queries), dim3(BM_THREADS), 0, (hipStream_t)stream,
                       rowptr, post_doc, post_tf, doclen, idf, avgdl, k1, b, id_base, query_terms,
                       max_terms, k, out_scores, out_ids, out_counts);
    return launch_status();
}
