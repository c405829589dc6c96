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
// gives the fixed summation order with no atomics and no hash table.  The doc
// ids are staged in LDS (posting-block staging, in doc-range passes that always
// fit) and the searches run there -- a search in HBM/L2 is a chain of ~13
// dependent loads per term; term frequencies are read from memory only for the
// postings that need them.
// Algorithmic bytes per query: sum_t df_t * (4 doc + 4 tf + 4 doclen) + T * 16.
#include "thr_common.hpp"

namespace thr {

constexpr int BM_THREADS = 512;
constexpr int BM_CAP = 1024;       // BlockTopK buffer (>= k + BM_THREADS)
constexpr int BM_STAGE = 8192;     // doc ids staged in LDS per doc-range pass (32 KiB)

struct TermRange {
    int64_t lo;   // first posting of the term
    int len;      // postings of the term
    int cur;      // postings already consumed by earlier doc-range passes
    int sub;      // postings of the current pass: [cur, cur + sub)
    int lds_off;  // offset of the current pass's doc ids in the staged array
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
// number of postings with doc < d
__device__ __forceinline__ int count_below(const int32_t* docs, int len, int64_t d) {
    int lo = 0, hi = len;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((int64_t)docs[mid] < d) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ double bm25_contrib(double idf, double tf, double dl, double avgdl,
                                               double k1, double b) {
    // nrm = k1*((1-b) + b*(dl/avgdl)); contrib = idf*((tf*(k1+1))/(tf+nrm))
    const double nrm = __dmul_rn(k1, __dadd_rn(__dsub_rn(1.0, b), __dmul_rn(b, __ddiv_rn(dl, avgdl))));
    return __dmul_rn(idf, __ddiv_rn(__dmul_rn(tf, __dadd_rn(k1, 1.0)), __dadd_rn(tf, nrm)));
}

// Upper bounds for WAND-style pruning, computed once at index set-up (thr_bm25_bounds) with the
// scoring formula itself: term_ub[t] = max over the postings of term t of bm25_contrib, and
// block_ub[j] = the same maximum over postings [128 j, 128 j + 128) of the posting array (a block
// that straddles two short lists bounds both).  Kept as order-preserving uint64 keys while the
// atomicMax passes run, decoded in place by bm25_bounds_decode.
constexpr int BM_BLOCK = 128;

__global__ __launch_bounds__(256) void bm25_bounds_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ post_tf, const float* __restrict__ doclen,
    const double* __restrict__ idf, double avgdl, double k1, double b, int64_t n_vocab, int64_t nnz,
    unsigned long long* __restrict__ term_key, unsigned long long* __restrict__ block_key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    int64_t lo = 0, hi = n_vocab;  // last term with rowptr[t] <= i
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= i) lo = mid; else hi = mid;
    }
    const double c = bm25_contrib(idf[lo], (double)post_tf[i], (double)doclen[post_doc[i]], avgdl, k1, b);
    const unsigned long long key = dkey(c);
    atomicMax(&term_key[lo], key);
    atomicMax(&block_key[i / BM_BLOCK], key);
}
__global__ void bm25_bounds_decode(unsigned long long* __restrict__ keys, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        // an untouched slot (term without postings) bounds nothing: 0
        const double v = keys[i] ? dkey_inv(keys[i]) : 0.0;
        reinterpret_cast<double*>(keys)[i] = v;
    }
}

// The query's postings are consumed in DOC-RANGE passes: a pass takes, from every term's list,
// the postings with doc id in [d_lo, d_hi) -- a contiguous piece of each doc-sorted list -- so
// that all pieces together fit the LDS stage (d_hi is halved towards d_lo until they do).  A doc's
// postings all fall into the same pass, so the owner search never leaves LDS, whatever the
// length of the lists.
//
// WAND-style pruning (exact): passes visit the docs in ascending id order, so once k docs have
// been scored every later doc has to BEAT the current k-th best score theta (a tie loses on the
// id).  An owner posting first learns from the staged doc ids -- LDS only -- which query terms
// hold its doc, and sums their bounds in query-term order: term_ub (already in LDS), and if that
// still exceeds theta the tighter block_ub of the blocks its postings sit in.  Rounding is
// monotone, so fl(sum of bounds) >= fl(sum of contributions): a doc whose bound does not exceed
// theta is dropped before any of its term frequencies / doc length / collection id is fetched
// -- the random 4-byte gathers that dominate long lists.  theta is refreshed at the end of a pass
// when enough new docs have entered the buffer.
__global__ __launch_bounds__(BM_THREADS) void bm25_topk_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ post_tf, const float* __restrict__ doclen,
    const double* __restrict__ idf, const double* __restrict__ term_ub,
    const double* __restrict__ block_ub, double avgdl, double k1, double b, int64_t n_docs,
    int64_t n_vocab, int64_t id_base, const int32_t* __restrict__ query_terms, int max_terms, int k,
    int conjunctive, const int32_t* __restrict__ doc_coll, const int32_t* __restrict__ query_coll,
    double* __restrict__ out_s, int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt) {
    __shared__ TermRange tr[THR_BM25_MAX_TERMS];
    __shared__ double t_idf[THR_BM25_MAX_TERMS], t_ub[THR_BM25_MAX_TERMS];
    __shared__ int t_prefix[THR_BM25_MAX_TERMS + 1];
    __shared__ int t_slot[THR_BM25_MAX_TERMS];
    __shared__ int n_terms, remaining, last_compact;
    __shared__ int64_t range_lo, range_hi;
    __shared__ double b_s[BM_CAP];
    __shared__ int64_t b_id[BM_CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    __shared__ int32_t st_doc[BM_STAGE];

    const int q = blockIdx.x;
    const int qc = query_coll ? query_coll[q] : -1;   // -1: no collection filter
    // set-up, one thread per query term: valid terms keep their query order (a ballot prefix)
    {
        int term = -1;
        if (threadIdx.x < max_terms) {
            term = query_terms[(int64_t)q * max_terms + threadIdx.x];
            if (term >= n_vocab) term = -1;   // unknown term: no postings
        }
        const uint64_t m = __ballot(term >= 0);   // (max_terms <= 32: all in wave 0)
        if (threadIdx.x < max_terms && term >= 0) {
            const int slot = __popcll(m & ((1ull << threadIdx.x) - 1ull));
            const int64_t lo = rowptr[term], hi = rowptr[term + 1];
            tr[slot].lo = lo;
            tr[slot].len = (int)(hi - lo);
            tr[slot].cur = 0;
            t_idf[slot] = idf[term];
            t_ub[slot] = term_ub ? term_ub[term] : INFINITY;
        }
        if (threadIdx.x == 0) {
            n_terms = __popcll(m);
            range_lo = 0;
            last_compact = 0;
        }
    }
    BlockTopK<BM_CAP> tk;
    tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);  // includes a barrier
    const int nt = n_terms;
    if (threadIdx.x == 0) {
        int total = 0;
        for (int t = 0; t < nt; ++t) total += tr[t].len;
        remaining = total;
    }
    __syncthreads();

    while (remaining > 0) {
        // ---- choose [range_lo, range_hi): everything left if it fits, else a share of the doc
        //      space proportional to the stage size, halved until the pieces fit ----
        if (threadIdx.x == 0) {
            int64_t span = n_docs - range_lo;
            if (remaining > BM_STAGE) {
                span = (int64_t)((double)span * (0.75 * BM_STAGE) / (double)remaining);
                if (span < 1) span = 1;
            }
            range_hi = range_lo + span;
        }
        __syncthreads();
        for (;;) {
            if (threadIdx.x < nt) {
                TermRange& r = tr[threadIdx.x];
                r.sub = range_hi >= n_docs
                            ? r.len - r.cur
                            : count_below(post_doc + r.lo + r.cur, r.len - r.cur, range_hi);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int total = 0;
                for (int t = 0; t < nt; ++t) {
                    tr[t].lds_off = total;
                    t_prefix[t] = total;
                    total += tr[t].sub;
                }
                t_prefix[nt] = total;
                // a single doc holds at most nt postings, so a one-doc range always fits
                if (total > BM_STAGE && range_hi - range_lo > 1)
                    range_hi = range_lo + (range_hi - range_lo) / 2;
                else
                    range_lo = -1 - range_lo;  // accepted (decoded below)
            }
            __syncthreads();
            if (range_lo < 0) break;
        }
        const int total = t_prefix[nt];
        for (int i = threadIdx.x; i < total; i += BM_THREADS) {
            int t = 0;
            while (i >= t_prefix[t + 1]) ++t;
            st_doc[i] = post_doc[tr[t].lo + tr[t].cur + (i - t_prefix[t])];
        }
        __syncthreads();
        const bool have_theta = b_cnt >= k && th_s > -INFINITY;
        const double theta = th_s;

        for (int base = 0; base < total; base += BM_THREADS) {
            const int i = base + threadIdx.x;
            bool owner = false;
            double score = 0.0;
            int32_t d = 0;
            if (i < total) {
                int t = 0;
                while (i >= t_prefix[t + 1]) ++t;
                const int off = i - t_prefix[t];
                d = st_doc[i];
                owner = true;
                for (int e = 0; e < t && owner; ++e)
                    if (find_doc(st_doc + tr[e].lds_off, tr[e].sub, d) >= 0) owner = false;
                if (owner) {
                    // which later terms hold the doc, and where (position inside the staged piece;
                    // kept in registers for the first 8 query terms -- static indexing only --
                    // and searched again for the rest)
                    int pos[8];
                    uint32_t present = 1u << t;
                    double ub = 0.0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        pos[e] = -1;
                        if (e == t) pos[e] = off;
                        if (e > t && e < nt) pos[e] = find_doc(st_doc + tr[e].lds_off, tr[e].sub, d);
                        if (pos[e] >= 0) {
                            present |= 1u << e;
                            ub = __dadd_rn(ub, t_ub[e]);
                        }
                    }
                    for (int e = 8; e < nt; ++e) {
                        const int f = e == t ? off : (e > t ? find_doc(st_doc + tr[e].lds_off, tr[e].sub, d) : -1);
                        if (f >= 0) {
                            present |= 1u << e;
                            ub = __dadd_rn(ub, t_ub[e]);
                        }
                    }
                    auto where = [&](int e) -> int64_t {   // posting index of the doc in term e (e >= 8)
                        const int f = e == t ? off : find_doc(st_doc + tr[e].lds_off, tr[e].sub, d);
                        return tr[e].lo + tr[e].cur + f;
                    };
                    if (conjunctive && __popc(present) < nt) owner = false;
                    if (owner && have_theta && !(ub > theta)) owner = false;
                    if (owner && have_theta && block_ub) {
                        double ub2 = 0.0;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (pos[e] >= 0)
                                ub2 = __dadd_rn(ub2, block_ub[(tr[e].lo + tr[e].cur + pos[e]) / BM_BLOCK]);
                        for (int e = 8; e < nt; ++e)
                            if (present & (1u << e)) ub2 = __dadd_rn(ub2, block_ub[where(e) / BM_BLOCK]);
                        if (!(ub2 > theta)) owner = false;
                    }
                    if (owner && qc != -1 && doc_coll[d] != qc) owner = false;
                    if (owner) {
                        const double dl = (double)doclen[d];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (pos[e] >= 0) {
                                const int32_t tf = post_tf[tr[e].lo + tr[e].cur + pos[e]];
                                score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)tf, dl, avgdl, k1, b));
                            }
                        for (int e = 8; e < nt; ++e)
                            if (present & (1u << e))
                                score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)post_tf[where(e)], dl,
                                                                      avgdl, k1, b));
                    }
                }
            }
            tk.push(owner, score, (int64_t)d);
        }
        __syncthreads();
        // a fresh theta pays for the sort once enough docs have entered since the last one
        if (b_cnt >= k && b_cnt - last_compact >= 64) {
            tk.compact();
            if (threadIdx.x == 0) last_compact = b_cnt;
        }
        if (threadIdx.x == 0) {
            for (int t = 0; t < nt; ++t) tr[t].cur += tr[t].sub;
            remaining -= total;
            range_lo = range_hi;  // (range_lo held the "accepted" marker)
        }
        __syncthreads();
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

extern "C" size_t thr_bm25_block_count(int64_t nnz) { return nnz > 0 ? (size_t)((nnz + BM_BLOCK - 1) / BM_BLOCK) : 0; }

extern "C" int thr_bm25_bounds(const int64_t* rowptr, const int32_t* post_doc, const int32_t* post_tf,
                               const float* doclen, const double* idf, double avgdl, double k1,
                               double b, int64_t n_vocab, int64_t nnz, double* term_ub,
                               double* block_ub, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!rowptr || !post_doc || !post_tf || !doclen || !idf || !term_ub || !block_ub,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_vocab <= 0 || nnz <= 0 || !(avgdl > 0.0), THR_ERR_INVALID);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = (int64_t)thr_bm25_block_count(nnz);
    hipError_t e = hipMemsetAsync(term_ub, 0, sizeof(double) * n_vocab, st);
    if (e == hipSuccess) e = hipMemsetAsync(block_ub, 0, sizeof(double) * nb, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bm25_bounds_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, rowptr,
                       post_doc, post_tf, doclen, idf, avgdl, k1, b, n_vocab, nnz,
                       (unsigned long long*)term_ub, (unsigned long long*)block_ub);
    hipLaunchKernelGGL(bm25_bounds_decode, dim3((unsigned)((n_vocab + 255) / 256)), dim3(256), 0, st,
                       (unsigned long long*)term_ub, n_vocab);
    hipLaunchKernelGGL(bm25_bounds_decode, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st,
                       (unsigned long long*)block_ub, nb);
    return launch_status();
}

extern "C" int thr_bm25_topk(const int64_t* rowptr, const int32_t* post_doc, const int32_t* post_tf,
                             const float* doclen, const double* idf, const double* term_ub,
                             const double* block_ub, double avgdl, double k1, double b,
                             int64_t n_docs, int64_t n_vocab, int64_t id_base,
                             const int32_t* query_terms, int n_queries, int max_terms, int k,
                             int conjunctive, const int32_t* doc_coll, const int32_t* query_coll,
                             double* out_scores, int64_t* out_ids, int32_t* out_counts,
                             thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!rowptr || !post_doc || !post_tf || !doclen || !idf || !query_terms ||
                      !out_scores || !out_ids || !out_counts,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_docs <= 0 || n_vocab <= 0 || n_queries <= 0 || k <= 0 || k > THR_TOPK_MAX ||
                      max_terms <= 0 || max_terms > THR_BM25_MAX_TERMS || !(avgdl > 0.0),
                  THR_ERR_INVALID);
    THR_RETURN_IF((query_coll != nullptr) != (doc_coll != nullptr), THR_ERR_INVALID);
    hipLaunchKernelGGL(bm25_topk_kernel, dim3(n_queries), dim3(BM_THREADS), 0, (hipStream_t)stream,
                       rowptr, post_doc, post_tf, doclen, idf, term_ub, term_ub ? block_ub : nullptr,
                       avgdl, k1, b, n_docs, n_vocab, id_base, query_terms, max_terms, k, conjunctive,
                       doc_coll, query_coll, out_scores, out_ids, out_counts);
    return launch_status();
}
