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
// contains the doc.  That gives the fixed summation order with no atomics and
// no hash table.  The doc ids are staged in LDS (posting-block staging, in
// doc-range passes that always fit).  How a pass finds the owners depends on
// its lists: dense lists (narrow doc range) OR a term bit into a doc-slot
// mask; sparse lists set a Bloom bit per (list, doc) and a posting whose doc
// shows in no other list's bits is scored at once (one contribution, no
// search), the few others are searched from a dense work list; in between,
// with a threshold to prune against, owners and the sum of their terms' score
// bounds come from binary searches in LDS and only the survivors touch memory.
// Term frequencies and doc lengths are read only for the postings that need them.
// Algorithmic bytes per query: sum_t df_t * (4 doc + 4 tf + 4 doclen) + T * 16.
#include <cstdlib>
#include "thr_common.hpp"

namespace thr {

// Block shapes (template arguments of bm25_topk_kernel):
//   BM_THREADS  threads per query
//   BM_STAGE    doc ids staged in LDS per doc-range pass
//   BM_WINDOW   doc slots of the mask path (BM_STAGE <= 3 * BM_WINDOW: the survivor list shares it)
//   BM_CAP      BlockTopK buffer (>= k + BM_THREADS)

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

// The query's postings are consumed in DOC-RANGE passes.  A pass stages, from every term's list,
// the next quota_t postings (quotas proportional to what is left of each list, together one LDS
// stage), then takes d_hi = the smallest "last staged doc + 1" among the lists that have more
// postings behind their quota: every posting with doc < d_hi of EVERY list is then on chip, so a
// doc's postings all fall into the same pass and the owner search never leaves LDS, whatever
// the length of the lists.  (No search in global memory picks the range: round 1 did that with
// a chain of ~20 dependent loads per term and pass, the dominant cost on long lists.)  The
// postings with doc >= d_hi stay for the next pass and are staged again.
//
// WAND-style pruning (exact): passes visit the docs in ascending id order, so once k docs have
// been scored every later doc has to BEAT the current k-th best score theta (a tie loses on the
// id).  Phase 1 of a pass is LDS-only: an owner posting (the posting of the first query term
// that holds its doc) learns from the staged doc ids which query terms hold the doc and sums
// their term_ub in query-term order; rounding is monotone, so fl(sum of bounds) >= fl(sum of
// contributions), and a doc whose bound does not exceed theta is dropped there.  The survivors
// are compacted into an LDS list; phase 2 walks that list 512 at a time: tighter block_ub check,
// collection filter, term-frequency / doc-length gathers, float64 score, top-k push.  theta is
// refreshed at the end of a pass when enough new docs have entered the buffer.
template <int BM_THREADS, int BM_STAGE, int BM_WINDOW, int BM_CAP>
__global__ __launch_bounds__(BM_THREADS, 4) void bm25_topk_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ post_tf, const float* __restrict__ doclen,
    const double* __restrict__ idf, const double* __restrict__ term_ub,
    const double* __restrict__ block_ub, double avgdl, double k1, double b, int64_t n_docs,
    int64_t n_vocab, int64_t id_base, const int32_t* __restrict__ query_terms, int max_terms, int k,
    int conjunctive, const int32_t* __restrict__ doc_coll, const int32_t* __restrict__ query_coll,
    double* __restrict__ out_s, int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt) {
    __shared__ TermRange tr[THR_BM25_MAX_TERMS];   // .sub = postings of this pass, .lds_off = where staged
    __shared__ double t_idf[THR_BM25_MAX_TERMS], t_ub[THR_BM25_MAX_TERMS];
    __shared__ int t_staged[THR_BM25_MAX_TERMS];   // postings of the term staged in this pass
    __shared__ int t_prefix[THR_BM25_MAX_TERMS + 1];
    __shared__ int n_terms, remaining, last_compact, n_surv;
    __shared__ int64_t d_hi, d_lo;
    __shared__ double b_s[BM_CAP];
    __shared__ int64_t b_id[BM_CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    __shared__ int32_t st_doc[BM_STAGE];
    // mask path: mask[BM_WINDOW] (which query terms hold doc d_lo + slot) + up to BM_WINDOW surviving
    // slots behind it; search path: up to BM_STAGE surviving staged indices.  One 24 KiB buffer.
    __shared__ uint32_t scratch[BM_WINDOW + BM_WINDOW / 2];
    static_assert(sizeof(uint32_t) * (BM_WINDOW + BM_WINDOW / 2) >= sizeof(uint16_t) * BM_STAGE, "survivor list must fit");
    static_assert(BM_CAP >= THR_TOPK_MAX + BM_THREADS && BM_STAGE <= 65536, "top-k buffer / 16-bit staged indices");
    uint32_t* mask = scratch;

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
            last_compact = 0;
        }
    }
    BlockTopK<BM_CAP, BM_THREADS> tk;
    tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);  // includes a barrier
    const int nt = n_terms;
    if (threadIdx.x == 0) {
        int total = 0;
        for (int t = 0; t < nt; ++t) total += tr[t].len;
        remaining = total;
    }
    __syncthreads();

    while (remaining > 0) {
        // ---- quotas: the stage is shared out in proportion to what is left of each list ----
        if (threadIdx.x == 0) {
            int off = 0;
            const int spare = BM_STAGE - 32 * nt;   // every list gets at least 32 slots
            for (int t = 0; t < nt; ++t) {
                const int rem = tr[t].len - tr[t].cur;
                int quota = 32 + (int)((int64_t)spare * rem / remaining);
                quota = quota < rem ? quota : rem;
                tr[t].lds_off = off;
                t_staged[t] = quota;
                off += quota;
            }
            d_hi = INT64_MAX;
            d_lo = INT64_MAX;
        }
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int32_t* src = post_doc + tr[t].lo + tr[t].cur;
            int32_t* dst = st_doc + tr[t].lds_off;
            for (int i = threadIdx.x; i < t_staged[t]; i += BM_THREADS) dst[i] = src[i];
        }
        __syncthreads();
        if (threadIdx.x < nt) {
            const int t = threadIdx.x;
            if (t_staged[t] > 0 && tr[t].cur + t_staged[t] < tr[t].len)   // more postings behind the quota
                atomicMin((unsigned long long*)&d_hi,
                          (unsigned long long)((int64_t)st_doc[tr[t].lds_off + t_staged[t] - 1] + 1));
            if (t_staged[t] > 0)
                atomicMin((unsigned long long*)&d_lo, (unsigned long long)st_doc[tr[t].lds_off]);
        }
        __syncthreads();
        if (threadIdx.x < nt) {
            TermRange& r = tr[threadIdx.x];
            r.sub = d_hi == INT64_MAX ? t_staged[threadIdx.x]
                                      : count_below(st_doc + r.lds_off, t_staged[threadIdx.x], d_hi);
        }
        if (threadIdx.x == 0) n_surv = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
            for (int t = 0; t < nt; ++t) {
                t_prefix[t] = total;
                total += tr[t].sub;
            }
            t_prefix[nt] = total;
        }
        __syncthreads();
        const int total = t_prefix[nt];
        const bool have_theta = b_cnt >= k && th_s > -INFINITY;
        const double theta = th_s;

        // ---- phase 1 (LDS only): owners, which terms hold the doc, bound against theta ----
        // Dense lists give narrow doc ranges: when the pass's docs fit BM_WINDOW slots, every
        // staged posting ORs its term's bit into the doc's slot -- O(1) per posting instead of a
        // binary search per (posting, other term) -- and the slots are then the candidate docs
        // (owner = lowest bit).  Wide ranges (sparse lists, few postings) keep the searches.
        int64_t last = d_hi;   // one past the last doc of the pass
        if (d_hi == INT64_MAX) {
            last = 0;
            for (int t = 0; t < nt; ++t)
                if (tr[t].sub > 0) {
                    const int64_t e = (int64_t)st_doc[tr[t].lds_off + tr[t].sub - 1] + 1;
                    last = e > last ? e : last;
                }
        }
        const int64_t first = d_lo;
        const bool masked = total > 0 && last - first <= BM_WINDOW;
        uint16_t* surv = masked ? reinterpret_cast<uint16_t*>(scratch + BM_WINDOW) : reinterpret_cast<uint16_t*>(scratch);
        if (masked) {
            const int w = (int)(last - first);
            for (int i = threadIdx.x; i < w; i += BM_THREADS) mask[i] = 0u;
            __syncthreads();
            for (int i = threadIdx.x; i < total; i += BM_THREADS) {
                int t = 0;
                while (i >= t_prefix[t + 1]) ++t;
                const int32_t d = st_doc[tr[t].lds_off + (i - t_prefix[t])];
                atomicOr(&mask[d - first], 1u << t);
            }
            __syncthreads();
            for (int slot = threadIdx.x; slot < w; slot += BM_THREADS) {
                uint32_t m = mask[slot];
                if (!m) continue;
                if (conjunctive && __popc(m) < nt) continue;
                if (have_theta) {
                    double ub = 0.0;
                    for (uint32_t r = m; r; r &= r - 1) ub = __dadd_rn(ub, t_ub[__ffs((int)r) - 1]);
                    if (!(ub > theta)) continue;
                }
                surv[atomicAdd(&n_surv, 1)] = (uint16_t)slot;
            }
        } else if (have_theta && (last - first) < 32 * (int64_t)total) {
            // moderately dense lists and a threshold to prune with: LDS-only owner / bound search,
            // survivors to phase 2 (a doc is dropped on the sum of its terms' bounds before any
            // gather; the sweep below would find most docs shared and score them all)
            for (int i = threadIdx.x; i < total; i += BM_THREADS) {
                int t = 0;
                while (i >= t_prefix[t + 1]) ++t;
                const int off = i - t_prefix[t];
                const int32_t d = st_doc[tr[t].lds_off + off];
                bool owner = true;
                for (int e = 0; e < t && owner; ++e)
                    if (find_doc(st_doc + tr[e].lds_off, tr[e].sub, d) >= 0) owner = false;
                if (!owner) continue;
                int present = 1;
                double ub = __dadd_rn(0.0, t_ub[t]);
                for (int e = t + 1; e < nt; ++e)
                    if (find_doc(st_doc + tr[e].lds_off, tr[e].sub, d) >= 0) {
                        ++present;
                        ub = __dadd_rn(ub, t_ub[e]);
                    }
                if (conjunctive && present < nt) continue;
                if (!(ub > theta)) continue;
                surv[atomicAdd(&n_surv, 1)] = (uint16_t)(tr[t].lds_off + off);
            }
        } else {
            // sparse lists (or no threshold yet): every owner is scored in the same sweep that finds it.
            // Sparse lists share few docs, so nearly every "is this doc in list e" question is
            // answered NO: a Bloom bit per (list, doc hash) in the idle scratch buffer answers
            // those with one LDS read instead of a binary search (a chain of ~11); a set bit is
            // confirmed by the search, so the result is exact.  The doc-length and own-tf gathers
            // of the NEXT sweep step are requested before the current one is worked on.
            // bits per list: the largest power of two (<= 32768) that fits the buffer nt times next
            // to a work list that could take every posting of the pass (16 bits each)
            constexpr int SCR_WORDS = BM_WINDOW + BM_WINDOW / 2;
            int bwords = 1024;
            while (bwords >= 128 && nt * bwords + (total + 1) / 2 > SCR_WORDS) bwords >>= 1;
            const bool bloom = bwords >= 128;
            const int bl2 = 31 - __clz(bwords * 32);
            if (bloom) {
                for (int i = threadIdx.x; i < nt * bwords; i += BM_THREADS) scratch[i] = 0u;
                __syncthreads();
                for (int i = threadIdx.x; i < total; i += BM_THREADS) {
                    int t = 0;
                    while (i >= t_prefix[t + 1]) ++t;
                    const uint32_t h = ((uint32_t)st_doc[tr[t].lds_off + (i - t_prefix[t])] * 2654435761u) >> (32 - bl2);
                    atomicOr(&scratch[t * bwords + (h >> 5)], 1u << (h & 31));
                }
                __syncthreads();
            }
            auto lookup = [&](int e, int32_t d) -> int {   // index of d in list e's staged ids, or -1
                if (bloom) {
                    const uint32_t h = ((uint32_t)d * 2654435761u) >> (32 - bl2);
                    if (!((scratch[e * bwords + (h >> 5)] >> (h & 31)) & 1u)) return -1;
                }
                return find_doc(st_doc + tr[e].lds_off, tr[e].sub, d);
            };
            // the searching version of "score posting (t, off) if it owns doc d"
            auto score_full = [&](int t, int off, int32_t d, float dl_own, int32_t tf_own, double& score) -> bool {
                for (int e = 0; e < t; ++e)
                    if (lookup(e, d) >= 0) return false;
                int present = 0;
                int64_t where[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    where[e] = -1;
                    if (e >= t && e < nt) {
                        const int f = e == t ? off : lookup(e, d);
                        if (f >= 0) {
                            where[e] = tr[e].lo + tr[e].cur + f;
                            ++present;
                        }
                    }
                }
                auto far = [&](int e) -> int64_t {   // terms beyond the 8th: searched when needed
                    const int f = e < t ? -1 : (e == t ? off : lookup(e, d));
                    return f >= 0 ? tr[e].lo + tr[e].cur + f : -1;
                };
                for (int e = 8; e < nt; ++e) present += far(e) >= 0 ? 1 : 0;
                if (conjunctive && present < nt) return false;
                if (qc != -1 && doc_coll[d] != qc) return false;
                const double dl = (double)dl_own;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (where[e] >= 0)
                        score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)(e == t ? tf_own : post_tf[where[e]]), dl, avgdl, k1, b));
                for (int e = 8; e < nt; ++e) {
                    const int64_t w = far(e);
                    if (w >= 0)
                        score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)(e == t ? tf_own : post_tf[w]), dl, avgdl, k1, b));
                }
                return true;
            };
            int n_t = 0, n_off = 0;
            int32_t n_d = 0, n_tf = 0;
            float n_dl = 0.f;
            auto fetch = [&](int i) {
                if (i >= 0 && i < total) {
                    n_t = 0;
                    while (i >= t_prefix[n_t + 1]) ++n_t;
                    n_off = i - t_prefix[n_t];
                    n_d = st_doc[tr[n_t].lds_off + n_off];
                    n_dl = doclen[n_d];
                    n_tf = post_tf[tr[n_t].lo + tr[n_t].cur + n_off];
                }
            };
            // With the filter, sweep 1 never searches: a posting whose doc shows in no other list's
            // bits is the doc's only posting (owner, one contribution) and is scored at once; the
            // few with a set bit -- which a wave would otherwise wait for, lane by lane -- go to a
            // work list (behind the bits in the same buffer) that sweep 2 walks densely.
            static_assert(2 * (BM_WINDOW + BM_WINDOW / 2) >= BM_STAGE, "work list of a pass without the filter");
            uint16_t* work = reinterpret_cast<uint16_t*>(bloom ? scratch + nt * bwords : scratch);
            if (threadIdx.x == 0) n_surv = 0;   // (work list length; phase 2 below sees 0 again)
            __syncthreads();
            if (bloom) {
                // list by list: everything that depends on the term is uniform (scalar registers)
                for (int t = 0; t < nt; ++t) {
                    const int sub = __builtin_amdgcn_readfirstlane(tr[t].sub);
                    const int off0 = __builtin_amdgcn_readfirstlane(tr[t].lds_off);
                    const int pre = __builtin_amdgcn_readfirstlane(t_prefix[t]);
                    const int32_t* tf_t = post_tf + tr[t].lo + tr[t].cur;
                    const double idf_t = t_idf[t], ub_t = t_ub[t];
                    const bool single_ok = !(conjunctive && nt > 1);
                    for (int base = 0; base < sub; base += BM_THREADS) {
                        const int i = base + (int)threadIdx.x;
                        bool owner = false;
                        double score = 0.0;
                        int32_t d = 0;
                        if (i < sub) {
                            d = st_doc[off0 + i];
                            const uint32_t h = ((uint32_t)d * 2654435761u) >> (32 - bl2);
                            const uint32_t w = h >> 5, bit = 1u << (h & 31);
                            bool alone = true;
                            for (int e = 0; e < nt; ++e)
                                if (e != t && (scratch[e * bwords + w] & bit)) alone = false;
                            if (!alone) {
                                work[atomicAdd(&n_surv, 1)] = (uint16_t)(pre + i);
                            } else if (single_ok && !(ub_t < th_s) && !(qc != -1 && doc_coll[d] != qc)) {
                                // (ub_t < threshold: no posting of this list can enter on its own)
                                owner = true;
                                score = __dadd_rn(score, bm25_contrib(idf_t, (double)tf_t[i], (double)doclen[d], avgdl, k1, b));
                            }
                        }
                        tk.push(owner, score, (int64_t)d);
                    }
                }
            } else {   // no room for the bits (many terms): every posting takes the searching sweep
                for (int i = threadIdx.x; i < total; i += BM_THREADS) work[i] = (uint16_t)i;
                if (threadIdx.x == 0) n_surv = total;
            }
            __syncthreads();
            const int n_work = n_surv;
            fetch(threadIdx.x < n_work ? (int)work[threadIdx.x] : -1);
            for (int base = 0; base < n_work; base += BM_THREADS) {
                const int j = base + threadIdx.x;
                const int t = n_t, off = n_off;
                const int32_t d = n_d, tf_own = n_tf;
                const float dl_own = n_dl;
                fetch(j + BM_THREADS < n_work ? (int)work[j + BM_THREADS] : -1);
                bool owner = false;
                double score = 0.0;
                if (j < n_work) owner = score_full(t, off, d, dl_own, tf_own, score);
                tk.push(owner, score, (int64_t)d);
            }
            __syncthreads();
            if (threadIdx.x == 0) n_surv = 0;
        }
        __syncthreads();

        // ---- phase 2: the survivors, densely ----
        const int ns = n_surv;
        for (int base = 0; base < ns; base += BM_THREADS) {
            const int j = base + threadIdx.x;
            bool keep = j < ns;
            double score = 0.0;
            int32_t d = 0;
            if (keep) {
                int t = 0, at;
                if (masked) {   // survivor = doc slot: owner = lowest term bit, position searched
                    d = (int32_t)(first + surv[j]);
                    t = __ffs((int)mask[surv[j]]) - 1;
                    at = tr[t].lds_off + find_doc(st_doc + tr[t].lds_off, tr[t].sub, d);
                } else {        // survivor = staged index of the owner posting
                    at = surv[j];
                    while (t + 1 < nt && at >= tr[t + 1].lds_off) ++t;   // lds_off ascends with t
                    d = st_doc[at];
                }
                // posting index of the doc in every term that holds it (first 8 terms in
                // registers -- static indexing only --, the rest searched again when needed)
                int64_t where[8];
                uint32_t present = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    where[e] = -1;
                    if (e >= t && e < nt) {
                        const int f = e == t ? at - tr[e].lds_off : find_doc(st_doc + tr[e].lds_off, tr[e].sub, d);
                        if (f >= 0) {
                            where[e] = tr[e].lo + tr[e].cur + f;
                            present |= 1u << e;
                        }
                    }
                }
                auto where_far = [&](int e) -> int64_t {   // e >= 8
                    const int f = e == t ? at - tr[e].lds_off : find_doc(st_doc + tr[e].lds_off, tr[e].sub, d);
                    return f >= 0 ? tr[e].lo + tr[e].cur + f : -1;
                };
                if (have_theta && block_ub) {
                    double ub2 = 0.0;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (where[e] >= 0) ub2 = __dadd_rn(ub2, block_ub[where[e] / BM_BLOCK]);
                    for (int e = 8 > t ? 8 : t; e < nt; ++e) {
                        const int64_t w = where_far(e);
                        if (w >= 0) ub2 = __dadd_rn(ub2, block_ub[w / BM_BLOCK]);
                    }
                    if (!(ub2 > theta)) keep = false;
                }
                if (keep && qc != -1 && doc_coll[d] != qc) keep = false;
                if (keep) {
                    const double dl = (double)doclen[d];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (where[e] >= 0)
                            score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)post_tf[where[e]], dl, avgdl, k1, b));
                    for (int e = 8 > t ? 8 : t; e < nt; ++e) {
                        const int64_t w = where_far(e);
                        if (w >= 0)
                            score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)post_tf[w], dl, avgdl, k1, b));
                    }
                }
            }
            tk.push(keep, score, (int64_t)d);
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
    // Block shape: 512 threads / 8192 staged ids per pass / 75 KiB of LDS, two queries per CU --
    // a four-term query of the bench (6.7 K postings) is one pass.  THR_BM25_SHAPE=small selects
    // 256 threads / 4096 ids / 39 KiB, four queries per CU: the fixed cost of a query (set-up,
    // staging, the final sort) overlaps four ways, which wins when every list is short (2048
    // queries over lists of <= 200 postings: 0.075 ms against 0.124 ms) and loses otherwise
    // (bench mix 0.46 ms against 0.42 ms; 256 stop-word queries 23 ms against 12 ms).
    static int small = -1;
    if (small < 0) {
        const char* e = getenv("THR_BM25_SHAPE");
        small = (e && e[0] == 's') ? 1 : 0;
    }
    const bool big = !small;
#define THR_BM25_LAUNCH(T, S, W, C)                                                                \
    hipLaunchKernelGGL((bm25_topk_kernel<T, S, W, C>), dim3(n_queries), dim3(T), 0, (hipStream_t)stream, \
                       rowptr, post_doc, post_tf, doclen, idf, term_ub, term_ub ? block_ub : nullptr, \
                       avgdl, k1, b, n_docs, n_vocab, id_base, query_terms, max_terms, k, conjunctive, \
                       doc_coll, query_coll, out_scores, out_ids, out_counts)
    if (big) THR_BM25_LAUNCH(512, 8192, 4096, 1024);
    else THR_BM25_LAUNCH(256, 4096, 2048, 512);
#undef THR_BM25_LAUNCH
    return launch_status();
}
