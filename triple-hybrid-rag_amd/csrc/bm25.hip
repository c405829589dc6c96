// Lexical channel: Okapi BM25 top-k over a CSR inverted index (gfx950).
//
// Stands where the reference calls SQL rag2_lexical_search
// (database/migrations/20260114_rag2_schema.sql:341-374, from
// src/voice_agent/rag2/retrieval.py:282-290).  The reference ranks with
// PostgreSQL's ts_rank_cd; the north-star mandates BM25, whose exact form is
// the oracle's (oracle/thr_oracle.py bm25_scores): OR semantics, float64,
// contributions added in query-term order, every operation one IEEE rounding.
//
// Work decomposition: a query whose lists hold more postings than one slice (24576 when the
// batch fills the chip, down to 8192 when it does not) is cut
// into DOC-RANGE slices of ~equal posting counts (the slice edges are docs of its longest
// list), one work item per slice; short queries are one item.  A persistent grid of
// workgroups pulls items from a device-side counter, so a stop-word query of millions of
// postings is the job of up to 128 workgroups instead of one; the slices of a query share
// the pruning threshold through a global atomic max and a last kernel merges their lists.
// (thr_bm25_plan_kernel / bm25_edges_kernel / bm25_topk_kernel / bm25_merge_kernel.)
//
// Inside an item the posting lists are doc-sorted, so a doc's score is
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
//
// Stop words (ABI 7): terms held by a large share of the docs also have per-doc ROWS of
// quantised impacts and term frequencies (bm25_dense_rows_kernel).  A query that holds such
// terms is split the MaxScore way: stage A walks only its other terms' postings and PROBES the
// rows where a doc is scored (bm25_topk_kernel<.., DP = true>); stage B sweeps the docs that hold
// none of the other terms in doc windows over the rows (bm25_window_kernel) -- unless the dense
// terms' bounds cannot reach stage A's threshold (bm25_sweep_filter_kernel), the usual case.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "thr_common.hpp"

namespace thr {

// Block shapes (template arguments of bm25_topk_kernel):
//   BM_THREADS  threads per query
//   BM_STAGE    doc ids staged in LDS per doc-range pass
//   BM_WINDOW   doc slots of the mask path (BM_STAGE <= 3 * BM_WINDOW: the survivor list shares it)
//   BM_CAP      BlockTopK buffer (>= k + BM_THREADS)

typedef unsigned short bm_u16x2 __attribute__((ext_vector_type(2)));

// What bm25_walk_wave_kernel needs of an item and of its terms, gathered by bm25_edges_kernel so that
// a wave's set-up is two dependent loads, not five (item -> query words -> term ids -> list heads).
struct WwItem {
    int32_t q, sl, SA, S, nt, pm, qc, pad;
};
struct WwTerm {
    int64_t lo;     // first posting of the term's slice (absolute)
    double idf, ub;
    int64_t row;    // probed term: offset of its per-doc row; else -1
    int32_t len;    // postings of the slice (0 for a probed term)
    int32_t pad;
};

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
    unsigned long long* __restrict__ term_key, unsigned long long* __restrict__ block_key,
    uint8_t* __restrict__ post_imp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    int64_t lo = 0, hi = n_vocab;  // last term with rowptr[t] <= i
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= i) lo = mid; else hi = mid;
    }
    const double tfv = (double)post_tf[i], dlv = (double)doclen[post_doc[i]];
    const double c = bm25_contrib(idf[lo], tfv, dlv, avgdl, k1, b);
    if (post_imp) {
        // the posting's IMPACT tf (k1+1) / (tf + nrm) -- its contribution is idf * impact, and the
        // impact does not depend on the query -- rounded UP to 8 bits of (k1 + 1) / 255 (one more
        // step than the ceiling, so no rounding of this arithmetic can leave it below the impact)
        const double nrm = __dmul_rn(k1, __dadd_rn(__dsub_rn(1.0, b), __dmul_rn(b, __ddiv_rn(dlv, avgdl))));
        const double imp = __ddiv_rn(__dmul_rn(tfv, __dadd_rn(k1, 1.0)), __dadd_rn(tfv, nrm));
        const int qv = (int)ceil(imp * (255.0 / (k1 + 1.0))) + 1;
        post_imp[i] = (uint8_t)(qv > 255 ? 255 : qv < 0 ? 0 : qv);
    }
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

// DENSE TERMS (stop words: a term held by at least an eighth of the docs, chosen by the caller at
// index set-up).  Besides its CSR postings such a term gets one byte and one 16-bit word PER DOC:
// its quantised impact (post_imp of the doc's posting, 0 where the doc does not hold the term)
// and its term frequency (0 likewise).  bm25_window_kernel then needs no posting of the term at
// all: the bound of doc d is a coalesced byte load at [row + d], the exact contribution comes
// from the frequency at [row + d] -- no staging, no LDS atomics, no position search.
__global__ __launch_bounds__(256) void bm25_dense_rows_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ post_tf, const uint8_t* __restrict__ post_imp,
    const int32_t* __restrict__ terms, int64_t stride, uint8_t* __restrict__ dense_imp,
    uint16_t* __restrict__ dense_tf) {
    const int row = blockIdx.y;
    const int term = terms[row];
    const int64_t lo = rowptr[term], hi = rowptr[term + 1];
    for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t at = (int64_t)row * stride + post_doc[i];
        const int32_t tf = post_tf[i];
        dense_imp[at] = post_imp[i];
        dense_tf[at] = (uint16_t)(tf > 65535 ? 65535 : tf);   // (the caller keeps terms with tf > 65535 out)
    }
}

// ---------------------------------------------------------------------------------------------
// Work decomposition (one launch each, no host round trip):
//   bm25_plan_kernel   per query: the valid term ids in query order, the total posting count,
//                      the number of doc-range slices S_q (1 up to one slice's postings, else
//                      ~total / target, <= BM_MAX_SLICES; the target doubles until all items
//                      fit the item list), the item list (query, slice);
//   bm25_edges_kernel  per (item, term): the first posting of the slice in the term's list
//                      (slice s of S starts at doc B_s = the (len * s / S)-th doc of the
//                      query's longest list: equal shares of the dominant list whatever the
//                      distribution of its docs; the other lists are cut by binary search);
//   bm25_topk_kernel   persistent workgroups pull items from ctl[1];
//   bm25_merge_kernel  per query with S_q > 1: the best k of its slices' lists.
constexpr int BM_MAX_SLICES = 128;
constexpr int BW_PAD = 65536;          // docs per window of bm25_window_kernel = zero padding of a dense row
constexpr int BM_EXTRA_ITEMS = 16384;  // item list capacity = 2 * n_queries + this (8 K / 16 K / 32 K / 64 K measured on 256 and
                                       // 2048 stop-word queries: 0.95 / 0.86 / 0.85 / 0.86 and 2.07 / 1.90 / 1.94 / 2.19 ms)
constexpr int WW_TARGET_MIN = 640, WW_TARGET_MAX = 1536;   // postings per slice of the wave walk (bm25_walk_wave_kernel)
constexpr int BM_TARGET0 = 24576;      // postings per slice aimed at when the batch fills the grid (3 passes)
constexpr int BM_TARGET_MIN = 8192;    // ... and at least (one pass), when it does not: a one-query
                                       // call spreads its 75 K postings over nine workgroups
constexpr int PLAN_THREADS = 256;       // bm25_plan_kernel: one query per thread in as many workgroups as that takes (<= 64);
constexpr int PLAN_MAX_BLOCKS = 64;     // the workgroup that finishes last cuts the slices and writes the item list

__device__ __forceinline__ int bm_slices(long long tot, long long target);
// stage-A slices of a query with dense terms: none when its other terms have no posting
__device__ __forceinline__ int bm_slices_a(long long sparse, long long target) {
    return sparse > 0 ? bm_slices(sparse, target) : 0;
}
__device__ __forceinline__ int bm_slices(long long tot, long long target) {
    if (tot <= target) return 1;   // (a query of at most one slice's postings is one work item)
    const long long s = (tot + target - 1) / target;
    return s < 1 ? 1 : s > BM_MAX_SLICES ? BM_MAX_SLICES : (int)s;
}

// first doc of slice s of S of a window-kernel query (a multiple of 4: the dword loads of the dense rows)
__device__ __forceinline__ int64_t bm_window_edge(int64_t n_docs, int s, int S) {
    return s >= S ? n_docs : (n_docs * s / S) & ~(int64_t)3;
}

__global__ __launch_bounds__(PLAN_THREADS) void bm25_plan_kernel(
    const int64_t* __restrict__ rowptr, int64_t n_vocab, const int32_t* __restrict__ query_terms,
    int nq, int mt, int cap, int cap_wave, int conjunctive, int n_slots, int target_max, int target_a0, int wave_mode, int walk_div,
    const int32_t* __restrict__ dense_slot, const double* __restrict__ term_ub, int64_t n_docs,
    int32_t* __restrict__ ctl, int64_t* __restrict__ q_tot, double* __restrict__ q_dub,
    int32_t* __restrict__ q_nt, int32_t* __restrict__ q_S, int32_t* __restrict__ q_SA,
    int32_t* __restrict__ q_pmask, int32_t* __restrict__ q_item0,
    int32_t* __restrict__ q_long, int32_t* __restrict__ q_terms, int2* __restrict__ items) {
    // Part 1, every workgroup: what a query is made of (its own load chains -- term ids, then list
    // lengths / bounds / row slots -- are the kernel's time: one query per thread, the workgroups of
    // the grid on different CUs; round 3 ran this on ONE workgroup, two queries per thread: 62 us)
    __shared__ int red[PLAN_THREADS];
    int n_dp = 0, n_blk = 0;
    for (int q = blockIdx.x * PLAN_THREADS + threadIdx.x; q < nq; q += gridDim.x * PLAN_THREADS) {
        int nt = 0, lng = 0;
        long long tot = 0, best = -1;
        bool dead = false;   // AND mode: a term outside the vocabulary is held by no doc
#pragma unroll 4
        for (int j = 0; j < mt; ++j) {
            const int term = query_terms[(int64_t)q * mt + j];
            if (term >= n_vocab && conjunctive) dead = true;
            if (term < 0 || term >= n_vocab) continue;   // padding / unknown term: no postings
            q_terms[(int64_t)q * mt + nt++] = term;
        }
        if (dead) nt = 0;   // (nothing to score: the item writes an empty list)
        // A query with dense terms (OR form, <= 8 terms) is split the MaxScore way.  Some of its
        // terms are PROBED -- never walked, read from their per-doc rows where a doc is scored --,
        // the others are WALKED.  Stage A walks the walked terms' postings (slices of those lists,
        // bm25_topk_kernel<.., true>); stage B sweeps the shard's docs that hold none of the walked
        // terms in doc windows (bm25_window_kernel) -- and is skipped when the probed terms' bounds
        // together cannot reach stage A's threshold.  Which terms are probed only decides the
        // cost, never the result: a term held by 1/64 of the docs always is (walking a posting costs
        // ~20x what a sweep spends on a doc); rarer terms with rows are walked, rarest first, until
        // the bounds of what is left sum to half the largest bound of a walked term held by >= 200
        // docs (a guess of stage A's threshold from below: then the sweep is very likely skipped).
        // q_SA = -1: not such a query; else the number of stage-A slices (the first q_SA of q_S).
        uint32_t pmask = 0;
        double dub = 0.0;
        long long walked = 0;
        if (nt <= 8) {
            // everything about the (up to eight) terms in registers, the loads of all of them in flight
            // together: this kernel is one workgroup, its time is the length of its load chains
            long long len_[8];
            double ub_[8];
            uint32_t cap_mask = 0;   // terms that have per-doc rows
            const bool rows = dense_slot && !conjunctive;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool on = t < nt;
                const int term = on ? q_terms[(int64_t)q * mt + t] : 0;
                len_[t] = on ? rowptr[term + 1] - rowptr[term] : 0;
                ub_[t] = on && rows ? term_ub[term] : 0.0;
                if (on && rows && dense_slot[term] >= 0) cap_mask |= 1u << t;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                tot += len_[t];
                if (t < nt && len_[t] > best) { best = len_[t]; lng = t; }
            }
            walked = tot;
            if (cap_mask) {
                double walk_ub = 0.0;
                pmask = cap_mask;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if ((cap_mask >> t) & 1u) dub += ub_[t];
                    else if (t < nt && len_[t] >= 200 && ub_[t] > walk_ub) walk_ub = ub_[t];
                }
                for (;;) {
                    if (!(dub > 0.5 * walk_ub)) break;
                    int pick = -1;
                    long long pick_len = 0;
                    double pick_ub = 0.0;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {   // the rarest probed term that may be walked
                        if (!((pmask >> t) & 1u)) continue;
                        if (len_[t] * walk_div >= n_docs) continue;   // (walking costs ~20x a sweep's per-doc work)
                        if (pick < 0 || len_[t] < pick_len) { pick = t; pick_len = len_[t]; pick_ub = ub_[t]; }
                    }
                    if (pick < 0) break;
                    pmask &= ~(1u << pick);
                    dub -= pick_ub;
                    if (pick_len >= 200 && pick_ub > walk_ub) walk_ub = pick_ub;
                }
                dub = 0.0;   // (summed again: no cancellation left over from the subtractions)
                walked = 0;
                best = -1;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t >= nt) continue;
                    if ((pmask >> t) & 1u) {
                        dub += ub_[t];
                    } else {
                        walked += len_[t];
                        if (len_[t] > best) { best = len_[t]; lng = t; }   // (the longest WALKED list cuts the stage-A slices)
                    }
                }
                if (pmask) ++n_dp;
            }
        } else {
            for (int t = 0; t < nt; ++t) {
                const int term = q_terms[(int64_t)q * mt + t];
                const long long len = rowptr[term + 1] - rowptr[term];
                if (len > best) { best = len; lng = t; }
                tot += len;
            }
        }
        // Wave mode (bm25_walk_wave_kernel): an OR query of <= 8 terms WITHOUT probed terms is walked by
        // waves too -- it is a stage A with nothing probed and no stage B: bit 30 marks it, all its
        // slices are stage-A slices (q_SA == q_S), cut with the waves' slice size.
        const bool wave_q = wave_mode && !conjunctive && nt >= 1 && nt <= 8 && !pmask;
        q_SA[q] = (pmask || wave_q) ? 0 : -1;      // (slice counts: below, once the target is known)
        if (!(pmask || wave_q)) ++n_blk;           // (left to the workgroup walk)
        q_pmask[q] = (int32_t)pmask | (wave_q ? (1 << 30) : 0);
        q_dub[q] = dub;
        q_tot[q] = pmask ? -(walked + 1) : tot;    // dense terms: -(postings of the walked terms + 1)
        q_nt[q] = nt;
        q_long[q] = lng;
    }
    {   // queries with probed terms: one atomic per wave
        for (int o = WAVE / 2; o > 0; o >>= 1) {
            n_dp += __shfl_down(n_dp, o, WAVE);
            n_blk += __shfl_down(n_blk, o, WAVE);
        }
        if ((threadIdx.x & (WAVE - 1)) == 0 && n_dp) atomicAdd(&ctl[3], n_dp);
        if ((threadIdx.x & (WAVE - 1)) == 0 && n_blk) atomicAdd(&ctl[8], n_blk);   // queries the workgroup walk takes
    }
    // Part 2, the workgroup that finishes last: slice size, item list.  (Its reads of the other
    // workgroups' per-query words go to L2: agent-scope atomic loads.)
    __shared__ int is_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(&ctl[6], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    auto tot_of = [&](int q) -> long long {
        return (long long)__hip_atomic_load(&q_tot[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wave_q_of = [&](int q) -> bool {   // an ordinary query the waves walk (bit 30 of its probe mask)
        return (__hip_atomic_load(&q_pmask[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 30) & 1;
    };
    const int per = (nq + PLAN_THREADS - 1) / PLAN_THREADS;
    const int q0 = threadIdx.x * per < nq ? threadIdx.x * per : nq;
    const int q1 = q0 + per < nq ? q0 + per : nq;
    // slice size: what gives every workgroup slot of the grid an item, between one pass and three
    __shared__ long long red64[PLAN_THREADS], red64w[PLAN_THREADS];
    {
        long long t = 0, tw = 0;   // (a stage-B sweep counts one unit per doc)
        for (int q = q0; q < q1; ++q) {
            const long long v = tot_of(q);
            t += v >= 0 ? v : -v - 1 + n_docs;
            tw += v >= 0 ? (wave_q_of(q) ? v : 0) : -v - 1;   // postings the waves will walk
        }
        red64[threadIdx.x] = t;
        red64w[threadIdx.x] = tw;
        __syncthreads();
        for (int o = PLAN_THREADS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                red64[threadIdx.x] += red64[threadIdx.x + o];
                red64w[threadIdx.x] += red64w[threadIdx.x + o];
            }
            __syncthreads();
        }
    }
    long long target = red64[0] / (n_slots > 0 ? n_slots : 1);
    target = target < BM_TARGET_MIN ? BM_TARGET_MIN : target > target_max ? target_max : target;
    // The waves' slices have their own size (bm25_walk_wave_kernel): what gives each of the ``target_a0``
    // wave slots of the chip an item, between WW_TARGET_MIN (a small batch spreads over many waves: 256
    // survey queries 0.46 -> 0.43 ms) and WW_TARGET_MAX (a full batch pays the per-item set-up less
    // often: 2048 survey queries 0.89 -> 0.87 ms); 0: no wave walk, stage A takes the shared size.
    long long target_a = target;
    if (target_a0 > 0) {
        target_a = red64w[0] / target_a0;
        target_a = target_a < WW_TARGET_MIN ? WW_TARGET_MIN : target_a > WW_TARGET_MAX ? WW_TARGET_MAX : target_a;
    }
    // ``cap`` items for the sweeps and the workgroup walk's items (the slice size that budget gives them
    // was tuned with it), ``cap_wave`` more for the waves' ~1 K-posting slices
    __shared__ int red_w[PLAN_THREADS];
    int total = 0, mine = 0;
    for (;;) {
        mine = 0;
        int mine_w = 0;
        for (int q = q0; q < q1; ++q) {
            const long long v = tot_of(q);
            if (v >= 0) {
                if (wave_q_of(q)) mine_w += bm_slices(v, target_a);
                else mine += bm_slices(v, target);
            } else {
                const int sa = bm_slices_a(-v - 1, target_a);
                if (wave_mode) mine_w += sa; else mine += sa;
                mine += bm_slices(n_docs, target);
            }
        }
        red[threadIdx.x] = mine;
        red_w[threadIdx.x] = mine_w;
        __syncthreads();
        for (int o = PLAN_THREADS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                red[threadIdx.x] += red[threadIdx.x + o];
                red_w[threadIdx.x] += red_w[threadIdx.x + o];
            }
            __syncthreads();
        }
        total = red[0] + red_w[0];
        const bool fits = red[0] <= cap && red_w[0] <= cap_wave;
        const bool grow_w = red_w[0] > cap_wave;
        __syncthreads();
        mine += mine_w;
        if (fits) break;   // (every query is one or two items once the targets reach its totals: terminates)
        if (grow_w) {
            target_a *= 2;
            continue;
        }
        target *= 2;
    }
    // exclusive prefix of the per-thread item counts
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < PLAN_THREADS; o <<= 1) {
        const int v = (int)threadIdx.x >= o ? red[threadIdx.x - o] : 0;
        __syncthreads();
        red[threadIdx.x] += v;
        __syncthreads();
    }
    // Item order: slice 0 of EVERY query first (item q), then the other slices query by query
    // (item q_item0[q] + s, s >= 1).  The slices of a query that are started together all begin
    // without a threshold and score every doc of their first pass in full; with this order a
    // query's first slice has published its threshold (theta_glob) long before most of its other
    // slices are taken, and those start with the pruning already in force.
    int rest = (red[threadIdx.x] - mine) - q0;   // slices s >= 1 of the queries before this thread's
    for (int q = q0; q < q1; ++q) {
        int S = 0;
        const long long v = tot_of(q);
        if (v >= 0) {
            const bool wq = wave_q_of(q);
            S = bm_slices(v, wq ? target_a : target);
            if (wq) q_SA[q] = S;
        } else {
            const int SA = bm_slices_a(-v - 1, target_a);
            q_SA[q] = SA;
            S = SA + bm_slices(n_docs, target);
        }
        q_S[q] = S;
        q_item0[q] = nq + rest - 1;
        items[q] = make_int2(q, 0);
        for (int s = 1; s < S; ++s) items[nq + rest + s - 1] = make_int2(q, s);
        rest += S - 1;
    }
    if (threadIdx.x == 0) {
        ctl[0] = total;
        ctl[7] = (int)(target_a > 0x7fffffff ? 0x7fffffff : target_a);   // (what a stage-A slice was aimed at)
    }
}

__global__ __launch_bounds__(256) void bm25_edges_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ ctl, const int32_t* __restrict__ q_nt,
    const int32_t* __restrict__ q_S, const int32_t* __restrict__ q_SA, const int32_t* __restrict__ q_long,
    const int32_t* __restrict__ q_terms, const int2* __restrict__ items, int mt,
    const int32_t* __restrict__ q_pmask, int64_t n_docs, int32_t* __restrict__ ipos,
    const double* __restrict__ idf, const double* __restrict__ term_ub, const int32_t* __restrict__ dense_slot,
    int64_t dense_stride, const int32_t* __restrict__ query_coll, WwItem* __restrict__ wrec,
    WwTerm* __restrict__ wterm) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int item = (int)(g / mt), slot = (int)(g % mt);
    if (item >= ctl[0]) return;
    const int2 it = items[item];
    const int q = it.x;
    if (wrec && slot == 0) {
        WwItem r;
        r.q = q; r.sl = it.y; r.SA = q_SA[q]; r.S = q_S[q]; r.nt = q_nt[q]; r.pm = q_pmask[q] & 0xFF;
        r.qc = query_coll ? query_coll[q] : -1; r.pad = 0;
        wrec[item] = r;
    }
    if (slot >= q_nt[q]) return;
    const int term = q_terms[(int64_t)q * mt + slot];
    const int64_t lo = rowptr[term];
    const int full = (int)(rowptr[term + 1] - lo);
    int start = 0, end = full;
    const int SA = q_SA[q];      // -1: an ordinary query; else its first SA slices are stage A
    int S = q_S[q], s = it.y;
    const bool sweep = SA >= 0 && s >= SA;
    if (SA >= 0) {
        if (sweep) s -= SA, S -= SA; else S = SA;
    }
    if (SA >= 0 && ((q_pmask[q] >> slot) & 1)) {
        start = end = 0;         // a probed term is read from its per-doc rows
    } else if (sweep) {
        // stage B: slice s is the doc range [bm_window_edge(s), bm_window_edge(s + 1))
        if (S > 1) {
            start = count_below(post_doc + lo, full, bm_window_edge(n_docs, s, S));
            end = count_below(post_doc + lo, full, bm_window_edge(n_docs, s + 1, S));
        }
    } else if (S > 1) {
        const int L = q_long[q];   // (stage A: the longest of the other terms' lists)
        const int tl = q_terms[(int64_t)q * mt + L];
        const int64_t lo_l = rowptr[tl], len_l = rowptr[tl + 1] - lo_l;
        // edge e of S: the (len * e / S)-th doc of the longest list (S > 1 only with > BM_TARGET_MIN
        // postings: len_l >= 256 > S, the edges are distinct)
        auto edge = [&](int e) -> int {
            if (e == 0) return 0;
            if (e == S) return full;
            const int64_t p = len_l * e / S;
            return slot == L ? (int)p : count_below(post_doc + lo, full, (int64_t)post_doc[lo_l + p]);
        };
        start = edge(s);
        end = edge(s + 1);
    }
    ipos[((int64_t)item * mt + slot) * 2] = start;
    ipos[((int64_t)item * mt + slot) * 2 + 1] = end;
    if (wterm && slot < 8 && SA >= 0 && !sweep) {
        const bool probed = (q_pmask[q] >> slot) & 1;
        WwTerm t;
        t.lo = lo + start;
        t.idf = idf[term];
        t.ub = term_ub[term];
        t.row = probed ? (int64_t)dense_slot[term] * dense_stride : -1;
        t.len = probed ? 0 : end - start;
        t.pad = 0;
        wterm[(int64_t)item * 8 + slot] = t;
    }
}

// An item's postings are consumed in DOC-RANGE passes.  A pass stages, from every term's list,
// the next quota_t postings (quotas proportional to what is left of each list, together one LDS
// stage), then takes d_hi = the smallest "last staged doc + 1" among the lists that have more
// postings behind their quota: every posting with doc < d_hi of EVERY list is then on chip, so a
// doc's postings all fall into the same pass and the owner search never leaves LDS, whatever
// the length of the lists.  The postings with doc >= d_hi stay for the next pass and are staged
// again.
//
// WAND-style pruning (exact): passes visit the docs in ascending id order, so once k docs have
// been scored every later doc of the item has to BEAT the item's k-th best score theta (a tie
// loses on the id); against the threshold shared by the query's other slices (th_glob, whose
// docs may have larger ids) a doc is dropped only when its bound is strictly BELOW it.  Phase 1
// of a pass is LDS-only: it learns from the staged doc ids which query terms hold a doc and sums
// their term_ub in query-term order; rounding is monotone, so fl(sum of bounds) >= fl(sum of
// contributions).  The survivors are compacted into an LDS list; phase 2 walks that list a
// workgroup's width at a time: tighter block_ub check, collection filter, term-frequency /
// doc-length gathers, float64 score, top-k push.
//
// Phase 1 has three forms.  DENSE lists: every staged posting ORs its term's bit into the
// doc's slot of a mask array -- O(1) per posting -- and the non-empty slots are the candidate
// docs.  The mask holds 8 bits per doc for queries of <= 8 terms (4 BM_WINDOW docs), 32 bits
// otherwise; a pass whose staged range is wider than the mask is CUT to the mask's width when
// that still consumes at least an eighth of the staged postings (so stop-word lists always
// take this path).  SPARSE lists: a Bloom bit per (list, doc) answers "is this doc in another
// list" with one LDS read; singletons are scored (or dropped on term_ub) at once, the others
// are searched from a dense work list.  In between, with a threshold: owners and bounds by
// binary search in LDS.
// BM_STAMPS (diagnostic build, _build.build_variant("stamps", ["BM_STAMPS"]); scripts/bm25_stamps.py):
// thread 0 of every workgroup adds the cycles between consecutive phase marks into buckets,
// written behind the workspace; thr_bm25_topk then waits for the launch and prints the shares.
#ifdef BM_STAMPS
constexpr int BM_NSTAMP = 20;   // 0-13 phases (cycles), 14-19 counters
#define BM_STAMP(i)                                                       \
    do {                                                                  \
        if (threadIdx.x == 0) {                                           \
            const unsigned long long now_ = __builtin_readcyclecounter(); \
            stamp_acc[i] += now_ - stamp_last;                            \
            stamp_last = now_;                                            \
        }                                                                 \
    } while (0)
#define BM_COUNT(i, v) do { if (threadIdx.x == 0) stamp_acc[i] += (unsigned long long)(v); } while (0)
#else
#define BM_STAMP(i)
#define BM_COUNT(i, v)
#endif

// DPM: 0 = ordinary queries only, 1 = stage-A slices only, 2 = both kinds in one launch (decided per item)
template <int BM_THREADS, int BM_STAGE, int BM_WINDOW, int BM_CAP, int DPM>
__global__ __launch_bounds__(BM_THREADS, 4) void bm25_topk_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ post_tf, const float* __restrict__ doclen,
    const double* __restrict__ idf, const double* __restrict__ term_ub,
    const double* __restrict__ block_ub, const uint8_t* __restrict__ post_imp,
    const int32_t* __restrict__ dense_slot, const uint16_t* __restrict__ dense_tf, int64_t dense_stride,
    double avgdl, double k1, double b,
    double imp_unit /* (k1 + 1) / 255 */, double imp_per_unit /* 255 / (k1 + 1): the host's divisions, same bits */,
    int64_t id_base, int max_terms, int k, int conjunctive, const int32_t* __restrict__ doc_coll,
    const int32_t* __restrict__ query_coll, int n_queries, int fuse_div, int32_t* __restrict__ ctl,
    const int32_t* __restrict__ q_nt, const int32_t* __restrict__ q_S, const int32_t* __restrict__ q_SA,
    const int32_t* __restrict__ q_pmask,
    const int32_t* __restrict__ q_terms, const int2* __restrict__ items,
    const int32_t* __restrict__ ipos, unsigned long long* __restrict__ theta_glob,
    double* __restrict__ slice_s, int64_t* __restrict__ slice_id, int32_t* __restrict__ slice_cnt,
    double* __restrict__ out_s, int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt
#ifdef BM_STAMPS
    , unsigned long long* __restrict__ stamps, unsigned long long* __restrict__ walk_log
#endif
    ) {
#ifdef BM_STAMPS
    unsigned long long stamp_acc[BM_NSTAMP] = {0};
    unsigned long long stamp_last = __builtin_readcyclecounter(), stamp_items = 0;
#endif
    __shared__ TermRange tr[THR_BM25_MAX_TERMS];   // .sub = postings of this pass, .lds_off = where staged
    __shared__ double t_idf[THR_BM25_MAX_TERMS], t_ub[THR_BM25_MAX_TERMS];
    __shared__ int t_staged[THR_BM25_MAX_TERMS];   // postings of the term staged in this pass
    __shared__ int t_subwin[THR_BM25_MAX_TERMS];   // ... of them inside the mask window
    __shared__ int t_prefix[THR_BM25_MAX_TERMS + 1];
    // DP (stage A of a query with dense terms): those terms have no postings here; their per-doc
    // rows are probed when a doc is scored, their bounds are added to every doc's bound
    __shared__ int64_t t_row[8];    // dense term: offset of its per-doc row; else -1
    __shared__ double p_dub;        // sum of the dense terms' term_ub
    __shared__ int p_dmaxq;         // ... of their largest quantised impacts, in accumulator units
    __shared__ int t_w[8];          // accumulator path: integer weight of a term's quantised impacts
    __shared__ double acc_scale;    // ... accumulated bound = acc_scale * (real bound), rounded up
    __shared__ int p_acc, p_thq;    // this pass takes the accumulator path; its threshold in acc units
    __shared__ int remaining, last_compact, n_surv, n_single, p_boot_q, cur_item;
    __shared__ int t_order[THR_BM25_MAX_TERMS];   // terms by descending term_ub
    __shared__ int64_t d_hi, d_lo, p_last;
    __shared__ double th_glob;
    __shared__ double b_s[BM_CAP];
    __shared__ int64_t b_id[BM_CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    __shared__ int32_t st_doc[BM_STAGE];
    // mask path: BM_WINDOW mask words (which query terms hold doc d_lo + slot; 1 or 4 slots per
    // word) + up to BM_WINDOW surviving slots behind them; search path: up to BM_STAGE surviving
    // staged indices; Bloom path: the bits + a work list.  One 24 KiB buffer.
    // accumulator path: ACC_WORDS words of two 16-bit doc accumulators, the survivor slots behind them
    constexpr int ACC_WORDS = BM_WINDOW, ACC_SLOTS = 2 * ACC_WORDS;   // (a wider window was measured: no gain)
    // survivor slots of a scan: SURV_CAP 16-bit entries behind the masks / accumulators (with a
    // threshold a window has ~100 survivors; a scan that finds more is redone SURV_CAP slots at a time)
    constexpr int SURV_CAP = BM_WINDOW;
    constexpr int SCR_WORDS = ACC_WORDS + SURV_CAP / 2;
    __shared__ uint32_t scratch[SCR_WORDS];
    static_assert(sizeof(uint32_t) * SCR_WORDS >= sizeof(uint16_t) * BM_STAGE, "survivor list of the search path must fit");
    static_assert(BM_CAP >= THR_TOPK_MAX + BM_THREADS && BM_STAGE <= 65536, "top-k buffer / 16-bit staged indices");
    static_assert(4 * BM_WINDOW <= 65536, "16-bit slot indices");
    uint32_t* mask = scratch;

    const int n_items = ctl[0];
    {   // Which launches work is decided on the device (no host round trip): when at least 1/fuse_div of
        // the batch's queries hold dense terms, ONE launch (DPM 2) takes the ordinary items and the
        // stage-A slices together -- two half-empty persistent grids, each with its own tail, cost
        // more than the row probes' registers cost the ordinary items (256 / 2048 survey queries:
        // 0.85 -> 0.60 / 1.83 -> 1.55 ms; a batch without dense terms: 0.49 -> 0.55 ms, hence the switch).
        // (fuse_div < 0: wave mode -- the waves took every OR query of <= 8 terms; this launch has the
        // rest, and nothing to do at all when the plan counted none)
        if (fuse_div < 0 && ctl[8] == 0) return;
        const int nd = ctl[3];
        const bool fuse = fuse_div > 0 && nd > 0 && (long long)nd * fuse_div >= n_queries;
        if (DPM == 2 ? !fuse : DPM == 1 ? (fuse || nd == 0) : fuse) return;
    }
    BlockTopK<BM_CAP, BM_THREADS> tk;
    for (;;) {
        __syncthreads();   // the previous item's LDS state is no longer read
        if (threadIdx.x == 0) cur_item = atomicAdd(&ctl[DPM == 1 ? 4 : 1], 1);
        __syncthreads();
        const int item = cur_item;
        if (item >= n_items) break;   // (uniform: every workgroup of the grid ends here)
#ifdef BM_STAMPS
        const unsigned long long item_t0 = __builtin_readcyclecounter();
        int item_passes = 0;
#endif
        const int2 it = items[item];
        const int q = it.x, sl = it.y;
        // a query with dense terms: its first q_SA slices are stage A's (DP), the rest bm25_window_kernel's
        const int SA_ = q_SA[q];
        if (DPM == 0 ? SA_ >= 0 : DPM == 1 ? !(SA_ >= 0 && sl < SA_) : (SA_ >= 0 && sl >= SA_)) continue;
        const bool DP = DPM == 1 || (DPM == 2 && SA_ >= 0);
        const int S = q_S[q];
        const int nt = q_nt[q];
        const int qc = query_coll ? query_coll[q] : -1;   // -1: no collection filter
        // 8 mask bits per doc for queries of <= 8 terms: 4 docs per mask word
        const int ms = nt <= 8 ? 2 : 0;
        const int spw = 1 << ms;
        const int64_t WIN = (int64_t)BM_WINDOW << ms;
        if ((int)threadIdx.x < nt) {
            const int slot = threadIdx.x;
            const int term = q_terms[(int64_t)q * max_terms + slot];
            const int64_t lo = rowptr[term];
            const int full = (int)(rowptr[term + 1] - lo);
            const int start = ipos[((int64_t)item * max_terms + slot) * 2];
            const int end = ipos[((int64_t)item * max_terms + slot) * 2 + 1];
            tr[slot].lo = lo + start;
            tr[slot].len = end - start;
            tr[slot].cur = 0;
            t_idf[slot] = idf[term];
            t_ub[slot] = term_ub ? term_ub[term] : INFINITY;
            if (DP && slot < 8) {
                const bool probed = (q_pmask[q] >> slot) & 1;
                t_row[slot] = probed ? (int64_t)dense_slot[term] * dense_stride : -1;   // (its slice is empty: bm25_edges_kernel)
            }
        }
        BM_STAMP(0);
        if (threadIdx.x == 0) {
            last_compact = 0;
            const unsigned long long g0 = S > 1 ? __hip_atomic_load(&theta_glob[q], __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            th_glob = g0 ? dkey_inv(g0) : -INFINITY;
        }
        tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);  // includes a barrier
        // Accumulator path (queries of <= 8 terms, OR form, impacts given): a doc's bound is the
        // sum over its postings of idf_t * impact_q * (k1+1)/255, accumulated in 16 bits per doc
        // slot as integers imp_q * w_t with w_t = ceil(idf_t * (k1+1)/255 * scale), scale chosen so
        // that the weights add up to <= 256 (255 * 256 < 2^16: a slot cannot overflow into its
        // neighbour).  It is an upper bound of the doc's score to within 1e-15, far tighter than
        // the sum of the per-term maxima: ~1 % of the docs of a stop-word query survive it
        // instead of ~16 %.
        const bool acc_ok = post_imp != nullptr && !conjunctive && nt >= 1 && nt <= 8;
        if (threadIdx.x == 0) {
            int total = 0;
            for (int t = 0; t < nt; ++t) total += tr[t].len;
            remaining = total;
            for (int t = 0; t < nt; ++t) t_order[t] = t;
            for (int a = 1; a < nt; ++a) {   // (insertion sort, <= 32 terms)
                const int ta = t_order[a];
                int c = a;
                for (; c > 0 && t_ub[t_order[c - 1]] < t_ub[ta]; --c) t_order[c] = t_order[c - 1];
                t_order[c] = ta;
            }
            if (acc_ok) {
                const double c = imp_unit;
                double sum = 0.0;
                for (int t = 0; t < nt; ++t) sum += t_idf[t] * c;
                const double scale = 248.0 / sum;
                for (int t = 0; t < nt; ++t) {
                    int w = (int)ceil(t_idf[t] * c * scale);
                    t_w[t] = w < 1 ? 1 : w;
                }
                acc_scale = scale;
            }
            if (DP) {
                double dub = 0.0;
                int dmaxq = 0;
                for (int t = 0; t < nt; ++t) {
                    if (t_row[t] < 0) continue;
                    dub += t_ub[t];
                    // the term's largest quantised impact: its bound / idf in steps of (k1+1)/255, as bm25_bounds_kernel rounds
                    const double im = t_idf[t] > 0.0 ? ceil(t_ub[t] / t_idf[t] * imp_per_unit) + 1.0 : 255.0;
                    dmaxq += t_w[t] * (im > 255.0 || !(im >= 0.0) ? 255 : (int)im);
                }
                p_dub = dub;
                p_dmaxq = dmaxq;
            }
        }
        __syncthreads();
        BM_STAMP(1);

        while (remaining > 0) {
#ifdef BM_STAMPS
            ++item_passes;
#endif
            // ---- quotas: the stage is shared out in proportion to what is left of each list ----
            if (threadIdx.x == 0) {
                // (th_glob: read at the item's start and again in every staging interval -- the load
                // of the other slices' threshold travels WITH the staging loads, it is never a round
                // trip of its own on the pass's critical path)
                const unsigned long long g = th_glob > -INFINITY ? 1ull : 0ull;
                // no threshold anywhere yet and a long way to go: a short first pass gets one cheaply
                // (without a threshold every staged doc is scored in full)
                // (sliced items only: an unsliced query is at most three passes long)
                const bool warm = g != 0ull || (b_cnt >= k && th_s > -INFINITY) || remaining <= BM_STAGE || S == 1;
                // with a threshold to hold them against, the pass accumulates per-doc impact bounds
                // (2 * BM_WINDOW 16-bit slots); without one every doc is scored anyway: the mask
                const bool have_th = g != 0ull || (b_cnt >= k && th_s > -INFINITY);
                p_acc = acc_ok && have_th ? 1 : 0;
                if (p_acc) {
                    double th = (b_cnt >= k && th_s > -INFINITY) ? th_s : -INFINITY;
                    th = th_glob > th ? th_glob : th;
                    // prune only what is below the threshold by more than the arithmetic's slack
                    const double tq = floor(th * acc_scale * (1.0 - 1e-12));
                    p_thq = tq < 0.0 ? 0 : tq > 70000.0 ? 70000 : (int)tq;
                }
                const int stage = warm ? BM_STAGE : BM_STAGE / 4;
                int off = 0;
                const int spare = stage - 32 * nt;   // every list gets at least 32 slots
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
            BM_STAMP(2);
            unsigned long long gth = 0ull;
            if (threadIdx.x == 0 && S > 1)
                gth = __hip_atomic_load(&theta_glob[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int t = 0; t < nt; ++t) {
                const int32_t* src = post_doc + tr[t].lo + tr[t].cur;
                int32_t* dst = st_doc + tr[t].lds_off;
                // (a variant that puts a term's 16 loads per thread in flight before the first LDS
                // store was A/B-measured on one box: 5 % slower -- the registers it holds cost more)
                for (int i = threadIdx.x; i < t_staged[t]; i += BM_THREADS) dst[i] = src[i];
            }
            if (threadIdx.x == 0 && S > 1) th_glob = gth ? dkey_inv(gth) : -INFINITY;
            __syncthreads();
            BM_STAMP(3);
            if ((int)threadIdx.x < nt) {
                const int t = threadIdx.x;
                if (t_staged[t] > 0 && tr[t].cur + t_staged[t] < tr[t].len)   // more postings behind the quota
                    atomicMin((unsigned long long*)&d_hi,
                              (unsigned long long)((int64_t)st_doc[tr[t].lds_off + t_staged[t] - 1] + 1));
                if (t_staged[t] > 0)
                    atomicMin((unsigned long long*)&d_lo, (unsigned long long)st_doc[tr[t].lds_off]);
            }
            __syncthreads();
            if ((int)threadIdx.x < nt) {
                TermRange& r = tr[threadIdx.x];
                const int stg = t_staged[threadIdx.x];
                r.sub = d_hi == INT64_MAX ? stg : count_below(st_doc + r.lds_off, stg, d_hi);
                const int64_t win = p_acc ? (int64_t)ACC_SLOTS : WIN;
                const int64_t dw = d_lo + win < d_hi ? d_lo + win : d_hi;
                t_subwin[threadIdx.x] = dw == INT64_MAX ? stg : count_below(st_doc + r.lds_off, stg, dw);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int total = 0, totw = 0;
                int64_t last = 0;
                for (int t = 0; t < nt; ++t) {
                    total += tr[t].sub;
                    totw += t_subwin[t];
                    if (tr[t].sub > 0) {
                        const int64_t e = (int64_t)st_doc[tr[t].lds_off + tr[t].sub - 1] + 1;
                        last = e > last ? e : last;
                    }
                }
                // wider than the mask, but the mask's width holds a fair share of the staged
                // postings: cut the pass to that width (the rest is staged again)
                if (total > 0 && last - d_lo > (p_acc ? (int64_t)ACC_SLOTS : WIN) && (int64_t)totw * 8 >= total) {
                    total = 0;
                    last = 0;
                    for (int t = 0; t < nt; ++t) {
                        tr[t].sub = t_subwin[t];
                        total += tr[t].sub;
                        if (tr[t].sub > 0) {
                            const int64_t e = (int64_t)st_doc[tr[t].lds_off + tr[t].sub - 1] + 1;
                            last = e > last ? e : last;
                        }
                    }
                }
                int acc = 0;
                for (int t = 0; t < nt; ++t) {
                    t_prefix[t] = acc;
                    acc += tr[t].sub;
                }
                t_prefix[nt] = acc;
                p_last = last;   // one past the last doc of the pass
                n_surv = 0;
                n_single = 0;
                p_boot_q = 0;
            }
            __syncthreads();
            BM_STAMP(4);
            const int total = t_prefix[nt];
            const double thg = th_glob;                 // the query's other slices' threshold (or -inf)
            const bool have_local = b_cnt >= k && th_s > -INFINITY;
            const double theta = have_local ? th_s : -INFINITY;
            const bool have_theta = have_local || thg > -INFINITY;
            // (a bound ub cannot make the top-k: it does not beat this item's threshold, or it is
            // below the threshold of the query's other slices)
            // (DP: plus the dense terms' bounds -- added out of query-term order, so with a margin far
            // above the rounding of an 8-term sum and far below anything that matters for pruning)
            const double dub = DP ? p_dub : 0.0;
            auto pruned = [&](double ub) -> bool {
                if (DP) ub = (ub + dub) * (1.0 + 1e-12);
                return !(ub > theta) || ub < thg;
            };
            // a dense term's contribution to doc d (DP), 0 when the doc does not hold it
            auto dense_add = [&](int e, int32_t d, double dl, double& score) {
                const int tfd = (int)dense_tf[t_row[e] + d];
                if (tfd > 0) score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)tfd, dl, avgdl, k1, b));
            };
            auto push = [&](bool ok, double sc, int64_t d) { tk.push(ok && !(sc < thg), sc, d); };
            const int64_t last = p_last;
            const int64_t first = d_lo;
            const bool use_acc = p_acc != 0;
            const bool masked = total > 0 && last - first <= (use_acc ? (int64_t)ACC_SLOTS : WIN);
            const uint32_t thq = (uint32_t)p_thq;
            const uint32_t dmaxq = DP ? (uint32_t)p_dmaxq : 0u;
            uint16_t* surv = reinterpret_cast<uint16_t*>(masked ? scratch + ACC_WORDS : scratch);   // (ACC_WORDS == BM_WINDOW)
            auto slot_mask = [&](int slot) -> uint32_t {
                const uint32_t v = mask[slot >> ms];
                return ms ? (v >> ((slot & 3) << 3)) & 0xFFu : v;
            };

            // ---- phase 2: the survivors, densely ----
            // survivors are staged indices of owner postings (mode 0), doc slots of the mask (1) or doc
            // slots of the accumulators (2: which terms hold the doc is not known, every list is searched)
            auto phase2 = [&](int mode, int ns) {
                for (int base = 0; base < ns; base += BM_THREADS) {
                    const int j = base + threadIdx.x;
                    bool keep = j < ns;
                    double score = 0.0;
                    int32_t d = 0;
                    if (keep) {
                        int t = 0, at;
                        uint32_t has = 0xFFFFFFFFu;   // terms that may hold the doc
                        if (mode == 2) {
                            d = (int32_t)(first + surv[j]);
                            at = 0;
                        } else if (mode == 1) {   // survivor = doc slot: owner = lowest term bit, position searched
                            d = (int32_t)(first + surv[j]);
                            has = slot_mask(surv[j]);
                            t = __ffs((int)has) - 1;
                            at = tr[t].lds_off + find_doc(st_doc + tr[t].lds_off, tr[t].sub, d);
                        } else {        // survivor = staged index of the owner posting
                            at = surv[j];
                            while (t + 1 < nt && at >= tr[t + 1].lds_off) ++t;   // lds_off ascends with t
                            d = st_doc[at];
                        }
                        // staged position of the doc in every term that holds it (first 8 terms in
                        // registers -- static indexing only --, the rest searched again when needed)
                        int wf[8];
                        if (mode == 2 && nt <= 4) {
                            // the (up to four) lower-bound searches advance in lockstep, branch-free: four
                            // independent LDS reads per step instead of four chains one after the other
                            int lo_[4], hi_[4], base_[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                lo_[e] = 0;
                                hi_[e] = e < nt ? tr[e].sub : 0;
                                base_[e] = e < nt ? tr[e].lds_off : 0;
                            }
                            int span = 0;
#pragma unroll
                            for (int e = 0; e < 4; ++e) span = hi_[e] > span ? hi_[e] : span;
#pragma unroll 1
                            for (; span > 0; span >>= 1) {
                                int mid[4];
                                int32_t v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    mid[e] = (lo_[e] + hi_[e]) >> 1;
                                    v[e] = st_doc[base_[e] + (lo_[e] < hi_[e] ? mid[e] : 0)];
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const bool live = lo_[e] < hi_[e], right = v[e] < d;
                                    lo_[e] = live && right ? mid[e] + 1 : lo_[e];
                                    hi_[e] = live && !right ? mid[e] : hi_[e];
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) wf[e] = -1;
#pragma unroll
                            for (int e = 0; e < 4; ++e)   // lo = the lower bound: the doc is there iff it equals d
                                if (e < nt && lo_[e] < tr[e].sub && st_doc[base_[e] + lo_[e]] == d) wf[e] = lo_[e];
                        } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            wf[e] = -1;
                            if (e >= t && e < nt && ((has >> e) & 1u))
                                wf[e] = (mode != 2 && e == t) ? at - tr[e].lds_off : find_doc(st_doc + tr[e].lds_off, tr[e].sub, d);
                        }
                        }
                        auto where_far = [&](int e) -> int64_t {   // e >= 8
                            if (!((has >> e) & 1u)) return -1;
                            const int f = e == t ? at - tr[e].lds_off : find_doc(st_doc + tr[e].lds_off, tr[e].sub, d);
                            return f >= 0 ? tr[e].lo + tr[e].cur + f : -1;
                        };
                        if (have_theta && block_ub && mode != 2) {   // (the impact bound is the tighter one)
                            double ub2 = 0.0;
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (wf[e] >= 0) ub2 = __dadd_rn(ub2, block_ub[(tr[e].lo + tr[e].cur + wf[e]) / BM_BLOCK]);
                            for (int e = 8 > t ? 8 : t; e < nt; ++e) {
                                const int64_t w = where_far(e);
                                if (w >= 0) ub2 = __dadd_rn(ub2, block_ub[w / BM_BLOCK]);
                            }
                            if (pruned(ub2)) keep = false;
                        }
                        if (keep && qc != -1 && doc_coll[d] != qc) keep = false;
                        if (keep) {
                            const double dl = (double)doclen[d];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                if (DP && e < nt && t_row[e] >= 0) dense_add(e, d, dl, score);
                                else if (wf[e] >= 0)
                                    score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)post_tf[tr[e].lo + tr[e].cur + wf[e]], dl, avgdl, k1, b));
                            }
                            for (int e = 8 > t ? 8 : t; e < nt; ++e) {
                                const int64_t w = where_far(e);
                                if (w >= 0)
                                    score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)post_tf[w], dl, avgdl, k1, b));
                            }
                        }
                    }
                    push(keep, score, (int64_t)d);
                }
            };

            // ---- phase 1 (LDS only): owners, which terms hold the doc, bound against theta ----
            const bool middle = !masked && have_theta && (last - first) < 32 * (int64_t)total;
            if (masked || middle) {
                const int w = masked ? (int)(last - first) : 1;
                if (masked && use_acc) {
                    const int words = (w + 1) >> 1;   // two 16-bit accumulators per word
                    for (int i = threadIdx.x; i < words; i += BM_THREADS) mask[i] = 0u;
                    __syncthreads();
                    // term by term: everything that depends on the term is uniform (scalar registers),
                    // a posting costs one LDS read, one byte from global memory and one LDS atomic
                    for (int t = 0; t < nt; ++t) {
                        const int sub = __builtin_amdgcn_readfirstlane(tr[t].sub);
                        const int off0 = __builtin_amdgcn_readfirstlane(tr[t].lds_off);
                        const uint32_t wt = (uint32_t)__builtin_amdgcn_readfirstlane(t_w[t]);
                        // (staging the impacts in LDS with the ids -- bytes, or aligned words -- was measured:
                        // what the fill gains the staging loses)
                        const uint8_t* imp_t = post_imp + tr[t].lo + tr[t].cur;
                        for (int i0 = threadIdx.x; i0 < sub; i0 += 4 * BM_THREADS) {
                            int slot[4];
                            uint32_t val[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int i = i0 + u * BM_THREADS;
                                slot[u] = -1;
                                if (i < sub) {
                                    slot[u] = (int)(st_doc[off0 + i] - first);
                                    val[u] = (uint32_t)imp_t[i] * wt;
                                }
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (slot[u] >= 0) atomicAdd(&mask[slot[u] >> 1], val[u] << ((slot[u] & 1) << 4));
                        }
                    }
                    __syncthreads();
                    BM_STAMP(12);
                    BM_COUNT(14, 1);
                    BM_COUNT(16, total);
                } else if (masked) {
                    const int words = (w + spw - 1) >> ms;
                    for (int i = threadIdx.x; i < words; i += BM_THREADS) mask[i] = 0u;
                    __syncthreads();
                    for (int i = threadIdx.x; i < total; i += BM_THREADS) {
                        int t = 0;
                        while (i >= t_prefix[t + 1]) ++t;
                        const int slot = (int)(st_doc[tr[t].lds_off + (i - t_prefix[t])] - first);
                        atomicOr(&mask[slot >> ms], 1u << (((slot & (spw - 1)) << 3) + t));
                    }
                    __syncthreads();
                    BM_STAMP(12);
                    BM_COUNT(15, 1);
                    BM_COUNT(16, total);
                }
                // masked: the candidate slots, BM_WINDOW slots (= the survivor list's capacity) at a
                // time; else one round over the staged postings
                // the whole window at once when its survivors fit the list (BM_WINDOW slots: they do
                // once a threshold prunes), else BM_WINDOW slots at a time
                int c_step = masked ? w : SURV_CAP;
                for (int c0 = 0; c0 < w;) {
                    const int c_next = c0 + c_step < w ? c0 + c_step : w;
                    if (masked && use_acc) {
                        const int cend = c_next;
                        for (int wd = (c0 >> 1) + (int)threadIdx.x; wd < ((cend + 1) >> 1); wd += BM_THREADS) {
                            const uint32_t v = mask[wd];
                            if (!v) continue;
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const uint32_t a = (v >> (u << 4)) & 0xFFFFu;
                                if (a != 0u && a + dmaxq >= thq) {
                                    const int at = atomicAdd(&n_surv, 1);
                                    if (at < SURV_CAP) surv[at] = (uint16_t)((wd << 1) + u);
                                }
                            }
                        }
                    } else if (masked) {
                        const int cend = c_next;
                        for (int wd = (c0 >> ms) + (int)threadIdx.x; wd < ((cend + spw - 1) >> ms); wd += BM_THREADS) {
                            const uint32_t v = mask[wd];
                            if (!v) continue;
                            for (int u = 0; u < spw; ++u) {
                                const uint32_t m = ms ? (v >> (u << 3)) & 0xFFu : v;
                                if (!m) continue;
                                if (conjunctive && __popc(m) < nt) continue;
                                if (have_theta) {
                                    double ub = 0.0;
                                    for (uint32_t r = m; r; r &= r - 1) ub = __dadd_rn(ub, t_ub[__ffs((int)r) - 1]);
                                    if (pruned(ub)) continue;
                                }
                                const int at = atomicAdd(&n_surv, 1);
                                if (at < SURV_CAP) surv[at] = (uint16_t)((wd << ms) + u);
                            }
                        }
                    } else {
                        // moderately dense lists and a threshold to prune with: LDS-only owner / bound
                        // search (a doc is dropped on the sum of its terms' bounds before any gather;
                        // the sweep below would find most docs shared and score them all)
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
                            if (pruned(ub)) continue;
                            surv[atomicAdd(&n_surv, 1)] = (uint16_t)(tr[t].lds_off + off);
                        }
                    }
                    __syncthreads();
                    BM_STAMP(13);
                    const int ns = n_surv;
                    if (masked && ns > SURV_CAP) {   // (only with c_step == w) too many: again, chunk by chunk
                        __syncthreads();
                        if (threadIdx.x == 0) n_surv = 0;
                        __syncthreads();
                        c_step = SURV_CAP;
                        continue;
                    }
                    BM_COUNT(17, ns);
                    BM_COUNT(18, (ns + BM_THREADS - 1) / BM_THREADS);
                    phase2(masked ? (use_acc ? 2 : 1) : 0, ns);
                    __syncthreads();
                    if (threadIdx.x == 0) n_surv = 0;
                    __syncthreads();
                    BM_STAMP(5);
                    c0 = c_next;
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
                BM_STAMP(6);
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
                    int wf[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        wf[e] = -1;
                        if (e >= t && e < nt) {
                            wf[e] = e == t ? off : lookup(e, d);
                            if (wf[e] >= 0) ++present;
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
                    for (int e = 0; e < 8; ++e) {
                        if (DP && e < nt && t_row[e] >= 0) dense_add(e, d, dl, score);
                        else if (wf[e] >= 0)
                            score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)(e == t ? tf_own : post_tf[tr[e].lo + tr[e].cur + wf[e]]), dl, avgdl, k1, b));
                    }
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
                static_assert(2 * SCR_WORDS >= BM_STAGE, "work list of a pass without the filter");
                uint16_t* work = reinterpret_cast<uint16_t*>(bloom ? scratch + nt * bwords : scratch);
                // (n_surv is 0 here: it counts the work list now)
                if (bloom) {
                    // list by list: everything that depends on the term is uniform (scalar registers).
                    // Nothing is scored in this sweep: a posting that is its doc's only one is held
                    // against the threshold with its own quantised impact (plus, DP, the dense terms'
                    // largest) and, when it may enter, listed -- from the BACK of the work list's
                    // buffer, the postings to be searched from the front (together at most ``total``).
                    auto list_term = [&](int t, bool use_q, uint32_t thq_now) {
                        const int sub = __builtin_amdgcn_readfirstlane(tr[t].sub);
                        const int off0 = __builtin_amdgcn_readfirstlane(tr[t].lds_off);
                        const int pre = __builtin_amdgcn_readfirstlane(t_prefix[t]);
                        const uint32_t wt = use_q ? (uint32_t)__builtin_amdgcn_readfirstlane(t_w[t]) : 0u;
                        const uint8_t* imp_t = post_imp + tr[t].lo + tr[t].cur;
                        const double ub_t = t_ub[t];
                        const double ubx_t = DP ? (ub_t + dub) * (1.0 + 1e-12) : ub_t;
                        // (ub_t < threshold: no posting of this list can enter on its own)
                        const bool single_ok = !(conjunctive && nt > 1) && !(ubx_t < th_s) && !(ubx_t < thg);
                        for (int base = 0; base < sub; base += BM_THREADS) {
                            const int i = base + (int)threadIdx.x;
                            if (i < sub) {
                                const int32_t d = st_doc[off0 + i];
                                const uint32_t h = ((uint32_t)d * 2654435761u) >> (32 - bl2);
                                const uint32_t w = h >> 5, bit = 1u << (h & 31);
                                bool alone = true;
                                for (int e = 0; e < nt; ++e)
                                    if (e != t && (scratch[e * bwords + w] & bit)) alone = false;
                                if (!alone) {
                                    work[atomicAdd(&n_surv, 1)] = (uint16_t)(pre + i);
                                } else if (single_ok && (!use_q || (uint32_t)imp_t[i] * wt + dmaxq >= thq_now)) {
                                    work[total - 1 - atomicAdd(&n_single, 1)] = (uint16_t)(pre + i);
                                }
                            }
                        }
                    };
                    // the listed singles [from, to), densely: collection filter, gathers, score, push
                    auto score_singles = [&](int from, int to) {
                        for (int base = from; base < to; base += BM_THREADS) {
                            const int j = base + (int)threadIdx.x;
                            bool owner = j < to;
                            double score = 0.0;
                            int32_t d = 0;
                            if (owner) {
                                const int idx = work[total - 1 - j];
                                int t = 0;
                                while (idx >= t_prefix[t + 1]) ++t;
                                const int off = idx - t_prefix[t];
                                d = st_doc[tr[t].lds_off + off];
                                if (qc != -1 && doc_coll[d] != qc) owner = false;
                                if (owner) {
                                    const double dl = (double)doclen[d];
                                    const double tf_own = (double)post_tf[tr[t].lo + tr[t].cur + off];
                                    if (DP) {   // its own posting and the dense terms, in query-term order
#pragma unroll
                                        for (int e = 0; e < 8; ++e) {
                                            if (e >= nt) continue;
                                            if (t_row[e] >= 0) dense_add(e, d, dl, score);
                                            else if (e == t) score = __dadd_rn(score, bm25_contrib(t_idf[e], tf_own, dl, avgdl, k1, b));
                                        }
                                    } else {
                                        score = __dadd_rn(score, bm25_contrib(t_idf[t], tf_own, dl, avgdl, k1, b));
                                    }
                                }
                            }
                            push(owner, score, (int64_t)d);
                        }
                    };
#if defined(BM_BOOT_NONE)
                    const bool boot = false;
#elif defined(BM_BOOT_ALL)
                    const bool boot = acc_ok && !have_theta && nt > 1;
#else
                    const bool boot = !DP && acc_ok && !have_theta && nt > 1;
#endif
                    if (!boot) {
                        // (use_acc: a threshold in accumulator units exists, p_thq)
                        for (int t = 0; t < nt; ++t) list_term(t, use_acc, thq);
                        __syncthreads();
                        BM_STAMP(7);
                        BM_COUNT(19, n_single);
                        score_singles(0, n_single);
                        BM_STAMP(8);
                    } else {
                        // No threshold yet (an item's first pass -- the only one of a short query): list by
                        // list, the largest bound first, and a select after each, so that the later lists
                        // (smaller bounds: the longer ones) are held against a threshold already.
                        int done = 0;
                        for (int oi = 0; oi < nt; ++oi) {
                            list_term(t_order[oi], p_boot_q != 0, (uint32_t)p_thq);
                            __syncthreads();
                            BM_STAMP(7);
                            const int upto = n_single;
                            BM_COUNT(19, upto - done);
                            score_singles(done, upto);
                            done = upto;
                            __syncthreads();
                            BM_STAMP(8);
                            if (b_cnt >= k && b_cnt - last_compact >= 64) {
                                tk.compact();
                                if (threadIdx.x == 0) {
                                    last_compact = b_cnt;
                                    if (S > 1 && th_s > -INFINITY) atomicMax(&theta_glob[q], (unsigned long long)dkey(th_s));
                                }
                            }
                            if (threadIdx.x == 0 && b_cnt >= k && th_s > -INFINITY) {
                                const double th = th_glob > th_s ? th_glob : th_s;
                                const double tq = floor(th * acc_scale * (1.0 - 1e-12));
                                p_thq = tq < 0.0 ? 0 : tq > 70000.0 ? 70000 : (int)tq;
                                p_boot_q = 1;
                            }
                            __syncthreads();
                        }
                    }
                } else {   // no room for the bits (many terms): every posting takes the searching sweep
                    for (int i = threadIdx.x; i < total; i += BM_THREADS) work[i] = (uint16_t)i;
                    if (threadIdx.x == 0) n_surv = total;
                }
                __syncthreads();
                const int n_work = n_surv;
                fetch((int)threadIdx.x < n_work ? (int)work[threadIdx.x] : -1);
                for (int base = 0; base < n_work; base += BM_THREADS) {
                    const int j = base + threadIdx.x;
                    const int t = n_t, off = n_off;
                    const int32_t d = n_d, tf_own = n_tf;
                    const float dl_own = n_dl;
                    fetch(j + BM_THREADS < n_work ? (int)work[j + BM_THREADS] : -1);
                    bool owner = false;
                    double score = 0.0;
                    if (j < n_work) owner = score_full(t, off, d, dl_own, tf_own, score);
                    push(owner, score, (int64_t)d);
                }
            }
            BM_STAMP(9);
            __syncthreads();
            // a fresh theta pays for the select once enough docs have entered since the last one
            if (b_cnt >= k && b_cnt - last_compact >= 64) {
                tk.compact();
                if (threadIdx.x == 0) {
                    last_compact = b_cnt;
                    // a lower bound of this slice's k-th best bounds the query's k-th best from below
                    if (S > 1 && th_s > -INFINITY) atomicMax(&theta_glob[q], (unsigned long long)dkey(th_s));
                }
            }
            if (threadIdx.x == 0) {
                for (int t = 0; t < nt; ++t) tr[t].cur += tr[t].sub;
                remaining -= total;
            }
            __syncthreads();
            BM_STAMP(10);
        }
        const int n = tk.finish();
        if (S == 1) {
            for (int i = threadIdx.x; i < k; i += BM_THREADS) {
                out_s[(int64_t)q * k + i] = i < n ? b_s[i] : -INFINITY;
                out_id[(int64_t)q * k + i] = i < n ? b_id[i] + id_base : -1;
            }
            if (threadIdx.x == 0) out_cnt[q] = n;
        } else {
            for (int i = threadIdx.x; i < n; i += BM_THREADS) {
                slice_s[(int64_t)item * k + i] = b_s[i];
                slice_id[(int64_t)item * k + i] = b_id[i] + id_base;
            }
            if (threadIdx.x == 0) {
                slice_cnt[item] = n;
                if (n >= k) atomicMax(&theta_glob[q], (unsigned long long)dkey(b_s[k - 1]));
            }
        }
        BM_STAMP(11);
#ifdef BM_STAMPS
        ++stamp_items;
        if (threadIdx.x == 0 && walk_log) {
            int tot_ = 0;
            for (int t = 0; t < nt; ++t) tot_ += tr[t].len;
            walk_log[4 * (size_t)item] = ((unsigned long long)q << 32) | (unsigned)((sl << 16) | ((DP ? 1 : 0) << 8) | nt);
            walk_log[4 * (size_t)item + 1] = (unsigned long long)tot_;
            walk_log[4 * (size_t)item + 2] = __builtin_readcyclecounter() - item_t0;
            walk_log[4 * (size_t)item + 3] = (unsigned long long)item_passes;
        }
#endif
    }
#ifdef BM_STAMPS
    if (threadIdx.x == 0) {
        for (int i = 0; i < BM_NSTAMP; ++i) stamps[(int64_t)blockIdx.x * (BM_NSTAMP + 1) + i] = stamp_acc[i];
        stamps[(int64_t)blockIdx.x * (BM_NSTAMP + 1) + BM_NSTAMP] = stamp_items;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// bm25_walk_wave_kernel: stage A (the walked terms of a query with probed terms) with a WAVE, not a
// workgroup, per work item.
//
// Why (round 4): the block walk above gives an item to 512 threads that run a chain of
// barrier-separated phases, each a dependent memory round trip; a stage-A item is ~3 K postings,
// its fixed latency ~60 us, two workgroups fit a CU -- the kernel's waves wait 82 % of their life
// and 40 % of its cycles are per-item latency.  The work itself is embarrassingly parallel over
// items, so the way to hide a latency chain is MORE CHAINS PER CU, not more threads per chain:
// here an item is a doc-range slice of ~1 K walked postings (the plan cuts stage A with its own,
// smaller target), ONE wave walks it without a single workgroup barrier, and sixteen waves --
// sixteen independent chains -- share a CU (10 KiB of LDS and <= 128 VGPRs per wave).
//
// Per pass (an item is one pass unless a doc-range slice came out longer than the stage):
//   stage     the next ids of every walked list into the wave's LDS (equal quotas; d_hi = the
//             smallest "last staged doc + 1" among lists with more behind: every posting below
//             d_hi of every list is on chip);
//   bits      a Bloom bit per (list, doc) when more than one list has postings;
//   classify  a posting whose doc shows in no other list's bits is its doc's only one: held against
//             the threshold with its own quantised impact + the probed terms' largest, listed when it
//             may enter; the others go to a work list;
//   score     listed singles 64 at a time: doc length, own tf, the probed terms' frequencies from
//             their rows, float64 in query-term order, wave-level top-k (128 slots, bitonic cut);
//             as soon as k docs are in, the cut gives a threshold and the classification goes on
//             with it; work-list postings find their owner and the other lists' positions by
//             binary search in LDS.
// Same arithmetic, same bounds (bm25_topk_kernel's accumulator units), same threshold sharing
// (theta_glob) and slice lists as the block walk: results are the same bits.  k <= 64.
// ---------------------------------------------------------------------------------------------
constexpr int WW_WAVES = 4;        // waves per workgroup (independent: they never synchronise)
constexpr int WW_STAGE = 1024;     // doc ids a wave stages per pass
constexpr int WW_CAP = 128;        // top-k slots of a wave (k <= 64: a batch of 64 always fits after a cut)
constexpr int WW_BLOOM = 256;      // words of Bloom bits per wave, shared out among the lists with postings
constexpr int BM_WAVE_ITEMS = 16384;   // item slots for the waves' slices, on top of the list's capacity

struct WwLds {
    int32_t st_doc[WW_STAGE];
    uint8_t st_imp[WW_STAGE];      // the staged postings' quantised impacts (the bound test never leaves LDS)
    uint16_t list[WW_STAGE];       // singles to score from the front, postings to search from the back
    double b_s[WW_CAP];
    int32_t b_id[WW_CAP];
    uint32_t bloom[WW_BLOOM];
    int64_t t_lo[8];               // first posting of the term's slice
    int64_t t_row[8];              // probed term: offset of its per-doc row; else -1
    double t_idf[8];
    int t_len[8], t_cur[8], t_sub[8], t_off[8], t_w[8], t_stg[8], t_bs[8];
};

__device__ __forceinline__ void ww_sync() {
    // lanes of ONE wave exchange data through LDS: the hardware keeps a wave's LDS accesses in
    // order; this keeps the compiler from moving or caching them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// descending (score, then ascending id) bitonic sort of the WW_CAP slots by one wave: two slots per lane
__device__ __forceinline__ void ww_sort(double* s, int32_t* id, int lane) {
    for (int k2 = 2; k2 <= WW_CAP; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const int i = ((lane & ~(j - 1)) << 1) | (lane & (j - 1)), p = i | j;
            const bool up = (i & k2) == 0;
            const double sa = s[i], sb = s[p];
            const int32_t ia = id[i], ib = id[p];
            const bool swap = up ? better(sb, (int64_t)ib, sa, (int64_t)ia) : better(sa, (int64_t)ia, sb, (int64_t)ib);
            if (swap) {
                s[i] = sb; s[p] = sa;
                id[i] = ib; id[p] = ia;
            }
            ww_sync();
        }
}

__global__ __launch_bounds__(WW_WAVES * 64, 4) void bm25_walk_wave_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const int32_t* __restrict__ post_tf, const float* __restrict__ doclen,
    const double* __restrict__ idf, const double* __restrict__ term_ub,
    const uint8_t* __restrict__ post_imp, const int32_t* __restrict__ dense_slot,
    const uint16_t* __restrict__ dense_tf, int64_t dense_stride, double avgdl, double k1, double b,
    double imp_unit /* (k1 + 1) / 255 */, double imp_per_unit /* 255 / (k1 + 1): the host's divisions, same bits */,
    int64_t id_base, int max_terms, int k, const int32_t* __restrict__ doc_coll,
    const int32_t* __restrict__ query_coll, int32_t* __restrict__ ctl,
    const int32_t* __restrict__ q_nt, const int32_t* __restrict__ q_S, const int32_t* __restrict__ q_SA,
    const int32_t* __restrict__ q_pmask, const int32_t* __restrict__ q_terms,
    const int2* __restrict__ items, const int32_t* __restrict__ ipos, const WwItem* __restrict__ wrec,
    const WwTerm* __restrict__ wterm,
    unsigned long long* __restrict__ theta_glob, double* __restrict__ slice_s,
    int64_t* __restrict__ slice_id, int32_t* __restrict__ slice_cnt, double* __restrict__ out_s,
    int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt
#ifdef BM_STAMPS
    , unsigned long long* __restrict__ wstamps
#endif
    ) {
#ifdef BM_STAMPS
    unsigned long long ws_acc[16] = {0};
    unsigned long long ws_last = __builtin_readcyclecounter();
#define WW_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); ws_acc[i] += now_ - ws_last; ws_last = now_; } while (0)
#define WW_C(i, v) do { ws_acc[i] += (unsigned long long)(v); } while (0)
#else
#define WW_T(i)
#define WW_C(i, v)
#endif
    __shared__ WwLds lds_all[WW_WAVES];
    const int lane = threadIdx.x & 63;
    WwLds& L = lds_all[threadIdx.x >> 6];
    const int n_items = ctl[0];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
        // a wave's first item is its own number (thousands of waves bumping one counter at launch spend
    // tens of microseconds in the L2's atomic unit: an EMPTY launch took 72 us that way); the counter
    // hands out the items behind those
    const int n_waves = (int)gridDim.x * WW_WAVES;
    int item = (int)blockIdx.x * WW_WAVES + (int)(threadIdx.x >> 6);
    bool first = true;
    for (;; first = false) {
        if (!first) {
            if (lane == 0) item = atomicAdd(&ctl[4], 1) + n_waves;
            item = __builtin_amdgcn_readfirstlane(item);
        }
        if (item >= n_items) break;
        const WwItem rec = wrec[item];
        const int q = rec.q, sl = rec.sl, SA = rec.SA;
        if (!(SA >= 0 && sl < SA)) {                    // not a stage-A slice: another kernel's item
            WW_T(7);
            continue;
        }
        const int S = rec.S, nt = rec.nt, pm = rec.pm, qc = rec.qc;
        // ---- the terms (lane t < 8), then everything uniform the passes need ----
        double scale_l = 0.0;
        int dm_l = 0;
        {
            const bool on = lane < nt && lane < 8;
            WwTerm tr_;
            tr_.lo = 0; tr_.idf = 0.0; tr_.ub = 0.0; tr_.row = -1; tr_.len = 0; tr_.pad = 0;
            if (on) tr_ = wterm[(int64_t)item * 8 + lane];
            const bool probed = on && ((pm >> lane) & 1);
            const int64_t lo = tr_.lo;
            const int start = 0, end = tr_.len;
            const double idf_l = tr_.idf;
            const double ubt = tr_.ub;
            const int64_t row_l = tr_.row;
            // integer weights of the quantised impacts, as bm25_topk_kernel computes them
            const double c = imp_unit;
            double sum = idf_l * c;
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, WAVE);   // (lanes 0..7 hold the terms; 8.. hold zeros)
            scale_l = 248.0 / sum;
            int w = on ? (int)ceil(idf_l * c * scale_l) : 0;
            w = on && w < 1 ? 1 : w;
            // the probed terms' largest quantised impacts in accumulator units (their bound / idf in
            // steps of (k1+1)/255, as bm25_bounds_kernel rounds)
            const double im = probed ? (idf_l > 0.0 ? ceil(ubt / idf_l * imp_per_unit) + 1.0 : 255.0) : 0.0;
            dm_l = probed ? w * (im > 255.0 || !(im >= 0.0) ? 255 : (int)im) : 0;
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) dm_l += __shfl_xor(dm_l, o, WAVE);
            if (lane < 8) {
                L.t_lo[lane] = lo + start;
                L.t_len[lane] = (on && !probed) ? end - start : 0;
                L.t_cur[lane] = 0;
                L.t_idf[lane] = idf_l;
                L.t_row[lane] = row_l;
                L.t_w[lane] = w;
            }
        }
        const double acc_scale = __shfl(scale_l, 0, WAVE);
        const uint32_t dmaxq = (uint32_t)__shfl(dm_l, 0, WAVE);
        ww_sync();
        WW_T(0);
        WW_C(10, 1);
        // the wave's top-k
        for (int i = lane; i < WW_CAP; i += 64) {
            L.b_s[i] = -INFINITY;
            L.b_id[i] = INT32_MAX;
        }
        int b_cnt = 0;                 // (uniform)
        double th_s = -INFINITY;       // this item's k-th best so far (exact after a cut)
        int32_t th_id = INT32_MAX;
        double thg = -INFINITY;        // the query's other slices' threshold
        {
            const unsigned long long g0 = __hip_atomic_load(&theta_glob[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g0) thg = dkey_inv(g0);
        }
        int remaining = 0;
        for (int t = 0; t < nt; ++t) remaining += L.t_len[t];
        ww_sync();
        auto thq_now = [&]() -> uint32_t {   // the threshold in accumulator units (0: none yet)
            double th = th_s > thg ? th_s : thg;
            if (!(th > -INFINITY)) return 0u;
            const double tq = floor(th * acc_scale * (1.0 - 1e-12));
            return tq < 0.0 ? 0u : tq > 70000.0 ? 70000u : (uint32_t)tq;
        };
        // cut the buffer back to the best k: exact sort, so the threshold is the k-th best itself
        auto cut = [&]() {
            WW_C(12, 1);
            ww_sync();
            ww_sort(L.b_s, L.b_id, lane);
            if (b_cnt > k) {
                for (int i = k + lane; i < WW_CAP; i += 64) {
                    L.b_s[i] = -INFINITY;
                    L.b_id[i] = INT32_MAX;
                }
                b_cnt = k;
            }
            ww_sync();
            if (b_cnt >= k) {
                th_s = L.b_s[k - 1];
                th_id = L.b_id[k - 1];
                if (lane == 0 && th_s > -INFINITY) atomicMax(&theta_glob[q], (unsigned long long)dkey(th_s));
            }
        };
        auto push = [&](bool ok, double sc, int32_t d) {
            // (precondition: b_cnt <= 64)
            ok = ok && !(sc < thg) && better(sc, (int64_t)d, th_s, (int64_t)th_id);
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int p = b_cnt + __popcll(m & lt_mask);
                L.b_s[p] = sc;
                L.b_id[p] = d;
            }
            b_cnt += __popcll(m);
            if (b_cnt > 64) cut();
        };
        // a probed term's contribution to doc d, 0 when the doc does not hold it
        auto probe_add = [&](int e, int32_t d, double dl, double& score) {
            const int tfd = (int)dense_tf[L.t_row[e] + d];
            if (tfd > 0) score = __dadd_rn(score, bm25_contrib(L.t_idf[e], (double)tfd, dl, avgdl, k1, b));
        };

        while (remaining > 0) {
            // ---- stage: the stage is shared out in proportion to what is left of each list (a slice is
            // cut at docs of its longest list: equal quotas would take a 900 + 100 slice in two passes) ----
            if (lane == 0) {
                int n_live = 0;
                for (int t = 0; t < nt; ++t) n_live += L.t_len[t] - L.t_cur[t] > 0 ? 1 : 0;
                const int spare = WW_STAGE - 16 * n_live;   // every list with postings left gets at least 16 slots
                int off = 0;
                for (int t = 0; t < nt; ++t) {
                    const int rem = L.t_len[t] - L.t_cur[t];
                    int stg = rem > 0 ? 16 + (int)((int64_t)spare * rem / remaining) : 0;
                    stg = stg < rem ? stg : rem;
                    L.t_off[t] = off;
                    L.t_stg[t] = stg;
                    off += stg;
                }
            }
            ww_sync();
            for (int t = 0; t < nt; ++t) {
                const int stg = L.t_stg[t];
                const int32_t* src = post_doc + L.t_lo[t] + L.t_cur[t];
                const uint8_t* srci = post_imp + L.t_lo[t] + L.t_cur[t];
                int32_t* dst = L.st_doc + L.t_off[t];
                uint8_t* dsti = L.st_imp + L.t_off[t];
                for (int i = lane; i < stg; i += 64) {
                    dst[i] = src[i];
                    dsti[i] = srci[i];
                }
            }
            // the other slices' threshold travels with the staging loads
            {
                const unsigned long long g1 = __hip_atomic_load(&theta_glob[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g1) {
                    const double g = dkey_inv(g1);
                    thg = g > thg ? g : thg;
                }
            }
            ww_sync();
            // d_hi: one past the last doc every list has fully staged
            int64_t d_hi = INT64_MAX;
            for (int t = 0; t < nt; ++t) {
                const int stg = L.t_stg[t];
                if (stg > 0 && L.t_cur[t] + stg < L.t_len[t]) {
                    const int64_t e = (int64_t)L.st_doc[L.t_off[t] + stg - 1] + 1;
                    d_hi = e < d_hi ? e : d_hi;
                }
            }
            if (lane < nt) {
                const int stg = L.t_stg[lane];
                L.t_sub[lane] = d_hi == INT64_MAX ? stg : count_below(L.st_doc + L.t_off[lane], stg, d_hi);
            }
            ww_sync();
            int total = 0, live = 0;
            for (int t = 0; t < nt; ++t) {
                total += L.t_sub[t];
                live += L.t_sub[t] > 0 ? 1 : 0;
            }
            WW_T(1);
            WW_C(11, 1);
            WW_C(14, total);
            // ---- Bloom bits (only with postings from more than one list) ----
            const bool bits = live > 1;
            int bwords = WW_BLOOM;   // per list: the largest power of two that fits `live` times
            while (bits && bwords * live > WW_BLOOM) bwords >>= 1;
            const int bl2 = 31 - __clz(bwords * 32);
            if (bits) {
                if (lane == 0) {
                    int s_ = 0;
                    for (int t = 0; t < nt; ++t) L.t_bs[t] = L.t_sub[t] > 0 ? s_++ : -1;
                }
                for (int i = lane; i < WW_BLOOM; i += 64) L.bloom[i] = 0u;
                ww_sync();
                for (int t = 0; t < nt; ++t) {
                    const int sub = L.t_sub[t], off = L.t_off[t], bs = L.t_bs[t];
                    for (int i = lane; i < sub; i += 64) {
                        const uint32_t dd = (uint32_t)L.st_doc[off + i];
                        const uint32_t h = (dd * 2654435761u) >> (32 - bl2), h2 = (dd * 0x85EBCA6Bu + 0x9E3779B9u) >> (32 - bl2);
                        atomicOr(&L.bloom[bs * bwords + (h >> 5)], 1u << (h & 31));
                        atomicOr(&L.bloom[bs * bwords + (h2 >> 5)], 1u << (h2 & 31));
                    }
                }
                ww_sync();
            }
            WW_T(2);
            // ---- classify and score ----
            int n_list = 0, n_work = 0;   // (uniform) singles from the front, postings to search from the back
            // the listed singles [0, n_list): gathers, float64 score, push -- 64 at a time
            auto score_listed = [&]() {
                WW_T(3);
                WW_C(13, (n_list + 63) / 64);
                WW_C(15, n_list);
                for (int base = 0; base < n_list; base += 64) {
                    const int j = base + lane;
                    bool ok = j < n_list;
                    double score = 0.0;
                    int32_t d = 0;
                    if (ok) {
                        const int idx = L.list[j];
                        int t = 0;
                        while (t + 1 < nt && idx >= L.t_off[t + 1]) ++t;   // (offsets ascend with t; an empty list shares the next one's)
                        d = L.st_doc[idx];
                        if (qc != -1 && doc_coll[d] != qc) ok = false;
                        if (ok) {
                            const double dl = (double)doclen[d];
                            const double tf_own = (double)post_tf[L.t_lo[t] + L.t_cur[t] + (idx - L.t_off[t])];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                if (e >= nt) continue;
                                if (L.t_row[e] >= 0) probe_add(e, d, dl, score);
                                else if (e == t) score = __dadd_rn(score, bm25_contrib(L.t_idf[e], tf_own, dl, avgdl, k1, b));
                            }
                        }
                    }
                    push(ok, score, d);
                }
                n_list = 0;
                WW_T(4);
            };
            for (int t = 0; t < nt; ++t) {
                const int sub = L.t_sub[t];
                if (sub == 0) continue;
                const int off = L.t_off[t];
                const uint32_t wt = (uint32_t)L.t_w[t];
                const uint8_t* imp_t = L.st_imp + off;
                for (int base = 0; base < sub; base += 64) {
                    const int i = base + lane;
                    const uint32_t thq = thq_now();
                    bool alone = i < sub, search = false;
                    if (alone) {
                        const int32_t d = L.st_doc[off + i];
                        if (bits) {   // two bits per (list, doc): ~5 % false positives where one bit gave 12 %
                            const uint32_t dd = (uint32_t)d;
                            const uint32_t h = (dd * 2654435761u) >> (32 - bl2), h2 = (dd * 0x85EBCA6Bu + 0x9E3779B9u) >> (32 - bl2);
                            const uint32_t w = h >> 5, bit = 1u << (h & 31), w2 = h2 >> 5, bit2 = 1u << (h2 & 31);
                            for (int e = 0; e < nt; ++e) {
                                const int bs = L.t_bs[e];
                                if (e != t && bs >= 0 && (L.bloom[bs * bwords + w] & bit) && (L.bloom[bs * bwords + w2] & bit2)) search = true;
                            }
                        }
                        alone = !search;
                        if (alone && thq != 0u && (uint32_t)imp_t[i] * wt + dmaxq < thq) alone = false;   // cannot enter
                    }
                    const unsigned long long ma = __ballot(alone), ms = __ballot(search);
                    if (alone) L.list[n_list + __popcll(ma & lt_mask)] = (uint16_t)(off + i);
                    if (search) L.list[WW_STAGE - 1 - (n_work + __popcll(ms & lt_mask))] = (uint16_t)(off + i);
                    n_list += __popcll(ma);
                    n_work += __popcll(ms);
                    // no threshold yet: score what is listed as soon as it can fill the top-k, so that
                    // the rest of the pass is classified against a threshold
                    ww_sync();
                    if (n_list >= 64 && (thq_now() == 0u || n_list + n_work + 64 > WW_STAGE)) score_listed();
                }
            }
            ww_sync();
            score_listed();
            WW_T(3);
            // the postings whose doc may be in another list: owner and positions by search
            for (int base = 0; base < n_work; base += 64) {
                const int j = base + lane;
                bool ok = j < n_work;
                double score = 0.0;
                int32_t d = 0;
                if (ok) {
                    const int idx = L.list[WW_STAGE - 1 - j];
                    int t = 0;
                    while (t + 1 < nt && idx >= L.t_off[t + 1]) ++t;
                    d = L.st_doc[idx];
                    int wf[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        wf[e] = -1;
                        if (e < nt && L.t_sub[e] > 0)
                            wf[e] = e == t ? idx - L.t_off[e] : find_doc(L.st_doc + L.t_off[e], L.t_sub[e], d);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < t && wf[e] >= 0) ok = false;             // an earlier list owns the doc
                    if (ok && qc != -1 && doc_coll[d] != qc) ok = false;
                    if (ok) {
                        const double dl = (double)doclen[d];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (e >= nt) continue;
                            if (L.t_row[e] >= 0) probe_add(e, d, dl, score);
                            else if (wf[e] >= 0)
                                score = __dadd_rn(score, bm25_contrib(L.t_idf[e], (double)post_tf[L.t_lo[e] + L.t_cur[e] + wf[e]], dl, avgdl, k1, b));
                        }
                    }
                }
                push(ok, score, d);
            }
            ww_sync();
            if (lane == 0)
                for (int t = 0; t < nt; ++t) L.t_cur[t] += L.t_sub[t];
            remaining -= total;
            ww_sync();
            WW_T(5);
        }
        // ---- the item's list: sorted best first ----
        ww_sync();
        ww_sort(L.b_s, L.b_id, lane);
        const int n = b_cnt < k ? b_cnt : k;
        // (the output addresses are formed HERE: formed at the top of the item, as the compiler
        // would, they are four registers carried -- spilled -- through the whole walk)
        int q_w = __builtin_amdgcn_readfirstlane(q), item_w = __builtin_amdgcn_readfirstlane(item);
        asm volatile("" : "+s"(q_w), "+s"(item_w));
        if (S == 1) {   // the query's only item: its list is the result
            for (int i = lane; i < k; i += 64) {
                out_s[(int64_t)q_w * k + i] = i < n ? L.b_s[i] : -INFINITY;
                out_id[(int64_t)q_w * k + i] = i < n ? (int64_t)L.b_id[i] + id_base : -1;
            }
            if (lane == 0) out_cnt[q_w] = n;
        } else {
            for (int i = lane; i < n; i += 64) {
                slice_s[(int64_t)item_w * k + i] = L.b_s[i];
                slice_id[(int64_t)item_w * k + i] = (int64_t)L.b_id[i] + id_base;
            }
            if (lane == 0) {
                slice_cnt[item_w] = n;
                if (n >= k) atomicMax(&theta_glob[q], (unsigned long long)dkey(L.b_s[k - 1]));
            }
        }
        ww_sync();
        WW_T(6);
    }
#ifdef BM_STAMPS
    WW_T(8);
    if (lane == 0) {
        const int w = blockIdx.x * WW_WAVES + (threadIdx.x >> 6);
        for (int i = 0; i < 16; ++i) wstamps[(size_t)w * 16 + i] = ws_acc[i];
    }
#endif
#undef WW_T
#undef WW_C
}

// Between stage A and stage B: the sweeps that are still needed.  The docs of a sweep hold none of
// the query's other terms, so a score there is at most the sum of the dense terms' bounds (added
// out of order: hence the margin); stage A is complete, and when that sum stays below its
// threshold no doc of the sweep can enter the top-k -- the query's sweep slices are closed with
// empty lists.  The others are listed for bm25_window_kernel SLICE-MAJOR: the first slice of every
// sweeping query, then the second of every one, ... -- the workgroups of the persistent grid then
// start on different queries, and a query's later slices find the threshold its first one has
// published (query-major, the first 512 items were the six slices of 85 queries, all started
// together and all without a threshold: every doc of their first windows scored in full).
// One workgroup: rank of a query among the sweeping ones by a block scan, no atomics, a
// deterministic list.
constexpr int FILTER_THREADS = 1024;
__global__ __launch_bounds__(FILTER_THREADS) void bm25_sweep_filter_kernel(
    int32_t* __restrict__ ctl, int nq, const int32_t* __restrict__ q_S, const int32_t* __restrict__ q_SA,
    const int32_t* __restrict__ q_item0, const double* __restrict__ q_dub,
    const unsigned long long* __restrict__ theta_glob, int32_t* __restrict__ slice_cnt,
    int32_t* __restrict__ sweep_items) {
    __shared__ int red[FILTER_THREADS];
    const int per = (nq + FILTER_THREADS - 1) / FILTER_THREADS;
    const int q0 = (int)threadIdx.x * per < nq ? (int)threadIdx.x * per : nq;
    const int q1 = q0 + per < nq ? q0 + per : nq;
    auto item_of = [&](int q, int s) -> int { return s == 0 ? q : q_item0[q] + s; };
    auto sweeps = [&](int q) -> bool {   // (and closes the slices of a sweep that is ruled out)
        const int SA = q_SA[q];
        if (SA < 0 || SA == q_S[q]) return false;   // not split / walked by waves without a stage B
        const unsigned long long g = theta_glob[q];
        if (g && q_dub[q] * (1.0 + 1e-12) < dkey_inv(g)) {
            // (a threshold exists: the query has stage-A slices, the lists are merged)
            for (int s = SA; s < q_S[q]; ++s) slice_cnt[item_of(q, s)] = 0;
            return false;
        }
        return true;
    };
    int mine = 0;
    unsigned long long live = 0ull;   // (per <= 64 for batches of up to 65536 queries; beyond, recomputed)
    for (int q = q0; q < q1; ++q)
        if (sweeps(q)) {
            ++mine;
            if (q - q0 < 64) live |= 1ull << (q - q0);
        }
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < FILTER_THREADS; o <<= 1) {
        const int v = (int)threadIdx.x >= o ? red[threadIdx.x - o] : 0;
        __syncthreads();
        red[threadIdx.x] += v;
        __syncthreads();
    }
    const int n_sw = red[FILTER_THREADS - 1];
    int rank = red[threadIdx.x] - mine;
    int n_items = 0;
    for (int q = q0; q < q1; ++q) {
        const bool on = q - q0 < 64 ? ((live >> (q - q0)) & 1ull) != 0ull : sweeps(q);
        if (!on) continue;
        const int SA = q_SA[q], SB = q_S[q] - SA;   // (SB is the same for every query of a batch)
        for (int s = 0; s < SB; ++s) sweep_items[(int64_t)s * n_sw + rank] = item_of(q, SA + s);
        n_items = SB;
        ++rank;
    }
    // the number of sweep items: n_sw * SB (any thread with a sweeping query knows SB)
    if (n_items && red[threadIdx.x] == n_sw && mine > 0) ctl[5] = n_sw * n_items;   // (the last thread that holds one)
}

// ---------------------------------------------------------------------------------------------
// bm25_window_kernel: stage B of a query with dense (probed) terms -- the docs of a doc range that
// hold none of the query's walked terms (those were scored by stage A).
// The range is taken in SEGMENTS of up to 256 K docs.  Per segment the walked terms' postings set
// one bit per doc in an LDS bitmap (32 KiB): the docs to leave out.  The segment is then swept in
// windows of up to 64 K docs: the probed terms add their quantised impacts straight from their
// per-doc rows into per-thread registers (coalesced dword loads: 4 docs each, v_perm_b32 +
// v_pk_mad_u16 into two 16-bit sums per word), BW_SCAN docs at a time; a doc whose summed bound
// reaches the threshold survives; phase 2 drops the survivors whose bit is set, reads the others'
// term frequencies from the rows and scores them with the oracle's arithmetic in query-term order.
// Nothing is staged per window and no accumulator is kept in LDS: a pass is the row loads, the
// scan, the (few) survivors and the select.  Same exactness argument as the accumulator path (the
// bound is >= acc_scale * score), same top-k / threshold sharing / slice merge as bm25_topk_kernel.
template <int BW_THREADS, int BW_SCAN, int BW_CAP>
__global__ __launch_bounds__(BW_THREADS, 4) void bm25_window_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ post_doc,
    const float* __restrict__ doclen, const double* __restrict__ idf,
    const int32_t* __restrict__ dense_slot, const uint8_t* __restrict__ dense_imp,
    const uint16_t* __restrict__ dense_tf, int64_t dense_stride, double avgdl, double k1, double b,
    int64_t n_docs, int64_t id_base, int max_terms, int k, const int32_t* __restrict__ doc_coll,
    const int32_t* __restrict__ query_coll, int32_t* __restrict__ ctl,
    const int32_t* __restrict__ q_nt, const int32_t* __restrict__ q_S, const int32_t* __restrict__ q_SA,
    const int32_t* __restrict__ q_pmask,
    const int32_t* __restrict__ q_terms, const int2* __restrict__ items, const int32_t* __restrict__ sweep_items,
    const int32_t* __restrict__ ipos, unsigned long long* __restrict__ theta_glob,
    double* __restrict__ slice_s, int64_t* __restrict__ slice_id, int32_t* __restrict__ slice_cnt,
    double* __restrict__ out_s, int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt
#ifdef BM_STAMPS
    , unsigned long long* __restrict__ stamps
#endif
    ) {
#ifdef BM_STAMPS
    unsigned long long stamp_acc[BM_NSTAMP] = {0};
    unsigned long long stamp_last = __builtin_readcyclecounter(), stamp_items = 0;
    unsigned long long* item_log = stamps + (size_t)2 * 4096 * (BM_NSTAMP + 1);   // behind the three stamp areas: 4 words per sweep item
#endif
    constexpr int BIT_WORDS = 8192;                   // the segment's bitmap: 256 K docs
    constexpr int SEG_DOCS = BIT_WORDS * 32;
    constexpr int QPT = BW_SCAN / 4 / BW_THREADS;     // dwords of a dense row per thread and scan step (4 docs each)
    constexpr int SURV_CAP = 4096;
    constexpr int CHUNK = 4 * BW_THREADS;             // walked postings looked at per step of the bitmap fill
    static_assert(QPT * 4 * BW_THREADS == BW_SCAN && BW_PAD % BW_SCAN == 0 && BW_SCAN % SURV_CAP == 0 &&
                  BW_PAD <= 65536 && SEG_DOCS % BW_PAD == 0, "window shape");
    static_assert(BW_CAP >= THR_TOPK_MAX + BW_THREADS, "top-k buffer");
    __shared__ TermRange tr[8];      // walked terms: .lo / .len = the slice's postings, .cur = consumed by earlier segments
    __shared__ double t_idf[8];
    __shared__ int64_t t_row[8];     // probed term: offset of its per-doc row; else -1
    __shared__ int t_w[8], p_w[8];
    __shared__ int64_t p_row[8];     // the probed terms' rows and weights, compactly
    __shared__ double acc_scale, th_glob;
    __shared__ int p_thq, p_wmax, n_surv, cur_item, last_compact, chunk_cnt;
    __shared__ double b_s[BW_CAP];
    __shared__ int64_t b_id[BW_CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    __shared__ uint32_t bits[BIT_WORDS];
    __shared__ uint16_t surv[SURV_CAP];

    const int n_sweeps = ctl[5];   // (bm25_sweep_filter_kernel)
    BlockTopK<BW_CAP, BW_THREADS> tk;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) cur_item = atomicAdd(&ctl[2], 1);
        __syncthreads();
        if (cur_item >= n_sweeps) break;
#ifdef BM_STAMPS
        const unsigned long long item_t0 = __builtin_readcyclecounter();
        int item_passes = 0, item_surv = 0;
#endif
        const int item = sweep_items[cur_item];
        const int2 it = items[item];
        const int q = it.x, sl = it.y;
        const int SA = q_SA[q];
        const int S = q_S[q];              // (> 1: the item writes a slice list and shares the threshold)
        const int nt = q_nt[q];
        const int qc = query_coll ? query_coll[q] : -1;
        const int64_t D0 = bm_window_edge(n_docs, sl - SA, S - SA), D1 = bm_window_edge(n_docs, sl - SA + 1, S - SA);
        if ((int)threadIdx.x < nt) {
            const int slot = threadIdx.x;
            const int term = q_terms[(int64_t)q * max_terms + slot];
            const int64_t lo = rowptr[term];
            const int ds = ((q_pmask[q] >> slot) & 1) ? dense_slot[term] : -1;   // a walked term: its docs are left out
            const int start = ipos[((int64_t)item * max_terms + slot) * 2];
            const int end = ipos[((int64_t)item * max_terms + slot) * 2 + 1];
            tr[slot].lo = lo + start;
            tr[slot].len = ds >= 0 ? 0 : end - start;
            tr[slot].cur = 0;
            t_row[slot] = ds >= 0 ? (int64_t)ds * dense_stride : -1;
            t_idf[slot] = idf[term];
        }
        if (threadIdx.x == 0) {
            last_compact = 0;
            chunk_cnt = 0;
            const unsigned long long g0 = S > 1 ? __hip_atomic_load(&theta_glob[q], __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            th_glob = g0 ? dkey_inv(g0) : -INFINITY;
        }
        BM_STAMP(0);
        tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);   // includes a barrier
        int n_walked = 0, np = 0;
        for (int t = 0; t < nt; ++t) {
            n_walked += t_row[t] < 0 && tr[t].len > 0 ? 1 : 0;
            np += t_row[t] >= 0 ? 1 : 0;
        }
        if (threadIdx.x == 0) {
            // integer weights of the quantised impacts (see bm25_topk_kernel) -- of the PROBED terms
            // only: a doc of the sweep holds no walked term, so its score is the probed terms' alone,
            // and the 248 units go to them.  (Shared out over all the query's terms -- round 3 -- a lone
            // stop word beside three rare walked terms got a weight of ceil(1.6) = 2: a bound 22 % above
            // the score, every doc of the shard "survived" and was scored in full: six items of 1.3 M
            // cycles each in a kernel whose workgroups average 0.64 M -- the sweep kernel's length.)
            const double c = (k1 + 1.0) / 255.0;
            double sum = 0.0;
            for (int t = 0; t < nt; ++t)
                if (t_row[t] >= 0) sum += t_idf[t] * c;
            const double scale = sum > 0.0 ? 248.0 / sum : 1.0;
            for (int t = 0; t < nt; ++t) {
                const int w = t_row[t] >= 0 ? (int)ceil(t_idf[t] * c * scale) : 0;
                t_w[t] = w < 1 ? 1 : w;
            }
            acc_scale = scale;
            int i = 0;   // the probed terms, compactly: row and weight
            for (int t = 0; t < nt; ++t)
                if (t_row[t] >= 0) {
                    p_row[i] = t_row[t];
                    p_w[i++] = t_w[t];
                }
        }
        __syncthreads();
        // what the coming pass needs: its window and its threshold in accumulator units
        auto prepare = [&](int last_w, int last_ns) {
            if (threadIdx.x == 0) {
                const bool have_local = b_cnt >= k && th_s > -INFINITY;
                const bool have_th = have_local || th_glob > -INFINITY;
                double th = have_local ? th_s : -INFINITY;
                th = th_glob > th ? th_glob : th;
                const double tq = have_th ? floor(th * acc_scale * (1.0 - 1e-12)) : 0.0;
                p_thq = tq < 0.0 ? 0 : tq > 70000.0 ? 70000 : (int)tq;
                // without a threshold every doc that holds a term is scored in full: a short window gets
                // one; and a threshold that let more than 1/16 of the last window through is still a poor
                // one (the k docs seen so far need not hold the term that decides the ranking: a stop
                // word with idf 0.01 beside a 2 % term with idf 3.8 had 35 K survivors in the 64 K window
                // that followed the first 2 K one): the window then grows fourfold per pass, not at once
                p_wmax = !(have_th || S == 1) ? 2048
                         : (last_w > 0 && last_ns * 16 > last_w && last_w * 4 < BW_PAD) ? (last_w * 4 > 2048 ? last_w * 4 : 2048)
                                                                                       : BW_PAD;
                n_surv = 0;
            }
        };
        BM_STAMP(1);
        for (int64_t g0 = D0; g0 < D1; g0 += SEG_DOCS) {
            const int64_t g1 = g0 + SEG_DOCS < D1 ? g0 + SEG_DOCS : D1;
            // ---- the segment's docs that hold a walked term: one bit each ----
            if (n_walked > 0) {
                const int nw = (int)((g1 - g0 + 31) >> 5);
                for (int i = threadIdx.x; i < nw; i += BW_THREADS) bits[i] = 0u;
                __syncthreads();
                for (int t = 0; t < nt; ++t) {
                    if (t_row[t] >= 0) continue;
                    for (;;) {   // the list's next postings, CHUNK at a time, up to the segment's end (the list is doc-sorted)
                        const int base = tr[t].cur, rem = tr[t].len - base;
                        if (rem <= 0) break;
                        const int n = rem < CHUNK ? rem : CHUNK;
                        const int32_t* src = post_doc + tr[t].lo + base;
                        int mine = 0;
#pragma unroll
                        for (int u = 0; u < CHUNK / BW_THREADS; ++u) {
                            const int i = u * BW_THREADS + (int)threadIdx.x;
                            if (i < n) {
                                const int64_t d = src[i];
                                if (d < g1) {
                                    const uint32_t bit = (uint32_t)(d - g0);
                                    atomicOr(&bits[bit >> 5], 1u << (bit & 31));
                                    ++mine;
                                }
                            }
                        }
                        if (mine) atomicAdd(&chunk_cnt, mine);
                        __syncthreads();
                        const int c = chunk_cnt;
                        __syncthreads();
                        if (threadIdx.x == 0) {
                            tr[t].cur = base + c;
                            chunk_cnt = 0;
                        }
                        __syncthreads();
                        if (c < n) break;   // the rest of the list belongs to later segments
                    }
                }
            }
            prepare(0, 0);
            __syncthreads();
            BM_STAMP(3);
            int64_t cursor = g0;
            while (cursor < g1) {
                const int wmax = p_wmax;
                const int64_t end = cursor + wmax < g1 ? cursor + wmax : g1;   // (g0, the window widths: multiples of 4)
                const int w = (int)(end - cursor);
                const double thg = th_glob;
                auto push = [&](bool ok, double sc, int64_t d) { tk.push(ok && !(sc < thg), sc, d); };
                // (the other slices' threshold for the NEXT pass: requested now, read in the tail)
                unsigned long long gth = 0ull;
                if (threadIdx.x == 0 && S > 1)
                    gth = __hip_atomic_load(&theta_glob[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ---- probed terms: 4 docs per load, straight into registers ----
                auto load_rows = [&](uint32_t (&v)[QPT], int i, int h0) {   // row i, docs [h0, h0 + BW_SCAN) of the window
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(dense_imp + p_row[i] + cursor + h0);   // a multiple of 4
#pragma unroll
                    for (int j = 0; j < QPT; ++j) {
                        const int dw = j * BW_THREADS + (int)threadIdx.x;
                        v[j] = h0 + 4 * dw < w ? src[dw] : 0u;
                    }
                };
                // four impact bytes -> two words of two 16-bit sums: v_perm_b32 spreads the bytes,
                // v_pk_mad_u16 multiplies both lanes by the weight and adds (a sum stays below 2^16)
                auto add_rows = [&](uint32_t (&dsum)[2 * QPT], const uint32_t (&v)[QPT], int i) {
                    const unsigned short wt = (unsigned short)p_w[i];
                    const bm_u16x2 w2 = {wt, wt};
#pragma unroll
                    for (int j = 0; j < QPT; ++j) {
                        const bm_u16x2 lo = __builtin_bit_cast(bm_u16x2, __builtin_amdgcn_perm(0u, v[j], 0x0c010c00u));
                        const bm_u16x2 hi = __builtin_bit_cast(bm_u16x2, __builtin_amdgcn_perm(0u, v[j], 0x0c030c02u));
                        dsum[2 * j] = __builtin_bit_cast(uint32_t, (bm_u16x2)(lo * w2 + __builtin_bit_cast(bm_u16x2, dsum[2 * j])));
                        dsum[2 * j + 1] = __builtin_bit_cast(uint32_t, (bm_u16x2)(hi * w2 + __builtin_bit_cast(bm_u16x2, dsum[2 * j + 1])));
                    }
                };
                auto dense_sums = [&](uint32_t (&dsum)[2 * QPT], int h0) {
#pragma unroll
                    for (int j = 0; j < 2 * QPT; ++j) dsum[j] = 0u;
                    for (int i = 0; i < np; ++i) {
                        uint32_t v[QPT];
                        load_rows(v, i, h0);
                        add_rows(dsum, v, i);
                    }
                };
                // ---- scan: the docs whose bound reaches the threshold ----
                const uint32_t thq = (uint32_t)p_thq;
                auto scan = [&](int c0, int c1, const uint32_t (&dsum)[2 * QPT], int h0) {
#pragma unroll
                    for (int j = 0; j < 2 * QPT; ++j) {
                        const int s0 = h0 + 4 * ((j >> 1) * BW_THREADS + (int)threadIdx.x) + 2 * (j & 1);
                        if (s0 >= c1 || s0 + 1 < c0 || s0 >= w) continue;
                        const uint32_t v = dsum[j];
                        if ((v & 0xFFFFu) < thq && (v >> 16) < thq) continue;   // (nearly every word)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const uint32_t a = (v >> (u << 4)) & 0xFFFFu;
                            const int slot = s0 + u;
                            if (a != 0u && a >= thq && slot >= c0 && slot < c1 && slot < w) {
                                const int at = atomicAdd(&n_surv, 1);
                                if (at < SURV_CAP) surv[at] = (uint16_t)slot;
                            }
                        }
                    }
                };
                auto phase2 = [&](int ns) {
                    for (int base = 0; base < ns; base += BW_THREADS) {
                        const int j = base + threadIdx.x;
                        bool keep = j < ns;
                        double score = 0.0;
                        int32_t d = 0;
                        if (keep) {
                            d = (int32_t)(cursor + surv[j]);
                            if (n_walked > 0) {   // (a doc that holds a walked term was scored by stage A)
                                const uint32_t bit = (uint32_t)(d - g0);
                                if ((bits[bit >> 5] >> (bit & 31)) & 1u) keep = false;
                            }
                            if (keep && qc != -1 && doc_coll[d] != qc) keep = false;
                            if (keep) {
                                const double dl = (double)doclen[d];
                                int tfv[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    tfv[e] = 0;
                                    if (e < nt) {
                                        const int64_t row = t_row[e];
                                        if (row >= 0) tfv[e] = (int)dense_tf[row + d];
                                    }
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    if (tfv[e] > 0)
                                        score = __dadd_rn(score, bm25_contrib(t_idf[e], (double)tfv[e], dl, avgdl, k1, b));
                            }
                        }
                        push(keep, score, (int64_t)d);
                    }
                };
                BM_COUNT(14, 1);
                BM_COUNT(16, w);
                // BW_SCAN docs per step; the rows of the first PF probed terms for the NEXT step are
                // requested before this step's sums and scan (a step is otherwise one round trip long)
                constexpr int PF = 4;
                uint32_t nxt[PF][QPT];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (i < np) load_rows(nxt[i], i, 0);
#pragma unroll 1
                for (int h0 = 0; h0 < w; h0 += BW_SCAN) {
                    uint32_t cur[PF][QPT];
#pragma unroll
                    for (int i = 0; i < PF; ++i)
#pragma unroll
                        for (int j = 0; j < QPT; ++j) cur[i][j] = nxt[i][j];
                    if (h0 + BW_SCAN < w) {
#pragma unroll
                        for (int i = 0; i < PF; ++i)
                            if (i < np) load_rows(nxt[i], i, h0 + BW_SCAN);
                    }
                    uint32_t dsum[2 * QPT];
#pragma unroll
                    for (int j = 0; j < 2 * QPT; ++j) dsum[j] = 0u;
#pragma unroll
                    for (int i = 0; i < PF; ++i)
                        if (i < np) add_rows(dsum, cur[i], i);
                    for (int i = PF; i < np; ++i) {
                        uint32_t v[QPT];
                        load_rows(v, i, h0);
                        add_rows(dsum, v, i);
                    }
                    scan(0, w, dsum, h0);
                }
                __syncthreads();
                BM_STAMP(13);
                const int ns = n_surv;
                BM_COUNT(17, ns);
                BM_COUNT(18, (ns + BW_THREADS - 1) / BW_THREADS);
#ifdef BM_STAMPS
                ++item_passes;
                item_surv += ns;
#endif
                if (ns <= SURV_CAP) {
                    phase2(ns);
                } else {   // (passes without a threshold) SURV_CAP slots at a time
                    for (int c0 = 0; c0 < w; c0 += SURV_CAP) {
                        __syncthreads();
                        if (threadIdx.x == 0) n_surv = 0;
                        __syncthreads();
                        {   // (the sums again: they are not kept across phase 2)
                            const int h0 = c0 / BW_SCAN * BW_SCAN;
                            uint32_t dsum[2 * QPT];
                            dense_sums(dsum, h0);
                            scan(c0, c0 + SURV_CAP, dsum, h0);
                        }
                        __syncthreads();
                        phase2(n_surv);
                    }
                }
                __syncthreads();
                BM_STAMP(5);
                if (b_cnt >= k && b_cnt - last_compact >= 64) {
                    BM_COUNT(15, 1);
                    tk.compact();
                    if (threadIdx.x == 0) {
                        last_compact = b_cnt;
                        if (S > 1 && th_s > -INFINITY) atomicMax(&theta_glob[q], (unsigned long long)dkey(th_s));
                    }
                }
                if (threadIdx.x == 0 && S > 1 && gth) {
                    const double g = dkey_inv(gth);
                    if (g > th_glob) th_glob = g;
                }
                cursor = end;
                prepare(w, ns);
                __syncthreads();
                BM_STAMP(10);
            }
        }
        const int n = tk.finish();
#ifdef BM_STAMPS
        if (threadIdx.x == 0) {
            item_log[4 * (size_t)cur_item] = ((unsigned long long)q << 32) | (unsigned)(((sl - SA) << 16) | (np << 8) | n_walked);
            item_log[4 * (size_t)cur_item + 1] = item_t0;
            item_log[4 * (size_t)cur_item + 2] = __builtin_readcyclecounter();
            item_log[4 * (size_t)cur_item + 3] = ((unsigned long long)item_passes << 32) | (unsigned)item_surv;
        }
#endif
        if (S == 1) {
            for (int i = threadIdx.x; i < k; i += BW_THREADS) {
                out_s[(int64_t)q * k + i] = i < n ? b_s[i] : -INFINITY;
                out_id[(int64_t)q * k + i] = i < n ? b_id[i] + id_base : -1;
            }
            if (threadIdx.x == 0) out_cnt[q] = n;
        } else {
            for (int i = threadIdx.x; i < n; i += BW_THREADS) {
                slice_s[(int64_t)item * k + i] = b_s[i];
                slice_id[(int64_t)item * k + i] = b_id[i] + id_base;
            }
            if (threadIdx.x == 0) {
                slice_cnt[item] = n;
                if (n >= k) atomicMax(&theta_glob[q], (unsigned long long)dkey(b_s[k - 1]));
            }
        }
        BM_STAMP(11);
#ifdef BM_STAMPS
        ++stamp_items;
#endif
    }
#ifdef BM_STAMPS
    if (threadIdx.x == 0) {
        for (int i = 0; i < BM_NSTAMP; ++i) stamps[(int64_t)blockIdx.x * (BM_NSTAMP + 1) + i] = stamp_acc[i];
        stamps[(int64_t)blockIdx.x * (BM_NSTAMP + 1) + BM_NSTAMP] = stamp_items;
    }
#endif
}

// The best k of a sliced query's per-slice lists (order: score desc, id asc -- the slices hold
// disjoint docs, so there are no duplicates to resolve).
constexpr int BMM_THREADS = 256, BMM_CAP = 512;
__global__ __launch_bounds__(BMM_THREADS) void bm25_merge_kernel(
    const int32_t* __restrict__ q_S, const int32_t* __restrict__ q_item0,
    const double* __restrict__ slice_s, const int64_t* __restrict__ slice_id,
    const int32_t* __restrict__ slice_cnt, int k, double* __restrict__ out_s,
    int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt) {
    static_assert(BMM_CAP >= THR_TOPK_MAX + BMM_THREADS, "merge buffer");
    __shared__ double b_s[BMM_CAP];
    __shared__ int64_t b_id[BMM_CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    const int q = blockIdx.x;
    const int S = q_S[q];
    if (S == 1) return;   // written by the item itself
    const int item0 = q_item0[q];
    BlockTopK<BMM_CAP, BMM_THREADS> tk;
    tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);
    for (int base = 0; base < S * k; base += BMM_THREADS) {
        const int idx = base + threadIdx.x;
        const int sl = idx / k, j = idx - sl * k;
        const int item = sl == 0 ? q : item0 + sl;   // (slice 0 is item q: bm25_plan_kernel)
        const bool ok = sl < S && j < slice_cnt[item];
        double sc = 0.0;
        int64_t id = 0;
        if (ok) {
            sc = slice_s[(int64_t)item * k + j];
            id = slice_id[(int64_t)item * k + j];
        }
        tk.push(ok, sc, id);
    }
    const int n = tk.finish();
    for (int i = threadIdx.x; i < k; i += BMM_THREADS) {
        out_s[(int64_t)q * k + i] = i < n ? b_s[i] : -INFINITY;
        out_id[(int64_t)q * k + i] = i < n ? b_id[i] : -1;
    }
    if (threadIdx.x == 0) out_cnt[q] = n;
}

// ---- workspace of thr_bm25_topk ----
struct BmLayout {
    size_t off_ctl, off_theta, off_tot, off_dub, off_sweep, off_nt, off_S, off_SA, off_pmask, off_item0, off_long, off_qterms, off_items,
        off_ipos, off_wrec, off_wterm, off_ss, off_sid, off_scnt, off_stamps, total;
    int cap, cap_base;
};
static BmLayout bm_layout(int nq, int mt, int k) {
    BmLayout L;
    static int extra = 0;
    if (!extra) {
        const char* ev = getenv("THR_BM25_ITEMS");   // item slots beyond two per query (A/B knob)
        extra = ev && atoi(ev) >= 1024 ? atoi(ev) : BM_EXTRA_ITEMS;
    }
    // (a query with dense terms is at least two items: stage A, stage B; the wave walk cuts stage A
    // into ~1 K-posting slices: 16 K more items for them)
    L.cap_base = 2 * nq + extra;
    L.cap = L.cap_base + BM_WAVE_ITEMS;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    L.off_ctl = take(sizeof(int32_t) * 16);                 // [0] items, [1] [2] [4] next item of a kernel, [3] queries with dense terms, [5] sweeps, [6] plan workgroups done, [7] stage-A slice size, [8] queries of the workgroup walk  } zeroed
    L.off_theta = take(sizeof(unsigned long long) * nq);   // shared thresholds (keys)   } per call
    L.off_tot = take(sizeof(int64_t) * nq);
    L.off_dub = take(sizeof(double) * nq);
    L.off_nt = take(sizeof(int32_t) * nq);
    L.off_S = take(sizeof(int32_t) * nq);
    L.off_SA = take(sizeof(int32_t) * nq);
    L.off_pmask = take(sizeof(int32_t) * nq);
    L.off_item0 = take(sizeof(int32_t) * nq);
    L.off_long = take(sizeof(int32_t) * nq);
    L.off_qterms = take(sizeof(int32_t) * (size_t)nq * mt);
    L.off_items = take(sizeof(int2) * (size_t)L.cap);
    L.off_sweep = take(sizeof(int32_t) * (size_t)L.cap);
    L.off_ipos = take(sizeof(int32_t) * 2 * (size_t)L.cap * mt);
    L.off_wrec = take(sizeof(WwItem) * (size_t)L.cap);
    L.off_wterm = take(sizeof(WwTerm) * (size_t)L.cap * 8);
    L.off_ss = take(sizeof(double) * (size_t)L.cap * k);
    L.off_sid = take(sizeof(int64_t) * (size_t)L.cap * k);
    L.off_scnt = take(sizeof(int32_t) * (size_t)L.cap);
#ifdef BM_STAMPS
    L.off_stamps = take(sizeof(unsigned long long) * (3 * 4096 * (BM_NSTAMP + 1) + 8 * (size_t)L.cap));
#endif
    L.total = off;
    return L;
}

static int bm_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace thr

using namespace thr;

extern "C" size_t thr_bm25_block_count(int64_t nnz) { return nnz > 0 ? (size_t)((nnz + BM_BLOCK - 1) / BM_BLOCK) : 0; }

extern "C" int thr_bm25_bounds(const int64_t* rowptr, const int32_t* post_doc, const int32_t* post_tf,
                               const float* doclen, const double* idf, double avgdl, double k1,
                               double b, int64_t n_vocab, int64_t nnz, double* term_ub,
                               double* block_ub, uint8_t* post_imp, thr_stream_t stream) {
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
                       (unsigned long long*)term_ub, (unsigned long long*)block_ub, post_imp);
    hipLaunchKernelGGL(bm25_bounds_decode, dim3((unsigned)((n_vocab + 255) / 256)), dim3(256), 0, st,
                       (unsigned long long*)term_ub, n_vocab);
    hipLaunchKernelGGL(bm25_bounds_decode, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st,
                       (unsigned long long*)block_ub, nb);
    return launch_status();
}

extern "C" int64_t thr_bm25_dense_stride(int64_t n_docs) {
    return n_docs > 0 ? ((n_docs + 15) & ~(int64_t)15) + BW_PAD : 0;
}

extern "C" int thr_bm25_dense_rows(const int64_t* rowptr, const int32_t* post_doc, const int32_t* post_tf,
                                   const uint8_t* post_imp, const int32_t* terms, int n_terms,
                                   int64_t n_docs, int64_t max_df, uint8_t* dense_imp, uint16_t* dense_tf,
                                   thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!rowptr || !post_doc || !post_tf || !post_imp || !terms || !dense_imp || !dense_tf,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_terms <= 0 || n_docs <= 0 || max_df <= 0, THR_ERR_INVALID);
    hipStream_t st = (hipStream_t)stream;
    const int64_t stride = thr_bm25_dense_stride(n_docs);
    hipError_t e = hipMemsetAsync(dense_imp, 0, (size_t)n_terms * stride, st);
    if (e == hipSuccess) e = hipMemsetAsync(dense_tf, 0, sizeof(uint16_t) * (size_t)n_terms * stride, st);
    if (e != hipSuccess) return (int)e;
    int bx = (int)((max_df + 256 * 16 - 1) / (256 * 16));
    bx = bx < 1 ? 1 : bx > 4096 ? 4096 : bx;
    hipLaunchKernelGGL(bm25_dense_rows_kernel, dim3(bx, n_terms), dim3(256), 0, st, rowptr, post_doc, post_tf,
                       post_imp, terms, stride, dense_imp, dense_tf);
    return launch_status();
}

extern "C" size_t thr_bm25_workspace_bytes(int n_queries, int max_terms, int k) {
    if (n_queries <= 0 || max_terms <= 0 || k <= 0) return 0;
    return bm_layout(n_queries, max_terms, k).total;
}

extern "C" int thr_bm25_topk(const int64_t* rowptr, const int32_t* post_doc, const int32_t* post_tf,
                             const float* doclen, const double* idf, const double* term_ub,
                             const double* block_ub, const uint8_t* post_imp, const int32_t* dense_slot,
                             const uint8_t* dense_imp, const uint16_t* dense_tf, int64_t dense_stride,
                             double avgdl, double k1, double b,
                             int64_t n_docs, int64_t n_vocab, int64_t id_base,
                             const int32_t* query_terms, int n_queries, int max_terms, int k,
                             int conjunctive, const int32_t* doc_coll, const int32_t* query_coll,
                             double* out_scores, int64_t* out_ids, int32_t* out_counts,
                             void* workspace, size_t workspace_bytes, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!rowptr || !post_doc || !post_tf || !doclen || !idf || !query_terms ||
                      !out_scores || !out_ids || !out_counts || !workspace,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_docs <= 0 || n_vocab <= 0 || n_queries <= 0 || k <= 0 || k > THR_TOPK_MAX ||
                      max_terms <= 0 || max_terms > THR_BM25_MAX_TERMS || !(avgdl > 0.0),
                  THR_ERR_INVALID);
    THR_RETURN_IF((query_coll != nullptr) != (doc_coll != nullptr), THR_ERR_INVALID);
    // the dense rows come as a set, need the impacts, and are padded by one window
    THR_RETURN_IF(dense_slot && (!dense_imp || !dense_tf || !post_imp || !term_ub ||
                                 dense_stride < n_docs + BW_PAD || (dense_stride & 3)),
                  THR_ERR_INVALID);
    const BmLayout L = bm_layout(n_queries, max_terms, k);
    THR_RETURN_IF(workspace_bytes < L.total, THR_ERR_WORKSPACE);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    int32_t* ctl = (int32_t*)(ws + L.off_ctl);
    unsigned long long* theta = (unsigned long long*)(ws + L.off_theta);
    int64_t* q_tot = (int64_t*)(ws + L.off_tot);
    double* q_dub = (double*)(ws + L.off_dub);
    int32_t* sweep_items = (int32_t*)(ws + L.off_sweep);
    int32_t* q_nt = (int32_t*)(ws + L.off_nt);
    int32_t* q_S = (int32_t*)(ws + L.off_S);
    int32_t* q_SA = (int32_t*)(ws + L.off_SA);
    int32_t* q_pmask = (int32_t*)(ws + L.off_pmask);
    int32_t* q_item0 = (int32_t*)(ws + L.off_item0);
    int32_t* q_long = (int32_t*)(ws + L.off_long);
    int32_t* q_terms = (int32_t*)(ws + L.off_qterms);
    int2* items = (int2*)(ws + L.off_items);
    int32_t* ipos = (int32_t*)(ws + L.off_ipos);
    WwItem* wrec = (WwItem*)(ws + L.off_wrec);
    WwTerm* wterm = (WwTerm*)(ws + L.off_wterm);
    double* slice_s = (double*)(ws + L.off_ss);
    int64_t* slice_id = (int64_t*)(ws + L.off_sid);
    int32_t* slice_cnt = (int32_t*)(ws + L.off_scnt);
    hipError_t e = hipMemsetAsync(ws + L.off_ctl, 0, L.off_tot - L.off_ctl, st);   // ctl + theta
    if (e != hipSuccess) return (int)e;
#ifdef BM_STAMPS
    (void)hipMemsetAsync(ws + L.off_stamps, 0, sizeof(unsigned long long) * (3 * 4096 * (BM_NSTAMP + 1) + 8 * (size_t)L.cap), st);
#endif
    static int small = -1, use_dense = 1, walk_div = 64, fuse_div = 8, use_wave = 1;
    if (small < 0) {
        const char* ei = getenv("THR_BM25_DENSE");    // 0: every term through its postings (A/B knob)
        use_dense = !(ei && ei[0] == '0');
        ei = getenv("THR_BM25_WALK_DIV");             // a term with rows may be walked when held by < 1/this of the docs
        if (ei && atoi(ei) > 0) walk_div = atoi(ei);
        ei = getenv("THR_BM25_WALK");                 // b(lock): stage A on the workgroup walk (the round-3 path; A/B knob)
        use_wave = !(ei && ei[0] == 'b');
        ei = getenv("THR_BM25_FUSE_DIV");             // one launch for ordinary items + stage A from 1/this of the queries (0: never)
        if (ei && atoi(ei) >= 0) fuse_div = atoi(ei);
        const char* ev = getenv("THR_BM25_SHAPE");
        small = (ev && ev[0] == 's') ? 1 : (ev && ev[0] == 'h') ? 2 : 0;   // s(mall) / h(uge)
    }
    const bool big = small == 0, huge = small == 2;
    // persistent grid: as many workgroups as the chip holds at once (never more than items can exist)
    int grid = bm_num_cus() * (huge ? 1 : big ? 2 : 4);
    if (grid > L.cap) grid = L.cap;
    const int32_t* dslot = use_dense ? dense_slot : nullptr;
    // OR queries of <= 8 terms by waves (bm25_walk_wave_kernel) when the impacts are there and k fits a
    // wave's buffer; AND queries, longer ones and calls without bounds keep the workgroup walk
    const bool wave = use_wave && k <= 64 && term_ub != nullptr && post_imp != nullptr && !conjunctive;
    int plan_blocks = (n_queries + PLAN_THREADS - 1) / PLAN_THREADS;
    plan_blocks = plan_blocks > PLAN_MAX_BLOCKS ? PLAN_MAX_BLOCKS : plan_blocks;
    hipLaunchKernelGGL(bm25_plan_kernel, dim3(plan_blocks), dim3(PLAN_THREADS), 0, st, rowptr, n_vocab, query_terms,
                       n_queries, max_terms, L.cap_base, BM_WAVE_ITEMS, conjunctive, grid, BM_TARGET0, wave ? bm_num_cus() * 4 * WW_WAVES : 0, wave ? 1 : 0, walk_div, dslot, term_ub, n_docs, ctl,
                       q_tot, q_dub, q_nt, q_S, q_SA, q_pmask, q_item0, q_long, q_terms, items);
    const int64_t edge_threads = (int64_t)L.cap * max_terms;
    hipLaunchKernelGGL(bm25_edges_kernel, dim3((unsigned)((edge_threads + 255) / 256)), dim3(256), 0, st,
                       rowptr, post_doc, ctl, q_nt, q_S, q_SA, q_long, q_terms, items, max_terms, q_pmask, n_docs, ipos,
                       idf, term_ub, dslot, dense_stride, query_coll, wave ? wrec : (WwItem*)nullptr,
                       wave ? wterm : (WwTerm*)nullptr);
    int rc = launch_status();
    if (rc) return rc;
#ifdef BM_STAMPS
    unsigned long long* d_stamps = (unsigned long long*)(ws + L.off_stamps);
    if (grid > 4096) grid = 4096;
#define BM_STAMP_ARG(DP) , d_stamps + (DP == 1 ? 2 : 0) * (size_t)4096 * (BM_NSTAMP + 1), d_stamps + (size_t)3 * 4096 * (BM_NSTAMP + 1) + 4 * (size_t)L.cap
#else
#define BM_STAMP_ARG(DP)
#endif
    // Block shape: 512 threads / 8192 staged ids per pass / 75 KiB of LDS, two workgroups per CU --
    // a four-term query of the bench (6.7 K postings) is one pass.  THR_BM25_SHAPE=small selects
    // 256 threads / 4096 ids / 39 KiB, four per CU: the fixed cost of an item (set-up, staging, the
    // final sort) overlaps four ways, which wins when every list is short (2048 queries over lists
    // of <= 200 postings: 0.075 ms against 0.124 ms) and loses otherwise.
#define THR_BM25_LAUNCH(T, S, W, C, DP)                                                            \
    hipLaunchKernelGGL((bm25_topk_kernel<T, S, W, C, DP>), dim3(grid), dim3(T), 0, st, rowptr, post_doc, \
                       post_tf, doclen, idf, term_ub, term_ub ? block_ub : nullptr,                 \
                       term_ub ? post_imp : nullptr, dslot, dense_tf, dense_stride,    \
                       avgdl, k1, b, (k1 + 1.0) / 255.0, 255.0 / (k1 + 1.0),                        \
                       id_base, max_terms, k, conjunctive, doc_coll, query_coll, n_queries,         \
                       wave ? -1 : dslot ? fuse_div : 0, ctl, q_nt, q_S,                            \
                       q_SA, q_pmask, q_terms, items, ipos, theta, slice_s, slice_id, slice_cnt,    \
                       out_scores, out_ids, out_counts BM_STAMP_ARG(DP))
#define THR_BM25_LAUNCH_SHAPE(DP)                                                                     \
    do {                                                                                              \
        if (huge) THR_BM25_LAUNCH(1024, 16384, 8192, 2048, DP);   /* one 16-wave workgroup per CU */  \
        else if (big) THR_BM25_LAUNCH(512, 8192, 4096, 1024, DP);                                     \
        else THR_BM25_LAUNCH(256, 4096, 2048, 512, DP);                                               \
    } while (0)
    // The walks.  Wave mode: every OR query of <= 8 terms -- stage A of the ones with probed terms and
    // the ones without -- is the wave kernel's; the workgroup walk further down takes what is left
    // (nothing, usually).  Else: queries with dense terms are stage A of the workgroup walk (fused with
    // the ordinary items, or on its own: the kernels themselves pick who works), the rest ordinary items.
    if (wave) {
        int wgrid_w = bm_num_cus() * 4;   // sixteen waves per CU; a small batch has no use for thousands of waves
        if ((long long)n_queries * 32 + 32 < wgrid_w) wgrid_w = n_queries * 32 + 32;
        hipLaunchKernelGGL(bm25_walk_wave_kernel, dim3(wgrid_w), dim3(WW_WAVES * 64), 0, st, rowptr, post_doc,
                           post_tf, doclen, idf, term_ub, post_imp, dslot, dense_tf, dense_stride, avgdl, k1, b,
                           (k1 + 1.0) / 255.0, 255.0 / (k1 + 1.0),
                           id_base, max_terms, k, doc_coll, query_coll, ctl, q_nt, q_S, q_SA, q_pmask, q_terms,
                           items, ipos, wrec, wterm, theta, slice_s, slice_id, slice_cnt, out_scores, out_ids, out_counts
#ifdef BM_STAMPS
                           , (unsigned long long*)(ws + L.off_stamps)
#endif
                           );
#ifdef BM_STAMPS
        {
            (void)hipStreamSynchronize(st);
            const int nw = wgrid_w * WW_WAVES;
            std::vector<unsigned long long> h((size_t)nw * 16);
            (void)hipMemcpy(h.data(), ws + L.off_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double tot[16] = {0};
            double mx = 0;
            for (int w = 0; w < nw; ++w) {
                double all = 0;
                for (int i = 0; i < 16; ++i) tot[i] += (double)h[(size_t)w * 16 + i];
                for (int i = 0; i < 9; ++i) all += (double)h[(size_t)w * 16 + i];
                mx = all > mx ? all : mx;
            }
            static const char* nm[9] = {"set-up", "stage+d_hi", "bloom", "classify", "score listed", "work list+advance", "finish", "skipped items", "idle tail"};
            double all = 0;
            for (int i = 0; i < 9; ++i) all += tot[i];
            fprintf(stderr, "[bm25 wave walk] %d waves, %.0f cycles per wave (max %.0f):", nw, all / nw, mx);
            for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * tot[i] / all);
            fprintf(stderr, " | items %.0f passes %.0f cuts %.0f batches %.0f postings %.0f listed %.0f\n", tot[10], tot[11], tot[12], tot[13], tot[14], tot[15]);
        }
#endif
    } else if (dslot) {
        THR_BM25_LAUNCH_SHAPE(2);
        THR_BM25_LAUNCH_SHAPE(1);
    }
    if (dslot) {
        // stage B: doc-window sweeps, skipped where stage A's threshold rules them out
        if ((rc = launch_status())) return rc;
        hipLaunchKernelGGL(bm25_sweep_filter_kernel, dim3(1), dim3(FILTER_THREADS), 0, st, ctl, n_queries, q_S,
                           q_SA, q_item0, q_dub, theta, slice_cnt, sweep_items);
        int wgrid = bm_num_cus() * 2;
        if (wgrid > L.cap) wgrid = L.cap;
#ifdef BM_STAMPS
#define BW_STAMP_ARG , (unsigned long long*)(ws + L.off_stamps) + (size_t)4096 * (BM_NSTAMP + 1)
#else
#define BW_STAMP_ARG
#endif
        hipLaunchKernelGGL((bm25_window_kernel<512, 8192, 1024>), dim3(wgrid), dim3(512), 0, st, rowptr, post_doc,
                           doclen, idf, dslot, dense_imp, dense_tf, dense_stride, avgdl, k1, b, n_docs, id_base,
                           max_terms, k, doc_coll, query_coll, ctl, q_nt, q_S, q_SA, q_pmask, q_terms, items,
                           sweep_items, ipos, theta, slice_s, slice_id, slice_cnt, out_scores, out_ids,
                           out_counts BW_STAMP_ARG);
        if ((rc = launch_status())) return rc;
    }
    THR_BM25_LAUNCH_SHAPE(0);
#ifdef BM_STAMPS
    {
        static const char* names[BM_NSTAMP] = {"item set-up", "init", "quotas", "staging", "edges/prefix", "phase 2 (+ chunk reset)",
                                               "bloom build", "singles listed", "singles scored", "work list / boot select", "compact/advance", "finish",
                                               "mask / acc fill", "slot scan / middle search", "#acc passes", "#mask passes",
                                               "#postings masked", "#survivors", "#phase2 rounds", ""};
        (void)hipStreamSynchronize(st);
        for (int pass = 0; pass < (dslot ? 3 : 1); ++pass) {
            const int g_n = pass == 1 ? bm_num_cus() * 2 : grid;
            std::vector<unsigned long long> h((size_t)g_n * (BM_NSTAMP + 1));
            (void)hipMemcpy(h.data(), d_stamps + (size_t)pass * 4096 * (BM_NSTAMP + 1), h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double tot[BM_NSTAMP + 1] = {0};
            for (int g = 0; g < g_n; ++g)
                for (int i = 0; i <= BM_NSTAMP; ++i) tot[i] += (double)h[(size_t)g * (BM_NSTAMP + 1) + i];
            double all = 0;
            for (int i = 0; i < 14; ++i) all += tot[i];
            fprintf(stderr, "[bm25 stamps%s] %d queries, %d workgroups, %.0f items, %.0f cycles per workgroup:", pass == 1 ? " window kernel (stage B)" : pass == 2 ? " stage A" : "", n_queries, g_n,
                    tot[BM_NSTAMP], all / g_n);
            for (int i = 0; i < 14; ++i)
                if (names[i][0]) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * tot[i] / all);
            for (int i = 14; i < 20; ++i) fprintf(stderr, " %s %.0f", names[i][0] ? names[i] : "#wmax|#singles", tot[i]);
            fprintf(stderr, "\n");
        }
        if (dslot) {   // the sweep items one by one: when each started and ended (cycles since the first), its passes and survivors
            int h_ctl[8];
            (void)hipMemcpy(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost);
            const int ns = h_ctl[5];
            std::vector<unsigned long long> lg((size_t)4 * (ns > 0 ? ns : 1));
            (void)hipMemcpy(lg.data(), d_stamps + (size_t)3 * 4096 * (BM_NSTAMP + 1), lg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int i = 0; i < ns; ++i) {
                if (lg[4 * i + 1] < t0) t0 = lg[4 * i + 1];
                if (lg[4 * i + 2] > t1) t1 = lg[4 * i + 2];
            }
            fprintf(stderr, "[bm25 sweep items] %d items, %llu cycles from the first start to the last end\n", ns, ns ? t1 - t0 : 0ull);
            if (getenv("THR_BM25_ITEM_LOG")) {
                const int ni = h_ctl[0];
                std::vector<unsigned long long> wl((size_t)4 * (ni > 0 ? ni : 1));
                (void)hipMemcpy(wl.data(), d_stamps + (size_t)3 * 4096 * (BM_NSTAMP + 1) + 4 * (size_t)L.cap, wl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                for (int i = 0; i < ni; ++i)
                    if (wl[4 * i + 2])
                        fprintf(stderr, "[walk] %d q %llu slice %llu dp %llu nt %llu postings %llu cycles %llu passes %llu\n", i, wl[4 * i] >> 32,
                                (wl[4 * i] >> 16) & 0xFFFF, (wl[4 * i] >> 8) & 0xFF, wl[4 * i] & 0xFF, wl[4 * i + 1], wl[4 * i + 2], wl[4 * i + 3]);
            }
            if (getenv("THR_BM25_ITEM_LOG"))
                for (int i = 0; i < ns; ++i)
                    fprintf(stderr, "[item] %d q %llu slice %llu np %llu walked %llu start %llu cycles %llu passes %llu survivors %llu\n", i,
                            lg[4 * i] >> 32, (lg[4 * i] >> 16) & 0xFFFF, (lg[4 * i] >> 8) & 0xFF, lg[4 * i] & 0xFF, lg[4 * i + 1] - t0,
                            lg[4 * i + 2] - lg[4 * i + 1], lg[4 * i + 3] >> 32, lg[4 * i + 3] & 0xFFFFFFFFull);
        }
    }
#endif
    if ((rc = launch_status())) return rc;
    hipLaunchKernelGGL(bm25_merge_kernel, dim3(n_queries), dim3(BMM_THREADS), 0, st, q_S, q_item0,
                       slice_s, slice_id, slice_cnt, k, out_scores, out_ids, out_counts);
    return launch_status();
}
