// Graph channel: bounded BFS over the entity CSR + mention scoring (gfx950).
//
// Stands where the reference calls GraphSearcher.search
// (src/voice_agent/rag2/retrieval.py:316-356 ->
// src/voice_agent/rag2/graph_search.py:290-418).  The reference returns an
// unordered set of chunk ids from PuppyGraph / SQL; the only score forms it
// holds are the standalone package's Cypher, 1/(1+distance) over 1..N hops and
// 1.0 for a direct mention (triple-hybrid-rag/src/triple_hybrid_rag/graph/
// puppygraph.py:152-167, 203-221).  The contract implemented here is the
// oracle's (oracle/thr_oracle.py graph_scores):
//     score(c) = sum over reached entities e, ascending e, mentions in CSR order,
//                of float64(conf(e,c)) / (1 + dist(e)),   dist(e) <= hops.
//
// One workgroup per query, everything on-chip except the contribution values:
//   1. level-synchronous BFS with an LDS open-addressing set (entity -> dist)
//   2. reached entities sorted ascending (LDS bitonic)
//   3. one contribution per (entity, mention) at an exclusive-scan position, so
//      position order == (entity asc, mention order)
//   4. keys (local chunk << 32 | position) sorted in LDS; each chunk's segment is
//      summed left to right in float64 by the thread that owns its head
//   5. streaming block top-k under (score desc, chunk asc)
// Algorithmic bytes per query: S*8 + sum_hops frontier*deg*4 + reached*(16 + mentions*8).
//
// Two launches, no host round trip: every query first runs with SMALL on-chip capacities
// (1024 entities / 2048 contributions: 5 workgroups per CU instead of 1), and only the queries
// that report an overflow there are redone by the second launch with the full capacities
// (its other workgroups exit at once).  Capacities never truncate silently: a query that
// overflows the full ones too keeps THR_FLAG_OVERFLOW.
#include "thr_common.hpp"

namespace thr {

constexpr int GR_THREADS = 256;
constexpr int GR_MAX_CON = 8192;   // contributions per query (full capacities; workspace stride)
constexpr uint32_t GR_EMPTY = 0xffffffffu;

// on-chip capacities of one launch flavour
struct GrSmall {
    static constexpr int SLOTS = 2048, MAX_ENT = 1024, MAX_CON = 2048, CAP = 512;
};
struct GrFull {
    static constexpr int SLOTS = 8192, MAX_ENT = 4096, MAX_CON = GR_MAX_CON, CAP = 1024;
};

// insert entity e at BFS level `lvl`; returns true if newly inserted
template <int GR_SLOTS>
__device__ __forceinline__ bool gr_insert(uint32_t* keys, uint8_t* dist, uint32_t e, int lvl) {
    uint32_t h = (e * 2654435761u) >> (32 - __builtin_ctz(GR_SLOTS));
    for (int probe = 0; probe < GR_SLOTS; ++probe) {
        uint32_t old = atomicCAS(&keys[h], GR_EMPTY, e);
        if (old == GR_EMPTY) {
            dist[h] = (uint8_t)lvl;
            return true;
        }
        if (old == e) return false;
        h = (h + 1) & (GR_SLOTS - 1);
    }
    return false;
}

// block-wide bitonic sort of uint64 keys, ascending, n = power of two.  Thread t takes PAIR t of a
// stage (i = t with a zero bit inserted at j, partner i | j): every thread of every trip does a
// compare-exchange (the i ^ j form leaves half of them idle).
__device__ inline void sort_u64_asc(uint64_t* a, int n) {
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const bool up = (i & k) == 0;
                const uint64_t x = a[i], y = a[p];
                if (up ? (x > y) : (x < y)) {
                    a[i] = y;
                    a[p] = x;
                }
            }
            __syncthreads();
        }
}

template <typename C, bool ONLY_OVERFLOWED>
__global__ __launch_bounds__(GR_THREADS) void graph_topk_kernel(
    const int64_t* __restrict__ ent_rowptr, const int32_t* __restrict__ ent_col, int64_t n_entities,
    const int64_t* __restrict__ men_rowptr, const int32_t* __restrict__ men_chunk,
    const float* __restrict__ men_conf, int64_t chunk_base, int64_t n_chunks,
    const int32_t* __restrict__ query_seeds, int max_seeds, int hops, int k,
    double* __restrict__ con_val_ws,  // [nq][GR_MAX_CON]
    double* __restrict__ out_s, int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt,
    uint32_t* __restrict__ out_flags) {
    constexpr int GR_SLOTS = C::SLOTS, GR_MAX_ENT = C::MAX_ENT, GR_MAX_CON = C::MAX_CON,
                  GR_CAP = C::CAP;
    static_assert(GR_SLOTS * 5 <= GR_MAX_CON * 8, "hash set must fit the sort-key bytes");
    // LDS: phase A (BFS) uses keys/dist/frontiers; phase B reuses the same bytes for sort keys
    __shared__ uint64_t big[GR_MAX_CON];            // hash set + lists, later sort keys
    __shared__ uint32_t reached[GR_MAX_ENT];         // (entity) list, later sorted
    __shared__ uint8_t reached_dist[GR_MAX_ENT];
    __shared__ int scan_tmp[GR_THREADS];
    __shared__ double b_s[GR_CAP];
    __shared__ int64_t b_id[GR_CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    __shared__ int n_reached, lvl_begin, lvl_end, overflow, n_con;

    uint32_t* keys = reinterpret_cast<uint32_t*>(big);            // [GR_SLOTS]
    uint8_t* hdist = reinterpret_cast<uint8_t*>(keys + GR_SLOTS);  // [GR_SLOTS]

    const int q = blockIdx.x;
    if (ONLY_OVERFLOWED && !(out_flags[q] & THR_FLAG_OVERFLOW)) return;
    double* con_val = con_val_ws + (int64_t)q * thr::GR_MAX_CON;
    for (int i = threadIdx.x; i < GR_SLOTS; i += GR_THREADS) keys[i] = GR_EMPTY;
    if (threadIdx.x == 0) {
        n_reached = 0;
        overflow = 0;
        n_con = 0;
    }
    __syncthreads();

    // ---- level 0: seeds ----
    if (threadIdx.x < max_seeds) {
        int32_t e = query_seeds[(int64_t)q * max_seeds + threadIdx.x];
        if (e >= 0 && e < n_entities && gr_insert<GR_SLOTS>(keys, hdist, (uint32_t)e, 0)) {
            int p = atomicAdd(&n_reached, 1);
            reached[p] = (uint32_t)e;
            reached_dist[p] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lvl_begin = 0;
        lvl_end = n_reached;
    }
    __syncthreads();

    // ---- BFS levels 1..hops: one wave per frontier entity, lanes over its edges ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = GR_THREADS / WAVE;
    for (int lvl = 1; lvl <= hops; ++lvl) {
        const int fb = lvl_begin, fe = lvl_end;
        for (int f = fb + wave; f < fe; f += nw) {
            const uint32_t e = reached[f];
            const int64_t lo = ent_rowptr[e], hi = ent_rowptr[e + 1];
            for (int64_t j = lo + lane; j < hi; j += WAVE) {
                const int32_t t = ent_col[j];
                if (*(volatile int*)&overflow) break;  // the hash set must not fill up
                if (t >= 0 && t < n_entities && gr_insert<GR_SLOTS>(keys, hdist, (uint32_t)t, lvl)) {
                    int p = atomicAdd(&n_reached, 1);
                    if (p < GR_MAX_ENT) {
                        reached[p] = (uint32_t)t;
                        reached_dist[p] = (uint8_t)lvl;
                    } else {
                        overflow = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            lvl_begin = fe;
            lvl_end = n_reached < GR_MAX_ENT ? n_reached : GR_MAX_ENT;
        }
        __syncthreads();
        if (overflow) break;
    }
    const int nr = n_reached < GR_MAX_ENT ? n_reached : GR_MAX_ENT;

    // ---- sort reached entities ascending (key = entity << 8 | dist) ----
    __syncthreads();
    {
        const int np = next_pow2(nr > 1 ? nr : 2);
        for (int i = threadIdx.x; i < np; i += GR_THREADS)
            big[i] = i < nr ? ((uint64_t)reached[i] << 8) | reached_dist[i] : ~0ull;
        __syncthreads();
        sort_u64_asc(big, np);
        for (int i = threadIdx.x; i < nr; i += GR_THREADS) {
            reached[i] = (uint32_t)(big[i] >> 8);
            reached_dist[i] = (uint8_t)(big[i] & 0xff);
        }
        __syncthreads();
    }

    // ---- contributions: exclusive scan of mention counts over the sorted entities ----
    int running = 0;  // same in every thread
    for (int base = 0; base < nr; base += GR_THREADS) {
        const int i = base + threadIdx.x;
        int cntm = 0;
        int64_t mlo = 0;
        if (i < nr) {
            mlo = men_rowptr[reached[i]];
            cntm = (int)(men_rowptr[reached[i] + 1] - mlo);
        }
        scan_tmp[threadIdx.x] = cntm;
        __syncthreads();
        for (int off = 1; off < GR_THREADS; off <<= 1) {  // Hillis-Steele inclusive scan
            int v = threadIdx.x >= off ? scan_tmp[threadIdx.x - off] : 0;
            __syncthreads();
            scan_tmp[threadIdx.x] += v;
            __syncthreads();
        }
        const int excl = running + scan_tmp[threadIdx.x] - cntm;
        const int chunk_total = scan_tmp[GR_THREADS - 1];
        if (i < nr) {
            const double w = __dadd_rn(1.0, (double)reached_dist[i]);
            for (int m = 0; m < cntm; ++m) {
                const int pos = excl + m;
                if (pos < GR_MAX_CON) {
                    const int64_t c = (int64_t)men_chunk[mlo + m] - chunk_base;
                    const bool mine = c >= 0 && c < n_chunks;
                    big[pos] = mine ? ((uint64_t)c << 32) | (uint32_t)pos : ~0ull;
                    con_val[pos] = __ddiv_rn((double)men_conf[mlo + m], w);
                } else {
                    overflow = 1;
                }
            }
        }
        running += chunk_total;
        __syncthreads();
    }
    const int nc = running < GR_MAX_CON ? running : GR_MAX_CON;
    const int ncp = next_pow2(nc > 1 ? nc : 2);
    for (int i = nc + threadIdx.x; i < ncp; i += GR_THREADS) big[i] = ~0ull;
    __syncthreads();
    sort_u64_asc(big, ncp);
    __threadfence_block();

    // ---- segmented left-to-right sums + top-k ----
    BlockTopK<GR_CAP, GR_THREADS> tk;
    tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);
    for (int base = 0; base < nc; base += GR_THREADS) {
        const int i = base + threadIdx.x;
        bool head = false;
        double score = 0.0;
        int64_t chunk = 0;
        if (i < nc && big[i] != ~0ull) {
            const uint32_t c = (uint32_t)(big[i] >> 32);
            head = (i == 0) || ((uint32_t)(big[i - 1] >> 32) != c);
            if (head) {
                chunk = c;
                for (int j = i; j < nc && big[j] != ~0ull && (uint32_t)(big[j] >> 32) == c; ++j)
                    score = __dadd_rn(score, con_val[(uint32_t)big[j]]);
            }
        }
        tk.push(head, score, chunk);
    }
    const int n = tk.finish();
    for (int i = threadIdx.x; i < k; i += GR_THREADS) {
        out_s[(int64_t)q * k + i] = i < n ? b_s[i] : -INFINITY;
        out_id[(int64_t)q * k + i] = i < n ? b_id[i] + chunk_base : -1;
    }
    if (threadIdx.x == 0) {
        out_cnt[q] = n;
        out_flags[q] = overflow ? THR_FLAG_OVERFLOW : THR_FLAG_CERTIFIED;
    }
}

// ---------------------------------------------------------------------------------------------
// Third tier: queries that overflow the full on-chip capacities too (a hub entity with tens of
// thousands of edges) are walked in GLOBAL memory -- no capacity left to overflow, so
// THR_FLAG_OVERFLOW never reaches the caller and nothing in a batch pipeline has to raise.
// GR_FB_BLOCKS workgroups share the overflowed queries of the batch; each owns one distance
// array (1 byte per entity, 0xFF = not reached) and
//   1. BFS, level-synchronous, by SCANNING the distance array for the entities of the previous
//      level (no frontier lists); a wave expands one frontier entity at a time, lanes over its
//      edges.  The distance bytes are written and read with agent-scope relaxed atomics (the
//      array is rewritten level after level by the same CU), fenced at every level barrier, and
//      only ever change 0xFF -> level, so racing writers agree;
//   2. scores every chunk of the shard from the TRANSPOSED mention CSR (chunk -> (entity, conf),
//      in (entity asc, mention) order -- built once at index set-up): one thread per chunk sums
//      conf/(1+dist) left to right in float64, which is the oracle's order, then the streaming
//      block top-k.  O(E + mentions) per query instead of O(reached): a rare, slow, exact path.
// ---------------------------------------------------------------------------------------------
// (64 workgroups: a quarter of the CUs -- a batch of many hub-seeded queries costs
// O(E * hops + mentions) per query and would serialise on fewer; 1 byte per entity and workgroup)
constexpr int GR_FB_BLOCKS = 64;
__device__ __forceinline__ void gr_set_dist(uint8_t* dist, int64_t e, uint8_t v) {
    __hip_atomic_store(dist + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t gr_dist(const uint8_t* dist, uint32_t e) {
    const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(dist) + (e >> 2),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (w >> (8 * (e & 3))) & 0xffu;
}

__global__ __launch_bounds__(GR_THREADS) void graph_fallback_kernel(
    const int64_t* __restrict__ ent_rowptr, const int32_t* __restrict__ ent_col, int64_t n_entities,
    const int64_t* __restrict__ tmen_rowptr, const int32_t* __restrict__ tmen_ent,
    const float* __restrict__ tmen_conf, int64_t chunk_base, int64_t n_chunks,
    const int32_t* __restrict__ query_seeds, int n_queries, int max_seeds, int hops, int k,
    uint8_t* __restrict__ dist_ws, int64_t e_pad, double* __restrict__ out_s,
    int64_t* __restrict__ out_id, int32_t* __restrict__ out_cnt, uint32_t* __restrict__ out_flags) {
    __shared__ double b_s[GrFull::CAP];
    __shared__ int64_t b_id[GrFull::CAP];
    __shared__ int b_cnt;
    __shared__ double th_s;
    __shared__ int64_t th_id;
    uint8_t* dist = dist_ws + (int64_t)blockIdx.x * e_pad;
    const int lane = threadIdx.x & 63;
    for (int q = blockIdx.x; q < n_queries; q += gridDim.x) {
        if (!(out_flags[q] & THR_FLAG_OVERFLOW)) continue;   // same answer in every thread
        __syncthreads();
        for (int64_t i = threadIdx.x; i < e_pad / 4; i += GR_THREADS)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(dist) + i, 0xffffffffu, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // (the distance bytes are written and read with agent-scope relaxed atomics; a release /
        // acquire fence pair around each barrier orders the levels)
        __threadfence();
        __syncthreads();
        if (threadIdx.x < max_seeds) {
            const int32_t e = query_seeds[(int64_t)q * max_seeds + threadIdx.x];
            if (e >= 0 && e < n_entities) gr_set_dist(dist, e, 0);
        }
        __threadfence();
        __syncthreads();
        for (int lvl = 1; lvl <= hops; ++lvl) {
            for (int64_t base = 0; base < n_entities; base += GR_THREADS) {
                const int64_t e = base + threadIdx.x;
                const bool in_frontier = e < n_entities && gr_dist(dist, (uint32_t)e) == (uint32_t)(lvl - 1);
                uint64_t m = __ballot(in_frontier);
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int64_t f = base + (threadIdx.x & ~63) + src;
                    const int64_t lo = ent_rowptr[f], hi = ent_rowptr[f + 1];
                    for (int64_t j = lo + lane; j < hi; j += WAVE) {
                        const int32_t t = ent_col[j];
                        if (t >= 0 && t < n_entities && gr_dist(dist, (uint32_t)t) == 0xffu)
                            gr_set_dist(dist, t, (uint8_t)lvl);
                    }
                }
            }
            __threadfence();
            __syncthreads();
        }
        BlockTopK<GrFull::CAP, GR_THREADS> tk;
        tk.init(b_s, b_id, &b_cnt, &th_s, &th_id, k);
        for (int64_t base = 0; base < n_chunks; base += GR_THREADS) {
            const int64_t c = base + threadIdx.x;
            bool any = false;
            double score = 0.0;
            if (c < n_chunks) {
                const int64_t lo = tmen_rowptr[c], hi = tmen_rowptr[c + 1];
                for (int64_t j = lo; j < hi; ++j) {
                    const uint32_t d = gr_dist(dist, (uint32_t)tmen_ent[j]);
                    if (d != 0xffu) {
                        score = __dadd_rn(score, __ddiv_rn((double)tmen_conf[j], __dadd_rn(1.0, (double)d)));
                        any = true;
                    }
                }
            }
            tk.push(any, score, c);
        }
        const int n = tk.finish();
        for (int i = threadIdx.x; i < k; i += GR_THREADS) {
            out_s[(int64_t)q * k + i] = i < n ? b_s[i] : -INFINITY;
            out_id[(int64_t)q * k + i] = i < n ? b_id[i] + chunk_base : -1;
        }
        if (threadIdx.x == 0) {
            out_cnt[q] = n;
            out_flags[q] = THR_FLAG_CERTIFIED | THR_FLAG_EXACT;
        }
    }
}

}  // namespace thr

using namespace thr;

static size_t graph_dist_pad(int64_t n_entities) { return (size_t)((n_entities + 255) / 256 * 256); }

extern "C" size_t thr_graph_workspace_bytes(int n_queries, int64_t n_entities) {
    if (n_queries <= 0) return 0;
    return (size_t)n_queries * GR_MAX_CON * sizeof(double) +
           (n_entities > 0 ? (size_t)GR_FB_BLOCKS * graph_dist_pad(n_entities) : 0);
}

extern "C" int thr_graph_topk(const int64_t* ent_rowptr, const int32_t* ent_col, int64_t n_entities,
                              const int64_t* men_rowptr, const int32_t* men_chunk,
                              const float* men_conf, const int64_t* tmen_rowptr,
                              const int32_t* tmen_ent, const float* tmen_conf, int64_t chunk_base,
                              int64_t n_chunks, const int32_t* query_seeds, int n_queries,
                              int max_seeds, int hops, int k, double* out_scores, int64_t* out_ids,
                              int32_t* out_counts, uint32_t* out_flags, void* workspace,
                              size_t workspace_bytes, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!ent_rowptr || !ent_col || !men_rowptr || !men_chunk || !men_conf ||
                      !query_seeds || !out_scores || !out_ids || !out_counts || !out_flags ||
                      !workspace,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_entities <= 0 || n_entities >= 0xffffffffll || n_chunks <= 0 ||
                      n_chunks >= 0xffffffffll || n_queries <= 0 || max_seeds <= 0 ||
                      max_seeds > THR_GRAPH_MAX_SEEDS || hops < 0 || hops > 8 || k <= 0 ||
                      k > THR_TOPK_MAX,
                  THR_ERR_INVALID);
    const bool fallback = tmen_rowptr && tmen_ent && tmen_conf;
    THR_RETURN_IF(workspace_bytes < thr_graph_workspace_bytes(n_queries, fallback ? n_entities : 0),
                  THR_ERR_WORKSPACE);
    hipLaunchKernelGGL((graph_topk_kernel<GrSmall, false>), dim3(n_queries), dim3(GR_THREADS), 0,
                       (hipStream_t)stream, ent_rowptr, ent_col, n_entities, men_rowptr, men_chunk,
                       men_conf, chunk_base, n_chunks, query_seeds, max_seeds, hops, k,
                       (double*)workspace, out_scores, out_ids, out_counts, out_flags);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL((graph_topk_kernel<GrFull, true>), dim3(n_queries), dim3(GR_THREADS), 0,
                       (hipStream_t)stream, ent_rowptr, ent_col, n_entities, men_rowptr, men_chunk,
                       men_conf, chunk_base, n_chunks, query_seeds, max_seeds, hops, k,
                       (double*)workspace, out_scores, out_ids, out_counts, out_flags);
    rc = launch_status();
    if (rc || !fallback) return rc;
    uint8_t* dist_ws = (uint8_t*)workspace + (size_t)n_queries * GR_MAX_CON * sizeof(double);
    hipLaunchKernelGGL(graph_fallback_kernel, dim3(GR_FB_BLOCKS), dim3(GR_THREADS), 0,
                       (hipStream_t)stream, ent_rowptr, ent_col, n_entities, tmen_rowptr, tmen_ent,
                       tmen_conf, chunk_base, n_chunks, query_seeds, n_queries, max_seeds, hops, k,
                       dist_ws, (int64_t)graph_dist_pad(n_entities), out_scores, out_ids, out_counts,
                       out_flags);
    return launch_status();
}
