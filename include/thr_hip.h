/* thr_hip.h -- C ABI of the MI355X-native retrieval hot path (libthr_hip.so).
 *
 * The reference (matheusfalcaopinto/triple-hybrid-rag) has no FFI layer: its
 * scorers run inside PostgreSQL / PuppyGraph / remote model servers and are
 * reached from Python over HTTP.  Each entry point
 * below replaces one of those out-of-process scorers; the reference call site
 * it stands behind is cited (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory (hipMalloc'd or a
 *     PyTorch-ROCm tensor's data_ptr) unless the name starts with ``h_``;
 *   - all work is enqueued on ``stream`` (a hipStream_t passed as void*; 0 =
 *     the null stream); nothing synchronises, allocates or frees;
 *   - return 0 on success, a negative THR_ERR_* for argument/capacity errors,
 *     or a positive hipError_t if a launch failed;
 *   - doc / chunk / entity ids are indices into the caller's arrays; outputs
 *     add ``id_base`` so a document-sharded index reports global ids;
 *   - total order of every ranked output: (score descending, id ascending).
 */
#ifndef THR_HIP_H
#define THR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THR_ABI_VERSION 9

typedef void *thr_stream_t;

enum {
    THR_OK = 0,
    THR_ERR_INVALID = -1,     /* null pointer / negative size / k out of range */
    THR_ERR_UNSUPPORTED = -2, /* shape the kernels are not built for */
    THR_ERR_WORKSPACE = -3,   /* workspace smaller than thr_*_workspace_bytes */
    THR_ERR_CAPACITY = -4     /* a fixed on-chip capacity would be exceeded */
};

/* per-query status bits written by thr_dense_topk */
#define THR_FLAG_CERTIFIED 1u /* top-k proven exact by the fp32-error certificate */
#define THR_FLAG_OVERFLOW 2u  /* candidate buffer overflowed (never certified)    */
#define THR_FLAG_EXACT 4u     /* produced by the exhaustive float64 path          */

#define THR_DENSE_MAX_K 256   /* k and k' (shortlist) upper bound; the certificate needs a margin
                               * k' - k of rows (28 for the fp32 scan, 92 for the f16 scans), so
                               * beyond k ~ 200 queries increasingly take the exhaustive path
                               * (still exact, much slower) */
#define THR_BM25_MAX_TERMS 32
#define THR_GRAPH_MAX_SEEDS 16
#define THR_RRF_MAX_PER_CHANNEL 128
#define THR_TOPK_MAX 128      /* bm25 / graph k upper bound */

int thr_abi_version(void);
const char *thr_error_string(int code);
/* CUs, HBM bytes and gcnArchName of the current device (host-side query). */
int thr_device_info(int *h_compute_units, int64_t *h_hbm_bytes, char *h_arch, int h_arch_len);

/* a1  query-embedding post-processing: prefix-truncate to ``store_dim`` and
 * L2-normalise in float32, zero rows stay zero.
 * Replaces truncate_matryoshka/normalize_l2, src/voice_agent/rag2/embedder.py:31-68
 * (the model forward that produces ``full`` is an external server). */
int thr_embed_postproc(const float *full, int n, int full_dim, int store_dim,
                       float *out /* [n, min(full_dim,store_dim)] */, thr_stream_t stream);

/* Index build helper: ||d|| per row as sqrt of the sequential float64 sum of
 * squares, and float32 1/||d|| (0 for a zero row = "embedding IS NULL",
 * database/migrations/20260114_rag2_schema.sql:404). */
int thr_doc_norms(const float *docs, int64_t n_docs, int dim, double *dnorm, float *inv_norm,
                  thr_stream_t stream);

/* a2  dense channel: exact brute-force cosine top-k of ``n_queries`` queries
 * over an HBM-resident float32 corpus [n_docs, dim] (dim % 256 == 0).
 * Replaces SQL rag2_semantic_search (`1 - (embedding_1024 <=> q)` ORDER BY
 * distance LIMIT k), database/migrations/20260114_rag2_schema.sql:377-410,
 * called from src/voice_agent/rag2/retrieval.py:304-312, and the client-side
 * np.dot fallback src/voice_agent/retrieval/hybrid_search.py:285-316.
 *
 * Pass 1 (this entry point: the fp32 matrix cores, 32 queries per workgroup and pass over a
 * row slice; thr_dense_topk_f16 below: f16 matrix cores over a float16 copy or over rows
 * rounded in flight) finds per query a threshold tau from a row sample, streams the corpus and
 * emits the rows scoring >= tau; pass 2 takes from those the band that can still reach the top
 * k' = ``kprime``, rescores it in float64 (sequential accumulation, the oracle's contract)
 * from the float32 rows and orders it.
 * out_flags[q] has THR_FLAG_CERTIFIED when the float32 error bound proves the
 * top-k exact; otherwise call thr_dense_topk_exact for that query.
 * Outputs are padded with (-inf, -1) beyond out_counts[q]. */
size_t thr_dense_workspace_bytes(int64_t n_docs, int dim, int n_queries, int kprime);
/* Collection filter of the RPC (``AND (p_collection IS NULL OR d.collection = p_collection)``,
 * rag2_schema.sql:404-408), applied BEFORE the ranking on the device: doc_coll int32 [n_docs]
 * (any labelling) and query_coll int32 [nq] (-1 = unfiltered) -- both or neither (NULL).  The
 * threshold sample, the shortlist and the exhaustive path all see only rows of the query's
 * collection; a collection too thin for the sampled threshold ends on the exhaustive path. */
int thr_dense_topk(const float *docs, const double *dnorm, const float *inv_norm, int64_t n_docs,
                   int dim, int64_t id_base, const float *queries, int n_queries, int k,
                   int kprime, const int32_t *doc_coll, const int32_t *query_coll,
                   double *out_scores /* [nq,k] */, int64_t *out_ids /* [nq,k] */,
                   int32_t *out_counts /* [nq] */, uint32_t *out_flags /* [nq] */,
                   void *workspace, size_t workspace_bytes, thr_stream_t stream);

/* Exhaustive float64 path (every row scored with the oracle's arithmetic);
 * used for queries thr_dense_topk could not certify (massive ties). */
size_t thr_dense_exact_workspace_bytes(int64_t n_docs, int n_queries);
int thr_dense_topk_exact(const float *docs, const double *dnorm, int64_t n_docs, int dim,
                         int64_t id_base, const float *queries, int n_queries, int k,
                         const int32_t *doc_coll, const int32_t *query_coll,
                         double *out_scores, int64_t *out_ids, int32_t *out_counts,
                         uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                         thr_stream_t stream);

/* Device-side completion of thr_dense_topk / thr_dense_topk_f16: every query whose flags lack
 * THR_FLAG_CERTIFIED is redone on the exhaustive float64 path and overwritten in place (flags
 * become CERTIFIED | EXACT); *n_rescued (a DEVICE int32, may be NULL) is incremented per redone
 * query.  No host read-back: a batch runs end to end without a synchronisation. */
size_t thr_dense_rescue_workspace_bytes(int n_queries, int k);
int thr_dense_rescue(const float *docs, const double *dnorm, int64_t n_docs, int dim,
                     int64_t id_base, const float *queries, int n_queries, int k,
                     const int32_t *doc_coll, const int32_t *query_coll,
                     double *io_scores, int64_t *io_ids, int32_t *io_counts, uint32_t *io_flags,
                     int32_t *n_rescued, void *workspace, size_t workspace_bytes,
                     thr_stream_t stream);

/* Shortlist scan on the f16 matrix cores (< 2^25 rows per shard).  The float32 corpus stays the
 * source of truth: scores are the same float64 rescoring of float32 rows, and the certificate's
 * error bound additionally covers row quantisation (doc_rel_err, measured by
 * thr_dense_quantize_f16 into *max_rel_err, a DEVICE float) and query quantisation (measured per
 * query on the device).  Two flavours:
 *   docs16 != NULL: the scan streams a float16 COPY written by thr_dense_quantize_f16: an opaque
 *                   FRAGMENT-MAJOR image of thr_dense_f16_copy_bytes(n_docs, dim) bytes,
 *                   [row tile of 32][dims 32 k .. +32][rows 16 a .. +16][lane = r + 16 g][8 halves]
 *                   = the register image of the v_mfma_f32_16x16x32_f16 row operand
 *                   (THR_DENSE_MFMA=32 in the environment of BOTH the quantiser and the scan:
 *                   [row tile of 32][k-step of 16 dims][lane = r + 32 h]
 *                   [8 halves], v_mfma_f32_32x32x16_f16), holding the NORMALISED rows
 *                   d/||d|| (NaN for rows without an embedding and for the padding of the last
 *                   tile; doc_rel_err = max_d ||fp16(d/||d||) - d/||d|| ||, any value range).
 *                   Row tiles reach LDS by LDS-DMA, once per CU; each wave keeps 32 queries in
 *                   registers as the other operand (256 queries per CU, 128 at dim 1024);
 *   docs16 == NULL: the scan streams the float32 rows and rounds them to float16 in registers
 *                   (no second copy; thr_dense_quantize_f16 with docs16 == NULL only measures
 *                   doc_rel_err = max_d ||fp16(d) - d|| / ||d||, +inf when a value leaves the
 *                   float16 range, which thr_dense_topk_f16 rejects; 64 queries per pass, 32 at
 *                   dim 1024).
 * dim in {512, 768, 1024}. */
size_t thr_dense_f16_copy_bytes(int64_t n_docs, int dim);
/* queries per workgroup (= per pass over a row slice) of the f16 scan */
int thr_dense_f16_query_tile(int dim, int packed /* docs16 != NULL */, int n_queries);
/* largest n_queries of ONE thr_dense_topk_f16 call: the copy scan addresses a lane's candidate
 * segment with a 32-bit byte offset (131072 bytes of candidate area per query), so a larger
 * batch returns THR_ERR_UNSUPPORTED and the caller splits it (GpuIndex.dense_search does). */
int thr_dense_f16_max_queries(int dim, int packed /* docs16 != NULL */);
int thr_dense_quantize_f16(const float *docs, int64_t n_docs, int dim,
                           uint16_t *docs16 /* thr_dense_f16_copy_bytes(...) bytes, or NULL */,
                           float *max_rel_err, thr_stream_t stream);
size_t thr_dense_f16_workspace_bytes(int64_t n_docs, int dim, int n_queries, int kprime);
int thr_dense_topk_f16(const float *docs, const uint16_t *docs16 /* or NULL */, double doc_rel_err,
                       const double *dnorm, const float *inv_norm, int64_t n_docs, int dim,
                       int64_t id_base, const float *queries, int n_queries, int k, int kprime,
                       const int32_t *doc_coll, const int32_t *query_coll,
                       double *out_scores, int64_t *out_ids, int32_t *out_counts,
                       uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                       thr_stream_t stream);
/* e1  The same search split around ONE exchange between document shards (ABI 9): what a shard of G
 * does NOT have to rescore.  The reference has a single table and a single `ORDER BY ... LIMIT k`
 * (rag2_schema.sql:404-410); sharded, every shard would rescore its own k best although only
 * about k / G of them are among the k best of all.
 *   thr_dense_shortlist_f16  thr_dense_topk_f16 up to and including the filter scan (the candidate
 *                            lists stay in ``workspace``), then per query the ``m`` largest scan
 *                            scores, each lowered by the scan's error bound to a LOWER bound of
 *                            ||q|| x cosine of its row: top_lb float32 [n_queries, m], -inf padded;
 *   (exchange)               all-gather top_lb over the shards -> [G, n_queries, m];
 *   thr_dense_floor          gfloor[q] = the k-th largest of a query's G * m values (-inf when
 *                            fewer than k are finite): a lower bound of ||q|| x the k-th best
 *                            cosine of the whole corpus;
 *   thr_dense_finish_f16     the rest of thr_dense_topk_f16 on the SAME workspace and arguments:
 *                            rows whose scan score is below gfloor[q] - 1.5 eps ||q|| are left out
 *                            of the float64 rescoring.  A shard's list may then hold FEWER than k
 *                            rows (out_counts); THR_FLAG_CERTIFIED now says "no row outside this
 *                            list can be among the k best of all shards" (proved against the
 *                            shard's own k-th score or against gfloor), and thr_dense_rescue
 *                            redoes the others exhaustively as before.  thr_merge_topk over the
 *                            shards' lists is then the exact top-k: no second exchange.
 * The floor reaches thr_dense_finish_f16 either as gfloor (thr_dense_floor's output) or as the
 * gathered bounds themselves (top_lb_all, n_shards, m; G * m <= 4096): the kernel that cuts the
 * band then finds the k-th largest itself and thr_dense_floor is not launched at all.  Neither
 * (or -inf entries) gives thr_dense_topk_f16's behaviour.  m <= THR_DENSE_MAX_K, G * m <= 8192
 * for thr_dense_floor; choose m >= k / G, about 2 k / G for a floor close to the true k-th score. */
int thr_dense_shortlist_f16(const float *docs, const uint16_t *docs16 /* or NULL */, double doc_rel_err,
                            const float *inv_norm, int64_t n_docs, int dim, const float *queries,
                            int n_queries, int kprime, const int32_t *doc_coll,
                            const int32_t *query_coll, int m, float *top_lb /* [nq, m] */,
                            void *workspace, size_t workspace_bytes, thr_stream_t stream);
int thr_dense_floor(const float *top_lb /* [n_shards, nq, m] */, int n_shards, int n_queries, int m,
                    int k, float *gfloor /* [nq] */, thr_stream_t stream);
int thr_dense_finish_f16(const float *docs, const uint16_t *docs16 /* or NULL */, double doc_rel_err,
                         const double *dnorm, const float *inv_norm, int64_t n_docs, int dim,
                         int64_t id_base, const float *queries, int n_queries, int k, int kprime,
                         const int32_t *doc_coll, const int32_t *query_coll,
                         const float *gfloor /* [nq] or NULL */,
                         const float *top_lb_all /* [n_shards, nq, m] or NULL */, int n_shards, int m,
                         double *out_scores, int64_t *out_ids, int32_t *out_counts,
                         uint32_t *out_flags, void *workspace, size_t workspace_bytes,
                         thr_stream_t stream);
int thr_dense_scan_probe_f16(const float *docs, const uint16_t *docs16 /* or NULL */,
                             const float *inv_norm, int64_t n_docs, int dim, const float *queries,
                             int n_queries, void *workspace, size_t workspace_bytes,
                             thr_stream_t stream);

/* Diagnostic build of the float16-copy scan (same launch as thr_dense_scan_probe_f16, on the
 * workspace of the last thr_dense_topk_f16): every wave sums s_memtime deltas around the phases
 * of its tile loop into stamps[wave * 8 + {0 wait for own DMA pieces, 1 barrier, 2 DMA issue,
 * 3 k-loop, 4 emit, 5 tiles, 6 whole loop, 7 HW_ID}] (device, uint64).  stamps == NULL: only
 * *h_n_waves (host) is set, to size the buffer.  Not a timing: the stamps drain the wave's
 * queues; it says where the time goes (profiles/README.md). */
int thr_dense_scan_stamps_f16(const uint16_t *docs16, int64_t n_docs, int dim, int n_queries,
                              void *workspace, size_t workspace_bytes,
                              unsigned long long *stamps, int *h_n_waves, thr_stream_t stream);

/* Timing/roofline probe: ONLY the streaming pass-1 kernel of thr_dense_topk
 * (threshold filter against ``tau``), for ``n_tiles`` query tiles. */
int thr_dense_scan_probe(const float *docs, const float *inv_norm, int64_t n_docs, int dim,
                         const float *queries, int n_queries, void *workspace,
                         size_t workspace_bytes, thr_stream_t stream);

/* f1  index build (the step before the path): the inverted index from tokenised rows, on the
 * device.  The reference leaves this to PostgreSQL -- rag_child_chunks rows go in
 * (src/voice_agent/rag2/ingest.py:361-470) and the `tsv` generated column + GIN index
 * (database/migrations/20260114_rag2_schema.sql:146-148, 171-172) become the posting lists.
 * ``doc`` / ``term`` / ``tf`` [n_pairs]: one entry per distinct (doc, term) of a chunk, or one per
 * token occurrence with tf == NULL (every entry counts 1) -- entries that repeat add up; entries
 * with doc outside [0, n_docs), term outside [0, n_vocab) or tf <= 0 are dropped.  Outputs:
 * rowptr [V + 1], post_doc / post_tf [n_pairs] of which the first *nnz_out are written (doc
 * ascending within a term), doclen [n_docs] = the docs' summed term frequencies (entries whose term
 * is outside the vocabulary still count: they are tokens of the chunk), df [V] = postings
 * per term of THIS shard (all-reduce it over the shards for the global idf), *nnz_out (device
 * memory).  No host round trip inside; ``workspace`` >= thr_lexical_build_workspace_bytes(). */
size_t thr_lexical_build_workspace_bytes(int64_t n_pairs, int64_t n_vocab);
int thr_lexical_build(const int32_t *doc, const int32_t *term, const int32_t *tf /* or NULL */,
                      int64_t n_pairs, int64_t n_docs, int64_t n_vocab, int64_t *rowptr,
                      int32_t *post_doc, int32_t *post_tf, float *doclen, int64_t *df,
                      int64_t *nnz_out, void *workspace, size_t workspace_bytes,
                      thr_stream_t stream);

/* a3  lexical channel: Okapi BM25 (k1, b) top-k over a CSR inverted index,
 * OR semantics, float64 accumulation in query-term order.
 * Stands where SQL rag2_lexical_search (ts_rank_cd ... ORDER BY rank DESC
 * LIMIT k) is called, rag2_schema.sql:341-374 / retrieval.py:282-290; the
 * north-star mandates BM25 in its place.  ``query_terms`` is [nq, max_terms]
 * term ids, negative = padding.  idf[] and avgdl are corpus-GLOBAL. */
/* Index set-up: upper bounds for the WAND-style pruning of thr_bm25_topk, computed with the
 * scoring formula itself: term_ub[t] = max contribution of a posting of term t, block_ub[j] = max
 * contribution among postings [128 j, 128 j + 128) of the posting array
 * (thr_bm25_block_count(nnz) doubles).  They depend on idf / avgdl / k1 / b. */
size_t thr_bm25_block_count(int64_t nnz);
int thr_bm25_bounds(const int64_t *rowptr, const int32_t *post_doc, const int32_t *post_tf,
                    const float *doclen, const double *idf, double avgdl, double k1, double b,
                    int64_t n_vocab, int64_t nnz, double *term_ub /* [V] */,
                    double *block_ub /* [thr_bm25_block_count(nnz)] */,
                    uint8_t *post_imp /* [nnz] or NULL */, thr_stream_t stream);
/* post_imp (ABI 6): per posting, its IMPACT tf (k1+1) / (tf + k1 ((1-b) + b dl/avgdl)) -- the
 * contribution is idf * impact, the impact does not depend on the query -- rounded UP to 8 bits
 * of (k1+1)/255.  Given to thr_bm25_topk, OR queries of <= 8 terms hold each candidate doc
 * against the SUM of its own postings' quantised impacts (accumulated in 16 bits per doc slot on
 * chip) instead of the sum of the per-term maxima: ~2.5 % of the postings of a stop-word query lead
 * to a survivor of that bound, ~16 % with the coarse one.  Results are the same bits either way. */
/* term ids outside [0, n_vocab) have no postings: the OR form ignores them, with conjunctive != 0 a
 * non-negative one makes the query unsatisfiable (empty result, as the SQL AND would give).  term_ub / block_ub (or NULL: score
 * every posting): a doc whose bound cannot beat the running k-th best score is dropped before its
 * term frequencies are fetched -- same results, bit for bit.  conjunctive != 0: only docs that
 * hold EVERY query term (the reference's plainto_tsquery is an AND, rag2_schema.sql:365; the
 * oracle's and north star's BM25 is the OR form, the default).  doc_coll [n_docs] / query_coll
 * [nq] (both or neither; query_coll -1 = unfiltered): only docs of the query's collection are
 * ranked, BEFORE the top-k -- ``AND (p_collection IS NULL OR d.collection = p_collection)``,
 * rag2_schema.sql:368-373. */
/* Work decomposition (ABI 5): queries whose lists hold more than one slice's postings (24576 when
 * the batch fills the chip, down to 8192 when it does not -- a one-query call spreads over several
 * workgroups) are cut into doc-range slices (up to 128, equal shares of the query's longest list), every slice is a work item of a
 * persistent grid, the slices of a query share the pruning threshold through a device-side
 * atomic max and are merged at the end -- a stop-word query is the job of many workgroups, not
 * of one.  ``workspace`` >= thr_bm25_workspace_bytes(n_queries, max_terms, k): item list, slice
 * edges, per-slice lists.
 * Wave walk (round 4, same ABI): with the bounds and impacts given, k <= 64 and the OR form, every
 * query of <= 8 terms is cut into slices of ~640-1536 postings instead and ONE WAVE walks a slice
 * (bm25_walk_wave_kernel: no workgroup barrier, sixteen waves -- sixteen independent latency chains --
 * per CU); queries of more terms, AND queries, k > 64 and calls without bounds keep the workgroup walk.
 * Same arithmetic, same bounds: the same bits either way (THR_BM25_WALK=block forces the latter). */
/* Dense terms (ABI 7).  A stop word's posting list is most of the corpus; walking it posting by
 * posting is the slowest way to learn that nearly all of its docs cannot make the top-k.  The
 * caller picks the terms held by a large share of the docs (the host layer: df >= 1 % of the
 * shard's docs, tf <= 65535, at most 512 of them), lists them in ``terms`` and gets, per term, one
 * ROW of thr_bm25_dense_stride(n_docs) entries: dense_imp[row][d] = post_imp of doc d's posting (0:
 * the doc does not hold the term), dense_tf[row][d] = its term frequency.  3 bytes per doc and
 * term.  dense_slot [V] maps a term to its row (-1: none).
 * Given to thr_bm25_topk, an OR query of <= 8 terms that holds such terms is evaluated the
 * MaxScore way, in two stages.  Stage A walks the postings of the query's OTHER terms only; where
 * a doc is scored, the dense terms' frequencies are read from their rows (no search), and every
 * bound carries the dense terms' largest impacts.  Stage B covers the docs that hold none of the
 * other terms: their score is at most the sum of the dense terms' bounds, so when that sum stays
 * below stage A's k-th best score -- the usual case: a stop word's idf is small -- the stage is
 * skipped; otherwise the shard is swept in doc windows, the dense terms' impacts arriving as
 * coalesced loads of the rows (no posting is staged for them), and the few docs whose summed
 * bound reaches the threshold are scored from dense_tf.  Both stages use the oracle's arithmetic
 * in query-term order: the results are the same bits as without the rows. */
int64_t thr_bm25_dense_stride(int64_t n_docs);
int thr_bm25_dense_rows(const int64_t *rowptr, const int32_t *post_doc, const int32_t *post_tf,
                        const uint8_t *post_imp, const int32_t *terms /* [n_terms] */, int n_terms,
                        int64_t n_docs, int64_t max_df /* longest of those lists */,
                        uint8_t *dense_imp /* [n_terms, stride] */,
                        uint16_t *dense_tf /* [n_terms, stride] */, thr_stream_t stream);
size_t thr_bm25_workspace_bytes(int n_queries, int max_terms, int k);
int thr_bm25_topk(const int64_t *rowptr /* [V+1] */, const int32_t *post_doc,
                  const int32_t *post_tf, const float *doclen /* [n_docs] */,
                  const double *idf /* [V] */, const double *term_ub, const double *block_ub,
                  const uint8_t *post_imp /* [nnz] or NULL */,
                  const int32_t *dense_slot /* [V] or NULL */, const uint8_t *dense_imp,
                  const uint16_t *dense_tf, int64_t dense_stride,
                  double avgdl, double k1, double b, int64_t n_docs, int64_t n_vocab,
                  int64_t id_base, const int32_t *query_terms, int n_queries, int max_terms, int k,
                  int conjunctive, const int32_t *doc_coll, const int32_t *query_coll,
                  double *out_scores, int64_t *out_ids, int32_t *out_counts, void *workspace,
                  size_t workspace_bytes, thr_stream_t stream);

/* a4  graph channel: bounded BFS (<= hops) from seed entities over the
 * entity CSR, then score(chunk) = sum_e conf(e,chunk)/(1+dist(e)) over the
 * entity->chunk mention CSR (ascending entity id, float64).
 * Stands where GraphSearcher.search is called, retrieval.py:316-356
 * (src/voice_agent/rag2/graph_search.py:290-418; score form
 * triple-hybrid-rag/src/triple_hybrid_rag/graph/puppygraph.py:152-167).
 * men_chunk holds GLOBAL chunk ids; only [chunk_base, chunk_base+n_chunks)
 * are scored (document sharding).  ``workspace`` >= thr_graph_workspace_bytes. */
size_t thr_graph_workspace_bytes(int n_queries, int64_t n_entities /* 0: no fallback tier */);
/* Three tiers, no host round trip: small on-chip capacities, full on-chip capacities for the
 * queries that overflowed them, and -- when the TRANSPOSED mention CSR is given (chunk ->
 * (entity, conf) in (entity asc, mention) order; tmen_rowptr indexed by LOCAL chunk) -- a
 * capacity-free walk in global memory for the rest, so THR_FLAG_OVERFLOW never comes back.
 * Without it (tmen_* NULL) such a query keeps THR_FLAG_OVERFLOW and its list is incomplete. */
int thr_graph_topk(const int64_t *ent_rowptr, const int32_t *ent_col, int64_t n_entities,
                   const int64_t *men_rowptr, const int32_t *men_chunk, const float *men_conf,
                   const int64_t *tmen_rowptr /* [n_chunks+1] or NULL */,
                   const int32_t *tmen_ent, const float *tmen_conf,
                   int64_t chunk_base, int64_t n_chunks, const int32_t *query_seeds,
                   int n_queries, int max_seeds, int hops, int k, double *out_scores,
                   int64_t *out_ids, int32_t *out_counts, uint32_t *out_flags, void *workspace,
                   size_t workspace_bytes, thr_stream_t stream);

/* a5+a6  candidate merge + weighted Reciprocal Rank Fusion, bit-for-bit the
 * float64 arithmetic and stable ordering of RAG2Retriever._retrieve_candidates
 * / _fuse_rrf, src/voice_agent/rag2/retrieval.py:203-271, 358-376.
 * Channel inputs are ranked id lists [nq, n_*] (negative id = end of list);
 * a null pointer or n_* == 0 disables the channel. */
int thr_rrf_fuse(const int64_t *lex_ids, int n_lex, const int64_t *sem_ids, int n_sem,
                 const int64_t *graph_ids, int n_graph, int n_queries, double w_lex, double w_sem,
                 double w_graph, int rrf_k, int top_k, int64_t *out_ids /* [nq,top_k] */,
                 double *out_scores /* [nq,top_k] */, int32_t *out_ranks /* [nq,top_k,3] or NULL */,
                 int32_t *out_counts, thr_stream_t stream);

/* f3  the standalone package's RRFFusion on the device (second fusion variant of the reference,
 * triple-hybrid-rag/src/triple_hybrid_rag/core/fusion.py): a channel's table value is
 * ``weight * (1.0 / (60 + rank))`` of the id's LAST rank (:167-185).  two_channels == 0: ``fuse``
 * (:52-165) -- the value is added once per OCCURRENCE of the id in the channel, channels in the
 * order lexical, semantic, graph; two_channels != 0: ``fuse_two_channels(a, b, wa, wb)``
 * (:249-292) with a in the lexical slot and b in the semantic slot (a's value assigned once, b's
 * added per occurrence; graph_ids must be NULL).  Sighting order and stable descending sort as in
 * thr_rrf_fuse; out_ranks (optional) feeds thr_fuse_post. */
int thr_rrf_fuse_standalone(const int64_t *lex_ids, int n_lex, const int64_t *sem_ids, int n_sem,
                            const int64_t *graph_ids, int n_graph, int n_queries, double w_lex,
                            double w_sem, double w_graph, int two_channels, int top_k,
                            int64_t *out_ids, double *out_scores, int32_t *out_ranks /* or NULL */,
                            int32_t *out_counts, thr_stream_t stream);
/* What RRFFusion.fuse does after its sort, and normalize_scores, for a batch of fused lists
 * [n_queries, n] (best first; counts[q] rows valid, NULL = n):
 *   safety_threshold > 0  keep rows whose best channel score max(semantic or 0, lexical or 0,
 *                         graph or 0) is >= it (fusion.py:187-216); the channel scores are read
 *                         at the id's last rank: ranks [n_queries, n, 3] from thr_rrf_fuse*,
 *                         *_scores [n_queries, n_*] float64 (NULL = channel absent);
 *   denoise != 0          with >= 3 rows left: cut = numpy.percentile(scores, 100 * quantile) with
 *                         numpy's linear interpolation, keep scores >= cut (:218-247); the caller
 *                         passes quantile = ((1 - denoise_alpha) * 100) / 100;
 *   top_k > 0             the first top_k of what is left;
 *   normalize != 0        min-max to [0, 1] over what is left, 1.0 when all equal (:294-318).
 * out_* are [n_queries, n] (ids -1 / scores -inf past out_counts[q]). */
int thr_fuse_post(const int64_t *ids, const double *scores, const int32_t *ranks,
                  const int32_t *counts, int n_queries, int n, const double *lex_scores, int n_lex,
                  const double *sem_scores, int n_sem, const double *graph_scores, int n_graph,
                  double safety_threshold, int denoise, double quantile, int normalize, int top_k,
                  int64_t *out_ids, double *out_scores, int32_t *out_counts, thr_stream_t stream);

/* a8  late-interaction rerank: MaxSim(q, c) = sum_i max_j <q_i, d_cj> with
 * float16 token matrices on MFMA (float32 accumulate).
 * Stands where Qwen3VLReranker._rerank_batch_native is called,
 * retrieval.py:427 (src/voice_agent/retrieval/reranker.py:287-354).
 * cand [nq, n_cand] are LOCAL doc indices into dtok, negative = padding
 * (score -inf).  q_tokens, d_tokens multiples of 32; tok_dim multiple of 16. */
int thr_maxsim(const uint16_t *qtok /* f16 [nq,q_tokens,tok_dim] */, int n_queries, int q_tokens,
               const uint16_t *dtok /* f16 [n_docs,d_tokens,tok_dim], or its packed image */,
               int64_t n_docs, int d_tokens, int tok_dim, const int32_t *cand, int n_cand,
               float *out_scores /* [nq,n_cand] */, int dtok_packed, thr_stream_t stream);
/* The same with the candidates as GLOBAL ids (what thr_rrf_fuse returns) of a shard whose local
 * doc 0 has id id_base: local = id - id_base; negative ids and ids of other shards score -inf.
 * (Saves the caller the id arithmetic: nine elementwise launches per batch in PyTorch.) */
int thr_maxsim_ids(const uint16_t *qtok, int n_queries, int q_tokens, const uint16_t *dtok,
                   int64_t n_docs, int d_tokens, int tok_dim, const int64_t *cand_ids /* [nq,n_cand] */,
                   int64_t id_base, int n_cand, float *out_scores, int dtok_packed,
                   thr_stream_t stream);
/* Ordering after the rerank (retrieval.py:449-455): candidate p of query q gets the maximum of
 * scores[l][q][p] over the n_lists lists (document-sharded index: the shard that owns the
 * candidate wrote its MaxSim score, the others -inf; list_stride in floats, 0 = n_queries * n),
 * ``rerank_score or 0`` for a candidate nobody scored, and the first counts[q] candidates
 * (NULL = n) are stably sorted by it, descending -> top_k ids / float64 scores / counts. */
int thr_rerank_order(const float *scores /* [n_lists][nq,n] */, int n_lists, int64_t list_stride,
                     const int64_t *ids /* [nq,n] */, const int32_t *counts /* [nq] or NULL */,
                     int n_queries, int n, int top_k, int64_t *out_ids /* [nq,top_k] */,
                     double *out_scores /* [nq,top_k] */, int32_t *out_counts, thr_stream_t stream);
/* Index build: re-lay the token store fragment-major ([doc][32-token tile][k-step][lane][8
 * halves], the register image of the MFMA operand) so thr_maxsim(dtok_packed = 1) reads 1 KiB
 * contiguous per load instruction.  Out of place; same size as dtok. */
int thr_maxsim_pack(const uint16_t *dtok, int64_t n_docs, int d_tokens, int tok_dim,
                    uint16_t *packed, thr_stream_t stream);

/* Merge step of the multi-GPU path: G per-shard ranked lists -> global top-k under
 * (score desc, id asc).  List l of query q starts at in_*[l * list_stride + q * k_in]
 * (elements); list_stride = 0 means n_queries * k_in, i.e. the [n_lists, n_queries, k_in]
 * layout an all-gather of each rank's [n_queries, k_in] block produces -- a larger stride lets
 * scores and ids be read in place from one gathered [n_lists, 2, n_queries, k_in] tile.
 * Entries with score -inf or a negative id are padding.  Ids must be distinct across lists
 * (disjoint shards).  Lists that arrive ranked (the normal case) are merged without a sort. */
int thr_merge_topk(const double *in_scores, const int64_t *in_ids, int n_queries, int n_lists,
                   int k_in, int64_t list_stride, int k_out, double *out_scores, int64_t *out_ids,
                   int32_t *out_counts, thr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* THR_HIP_H */
