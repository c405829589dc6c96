// Index build on the device: (doc, term, tf) rows -> the CSR inverted index thr_bm25_topk reads.
//
// Stands where the reference's ingestion leaves the lexical side to PostgreSQL: it inserts
// rag_child_chunks rows (src/voice_agent/rag2/ingest.py:361-470) and the `tsv` generated column +
// GIN index (database/migrations/20260114_rag2_schema.sql:146-148, 171-172) turn the text into
// posting lists inside the database.  Here the tokenised rows -- one (doc, term) pair per distinct
// term of a chunk, or one per token occurrence: pairs that repeat add up -- become
//
//     rowptr  int64 [V + 1]   postings of term t are [rowptr[t], rowptr[t + 1])
//     post_doc int32 [nnz]    ascending within a term
//     post_tf  int32 [nnz]    term frequency
//     doclen   float32 [n]    sum of the term frequencies of a doc
//     df       int64 [V]      postings per term (LOCAL to this shard: the caller all-reduces it
//                             over the shards for the global idf)
//
// Pipeline (one stream, no host round trip; the number of postings is left in device memory):
//   pack keys (term << 32 | doc), pairs out of range get the largest key and are counted out;
//   rocprim::radix_sort_pairs over the key bits that can differ; rocprim::reduce_by_key adds the
//   frequencies of equal keys; one kernel writes post_doc / post_tf, one kernel finds every term's
//   first posting by binary search in the sorted keys (rowptr), df = its differences; the doc
//   lengths are summed as the keys are packed (integer sums in float32: exact below 2^24).
// rocPRIM supplies the sort / segmented reduce (the ROCm device-primitive library, as hipBLASLt
// would supply a plain GEMM); the kernels around them are written here.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "thr_common.hpp"

namespace thr {

constexpr uint64_t LB_DROPPED = ~0ull;

__global__ __launch_bounds__(256) void lb_pack_keys(const int32_t* __restrict__ doc,
                                                    const int32_t* __restrict__ term,
                                                    const int32_t* __restrict__ tf, int64_t n_pairs,
                                                    int64_t n_docs, int64_t n_vocab,
                                                    uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                    float* __restrict__ doclen) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const int64_t d = doc[i], t = term[i];
    const int32_t f = tf ? tf[i] : 1;
    const bool in_doc = d >= 0 && d < n_docs && f > 0;
    const bool ok = in_doc && t >= 0 && t < n_vocab;
    keys[i] = ok ? ((uint64_t)t << 32) | (uint64_t)d : LB_DROPPED;
    vals[i] = ok ? f : 0;
    // a token outside the vocabulary (a shard built against the global one) is still a token of
    // its chunk: it counts toward the length (integer sums in float32: exact below 2^24)
    if (in_doc) atomicAdd(&doclen[d], (float)f);
}

// postings of the unique keys; *n_unique comes from reduce_by_key (device memory)
__global__ __launch_bounds__(256) void lb_write_postings(const uint64_t* __restrict__ ukeys,
                                                         const int32_t* __restrict__ sums,
                                                         const int64_t* __restrict__ n_unique,
                                                         int32_t* __restrict__ post_doc,
                                                         int32_t* __restrict__ post_tf,
                                                         int64_t* __restrict__ nnz_out) {
    int64_t n = *n_unique;
    if (n > 0 && ukeys[n - 1] == LB_DROPPED) --n;   // (the run of dropped pairs sorts last)
    if (blockIdx.x == 0 && threadIdx.x == 0) *nnz_out = n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = ukeys[i];
        const int32_t d = (int32_t)(k & 0xFFFFFFFFull);
        post_doc[i] = d;
        post_tf[i] = sums[i];
    }
}

// rowptr[t] = first unique key >= (t << 32); rowptr[V] = nnz
__global__ __launch_bounds__(256) void lb_rowptr(const uint64_t* __restrict__ ukeys,
                                                 const int64_t* __restrict__ nnz, int64_t n_vocab,
                                                 int64_t* __restrict__ rowptr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_vocab) return;
    const int64_t n = *nnz;
    const uint64_t want = (uint64_t)t << 32;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ukeys[mid] < want) lo = mid + 1; else hi = mid;
    }
    rowptr[t] = t == n_vocab ? n : lo;
}
__global__ __launch_bounds__(256) void lb_df(const int64_t* __restrict__ rowptr, int64_t n_vocab,
                                             int64_t* __restrict__ df) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_vocab) df[t] = rowptr[t + 1] - rowptr[t];
}

struct LbLayout {
    size_t off_keys, off_vals, off_skeys, off_svals, off_ukeys, off_sums, off_count, off_tmp, tmp_bytes, total;
};
static int key_bits(int64_t n_vocab) {
    int b = 0;
    while (((int64_t)1 << b) < n_vocab) ++b;
    return 32 + (b < 1 ? 1 : b) + 1 > 64 ? 64 : 32 + (b < 1 ? 1 : b) + 1;   // (+1: the dropped key's bits above V)
}
static LbLayout lb_layout(int64_t n_pairs, int64_t n_vocab) {
    LbLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t n = (size_t)n_pairs;
    L.off_keys = take(sizeof(uint64_t) * n);
    L.off_vals = take(sizeof(int32_t) * n);
    L.off_skeys = take(sizeof(uint64_t) * n);
    L.off_svals = take(sizeof(int32_t) * n);
    L.off_ukeys = take(sizeof(uint64_t) * n);
    L.off_sums = take(sizeof(int32_t) * n);
    L.off_count = take(sizeof(int64_t) * 2);
    size_t t_sort = 0, t_red = 0;
    (void)rocprim::radix_sort_pairs((void*)nullptr, t_sort, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (const int32_t*)nullptr, (int32_t*)nullptr, n, 0u, (unsigned)key_bits(n_vocab));
    (void)rocprim::reduce_by_key((void*)nullptr, t_red, (const uint64_t*)nullptr, (const int32_t*)nullptr, n,
                                 (uint64_t*)nullptr, (int32_t*)nullptr, (int64_t*)nullptr);
    L.tmp_bytes = t_sort > t_red ? t_sort : t_red;
    L.off_tmp = take(L.tmp_bytes);
    L.total = off;
    return L;
}

}  // namespace thr

using namespace thr;

extern "C" size_t thr_lexical_build_workspace_bytes(int64_t n_pairs, int64_t n_vocab) {
    if (n_pairs <= 0 || n_vocab <= 0) return 0;
    return lb_layout(n_pairs, n_vocab).total;
}

extern "C" int thr_lexical_build(const int32_t* doc, const int32_t* term, const int32_t* tf,
                                 int64_t n_pairs, int64_t n_docs, int64_t n_vocab, int64_t* rowptr,
                                 int32_t* post_doc, int32_t* post_tf, float* doclen, int64_t* df,
                                 int64_t* nnz_out, void* workspace, size_t workspace_bytes,
                                 thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!doc || !term || !rowptr || !post_doc || !post_tf || !doclen || !df || !nnz_out || !workspace,
                  THR_ERR_INVALID);
    THR_RETURN_IF(n_pairs <= 0 || n_docs <= 0 || n_vocab <= 0 || n_docs > ((int64_t)1 << 31) - 1 ||
                      n_vocab > ((int64_t)1 << 31) - 1,
                  THR_ERR_INVALID);
    const LbLayout L = lb_layout(n_pairs, n_vocab);
    THR_RETURN_IF(workspace_bytes < L.total, THR_ERR_WORKSPACE);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    uint64_t* keys = (uint64_t*)(ws + L.off_keys);
    int32_t* vals = (int32_t*)(ws + L.off_vals);
    uint64_t* skeys = (uint64_t*)(ws + L.off_skeys);
    int32_t* svals = (int32_t*)(ws + L.off_svals);
    uint64_t* ukeys = (uint64_t*)(ws + L.off_ukeys);
    int32_t* sums = (int32_t*)(ws + L.off_sums);
    int64_t* count = (int64_t*)(ws + L.off_count);
    hipError_t e = hipMemsetAsync(doclen, 0, sizeof(float) * (size_t)n_docs, st);
    if (e != hipSuccess) return (int)e;
    const unsigned blocks = (unsigned)((n_pairs + 255) / 256);
    hipLaunchKernelGGL(lb_pack_keys, dim3(blocks), dim3(256), 0, st, doc, term, tf, n_pairs, n_docs, n_vocab, keys, vals, doclen);
    size_t tmp = L.tmp_bytes;
    e = rocprim::radix_sort_pairs(ws + L.off_tmp, tmp, keys, skeys, vals, svals, (size_t)n_pairs, 0u,
                                  (unsigned)key_bits(n_vocab), st);
    if (e != hipSuccess) return (int)e;
    tmp = L.tmp_bytes;
    e = rocprim::reduce_by_key(ws + L.off_tmp, tmp, skeys, svals, (size_t)n_pairs, ukeys, sums, count,
                               rocprim::plus<int32_t>(), rocprim::equal_to<uint64_t>(), st);
    if (e != hipSuccess) return (int)e;
    int wblocks = (int)(blocks < 4096u ? blocks : 4096u);
    hipLaunchKernelGGL(lb_write_postings, dim3(wblocks), dim3(256), 0, st, ukeys, sums, count, post_doc, post_tf,
                       nnz_out);
    hipLaunchKernelGGL(lb_rowptr, dim3((unsigned)((n_vocab + 1 + 255) / 256)), dim3(256), 0, st, ukeys, nnz_out,
                       n_vocab, rowptr);
    hipLaunchKernelGGL(lb_df, dim3((unsigned)((n_vocab + 255) / 256)), dim3(256), 0, st, rowptr, n_vocab, df);
    return launch_status();
}
