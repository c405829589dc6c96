// a1 query-embedding post-processing and the index-build norm helper.
// Stands where truncate_matryoshka / normalize_l2 run on the host in the
// reference (src/voice_agent/rag2/embedder.py:31-68).
#include "thr_common.hpp"

namespace thr {

// one wave per row: prefix-truncate to store_dim, L2-normalise in float32.
// The squared norm is accumulated in float64 and rounded once (numpy's own
// float32 reduction order is unspecified; tests hold this to 2 ulp of it).
__global__ __launch_bounds__(256) void embed_postproc(const float* __restrict__ full, int n,
                                                      int full_dim, int out_dim,
                                                      float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = full + (int64_t)row * full_dim;
    double ss = 0.0;
    for (int i = lane; i < out_dim; i += WAVE) {
        double v = (double)x[i];
        ss += v * v;
    }
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, WAVE);
    const float nrm = (float)sqrt(ss);
    float* o = out + (int64_t)row * out_dim;
    for (int i = lane; i < out_dim; i += WAVE) o[i] = nrm > 0.f ? x[i] / nrm : x[i];
}

// one thread per row: ||d|| = sqrt(sequential float64 sum of squares) -- the
// oracle's doc_norms_f64 -- and float32 1/||d|| for the fp32 scan (0 = no embedding).
__global__ __launch_bounds__(256) void doc_norms(const float* __restrict__ docs, int64_t n,
                                                 int dim, double* __restrict__ dnorm,
                                                 float* __restrict__ inv_norm) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const float4* a = reinterpret_cast<const float4*>(docs + row * dim);
    double s = 0.0;
    for (int i = 0; i < dim / 4; ++i) {
        float4 x = a[i];
        s = __dadd_rn(s, __dmul_rn((double)x.x, (double)x.x));
        s = __dadd_rn(s, __dmul_rn((double)x.y, (double)x.y));
        s = __dadd_rn(s, __dmul_rn((double)x.z, (double)x.z));
        s = __dadd_rn(s, __dmul_rn((double)x.w, (double)x.w));
    }
    const double nrm = __dsqrt_rn(s);
    dnorm[row] = nrm;
    inv_norm[row] = nrm > 0.0 ? (float)__ddiv_rn(1.0, nrm) : 0.f;
}

}  // namespace thr

using namespace thr;

extern "C" int thr_embed_postproc(const float* full, int n, int full_dim, int store_dim, float* out,
                                  thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!full || !out || n <= 0 || full_dim <= 0 || store_dim <= 0, THR_ERR_INVALID);
    const int out_dim = full_dim < store_dim ? full_dim : store_dim;
    hipLaunchKernelGGL(embed_postproc, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, full, n,
                       full_dim, out_dim, out);
    return launch_status();
}

extern "C" int thr_doc_norms(const float* docs, int64_t n_docs, int dim, double* dnorm,
                             float* inv_norm, thr_stream_t stream) {
    clear_status();
    THR_RETURN_IF(!docs || !dnorm || !inv_norm || n_docs <= 0, THR_ERR_INVALID);
    THR_RETURN_IF(dim <= 0 || dim % 4 != 0, THR_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(doc_norms, dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, docs, n_docs, dim, dnorm, inv_norm);
    return launch_status();
}
