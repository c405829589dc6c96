// Shared device helpers for the gfx950 retrieval kernels (wave64, LDS-resident
// block primitives).  Everything here is written for CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/thr_hip.h"

namespace thr {

constexpr int WAVE = 64;

#define THR_RETURN_IF(cond, code) \
    do {                          \
        if (cond) return (code);  \
    } while (0)

// entry points call this first: an earlier failed HIP call must not be reported as theirs
inline void clear_status() { (void)hipGetLastError(); }

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? THR_OK : (int)e;
}

// ---- order-preserving float <-> uint keys (larger float => larger key) ----
__device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ uint64_t dkey(double d) {
    uint64_t u = (uint64_t)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ double dkey_inv(uint64_t k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// ---- ranked item under the total order (score desc, id asc) ----
struct Item {
    double s;
    int64_t id;
};
__device__ __forceinline__ bool better(double sa, int64_t ia, double sb, int64_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// Block-wide bitonic sort of N (power of two) items held in LDS arrays, best
// first.  Empty slots carry (-inf, INT64_MAX) and sink to the end.
template <int N>
__device__ void bitonic_sort_desc(double* s, int64_t* id) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N; i += blockDim.x) {
                int p = i ^ j;
                if (p > i) {
                    bool up = (i & k) == 0;  // this run sorted best-first
                    double sa = s[i], sb = s[p];
                    int64_t ia = id[i], ib = id[p];
                    bool swap = up ? better(sb, ib, sa, ia) : better(sa, ia, sb, ib);
                    if (swap) {
                        s[i] = sb; s[p] = sa;
                        id[i] = ib; id[p] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Same sort with a runtime size (power of two <= capacity).
__device__ inline void bitonic_sort_desc_n(double* s, int64_t* id, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                int p = i ^ j;
                if (p > i) {
                    bool up = (i & k) == 0;
                    double sa = s[i], sb = s[p];
                    int64_t ia = id[i], ib = id[p];
                    bool swap = up ? better(sb, ib, sa, ia) : better(sa, ia, sb, ib);
                    if (swap) {
                        s[i] = sb; s[p] = sa;
                        id[i] = ib; id[p] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Streaming block top-k: threads push (score,id) pairs that beat the current
// k-th best; when the LDS buffer fills it is sorted and cut back to k.
// All threads of the block must call push()/flush_if_needed() together.
template <int CAP>
struct BlockTopK {
    double* s;     // [CAP] LDS
    int64_t* id;   // [CAP] LDS
    int* count;    // LDS
    double* thr_s; // LDS: k-th best so far (score)
    int64_t* thr_id;
    int k;

    __device__ void init(double* s_, int64_t* id_, int* count_, double* ts, int64_t* ti, int k_) {
        s = s_; id = id_; count = count_; thr_s = ts; thr_id = ti; k = k_;
        for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
            s[i] = -INFINITY;
            id[i] = INT64_MAX;
        }
        if (threadIdx.x == 0) {
            *count = 0;
            *thr_s = -INFINITY;
            *thr_id = INT64_MAX;
        }
        __syncthreads();
    }
    // Call with valid=false for threads that have nothing this round.
    // Precondition (kept by compact()): count + blockDim.x <= CAP.
    __device__ void push(bool valid, double sc, int64_t i) {
        if (valid && better(sc, i, *thr_s, *thr_id)) {
            int p = atomicAdd(count, 1);
            s[p] = sc;
            id[p] = i;
        }
        __syncthreads();
        if (*count + (int)blockDim.x > CAP) compact();
    }
    __device__ void compact() {
        // slots past *count hold (-inf, INT64_MAX) already: sort only the occupied power of two
        int c = *count;
        int n2 = next_pow2(c < 2 ? 2 : c);
        bitonic_sort_desc_n(s, id, n2 < CAP ? n2 : CAP);
        __syncthreads();
        for (int i = threadIdx.x; i < CAP; i += blockDim.x)
            if (i >= k) {
                s[i] = -INFINITY;
                id[i] = INT64_MAX;
            }
        if (threadIdx.x == 0) {
            *count = c < k ? c : k;
            if (c >= k) {
                *thr_s = s[k - 1];
                *thr_id = id[k - 1];
            }
        }
        __syncthreads();
    }
    // Final: sorted best-first in s/id[0..n), returns n = min(k, pushed).
    __device__ int finish() {
        compact();
        return *count;
    }
};

}  // namespace thr
