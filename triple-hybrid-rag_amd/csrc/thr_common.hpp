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
            // thread t takes pair t of the stage: i = t with a zero bit inserted at j, partner i | j
            for (int t = threadIdx.x; t < N / 2; t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                bool up = (i & k) == 0;  // this run sorted best-first
                double sa = s[i], sb = s[p];
                int64_t ia = id[i], ib = id[p];
                bool swap = up ? better(sb, ib, sa, ia) : better(sa, ia, sb, ib);
                if (swap) {
                    s[i] = sb; s[p] = sa;
                    id[i] = ib; id[p] = ia;
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
            for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                bool up = (i & k) == 0;
                double sa = s[i], sb = s[p];
                int64_t ia = id[i], ib = id[p];
                bool swap = up ? better(sb, ib, sa, ia) : better(sa, ia, sb, ib);
                if (swap) {
                    s[i] = sb; s[p] = sa;
                    id[i] = ib; id[p] = ia;
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

// Streaming block top-k: threads push (score,id) pairs that beat the current threshold; when the
// LDS buffer fills it is cut back.  All threads of the block (THREADS of them) must call push() /
// compact() / finish() together.
//
// compact() does NOT sort (a 1024-entry bitonic sort of (double, int64) pairs is 55 barrier
// stages -- profiling the BM25 kernel showed it spending more than half of a short query's time
// there): a 4-pass radix select over the top 32 bits of the order-preserving score key finds the
// prefix P of the k-th best score, the entries with prefix >= P are kept (at least k: the k best
// and the few that share the k-th's sign, exponent and 20 mantissa bits), and the threshold
// becomes the smallest score with that prefix -- a LOWER bound of the k-th best, which is all a
// pruning threshold has to be.  Should that not free a block's worth of slots (a flood of
// near-equal scores), the exact sort-and-cut runs instead.  finish() sorts what is left, exactly.
template <int CAP, int THREADS>
struct BlockTopK {
    static constexpr int PER = (CAP + THREADS - 1) / THREADS;   // buffer entries per thread (blockDim.x == THREADS)
    double* s;     // [CAP] LDS
    int64_t* id;   // [CAP] LDS
    int* count;    // LDS
    double* thr_s; // LDS: lower bound of the k-th best so far (score)
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
        // one LDS atomic per wave (not per entry: they all hit the same address)
        const bool ok = valid && better(sc, i, *thr_s, *thr_id);
        const unsigned long long m = __ballot(ok);
        if (m) {
            const int lane = threadIdx.x & (WAVE - 1);
            int base = 0;
            if (lane == 0) base = atomicAdd(count, __popcll(m));
            base = __shfl(base, 0, WAVE);
            if (ok) {
                const int p = base + __popcll(m & ((1ull << lane) - 1ull));
                s[p] = sc;
                id[p] = i;
            }
        }
        __syncthreads();
        // every thread takes the same decision: the count is read between two barriers, so no
        // wave can be back in push() (and add to it) while another still reads it
        const int c = *count;
#ifndef THR_PUSH_SINGLE_BARRIER
        __syncthreads();
#endif
        if (c + (int)blockDim.x > CAP) compact();
    }
    __device__ void compact() {
        __shared__ int hist[256];
        __shared__ int sel[2];      // bin of the k-th, entries in the bins above it
        __shared__ int kept;
        const int c = *count;       // (uniform: nobody adds to it between a barrier and this call)
        if (c < k) return;
        const int lane = threadIdx.x & (WAVE - 1);
        uint32_t prefix = 0;
        int remaining = k;
#pragma unroll 1
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (threadIdx.x < 256) hist[threadIdx.x] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < c; i += blockDim.x) {
                const uint32_t key = (uint32_t)(dkey(s[i]) >> 32);
                if (shift == 24 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
                    atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            if (threadIdx.x < WAVE) {   // lane l owns bins 255 - 4l .. 252 - 4l, largest first
                const int top = 255 - 4 * lane;
                const int h0 = hist[top], h1 = hist[top - 1], h2 = hist[top - 2], h3 = hist[top - 3];
                const int mine = h0 + h1 + h2 + h3;
                int incl = mine;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) {
                    const int v = __shfl_up(incl, o, WAVE);
                    if (lane >= o) incl += v;
                }
                const int excl = incl - mine;
                if (excl < remaining && remaining <= incl) {
                    int cum = excl, bin = top;
                    if (cum + h0 < remaining) {
                        cum += h0; bin = top - 1;
                        if (cum + h1 < remaining) {
                            cum += h1; bin = top - 2;
                            if (cum + h2 < remaining) { cum += h2; bin = top - 3; }
                        }
                    }
                    sel[0] = bin;
                    sel[1] = cum;
                }
            }
            __syncthreads();
            prefix |= (uint32_t)sel[0] << shift;
            remaining -= sel[1];
        }
        // keep the entries whose prefix is >= that of the k-th best, packed to the front
        double rs[PER];
        int64_t ri[PER];
        bool keep[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = (int)threadIdx.x + u * (int)blockDim.x;
            keep[u] = false;
            if (i < c) {
                rs[u] = s[i];
                ri[u] = id[i];
                keep[u] = (uint32_t)(dkey(rs[u]) >> 32) >= prefix;
            }
        }
        if (threadIdx.x == 0) kept = 0;
        __syncthreads();   // every entry is in a register before any slot is overwritten
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const unsigned long long m = __ballot(keep[u]);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&kept, __popcll(m));
                base = __shfl(base, 0, WAVE);
                if (keep[u]) {
                    const int p = base + __popcll(m & ((1ull << lane) - 1ull));
                    s[p] = rs[u];
                    id[p] = ri[u];
                }
            }
        }
        __syncthreads();
        const int nk = kept;
        if (nk + (int)blockDim.x > CAP) {   // not enough room won: the exact cut, on the original entries
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = (int)threadIdx.x + u * (int)blockDim.x;
                if (i < c) {
                    s[i] = rs[u];
                    id[i] = ri[u];
                }
            }
            __syncthreads();
            sort_and_cut();
            return;
        }
        for (int i = nk + (int)threadIdx.x; i < c; i += blockDim.x) {
            s[i] = -INFINITY;
            id[i] = INT64_MAX;
        }
        if (threadIdx.x == 0) {
            *count = nk;
            *thr_s = dkey_inv((uint64_t)prefix << 32);   // smallest score with this prefix
            *thr_id = INT64_MAX;                          // (a score equal to it still passes)
        }
        __syncthreads();
    }
    // exact: sorted best-first, cut to k, threshold = the k-th best itself
    __device__ void sort_and_cut() {
        // slots past *count hold (-inf, INT64_MAX) already: sort only the occupied power of two
        int c = *count;
        int n2 = next_pow2(c < 2 ? 2 : c);
        __syncthreads();
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
        sort_and_cut();
        return *count;
    }
};

}  // namespace thr
