// Shared device helpers for the gfx950 retrieval kernels (wave64, LDS-resident selection).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ERH_WAVE 64

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ERH_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// Dense candidate record written by the scan epilogue: fp32 MFMA score + document index.
struct __attribute__((aligned(8))) ErhCand {
    float s;
    int32_t idx;
};

// ---- order-preserving float <-> uint32 (total order, -inf lowest among real numbers) --------
__device__ __forceinline__ uint32_t erh_f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float erh_ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
// (score, idx) -> 64-bit key whose descending order is (score desc, idx asc). key 0 = "empty".
__device__ __forceinline__ uint64_t erh_key32(float s, int32_t idx) {
    return ((uint64_t)erh_f2ord(s) << 32) | (uint64_t)(0xffffffffu - (uint32_t)idx);
}
__device__ __forceinline__ float erh_key32_score(uint64_t k) { return erh_ord2f((uint32_t)(k >> 32)); }
__device__ __forceinline__ int32_t erh_key32_idx(uint64_t k) { return (int32_t)(0xffffffffu - (uint32_t)k); }

// ---- block-wide bitonic sort, descending, n a power of two, data in LDS -----------------------
// Every thread of the block must call it (contains __syncthreads()).
// Compare-exchange steps with distance j <= 64 stay inside an aligned block of 128 elements, which one wave owns
// (64 pairs, one per lane): those steps need no workgroup barrier -- a wave's LDS operations execute in order, so
// only the compiler has to be kept from reordering them.  All phases k <= 128 therefore run back to back inside
// the wave; the larger phases pay one barrier per step with j >= 128 plus one around their wave-local tail.
// For n = 2048 that is about 20 barriers instead of 66.
#define ERH_WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <typename T>
__device__ __forceinline__ void erh_bitonic_step(T *a, int t, int j, int k) {
    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // t-th compare-exchange pair of this (k, j) step
    const int p = i | j;
    const bool desc = ((i & k) == 0);
    const T x = a[i], y = a[p];
    const bool sw = desc ? (x < y) : (y < x);
    if (sw) { a[i] = y; a[p] = x; }
}

template <typename T>
__device__ __forceinline__ void erh_bitonic_desc(T *a, int n) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const int half = n >> 1;
    __syncthreads();
    // phases k = 2 .. min(n, 128): entirely wave-local
    for (int t0 = (tid & ~63); t0 < half; t0 += nth) {
        const int t = t0 + (tid & 63);
        for (int k = 2; k <= n && k <= 128; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (t < half) erh_bitonic_step<T>(a, t, j, k);
                ERH_WAVE_LDS_FENCE();
            }
    }
    for (int k = 256; k <= n; k <<= 1) {
        for (int j = k >> 1; j >= 128; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < half; t += nth) erh_bitonic_step<T>(a, t, j, k);
        }
        __syncthreads();
        for (int t0 = (tid & ~63); t0 < half; t0 += nth) {
            const int t = t0 + (tid & 63);
            for (int j = 64; j > 0; j >>= 1) {
                if (t < half) erh_bitonic_step<T>(a, t, j, k);
                ERH_WAVE_LDS_FENCE();
            }
        }
    }
    __syncthreads();
}

// Two-array variant for (double score, int idx) records: order (score desc, idx asc).
__device__ __forceinline__ bool erh_rec_before(double s1, int32_t i1, double s2, int32_t i2) {
    return (s1 > s2) || (s1 == s2 && i1 < i2);
}
template <typename ST>
__device__ __forceinline__ void erh_bitonic_rec_step(ST *s, int32_t *ix, int t, int j, int k) {
    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const int p = i | j;
    const bool desc = ((i & k) == 0);
    const ST sx = s[i], sy = s[p];
    const int32_t ixx = ix[i], iyy = ix[p];
    // "x before y" in the final descending order?
    const bool x_first = (sx > sy) || (sx == sy && ixx < iyy);
    const bool y_first = (sy > sx) || (sx == sy && iyy < ixx);
    const bool sw = desc ? y_first : x_first;
    if (sw) { s[i] = sy; s[p] = sx; ix[i] = iyy; ix[p] = ixx; }
}
template <typename ST>
__device__ __forceinline__ void erh_bitonic_rec_desc(ST *s, int32_t *ix, int n) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const int half = n >> 1;
    __syncthreads();
    for (int t0 = (tid & ~63); t0 < half; t0 += nth) {
        const int t = t0 + (tid & 63);
        for (int k = 2; k <= n && k <= 128; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (t < half) erh_bitonic_rec_step<ST>(s, ix, t, j, k);
                ERH_WAVE_LDS_FENCE();
            }
    }
    for (int k = 256; k <= n; k <<= 1) {
        for (int j = k >> 1; j >= 128; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < half; t += nth) erh_bitonic_rec_step<ST>(s, ix, t, j, k);
        }
        __syncthreads();
        for (int t0 = (tid & ~63); t0 < half; t0 += nth) {
            const int t = t0 + (tid & 63);
            for (int j = 64; j > 0; j >>= 1) {
                if (t < half) erh_bitonic_rec_step<ST>(s, ix, t, j, k);
                ERH_WAVE_LDS_FENCE();
            }
        }
    }
    __syncthreads();
}

// Row placement of the chunk matrix: original row o is stored at position (o * mul) mod n (mul coprime with n), so
// that every prefix of the stored order is an evenly spread sample of the original order (api.hip: erh_set_dense).
__device__ __forceinline__ int64_t erh_mulmod(int64_t a, int64_t mul, int64_t n) {
    return (int64_t)(((uint64_t)a * (uint64_t)mul) % (uint64_t)n);        // a, mul < 2^31
}

__device__ __forceinline__ int erh_next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

__device__ __forceinline__ int erh_lane() { return (int)(threadIdx.x & 63); }
