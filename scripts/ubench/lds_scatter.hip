// Micro-benchmark (measurement only, not part of the library): what does one random-address LDS scatter cost per
// wave-instruction on a full CU (1024 threads, 128 KiB of accumulators), for the candidate ways to add a posting's
// payload to its document's sum?
//   0 ds_add_f32 (no return)     1 ds_add_rtn_f32     2 ds_add_u32 (no return)     3 ds_add_rtn_u32
//   4 ds_read_b32 + add + ds_write_b32 (dependent chain, what the order-keeping scans do)
//   5 ds_read_b32 x8 in flight, then 8 adds + ds_write_b32 (independent postings of one token)
//   6 ds_add_u32 + ds_max_u32 pair      7 ds_add_rtn_u64 (fixed point 32.32)   8 ds_add_u64 (no return)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_scatter.hip -o scripts/ubench/lds_scatter && scripts/ubench/lds_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kSlots = 32768;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(1024) void scatter(int iters, unsigned long long *__restrict__ out, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *accf = reinterpret_cast<float *>(smem);
    uint32_t *accu = reinterpret_cast<uint32_t *>(smem);
    unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(smem);
    for (int i = threadIdx.x; i < kSlots; i += 1024) accu[i] = 0;
    __syncthreads();
    const long long t0 = clock64();
    float keep = 0.f;
    uint32_t seed = mix(threadIdx.x * 2654435761u + blockIdx.x);
    for (int it = 0; it < iters; ++it) {
        uint32_t s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { seed = seed * 1664525u + 1013904223u; s[u] = (seed >> 9) & (kSlots - 1); }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&accf[s[u]], 1.25f);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) keep += atomicAdd(&accf[s[u]], 1.25f);
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&accu[s[u]], 77u);
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) keep += (float)atomicAdd(&accu[s[u]], 77u);
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                volatile float *p = &accf[s[u]];
                const float o = *p;
                *p = o + 1.25f;
            }
        } else if (MODE == 5) {
            float o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = accf[s[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) accf[s[u]] = o[u] + 1.25f;
            asm volatile("" ::: "memory");
        } else if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { atomicAdd(&accu[s[u]], 77u); atomicMax(&accu[s[u] ^ 1], 77u); }
        } else if (MODE == 7) {
#pragma unroll
            for (int u = 0; u < 8; ++u) keep += (float)atomicAdd(&acc64[s[u] >> 1], 77ull);
        } else if (MODE == 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&acc64[s[u] >> 1], 77ull);
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (keep == 1.2345f || accf[threadIdx.x] == 3.3f) sink[0] = keep;
}

template <int MODE>
void run(const char *name, unsigned long long *d_out, float *sink) {
    const int iters = 256, wgs = 256;
    hipFuncSetAttribute((const void *)scatter<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(scatter<MODE>, dim3(wgs), dim3(1024), kSlots * 4, 0, iters, d_out, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[256];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < wgs; ++i) cyc += (double)h[i];
    cyc /= wgs;
    const double wave_instr = 16.0 * iters * 8;      // scatter wave-instructions per CU (pairs count once)
    printf("%-58s %8.3f ms  %10.0f cycles/WG  %7.1f cycles per wave-instruction per CU  (%.2f lanes/clk)\n", name, ms, cyc,
           cyc / wave_instr, 64.0 * wave_instr / cyc);
}

int main() {
    unsigned long long *d_out;
    float *sink;
    hipMalloc(&d_out, 256 * 8);
    hipMalloc(&sink, 64);
    run<0>("0 ds_add_f32", d_out, sink);
    run<1>("1 ds_add_rtn_f32", d_out, sink);
    run<2>("2 ds_add_u32", d_out, sink);
    run<3>("3 ds_add_rtn_u32", d_out, sink);
    run<4>("4 read-add-write chain (volatile)", d_out, sink);
    run<5>("5 8 reads in flight, then 8 writes", d_out, sink);
    run<6>("6 ds_add_u32 + ds_max_u32 pair", d_out, sink);
    run<7>("7 ds_add_rtn_u64", d_out, sink);
    run<8>("8 ds_add_u64", d_out, sink);
    return 0;
}
