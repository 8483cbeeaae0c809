// Micro-benchmark (measurement only, not part of the library): how fast can one CU pull an L2-resident operand
// tile into LDS?   mode 0: global_load_lds_dwordx4 (LDS-DMA)   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
// mode 2: global_load_dwordx4 -> VGPR only.   `rb` = contiguous bytes per row piece (64, 128 or 1024); rows are
// 2048 bytes apart like the fp16 chunk matrix at d = 1024.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/ldsdma.hip -o /tmp/ldsdma && /tmp/ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
typedef int int4v __attribute__((ext_vector_type(4)));

template <int MODE, int INFLIGHT, bool PAIR = false>
__global__ __launch_bounds__(512) void pull(const char *__restrict__ src, size_t wg_stride, int rb, int iters,
                                            int *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *base = src + (size_t)blockIdx.x * wg_stride;
    const int ppr = rb / 16;                                 // 16-byte pieces per row
    // one "stage" = 8 waves x INFLIGHT instructions x 1 KiB
    int4v keep = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int kcol = (it & 7) * rb;                      // walk along the row like K-steps do
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            int piece = (u * 8 + wave) * 64 + lane;
            int row = piece / ppr, p = piece % ppr;
            const char *g = base + (size_t)row * 2048 + kcol % 2048 + p * 16;
            if (PAIR) {   // rb == 64: instructions 2v and 2v+1 fetch the two halves of the same 128-byte lines back to back
                piece = ((u >> 1) * 8 + wave) * 64 + lane;
                row = piece / 4; p = piece % 4;
                g = base + (size_t)row * 2048 + ((it & 3) * 128) + (u & 1) * 64 + p * 16;
            }
            char *l = lds + ((it & 1) * INFLIGHT * 8 + (u * 8 + wave)) * 1024;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const void *)g, LDS_PTR(l), 16, 0, 0);
            } else {
                const int4v v = *reinterpret_cast<const int4v *>(g);
                if (MODE == 1) *reinterpret_cast<int4v *>(l + lane * 16) = v;
                else keep += v;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // previous stage landed
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (keep[0] + keep[1] + keep[2] + keep[3] == 0x12345678 || lds[threadIdx.x] == 77) sink[0] = 1;
}

template <int MODE, int INFLIGHT, bool PAIR = false>
void run(const char *name, const char *src, size_t wg_stride, int rb, int *sink, int wgs) {
    const int iters = 2000;
    const size_t lds_bytes = 2 * INFLIGHT * 8 * 1024;
    hipFuncSetAttribute((const void *)pull<MODE, INFLIGHT, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((pull<MODE, INFLIGHT, PAIR>), dim3(wgs), dim3(512), lds_bytes, 0, src, wg_stride, rb, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * iters * INFLIGHT * 8 * 1024;
    printf("%-34s rb=%4d inflight=%d wgs=%d: %7.3f ms  %7.1f GB/s/CU  %6.2f TB/s chip  (%s)\n", name, rb, INFLIGHT, wgs, ms,
           bytes / wgs / ms * 1e-6, bytes / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int wgs = 256;
    const size_t wg_stride = 1 << 20;                        // 512 rows x 2048 B per workgroup: stays in L2/MALL
    char *src; int *sink;
    hipMalloc(&src, wg_stride * wgs + (1 << 20));
    hipMemset(src, 1, wg_stride * wgs + (1 << 20));
    hipMalloc(&sink, 64);
    for (int rb : {64, 128, 1024}) {
        run<0, 2>("lds-dma x4", src, wg_stride, rb, sink, wgs);
        run<0, 4>("lds-dma x4", src, wg_stride, rb, sink, wgs);
        run<0, 8>("lds-dma x4", src, wg_stride, rb, sink, wgs);
        run<1, 4>("vgpr load + ds_write_b128", src, wg_stride, rb, sink, wgs);
        run<1, 8>("vgpr load + ds_write_b128", src, wg_stride, rb, sink, wgs);
        run<2, 8>("vgpr load only", src, wg_stride, rb, sink, wgs);
    }
    run<0, 8, true>("lds-dma x4 paired halves", src, wg_stride, 64, sink, wgs);
    run<0, 8, true>("lds-dma x4 paired halves, shared", src, 0, 64, sink, wgs);
    run<0, 4, true>("lds-dma x4 paired halves, shared", src, 0, 64, sink, wgs);
    // every workgroup reads the SAME 1 MiB (pure L2 hits)
    for (int rb : {64, 128}) {
        run<0, 8>("lds-dma x4, shared source", src, 0, rb, sink, wgs);
        run<1, 8>("vgpr+ds_write, shared source", src, 0, rb, sink, wgs);
    }
    return 0;
}
