// Micro-benchmark (measurement only, not part of the library): does more waves per SIMD hide the LDS-DMA issue blocking
// that the dense scan suffers from?  256 persistent workgroups stream the 2 GB chunk matrix with the scan's DMA volume
// (32 KiB per stage: 16 KiB chunk side from the tiled layout, 16 KiB query side from an L2-resident tile) and issue the
// scan's MFMA volume (128 x v_mfma_f32_32x32x16_f16 per stage and CU-workgroup) from register operands -- no fragment
// reads, no epilogue -- with one barrier per stage and a counted vmcnt.  NW = 8 waves (two per SIMD, 16 MFMAs + 4 DMA
// instructions per wave and stage, 128 accumulator registers: the scan's shape) or NW = 16 waves (four per SIMD, 8 MFMAs
// + 2 DMA instructions, 64 accumulator registers).  MODE: 1 MFMA only, 2 DMA only, 3 both.  SHARE: workgroups per
// chunk stream (1 = the B = 256 shape, 4 = the B = 1024 shape).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_dma.hip -o scripts/ubench/mfma_dma && scripts/ubench/mfma_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const void *)(SRC), LDS_PTR(DST), 16, 0, 0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileBytes = 256 * 1024 * 2;        // 512 KiB: 256 rows x 1024 halves
constexpr int kStages = 32;                       // stages of 32 halves per tile

// DIST 4: the chunk side does not go through LDS at all -- every wave loads its 2 KiB of the stage straight into
// registers (4-deep ring, the data is the MFMA A operand), the query side stays LDS-DMA.
// DIST: who issues the 32 DMA instructions of a stage -- 0 every wave its share, 1 the second half of the waves only
// (one MFMA-only and one MFMA+DMA wave per SIMD), 2 four extra loader waves that issue no MFMA at all (NW + 4 waves)
template <int NW, int MODE, int SHARE, int DIST>
__global__ __launch_bounds__((NW + (DIST == 2 ? 4 : 0)) * 64) void k(const char *__restrict__ X, int64_t n_tiles, const char *__restrict__ Q,
                                             float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NT = 64 / NW;                   // accumulator tiles of 32 x 32; 2 * NT MFMAs per wave and stage
    constexpr int NL = (DIST == 0 || DIST == 3 || DIST == 4) ? NW : DIST == 1 ? NW / 2 : 4;   // waves that issue DMA
    constexpr int DPW = 32 / NL;                  // DMA instructions per issuing wave and stage (half chunk side, half query side)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = (DIST == 0 || DIST == 3 || DIST == 4) ? true : DIST == 1 ? (wave >= NW / 2) : (wave >= NW);
    const bool mfma_first = DIST == 3 && wave < NW / 2;         // 3: ping-pong order (one wave of each kind per SIMD)
    const bool computer = DIST == 2 ? (wave < NW) : true;
    const int lw = (DIST == 0 || DIST == 3 || DIST == 4) ? wave : DIST == 1 ? wave - NW / 2 : wave - NW;   // rank among the issuing waves
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int qt = jx % SHARE;
    const int stream = (jx / SHARE) * 8 + xcd;
    const int n_streams = gridDim.x / SHARE;
    Q += (int64_t)qt * kTileBytes;
    const long long t_start = clock64();
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    half8 a, b;
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = (_Float16)(0.01f * (lane + u)); b[u] = (_Float16)(0.02f * (lane - u)); }
    // this wave's share of a 16 KiB stage image: instructions i = 0 .. DPW/2-1 move bytes [(i * NW + wave) * 1024, +1024)
    constexpr int kA = 5 * 16384, kB = 4 * 16384;
    int a_dst = 0, b_dst = 0;
#define DO_DMA()                                                                              \
            if ((MODE & 2) && loader) {                                                          \
                _Pragma("unroll") for (int i = 0; i < DPW / 2; ++i) {                            \
                    const int off = (i * NL + lw) * 1024 + lane * 16;                            \
                    GLDS(xt + s * 16384 + off, lds + a_dst + (i * NL + lw) * 1024);              \
                    GLDS(Q + s * 16384 + off, lds + kA + b_dst + (i * NL + lw) * 1024);          \
                }                                                                                \
                a_dst += 16384; if (a_dst == kA) a_dst = 0;                                      \
                b_dst = (b_dst + 16384) & (kB - 1);                                              \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);
#define DO_MFMA()                                                                             \
            if ((MODE & 1) && computer) {                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                    \
                    _Pragma("unroll") for (int u = 0; u < NT; ++u)                               \
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);  \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);
#define STAGE_END()                                                                           \
            if ((MODE & 2) && loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW) : "memory"); \
            asm volatile("s_barrier" ::: "memory");
    if (DIST == 4) {
        // flattened (tile, stage) sequence; register ring of 4 stages x 2 fragments
        half8 ra[4][2];
        const int64_t tiles_mine = (n_tiles - stream + n_streams - 1) / n_streams;
        const int64_t total = tiles_mine * kStages;
        auto src = [&](int64_t g) -> const char * {
            const int64_t t = stream + (g / kStages) * n_streams;
            return X + t * (int64_t)kTileBytes + (g % kStages) * 16384 + wave * 2048 + lane * 16;
        };
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            ra[g][0] = *reinterpret_cast<const half8 *>(src(g));
            ra[g][1] = *reinterpret_cast<const half8 *>(src(g) + 1024);
        }
        for (int64_t g0 = 0; g0 < total; g0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u;
                const int s = (int)(g % kStages);
                if ((MODE & 2) && g + 3 < total) {
                    ra[(u + 3) & 3][0] = *reinterpret_cast<const half8 *>(src(g + 3));
                    ra[(u + 3) & 3][1] = *reinterpret_cast<const half8 *>(src(g + 3) + 1024);
                }
                if (MODE & 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) GLDS(Q + s * 16384 + (i * NW + wave) * 1024 + lane * 16, lds + kA + b_dst + (i * NW + wave) * 1024);
                    b_dst = (b_dst + 16384) & (kB - 1);
                }
                if (MODE & 1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int w = 0; w < NT; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_f16((MODE & 2) ? ra[u][j] : a, b, acc[w], 0, 0, 0);
                }
                asm volatile("s_barrier" ::: "memory");
            }
        }
    } else if (mfma_first) {
        for (int64_t t = stream; t < n_tiles; t += n_streams) {
            const char *xt = X + t * (int64_t)kTileBytes;
            for (int s = 0; s < kStages; ++s) { DO_MFMA() DO_DMA() STAGE_END() }
        }
    } else {
        for (int64_t t = stream; t < n_tiles; t += n_streams) {
            const char *xt = X + t * (int64_t)kTileBytes;
            for (int s = 0; s < kStages; ++s) { DO_DMA() DO_MFMA() STAGE_END() }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[t][r];
    if (keep == 1.2345e-30f || lds[threadIdx.x * 16] == 77) sink[0] = keep;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(sink)[1] = clock64() - t_start;   // shader clocks of workgroup 0
}

template <int NW, int MODE, int SHARE, int DIST = 0>
void run(const char *X, int64_t n_tiles, const char *Q, float *sink) {
    auto kern = k<NW, MODE, SHARE, DIST>;
    const size_t lds_bytes = 9 * 16384;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    long long cyc = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3((NW + (DIST == 2 ? 4 : 0)) * 64), lds_bytes, 0, X, n_tiles, Q, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) { best = ms; hipMemcpy(&cyc, reinterpret_cast<long long *>(sink) + 1, 8, hipMemcpyDeviceToHost); }
    }
    const double flops = (double)n_tiles * SHARE * 2.0 * 256 * 256 * 1024;
    printf("waves=%2d dist=%d mode=%s share=%d: %7.3f ms   %6.2f PF   chunk side %5.2f TB/s   clock %4.0f MHz   %s\n", NW, DIST,
           MODE == 1 ? "mfma    " : MODE == 2 ? "dma     " : "mfma+dma", SHARE, best, (MODE & 1) ? flops / best * 1e-12 : 0.0,
           (MODE & 2) ? (double)n_tiles * SHARE * kTileBytes / best * 1e-9 : 0.0, (double)cyc / best * 1e-3,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    const int64_t n_tiles = 3840;
    char *X, *Q; float *sink;
    hipMalloc(&X, (size_t)n_tiles * kTileBytes + (1 << 20));
    hipMemset(X, 1, (size_t)n_tiles * kTileBytes + (1 << 20));
    hipMalloc(&Q, 4 * kTileBytes);
    hipMemset(Q, 1, 4 * kTileBytes);
    hipMalloc(&sink, 64);
#define ALL(NW, SH, D) run<NW, 1, SH, D>(X, n_tiles, Q, sink); run<NW, 2, SH, D>(X, n_tiles, Q, sink); run<NW, 3, SH, D>(X, n_tiles, Q, sink);
    ALL(8, 4, 3) ALL(8, 4, 4)
    ALL(8, 1, 3) ALL(8, 1, 4)
    ALL(8, 4, 3) ALL(8, 4, 4)
    return 0;
}
