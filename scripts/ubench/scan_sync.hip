// Micro-benchmark (measurement only, not part of the library): how should the eight waves of the dense scan's mainloop
// synchronise?  Same work per 32-half stage and CU as dense_scan_pp3_kernel at its 256 x 256 tile: 32 LDS-DMA instructions
// (16 KiB chunk side from HBM, 16 KiB query side from an L2-resident tile), 8 x 12 ds_read_b128 fragment reads out of the
// rings (FRAG = 1; the data is what the MFMAs consume, random fp16 values) and 8 x 16 v_mfma_f32_32x32x16_f16 -- no
// epilogue.  SYNC:
//   0  strict alternation, two s_barrier per stage: one wave group in its matrix segment (MFMAs + the next stage's fragment
//      reads), the other in its memory segment (its four DMA instructions + the counted wait)  -- what pp3 does
//   1  one s_barrier per stage, ping-pong order (group 0 MFMA then DMA, group 1 DMA then MFMA)
//   2  NO barrier: every wave runs free.  A wave confirms its own pieces of stage g+1 (counted vmcnt) with an LDS add on
//      landed[g+1], reads the stage once all eight have confirmed, and reports done[g+1] behind its fragment reads (LDS
//      operations of one wave execute in order); a ring slot is re-filled once done[] of the stage it held is complete.
//      A slow wave never waits for a fast one (whatever it polls was signalled earlier in the fast wave's program order).
//   PLACE (SYNC 2): 0 the four DMA instructions ahead of the MFMAs, 1 one behind every fourth MFMA
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/scan_sync.hip -o scripts/ubench/scan_sync && scripts/ubench/scan_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const void *)(SRC), LDS_PTR(DST), 16, 0, 0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileBytes = 256 * 1024 * 2;        // 512 KiB: 256 rows x 1024 halves, as 32 stage images of 16 KiB
constexpr int kStages = 32;
constexpr int kA = 5 * 16384, kB = 4 * 16384;     // ring bytes
constexpr int kFlags = kA + kB;                   // landed[16], done[16] (uint32)
constexpr int kLds = kFlags + 256;

#define FLAG(IDX) (*reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t *>(LDS_PTR(lds + kFlags + 4 * (IDX))))
// one arrival on a flag word: a plain LDS add (ds_add_u32, no return) written as inline assembly -- for an LDS atomic the
// compiler waits vmcnt(0) first (it cannot tell the word from the LDS-DMA destinations), which would drain the DMA queue
#define ARRIVE(IDX)                                                                                                  \
    do {                                                                                                             \
        const uint32_t addr_ = (uint32_t)(uintptr_t)LDS_PTR(lds + kFlags + 4 * (IDX));                               \
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr_), "v"(1u) : "memory");                          \
    } while (0)

// MODE 0: one binade (+-[2^-6, 2^-5), random mantissa and sign); MODE 1: approximately N(0, 1/32) like a unit-norm row of
// 1024 components (sum of four uniforms), the bench corpus' distribution -- the matrix pipe's power draw depends on the data
__global__ void fill_kernel(uint16_t *p, size_t n, uint32_t seed, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        if (mode == 0) {
            p[i] = (uint16_t)(((h & 1u) << 15) | (9u << 10) | ((h >> 8) & 0x3ffu));
        } else {
            const float u = (float)(h & 255u) + (float)((h >> 8) & 255u) + (float)((h >> 16) & 255u) + (float)(h >> 24);   // mean 510, sd 147.8
            const _Float16 v = (_Float16)((u - 510.f) * (1.f / (147.8f * 32.f)));
            p[i] = *reinterpret_cast<const uint16_t *>(&v);
        }
    }
}

template <int SYNC, int FRAG, int PLACE, int SHARE>
__global__ __launch_bounds__(512) void k(const char *__restrict__ X, int64_t n_tiles, const char *__restrict__ Q,
                                         float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wave_n = wave & 3;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int qt = jx % SHARE;
    const int stream = (jx / SHARE) * 8 + xcd;
    const int n_streams = gridDim.x / SHARE;
    Q += (int64_t)qt * kTileBytes;
    const int tiles_mine = (int)((n_tiles - stream + n_streams - 1) / n_streams);
    const int total = tiles_mine * kStages;
    const long long t_start = clock64();
    if (threadIdx.x < 64) FLAG(threadIdx.x) = 0;
    __syncthreads();

    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    half8 fa[4][2], fb[2][2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const _Float16 va = (_Float16)(0.01f * (lane + u)), vb = (_Float16)(0.02f * (lane - u));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) fa[mt][j][u] = va;
            fb[0][j][u] = vb; fb[1][j][u] = vb;
        }
    }
    // fragment read addresses (the scan's: 64-byte rows, 16-byte slot XOR-swizzled by row)
    const int l31 = lane & 31, hh = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    const int a_rd0 = (grp * 128 + l31) * 64 + ((hh ^ sw) << 4), a_rd1 = (grp * 128 + l31) * 64 + (((2 + hh) ^ sw) << 4);
    const int b_rd0 = kA + (wave_n * 64 + l31) * 64 + ((hh ^ sw) << 4), b_rd1 = kA + (wave_n * 64 + l31) * 64 + (((2 + hh) ^ sw) << 4);

    // source of this lane for stage s (flattened): tile stream + (s / 32) * n_streams, stage image s % 32
    const char *xa = X + (int64_t)stream * kTileBytes + wave * 1024 + lane * 16;     // + 8192 for the second instruction
    const char *qb = Q + wave * 1024 + lane * 16;
    int s_a = 0, s_b = 0;                          // next stage to issue, per operand
    int a_dst = 0, b_dst = 0;
    char *const my_dst = lds + wave * 1024;
    const int64_t a_tile_jump = (int64_t)n_streams * kTileBytes - (int64_t)kTileBytes;
#define ISSUE_A()                                                                  \
    do {                                                                           \
        if (s_a < total) {                                                         \
            GLDS(xa, my_dst + a_dst);                                              \
            GLDS(xa + 8192, my_dst + a_dst + 8192);                                \
            xa += 16384;                                                           \
            ++s_a;                                                                 \
            if ((s_a & (kStages - 1)) == 0) xa += a_tile_jump;                     \
            a_dst += 16384; if (a_dst == kA) a_dst = 0;                            \
        }                                                                          \
    } while (0)
#define ISSUE_B()                                                                  \
    do {                                                                           \
        if (s_b < total) {                                                         \
            const char *q_ = qb + (s_b & (kStages - 1)) * 16384;                   \
            GLDS(q_, my_dst + kA + b_dst);                                         \
            GLDS(q_ + 8192, my_dst + kA + b_dst + 8192);                           \
            ++s_b;                                                                 \
            b_dst = (b_dst + 16384) & (kB - 1);                                    \
        }                                                                          \
    } while (0)
    int fa_off = 0, fb_off = 0;                    // ring offsets of the stage the NEXT fragment reads take
#define ADVANCE_READ() do { fa_off += 16384; if (fa_off == kA) fa_off = 0; fb_off = (fb_off + 16384) & (kB - 1); } while (0)
#define READ_ALL()                                                                                     \
    do {                                                                                               \
        if (FRAG) {                                                                                    \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                         \
                fa[mt][0] = *reinterpret_cast<const half8 *>(lds + a_rd0 + fa_off + mt * 2048);        \
                fa[mt][1] = *reinterpret_cast<const half8 *>(lds + a_rd1 + fa_off + mt * 2048);        \
            }                                                                                          \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                         \
                fb[nt][0] = *reinterpret_cast<const half8 *>(lds + b_rd0 + fb_off + nt * 2048);        \
                fb[nt][1] = *reinterpret_cast<const half8 *>(lds + b_rd1 + fb_off + nt * 2048);        \
            }                                                                                          \
        }                                                                                              \
        ADVANCE_READ();                                                                                \
    } while (0)
// matrix segment of the current stage; every fragment register is re-loaded with the next stage's contents behind the last
// MFMA that reads it (DMAQ: one DMA instruction behind every fourth MFMA, SYNC 2 / PLACE 1)
#define HALF(J, PA_, PB_, DMAQ)                                                                        \
    do {                                                                                               \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                             \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[0][J], acc[mt][0], 0, 0, 0); \
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[1][J], acc[mt][1], 0, 0, 0); \
            if (FRAG) fa[mt][J] = *reinterpret_cast<const half8 *>(PA_ + mt * 2048);                   \
            if (DMAQ && mt == 1) { if (J == 0) ISSUE_A(); else ISSUE_B(); }                            \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
        if (FRAG) {                                                                                    \
            fb[0][J] = *reinterpret_cast<const half8 *>(PB_);                                          \
            fb[1][J] = *reinterpret_cast<const half8 *>(PB_ + 2048);                                   \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#define COMPUTE(DMAQ)                                                                                  \
    do {                                                                                               \
        const char *pa0_ = lds + a_rd0 + fa_off, *pa1_ = lds + a_rd1 + fa_off;                         \
        const char *pb0_ = lds + b_rd0 + fb_off, *pb1_ = lds + b_rd1 + fb_off;                         \
        HALF(0, pa0_, pb0_, DMAQ);                                                                     \
        HALF(1, pa1_, pb1_, DMAQ);                                                                     \
        ADVANCE_READ();                                                                                \
    } while (0)
#define BARRIER() do { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    unsigned spins = 0;
    if (SYNC == 2) {
        // chunk side 4 stages ahead (ring 5), query side 3 (ring 4); order per iteration: A(g+4), B(g+3)
        ISSUE_A(); ISSUE_B(); ISSUE_A(); ISSUE_B(); ISSUE_A(); ISSUE_B(); ISSUE_A();      // A0 B0 A1 B1 A2 B2 A3
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                  // A0 B0 landed
        ARRIVE(0);
        for (;;) { if (FLAG(0) >= 8u || ++spins > (1u << 22)) break; __builtin_amdgcn_s_sleep(1); }
        READ_ALL();                                                                         // stage 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ARRIVE(16);
        for (int g = 0; g < total; ++g) {
            // own pieces of stage g+1: everything but the four instructions of iteration g-1 (A(g+3), B(g+2)) has landed
            if (g + 1 < total) {
                if (g + 3 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int i1 = (g + 1) & 15;
                const uint32_t want1 = 8u * (uint32_t)(((g + 1) >> 4) + 1);
                ARRIVE(i1);
                // slots of A(g+4) / B(g+3) held stage g-1: every wave must have finished reading it
                const uint32_t want0 = g >= 1 ? 8u * (uint32_t)(((g - 1) >> 4) + 1) : 0u;
                const int i0 = (g - 1) & 15;
                for (;;) {
                    const uint32_t l_ = FLAG(i1), d_ = FLAG(16 + i0);
                    if ((l_ >= want1 && d_ >= want0) || ++spins > (1u << 22)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (PLACE == 0) { ISSUE_A(); ISSUE_B(); __builtin_amdgcn_sched_barrier(0); COMPUTE(0); }
            else { COMPUTE(1); }
            if (g + 1 < total) ARRIVE(16 + ((g + 1) & 15));
        }
    } else {
        // barrier variants: the same prefetch distances (chunk side 4 stages ahead, query side 3); before the barrier that ends
        // iteration g everything but that iteration's four instructions has landed, i.e. stage g+2 is complete
        ISSUE_A(); ISSUE_B(); ISSUE_A(); ISSUE_B(); ISSUE_A(); ISSUE_B(); ISSUE_A();      // A0 B0 A1 B1 A2 B2 A3
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                  // stages 0 and 1 landed
        BARRIER();
        READ_ALL();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BARRIER();
        if (SYNC == 0) {
            if (grp == 0) {
                for (int g = 0; g < total; ++g) {
                    COMPUTE(0);
                    BARRIER();
                    ISSUE_A(); ISSUE_B();
                    if (g + 4 < total) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    BARRIER();
                }
            } else {
                for (int g = 0; g < total; ++g) {
                    ISSUE_A(); ISSUE_B();
                    __builtin_amdgcn_sched_barrier(0);
                    BARRIER();
                    COMPUTE(0);
                    if (g + 4 < total) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    BARRIER();
                }
            }
        } else {
            if (grp == 0) {
                for (int g = 0; g < total; ++g) {
                    COMPUTE(0);
                    ISSUE_A(); ISSUE_B();
                    if (g + 4 < total) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    BARRIER();
                }
            } else {
                for (int g = 0; g < total; ++g) {
                    ISSUE_A(); ISSUE_B();
                    __builtin_amdgcn_sched_barrier(0);
                    COMPUTE(0);
                    if (g + 4 < total) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    BARRIER();
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float keep = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[mt][nt][r];
    if (keep == 1.2345e-30f) sink[0] = keep;
    if (spins > (1u << 22)) sink[4] = 1.f;                                              // a poll gave up: the run is invalid
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(sink)[1] = clock64() - t_start;
}

template <int SYNC, int FRAG, int PLACE, int SHARE>
void run(const char *X, int64_t n_tiles, const char *Q, float *sink, const char *what) {
    auto kern = k<SYNC, FRAG, PLACE, SHARE>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, sum = 0;
    long long cyc = 0;
    hipMemset(sink, 0, 64);
    const int reps = 6;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), kLds, 0, X, n_tiles, Q, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0) sum += ms;
        if (rep > 0 && ms < best) { best = ms; hipMemcpy(&cyc, reinterpret_cast<long long *>(sink) + 1, 8, hipMemcpyDeviceToHost); }
    }
    float bad = 0;
    hipMemcpy(&bad, sink + 4, 4, hipMemcpyDeviceToHost);
    const double flops = (double)n_tiles * SHARE * 2.0 * 256 * 256 * 1024;
    printf("sync=%d frag=%d place=%d share=%d %-34s best %7.3f ms  mean %7.3f ms  %5.2f PF  chunk side %5.2f TB/s  clock %4.0f MHz  %s%s\n",
           SYNC, FRAG, PLACE, SHARE, what, best, sum / (reps - 1), flops / best * 1e-12, (double)n_tiles * SHARE * kTileBytes / best * 1e-9,
           (double)cyc / best * 1e-3, hipGetErrorString(hipGetLastError()), bad != 0.f ? "  POLL TIMEOUT: INVALID" : "");
    fflush(stdout);
}

int main() {
    const int64_t n_tiles = 3840;
    char *X, *Q; float *sink;
    hipMalloc(&X, (size_t)n_tiles * kTileBytes + (1 << 20));
    hipMalloc(&Q, 4 * kTileBytes);
    hipMalloc(&sink, 64);
    for (int mode = 0; mode < 2; ++mode) {
        printf("--- data: %s ---\n", mode == 0 ? "one binade, random mantissa and sign" : "approximately normal, sd 1/32 (the bench corpus' distribution)");
        fill_kernel<<<4096, 256>>>((uint16_t *)X, ((size_t)n_tiles * kTileBytes + (1 << 20)) / 2, 1u, mode);
        fill_kernel<<<256, 256>>>((uint16_t *)Q, (size_t)4 * kTileBytes / 2, 7u, mode);
        hipDeviceSynchronize();
        for (int round = 0; round < 2; ++round) {
            run<0, 1, 0, 1>(X, n_tiles, Q, sink, "strict alternation, 2 barriers");
            run<1, 1, 0, 1>(X, n_tiles, Q, sink, "one barrier, ping-pong order");
            run<2, 1, 0, 1>(X, n_tiles, Q, sink, "free-running, DMA ahead of MFMAs");
            run<2, 1, 1, 1>(X, n_tiles, Q, sink, "free-running, DMA between MFMAs");
            run<0, 0, 0, 1>(X, n_tiles, Q, sink, "strict, register operands");
        }
        for (int round = 0; round < 2; ++round) {
            run<0, 1, 0, 4>(X, n_tiles, Q, sink, "strict alternation, 2 barriers");
            run<1, 1, 0, 4>(X, n_tiles, Q, sink, "one barrier, ping-pong order");
            run<2, 1, 1, 4>(X, n_tiles, Q, sink, "free-running, DMA between MFMAs");
        }
    }
    return 0;
}
