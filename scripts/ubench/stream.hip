// Micro-benchmark (measurement only, not part of the library): whole-chip LDS-DMA streaming of a 2 GB fp16 chunk
// matrix (1M x 1024) by 256 persistent workgroups, the way the dense scan's memory side does it -- without MFMA,
// fragment reads or epilogue -- to separate "what the access pattern can deliver" from "what the kernel schedule
// loses".  Patterns:
//   LAYOUT 0  row-major [N][1024]: a stage pair = 128-byte column slice of 256 rows (2 KiB apart), issued as the two
//             64-byte halves back to back (what dense_scan_pp2_kernel does today)
//   LAYOUT 1  tiled: a stage pair of a 256-row tile is one contiguous 32 KiB block, every DMA instruction moves
//             1 KiB of consecutive bytes (tile = 512 KiB contiguous)
//   QSIDE  0  chunk stream only     1  + the query-tile stream (512 KiB, L2 resident, same layout as the chunks)
//   BAR       one s_barrier per stage pair (as the scan has)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/stream.hip -o scripts/ubench/stream && scripts/ubench/stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const void *)(SRC), LDS_PTR(DST), 16, 0, 0)

constexpr int kRows = 256, kD = 1024, kRowBytes = kD * 2, kTileBytes = kRows * kRowBytes;   // 512 KiB
constexpr int kPairs = kD / 64;                                                               // 16 stage pairs per tile

template <int LAYOUT, int QSIDE, int DEPTH, bool BAR, int SHARE = 1>
__global__ __launch_bounds__(512) void stream_kernel(const char *__restrict__ X, int64_t n_tiles, const char *__restrict__ Q,
                                                     int *__restrict__ sink, int64_t tile_mod, int rot, int skew) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // SHARE workgroups (consecutive slots of one XCD: block b runs on XCD b % 8) walk the same tile stream, each with its
    // own query tile -- the B = 256 * SHARE shape of the scan
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int qt = jx % SHARE;
    const int stream = (jx / SHARE) * 8 + xcd;
    const int n_streams = gridDim.x / SHARE;
    Q += (int64_t)qt * kTileBytes;
    // per-lane source offsets inside a tile for this wave's two instructions of a stage
    int off[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int piece = (it * 8 + wave) * 64 + lane;
        if (LAYOUT == 0) {
            const int r = piece >> 2, p4 = piece & 3;
            const int ls = p4 ^ ((r >> 2) & 3);
            off[it] = r * kRowBytes + ls * 16;
        } else {
            off[it] = piece * 16;                       // linear inside the 16 KiB stage image
        }
    }
    char *const my_dst = lds + wave * 1024;
    constexpr int kABytes = 6 * 16384, kBBase = kABytes, kBBytes = 4 * 16384;
    int a_dst = 0, b_dst = 0;
    int64_t done = 0;
    // skew: workgroup qt of a stream starts qt * skew * ~1000 cycles late; rot: it walks the K pairs starting at qt * rot
    for (int i = 0; i < qt * skew; ++i) __builtin_amdgcn_s_sleep(16);
    const int kp0 = (qt * rot) % kPairs;
    for (int64_t t = stream; t < n_tiles; t += n_streams) {
        const char *xt = X + (t % tile_mod) * (int64_t)kTileBytes;
        for (int kq = 0; kq < kPairs; ++kq) {
            const int kp = (kq + kp0) % kPairs;
            if (QSIDE == 2) {
            } else if (LAYOUT == 0) {
                const char *s = xt + kp * 128;
                GLDS(s + off[0], my_dst + a_dst);
                GLDS(s + off[0] + 64, my_dst + a_dst + 16384);
                GLDS(s + off[1], my_dst + a_dst + 8192);
                GLDS(s + off[1] + 64, my_dst + a_dst + 16384 + 8192);
            } else {
                const char *s = xt + kp * 32768;
                GLDS(s + off[0], my_dst + a_dst);
                GLDS(s + off[1], my_dst + a_dst + 8192);
                GLDS(s + 16384 + off[0], my_dst + a_dst + 16384);
                GLDS(s + 16384 + off[1], my_dst + a_dst + 16384 + 8192);
            }
            a_dst += 32768;
            if (a_dst >= kABytes) a_dst = 0;
            if (QSIDE) {
                if (LAYOUT == 0) {
                    const char *s = Q + kp * 128;
                    GLDS(s + off[0], my_dst + kBBase + b_dst);
                    GLDS(s + off[0] + 64, my_dst + kBBase + b_dst + 16384);
                    GLDS(s + off[1], my_dst + kBBase + b_dst + 8192);
                    GLDS(s + off[1] + 64, my_dst + kBBase + b_dst + 16384 + 8192);
                } else {
                    const char *s = Q + kp * 32768;
                    GLDS(s + off[0], my_dst + kBBase + b_dst);
                    GLDS(s + off[1], my_dst + kBBase + b_dst + 8192);
                    GLDS(s + 16384 + off[0], my_dst + kBBase + b_dst + 16384);
                    GLDS(s + 16384 + off[1], my_dst + kBBase + b_dst + 16384 + 8192);
                }
                b_dst = (b_dst + 32768) & (kBBytes - 1);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
            if (BAR) asm volatile("s_barrier" ::: "memory");
            ++done;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x * 16] == 77 && done == 12345) sink[0] = 1;
}

template <int LAYOUT, int QSIDE, int DEPTH, bool BAR, int SHARE = 1>
void run(const char *X, int64_t n_tiles, const char *Q, int *sink, int wgs, int64_t tile_mod = 1 << 30, int rot = 0, int skew = 0) {
    const size_t lds_bytes = 160 * 1024;
    auto kern = stream_kernel<LAYOUT, QSIDE, DEPTH, BAR, SHARE>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds_bytes, 0, X, n_tiles, Q, sink, tile_mod, rot, skew);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n_tiles * kTileBytes * SHARE;   // bytes through L1 -> LDS on the chunk side
    printf("rot=%d skew=%-3d mod=%-5lld share=%d layout=%s qside=%d vmcnt=%2d barrier=%d wgs=%d: %7.3f ms  chunk side %5.2f TB/s  (L1->LDS total %5.2f TB/s, %5.1f B/clk/CU at 1.9 GHz)  %s\n",
           rot, skew, (long long)(tile_mod > n_tiles ? 0 : tile_mod), SHARE, LAYOUT ? "tiled   " : "rowmajor", QSIDE, DEPTH, (int)BAR, wgs, best, bytes / best * 1e-9,
           bytes * (QSIDE == 1 ? 2 : 1) / best * 1e-9,
           bytes * (QSIDE == 1 ? 2 : 1) / best * 1e3 / wgs / 1.9e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int64_t n_tiles = 3904;                            // 3904 x 256 rows = 999424 chunks, 2.05 GB
    char *X, *Q; int *sink;
    hipMalloc(&X, (size_t)n_tiles * kTileBytes + (1 << 20));
    hipMemset(X, 1, (size_t)n_tiles * kTileBytes + (1 << 20));
    hipMalloc(&Q, 4 * kTileBytes);
    hipMemset(Q, 1, 4 * kTileBytes);
    hipMalloc(&sink, 64);
    const int wgs = 256;
    const int64_t big = 1 << 30;
#define VAR(L, Q_, SH)                                                  \
    run<L, Q_, 8, true, SH>(X, n_tiles, Q, sink, wgs, big, 0, 0);       \
    run<L, Q_, 8, true, SH>(X, n_tiles, Q, sink, wgs, big, 4, 0);       \
    run<L, Q_, 8, true, SH>(X, n_tiles, Q, sink, wgs, big, 1, 0);       \
    run<L, Q_, 8, true, SH>(X, n_tiles, Q, sink, wgs, big, 0, 2);       \
    run<L, Q_, 8, true, SH>(X, n_tiles, Q, sink, wgs, big, 0, 8);       \
    run<L, Q_, 8, true, SH>(X, n_tiles, Q, sink, wgs, big, 0, 32);
    VAR(0, 0, 4)
    VAR(0, 1, 4)
    VAR(1, 0, 4)
    VAR(1, 1, 4)
    VAR(0, 1, 2)
    return 0;
}
