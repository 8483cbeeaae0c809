// Micro-benchmark (measurement only, not part of the library): would a 384 x 256 workgroup tile raise the ceiling of the dense
// scan's mainloop?  (17 % fewer L1 -> LDS fill bytes per MAC than the 256 x 256 tile of
// dense_scan_pp3_kernel.)  Same structure as scan_sync.hip's strict alternation -- two wave groups, one in its matrix segment
// while the other is in its memory segment, two s_barrier per 32-half stage, LDS-DMA fills, fragment reads out of the rings
// with the scan's swizzle, no epilogue -- at two tile shapes:
//   ROWS = 256: 4 x 2 accumulator tiles per wave (128 VGPRs), fragment registers for a whole stage (64), rings of 5 / 4 stages
//               (chunk side 4 stages ahead, query side 3)                                     -- what pp3 does
//   ROWS = 384: 6 x 2 accumulator tiles per wave (192 VGPRs), fragment registers for HALF a stage (32: each is re-loaded behind
//               its last MFMA with the next half-stage's contents), rings of 4 x 24 KiB / 3 x 16 KiB (3 / 2 stages ahead): what
//               fits 256 VGPRs per wave and 160 KiB of LDS
// Both scan the same number of chunk rows against one 256-query tile (SHARE 1, the 256-query configuration) or four (SHARE 4).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/scan_tile384.hip -o scripts/ubench/scan_tile384 && scripts/ubench/scan_tile384
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const void *)(SRC), LDS_PTR(DST), 16, 0, 0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kStages = 32;                       // 1024 halves per row = 32 stages of 32 halves
constexpr int kQTileBytes = 256 * 1024 * 2;       // one 256-query tile: 32 stage images of 16 KiB

// approximately N(0, 1/32) like a unit-norm row of 1024 components (the bench corpus' distribution: the matrix pipe's power
// draw depends on the data), or one binade with random mantissa and sign
__global__ void fill_kernel(uint16_t *p, size_t n, uint32_t seed, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        if (mode == 0) {
            p[i] = (uint16_t)(((h & 1u) << 15) | (9u << 10) | ((h >> 8) & 0x3ffu));
        } else {
            const float u = (float)(h & 255u) + (float)((h >> 8) & 255u) + (float)((h >> 16) & 255u) + (float)(h >> 24);
            const _Float16 v = (_Float16)((u - 510.f) * (1.f / (147.8f * 32.f)));
            p[i] = *reinterpret_cast<const uint16_t *>(&v);
        }
    }
}

#define WAITVM(N_)                                                                                     \
    do {                                                                                               \
        if ((N_) == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                     \
        else if ((N_) == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                \
        else if ((N_) == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                \
        else if ((N_) == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");                \
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                               \
    } while (0)

template <int ROWS, int SHARE, int NA, int NB>
__global__ __launch_bounds__(512) void k(const char *__restrict__ X, int64_t n_tiles, const char *__restrict__ Q,
                                         float *__restrict__ sink) {
    constexpr int MT = ROWS / 64;                                  // 32-row accumulator tiles per wave (two wave groups)
    constexpr int kAStage = ROWS * 64;                             // bytes of one chunk-side stage image (64-byte rows)
    constexpr int kA = NA * kAStage, kB = NB * 16384;
    constexpr int kAIns = kAStage / 8192;                          // DMA instructions per wave and chunk-side stage
    // DMA instructions of an iteration that may still be in flight at its end: the next matrix segment re-loads from stage
    // g + 2, whose query side was issued THIS iteration when that ring is only three deep (then it is issued first and
    // waited for; only the chunk side stays in flight)
    constexpr bool B_FIRST = NB == 3;
    constexpr int kInFlight = B_FIRST ? kAIns : kAIns + 2;
    constexpr int kPrologueWait = (NA - 3) * kAIns + (NB - 3) * 2;  // instructions behind stages 0 and 1 in the prologue
    static_assert(NA >= 4 && NB >= 3, "ring depths");
    constexpr int kTileBytes = ROWS * 1024 * 2;
    constexpr bool HALFSET = ROWS != 256;                          // fragment registers for half a stage
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wave_n = wave & 3;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int qt = jx % SHARE;
    const int stream = (jx / SHARE) * 8 + xcd;
    const int n_streams = gridDim.x / SHARE;
    Q += (int64_t)qt * kQTileBytes;
    const int tiles_mine = (int)((n_tiles - stream + n_streams - 1) / n_streams);
    const int total = tiles_mine * kStages;
    const long long t_start = clock64();

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    constexpr int NJ = HALFSET ? 1 : 2;
    half8 fa[MT][NJ], fb[2][NJ];
    const int l31 = lane & 31, hh = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    // fragment read addresses (the scan's: 64-byte rows, 16-byte slot XOR-swizzled by row); half J reads slot (2 J + hh) ^ sw
    // (the second half's slot is the first one's with bit 1 flipped, i.e. address ^ 32: re-computed where it is needed -- by an
    // asm statement the compiler cannot hoist -- instead of held in registers; the 384-row variant has none to spare)
    const int a_adr0 = (grp * (ROWS / 2) + l31) * 64 + ((hh ^ sw) << 4), b_adr0 = kA + (wave_n * 64 + l31) * 64 + ((hh ^ sw) << 4);
#define ADR1(DST, SRC) asm volatile("v_xor_b32 %0, 32, %1" : "=v"(DST) : "v"(SRC))

    // (wave-uniform pointers + one per-lane offset: the loads take the scalar-base form, no 64-bit address registers)
    const char *xa = X + (int64_t)stream * kTileBytes + wave * 1024;
    const char *qb = Q + wave * 1024;
    const int lane_off = lane * 16;
    int s_a = 0, s_b = 0, a_dst = 0, b_dst = 0;
    char *const my_dst = lds + wave * 1024;
    const int64_t a_tile_jump = (int64_t)n_streams * kTileBytes - (int64_t)kTileBytes;
#define ISSUE_A()                                                                  \
    do {                                                                           \
        if (s_a < total) {                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < kAIns; ++i_) GLDS(xa + i_ * 8192 + lane_off, my_dst + a_dst + i_ * 8192); \
            xa += kAStage;                                                         \
            ++s_a;                                                                 \
            if ((s_a & (kStages - 1)) == 0) xa += a_tile_jump;                     \
            a_dst += kAStage; if (a_dst == kA) a_dst = 0;                          \
        }                                                                          \
    } while (0)
#define ISSUE_B()                                                                  \
    do {                                                                           \
        if (s_b < total) {                                                         \
            const char *q_ = qb + (s_b & (kStages - 1)) * 16384;                   \
            GLDS(q_ + lane_off, my_dst + kA + b_dst);                              \
            GLDS(q_ + 8192 + lane_off, my_dst + kA + b_dst + 8192);                \
            ++s_b;                                                                 \
            b_dst += 16384; if (b_dst == kB) b_dst = 0;                            \
        }                                                                          \
    } while (0)
    int fa_off = 0, fb_off = 0;                    // ring offsets of the stage the fragment reads take next
#define ADVANCE_READ() do { fa_off += kAStage; if (fa_off == kA) fa_off = 0; fb_off += 16384; if (fb_off == kB) fb_off = 0; } while (0)
#define BARRIER() do { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    // one K = 16 half of the current stage: MT x 2 MFMAs; behind its last MFMA every fragment register is re-loaded from
    // (PA_, PB_) -- the same half of the NEXT stage (two register sets, ROWS = 256) or the next half-stage (one set)
#define HALF(J, PA_, PB_)                                                                              \
    do {                                                                                               \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                            \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[0][J], acc[mt][0], 0, 0, 0); \
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[1][J], acc[mt][1], 0, 0, 0); \
            fa[mt][J] = *reinterpret_cast<const half8 *>(PA_ + mt * 2048);                             \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
        fb[0][J] = *reinterpret_cast<const half8 *>(PB_);                                              \
        fb[1][J] = *reinterpret_cast<const half8 *>(PB_ + 2048);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
    // matrix segment of stage g.  Two sets: both halves re-load from stage g + 1 (fa_off points there).  One set: the first
    // half re-loads with the second half of stage g (cur_a / cur_b), the second half with the first half of stage g + 1.
#define COMPUTE()                                                                                      \
    do {                                                                                               \
        int a1_, b1_;                                                                                  \
        ADR1(a1_, a_adr0);                                                                             \
        ADR1(b1_, b_adr0);                                                                             \
        if (HALFSET) {                                                                                 \
            HALF(0, (lds + a1_ + cur_a), (lds + b1_ + cur_b));                                         \
            HALF(0, (lds + a_adr0 + fa_off), (lds + b_adr0 + fb_off));                                 \
            cur_a = __builtin_amdgcn_readfirstlane(fa_off); cur_b = __builtin_amdgcn_readfirstlane(fb_off);   /* (scalars) */ \
        } else {                                                                                       \
            HALF(0, (lds + a_adr0 + fa_off), (lds + b_adr0 + fb_off));                                 \
            HALF(NJ - 1, (lds + a1_ + fa_off), (lds + b1_ + fb_off));                                  \
        }                                                                                              \
        ADVANCE_READ();                                                                                \
    } while (0)

    // prologue: chunk side NA - 1 stages ahead, query side NB - 1; stages 0 and 1 complete before the first reads
    int cur_a = 0, cur_b = 0;                      // (one set) ring offsets of the stage whose second half is read next
#pragma unroll
    for (int s_ = 0; s_ < (NA > NB ? NA : NB) - 1; ++s_) {         // A0 B0 A1 B1 A2 [B2] [A3]
        if (s_ < NA - 1) ISSUE_A();
        if (s_ < NB - 1) ISSUE_B();
    }
    WAITVM(kPrologueWait);
    BARRIER();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        fa[mt][0] = *reinterpret_cast<const half8 *>(lds + a_adr0 + mt * 2048);
        if (!HALFSET) fa[mt][NJ - 1] = *reinterpret_cast<const half8 *>(lds + (a_adr0 ^ 32) + mt * 2048);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        fb[nt][0] = *reinterpret_cast<const half8 *>(lds + b_adr0 + nt * 2048);
        if (!HALFSET) fb[nt][NJ - 1] = *reinterpret_cast<const half8 *>(lds + (b_adr0 ^ 32) + nt * 2048);
    }
    ADVANCE_READ();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BARRIER();
    // before the barrier that ends iteration g everything but that iteration's kPerIter instructions has landed, i.e. the
    // stage the NEXT matrix segment re-loads from (g + 2) is complete
#define WAIT_ITER(G) do { if ((G) + NA < total) WAITVM(kInFlight); else WAITVM(0); } while (0)
#define ISSUE_BOTH() do { if (B_FIRST) { ISSUE_B(); ISSUE_A(); } else { ISSUE_A(); ISSUE_B(); } } while (0)
    if (grp == 0) {
        for (int g = 0; g < total; ++g) {
            COMPUTE();
            BARRIER();
            ISSUE_BOTH();
            WAIT_ITER(g);
            BARRIER();
        }
    } else {
        for (int g = 0; g < total; ++g) {
            ISSUE_BOTH();
            __builtin_amdgcn_sched_barrier(0);
            BARRIER();
            COMPUTE();
            WAIT_ITER(g);
            BARRIER();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float keep = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[mt][nt][r];
    if (keep == 1.2345e-30f) sink[0] = keep;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(sink)[1] = clock64() - t_start;
}

template <int ROWS, int SHARE, int NA, int NB>
void run(const char *X, int64_t n_rows, const char *Q, float *sink, const char *what) {
    auto kern = k<ROWS, SHARE, NA, NB>;
    constexpr int lds_bytes = NA * ROWS * 64 + NB * 16384;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int64_t n_tiles = n_rows / ROWS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, sum = 0;
    long long cyc = 0;
    hipMemset(sink, 0, 64);
    const int reps = 6;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds_bytes, 0, X, n_tiles, Q, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0) sum += ms;
        if (rep > 0 && ms < best) { best = ms; hipMemcpy(&cyc, reinterpret_cast<long long *>(sink) + 1, 8, hipMemcpyDeviceToHost); }
    }
    const double flops = (double)n_rows * SHARE * 2.0 * 256 * 1024;
    printf("rows=%d share=%d rings=%d/%d %-46s best %7.3f ms  mean %7.3f ms  %5.2f PF  chunk side %5.2f TB/s  clock %4.0f MHz  %s\n",
           ROWS, SHARE, NA, NB, what, best, sum / (reps - 1), flops / best * 1e-12, (double)n_rows * 2048.0 * SHARE / best * 1e-9,
           (double)cyc / best * 1e-3, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    const int64_t n_rows = 983040;                // 3840 tiles of 256 = 2560 tiles of 384
    const size_t bytes = (size_t)n_rows * 2048 + (2 << 20);
    char *X, *Q; float *sink;
    hipMalloc(&X, bytes);
    hipMalloc(&Q, 4 * kQTileBytes);
    hipMalloc(&sink, 64);
    for (int mode = 0; mode < 2; ++mode) {
        printf("--- data: %s ---\n", mode == 0 ? "one binade, random mantissa and sign" : "approximately normal, sd 1/32 (the bench corpus' distribution)");
        fill_kernel<<<4096, 256>>>((uint16_t *)X, bytes / 2, 1u, mode);
        fill_kernel<<<256, 256>>>((uint16_t *)Q, (size_t)4 * kQTileBytes / 2, 7u, mode);
        hipDeviceSynchronize();
        for (int round = 0; round < 3; ++round) {
            run<256, 1, 5, 4>(X, n_rows, Q, sink, "256 x 256 (pp3: 144 KiB rings + 16 KiB records)");
            run<384, 1, 4, 3>(X, n_rows, Q, sink, "384 x 256 (144 KiB rings + 16 KiB records)");
            run<384, 1, 4, 4>(X, n_rows, Q, sink, "384 x 256 (160 KiB rings, no room for records)");
        }
        for (int round = 0; round < 2; ++round) {
            run<256, 2, 5, 4>(X, n_rows, Q, sink, "256 x 256 (pp3: 144 KiB rings + 16 KiB records)");
            run<384, 2, 4, 3>(X, n_rows, Q, sink, "384 x 256 (144 KiB rings + 16 KiB records)");
        }
        for (int round = 0; round < 2; ++round) {
            run<256, 4, 5, 4>(X, n_rows, Q, sink, "256 x 256 (pp3: 144 KiB rings + 16 KiB records)");
            run<384, 4, 4, 3>(X, n_rows, Q, sink, "384 x 256 (144 KiB rings + 16 KiB records)");
            run<384, 4, 4, 4>(X, n_rows, Q, sink, "384 x 256 (160 KiB rings, no room for records)");
        }
    }
    return 0;
}
