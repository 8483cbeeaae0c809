// Dense route, small batches (B <= 64): the append scan as a skinny GEMM on the 16x16x32 matrix-core tile.
// The reference issues ONE query at a time (src/main.py:48-52 -> QdrantRetriever._aretrieve,
// /root/reference/src/easyrag/custom/retrievers.py:37-52); padding such a batch to the 256-query tile of
// dense_scan.hip costs a full MFMA scan (~0.8 ms at 1M x 1024), while the work is a 2 GB stream.
//
// Shape per wave: 16 chunk rows x 16 query columns (columns >= B are zero), K = d in steps of 32 halves with
// v_mfma_f32_16x16x32_f16.  The chunk-side fragment of a step is exactly what a coalesced load delivers: lane
// (r = lane & 15, s = lane >> 4) holds halves [32 j + 8 s, +8) of row r, i.e. a wave instruction fetches 64
// contiguous bytes of each of 16 consecutive rows straight into VGPRs (no LDS round trip: nothing is shared
// between waves on the chunk side, cdna_hip_programming.md "GEMV / M <= 16 decode weights"), and consecutive
// steps take the neighbouring 64 bytes of the same lines.  The query-side fragments are the same for every row
// group: they are laid out once per workgroup in LDS in fragment order ([step][lane] x 16 bytes, lane-linear =
// conflict-free ds_read_b128) and re-read per step -- 1 KiB of LDS traffic per KiB streamed from HBM.
// The 4 fp32 results per lane (rows 4 * (lane >> 4) + i, column lane & 15) go through the same threshold /
// filter / candidate-list epilogue as the big scan; everything downstream (refine, finalize, exhaustive path) is
// shared, so results are bit-identical to the padded scan's.
// Batches of 17 .. 64 queries (round 4: the padded 256-query MFMA scan took 0.455 ms from 17 queries on, against 0.35 for
// 16): G = 2 or 4 column groups of 16 queries share every chunk fragment -- one load, G matrix instructions with G query
// fragments out of LDS (G x d x 32 bytes: 128 KiB at d = 1024 and 49 .. 64 queries, one workgroup per CU).
// Roofline: HBM (arithmetic intensity = B flop/byte <= 64); algorithmic bytes = rows * d * 2.
#include "common.h"
#include "kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGvThreads = 256;             // 4 waves per workgroup, each with its own row groups
// KB = steps (of 32 halves) whose loads are issued together: 32 -> 1024 halves, 128 VGPRs in flight per lane;
// 16 -> half of that and more resident waves

// STORE: scores of rows [c0, c1) go to S0[q][row - c0] (the threshold-seeding prefix) instead of the candidate lists.
// PIPE: the chunk-side loads of the NEXT 16 steps are issued before the matrix instructions of the current 16 run (two
// register sets of 16 fragments, walked over the flattened (row group, half of K) sequence), so a wave's load and compute
// phases overlap -- with four column groups only four waves fit a CU and nothing else would cover the HBM latency.
// NT: the chunk-side loads carry the non-temporal hint (the matrix is read once per call and is far larger than L2 + MALL).
template <bool NT>
__device__ __forceinline__ half8 gv_load(const _Float16 *p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const half8 *>(p));
    else return *reinterpret_cast<const half8 *>(p);
}

template <bool STORE, int kGvKB, int G, bool PIPE, bool NT = false>
__global__ __launch_bounds__(kGvThreads) void dense_gemv_kernel(
    const _Float16 *__restrict__ X, int64_t N, int d, int64_t c0, int64_t c1,
    const _Float16 *__restrict__ Q, int B,
    const float *__restrict__ tau, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap, uint32_t *__restrict__ overflow,
    float *__restrict__ S0, int ld_s0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8 *qf = reinterpret_cast<half8 *>(smem);                        // [group][steps][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int steps = d / 32;
    const int col = lane & 15, ks = lane >> 4;
    // query fragments in fragment order (zero columns beyond B)
    for (int i = threadIdx.x; i < G * steps * 64; i += kGvThreads) {
        const int gq = i / (steps * 64), r = i - gq * (steps * 64);
        const int j = r >> 6, l = r & 63;
        const int c = gq * 16 + (l & 15), s = l >> 4;
        half8 v;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (_Float16)0.f;
        if (c < B) v = *reinterpret_cast<const half8 *>(Q + (int64_t)c * d + 32 * j + 8 * s);
        qf[i] = v;
    }
    float t_q[G];
    int fd[G];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        const int c = gq * 16 + col;
        t_q[gq] = (!STORE && c < B) ? tau[c] : INFINITY;
        fd[gq] = (!STORE && filter_dir && c < B) ? (int)filter_dir[c] : -1;
    }
    __syncthreads();

    const int64_t lim = (c1 < N) ? c1 : N;
    const int64_t n_groups = ((STORE ? c1 : lim) - c0 + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * (kGvThreads / 64);
    half8 pa[2][16];                                                     // PIPE only (steps must be a multiple of 16)
    bool primed = false;
    int cur = 0;                                                         // which set holds the chunk about to be computed
    for (int64_t g = (int64_t)blockIdx.x * (kGvThreads / 64) + wave; g < n_groups; g += n_waves) {
        const int64_t row0 = c0 + g * 16;
        int64_t my_row = row0 + col;                                     // this lane's chunk row on the load side
        if (my_row > N - 1) my_row = N - 1;                              // clamp: rows past the end are masked below
        const _Float16 *src = X + my_row * (int64_t)d + 8 * ks;
        f32x4 acc[G];
#pragma unroll
        for (int gq = 0; gq < G; ++gq) acc[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (PIPE) {
            // two sets of 16 fragments: while set `cur` feeds the matrix unit, the other one is being filled with the next 16
            // steps -- of this row group, or the first 16 of the wave's next row group (pa carried across the outer loop)
            constexpr int HB = 16;
            if (!primed) {
#pragma unroll
                for (int u = 0; u < HB; ++u) pa[0][u] = gv_load<NT>(src + 32 * u);
                primed = true;
            }
            for (int j0 = 0; j0 < steps; j0 += HB) {
                const bool more_here = j0 + HB < steps;
                const int64_t g_next = g + n_waves;
                const _Float16 *nsrc = src + 32 * (j0 + HB);
                bool fetch = more_here;
                if (!more_here && g_next < n_groups) {
                    int64_t nrow = c0 + g_next * 16 + col;
                    if (nrow > N - 1) nrow = N - 1;
                    nsrc = X + nrow * (int64_t)d + 8 * ks;
                    fetch = true;
                }
                if (cur == 0) {
                    if (fetch) {
#pragma unroll
                        for (int u = 0; u < HB; ++u) pa[1][u] = gv_load<NT>(nsrc + 32 * u);
                    }
#pragma unroll
                    for (int u = 0; u < HB; ++u) {
#pragma unroll
                        for (int gq = 0; gq < G; ++gq)
                            acc[gq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[0][u], qf[(gq * steps + j0 + u) * 64 + lane], acc[gq], 0, 0, 0);
                    }
                } else {
                    if (fetch) {
#pragma unroll
                        for (int u = 0; u < HB; ++u) pa[0][u] = gv_load<NT>(nsrc + 32 * u);
                    }
#pragma unroll
                    for (int u = 0; u < HB; ++u) {
#pragma unroll
                        for (int gq = 0; gq < G; ++gq)
                            acc[gq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[1][u], qf[(gq * steps + j0 + u) * 64 + lane], acc[gq], 0, 0, 0);
                    }
                }
                cur ^= 1;
            }
        } else {
        for (int j0 = 0; j0 < steps; j0 += kGvKB) {
            half8 a[kGvKB];
#pragma unroll
            for (int u = 0; u < kGvKB; ++u)
                if (j0 + u < steps) a[u] = gv_load<NT>(src + 32 * (j0 + u));
#pragma unroll
            for (int u = 0; u < kGvKB; ++u)
                if (j0 + u < steps) {
#pragma unroll
                    for (int gq = 0; gq < G; ++gq)
                        acc[gq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], qf[(gq * steps + j0 + u) * 64 + lane], acc[gq], 0, 0, 0);
                }
        }
        }
        // lane: query columns 16 * gq + `col`, chunk rows row0 + 4 * ks + i
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
            const int c = gq * 16 + col;
            if (STORE) {
                if (c < B) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t chunk = row0 + 4 * ks + i;
                        if (chunk < c1) S0[(int64_t)c * ld_s0 + (chunk - c0)] = chunk < N ? acc[gq][i] : -INFINITY;
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sc = acc[gq][i];
                if (sc >= t_q[gq]) {                                     // rare: a fraction of a percent of the scores
                    const int64_t chunk = row0 + 4 * ks + i;
                    if (chunk < lim && (fd[gq] < 0 || (int)dir_id[chunk] == fd[gq])) {
                        const uint32_t pos = atomicAdd(&cand_cnt[c], 1u);
                        if (pos < (uint32_t)cap) {
                            ErhCand cd;
                            cd.s = sc;
                            cd.idx = (int32_t)chunk;
                            cand[(int64_t)c * cap + pos] = cd;
                        } else {
                            atomicOr(overflow, 1u);
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

namespace erh {

int dense_gemv_max_queries() { return 64; }

// kb: steps of 32 halves whose loads are issued together (16 / 32); wgs: workgroups per CU at most (options dense_gemv_kb /
// dense_gemv_wgs of the calling handle)
static hipError_t gemv_launch(bool store, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q,
                              int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                              uint32_t *cand_cnt, int cap, uint32_t *overflow, float *S0, int ld_s0, int n_cus, int kb, int wgs,
                              int pipe_opt, hipStream_t st, int nt = 0) {
    if (c1 <= c0) return hipSuccess;
    const int groups = B <= 16 ? 1 : B <= 32 ? 2 : 4;                    // column groups of 16 queries
    // software-pipelined loads (two sets of 16 fragments): default for 2 / 4 column groups, where few waves fit a CU
    const bool pipe = (d % 512 == 0) && (pipe_opt < 0 ? groups >= 2 : pipe_opt != 0);
    const size_t lds = (size_t)groups * d * 32;                          // per group d/32 steps x 1 KiB
    if (B > 64 || d % 32 != 0 || lds > 128 * 1024) return hipErrorInvalidValue;
    // resident grid: LDS allows 160 KiB / lds workgroups per CU; registers 2-3 (KB 32) or 4-5 (KB 16) waves per SIMD
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > wgs) per_cu = wgs;
    if (per_cu < 1) per_cu = 1;
    const int64_t n_groups = (c1 - c0 + 15) / 16;
    int64_t grid = (int64_t)n_cus * per_cu;
    if (grid * 4 > n_groups) grid = (n_groups + 3) / 4;
#define ERH_GV_LAUNCH(ST, KB, G, P)                                                                                    \
    hipLaunchKernelGGL((dense_gemv_kernel<ST, KB, G, P>), dim3((unsigned)grid), dim3(kGvThreads), lds, st, X, N, d, c0, c1, Q, \
                       B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, S0, ld_s0)
#define ERH_GV_LAUNCH_NT(KB, G, P)                                                                                     \
    hipLaunchKernelGGL((dense_gemv_kernel<false, KB, G, P, true>), dim3((unsigned)grid), dim3(kGvThreads), lds, st, X, N, d, c0, c1, Q, \
                       B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, S0, ld_s0)
#define ERH_GV_PICK(G)                                                                                                 \
    do {                                                                                                               \
        if (nt && !store) { if (pipe) ERH_GV_LAUNCH_NT(16, G, true); else if (kb == 16) ERH_GV_LAUNCH_NT(16, G, false); else ERH_GV_LAUNCH_NT(32, G, false); } \
        else if (pipe) { if (store) ERH_GV_LAUNCH(true, 16, G, true); else ERH_GV_LAUNCH(false, 16, G, true); }             \
        else if (kb == 16) { if (store) ERH_GV_LAUNCH(true, 16, G, false); else ERH_GV_LAUNCH(false, 16, G, false); }  \
        else { if (store) ERH_GV_LAUNCH(true, 32, G, false); else ERH_GV_LAUNCH(false, 32, G, false); }                \
    } while (0)
    if (groups == 1) ERH_GV_PICK(1); else if (groups == 2) ERH_GV_PICK(2); else ERH_GV_PICK(4);
#undef ERH_GV_PICK
#undef ERH_GV_LAUNCH_NT
#undef ERH_GV_LAUNCH
    return hipGetLastError();
}

// dynamic-LDS limit of the kernels on the CURRENT device (erh_create calls it for every handle: attributes are per device)
hipError_t dense_gemv_init() {
#define ERH_GV_FNS(G)                                                                                                  \
    (const void *)dense_gemv_kernel<false, 32, G, false>, (const void *)dense_gemv_kernel<true, 32, G, false>,         \
    (const void *)dense_gemv_kernel<false, 16, G, false>, (const void *)dense_gemv_kernel<true, 16, G, false>,         \
    (const void *)dense_gemv_kernel<false, 16, G, true>, (const void *)dense_gemv_kernel<true, 16, G, true>,           \
    (const void *)dense_gemv_kernel<false, 32, G, false, true>, (const void *)dense_gemv_kernel<false, 16, G, false, true>, \
    (const void *)dense_gemv_kernel<false, 16, G, true, true>
    const void *fns[] = {ERH_GV_FNS(1), ERH_GV_FNS(2), ERH_GV_FNS(4)};
#undef ERH_GV_FNS
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// hipErrorInvalidValue when the shape does not qualify (more than 64 queries, d not a multiple of 32, query
// fragments larger than LDS): the caller then uses the padded MFMA scan.
hipError_t launch_dense_gemv_append(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q, int B,
                                    const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                                    uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int kb, int wgs, int pipe,
                                    hipStream_t st, int nt) {
    return gemv_launch(false, X, N, d, c0, c1, Q, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, nullptr, 0,
                       n_cus, kb, wgs, pipe, st, nt);
}

// Seed prefix of the same small batch: S0[q][chunk - c0] for chunks [c0, c0 + nc), q < B.
hipError_t launch_dense_gemv_store(const _Float16 *X, int64_t N, int d, int64_t c0, int nc, const _Float16 *Q, int B,
                                   float *S0, int ld_s0, int n_cus, int kb, int wgs, int pipe, hipStream_t st) {
    return gemv_launch(true, X, N, d, c0, c0 + nc, Q, B, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, S0, ld_s0,
                       n_cus, kb, wgs, pipe, st);
}

}  // namespace erh
