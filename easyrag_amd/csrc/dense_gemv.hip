// Dense route, small batches (B <= 16): the append scan as a skinny GEMM on the 16x16x32 matrix-core tile.
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
// Roofline: HBM (arithmetic intensity = B flop/byte <= 16); algorithmic bytes = rows * d * 2.
#include "common.h"
#include "kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGvThreads = 256;             // 4 waves per workgroup, each with its own row groups
// KB = steps (of 32 halves) whose loads are issued together: 32 -> 1024 halves, 128 VGPRs in flight per lane;
// 16 -> half of that and more resident waves

// STORE: scores of rows [c0, c1) go to S0[q][row - c0] (the threshold-seeding prefix) instead of the candidate lists.
template <bool STORE, int kGvKB>
__global__ __launch_bounds__(kGvThreads) void dense_gemv_kernel(
    const _Float16 *__restrict__ X, int64_t N, int d, int64_t c0, int64_t c1,
    const _Float16 *__restrict__ Q, int B,
    const float *__restrict__ tau, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap, uint32_t *__restrict__ overflow,
    float *__restrict__ S0, int ld_s0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8 *qf = reinterpret_cast<half8 *>(smem);                        // [steps][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int steps = d / 32;
    const int col = lane & 15, ks = lane >> 4;
    // query fragments in fragment order (zero columns beyond B)
    for (int i = threadIdx.x; i < steps * 64; i += kGvThreads) {
        const int j = i >> 6, l = i & 63;
        const int c = l & 15, s = l >> 4;
        half8 v;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (_Float16)0.f;
        if (c < B) v = *reinterpret_cast<const half8 *>(Q + (int64_t)c * d + 32 * j + 8 * s);
        qf[i] = v;
    }
    const float t_q = (!STORE && col < B) ? tau[col] : INFINITY;
    const int fd = (!STORE && filter_dir && col < B) ? (int)filter_dir[col] : -1;
    __syncthreads();

    const int64_t lim = (c1 < N) ? c1 : N;
    const int64_t n_groups = ((STORE ? c1 : lim) - c0 + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * (kGvThreads / 64);
    for (int64_t g = (int64_t)blockIdx.x * (kGvThreads / 64) + wave; g < n_groups; g += n_waves) {
        const int64_t row0 = c0 + g * 16;
        int64_t my_row = row0 + col;                                     // this lane's chunk row on the load side
        if (my_row > N - 1) my_row = N - 1;                              // clamp: rows past the end are masked below
        const _Float16 *src = X + my_row * (int64_t)d + 8 * ks;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < steps; j0 += kGvKB) {
            half8 a[kGvKB];
#pragma unroll
            for (int u = 0; u < kGvKB; ++u)
                if (j0 + u < steps) a[u] = *reinterpret_cast<const half8 *>(src + 32 * (j0 + u));
#pragma unroll
            for (int u = 0; u < kGvKB; ++u)
                if (j0 + u < steps)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], qf[(j0 + u) * 64 + lane], acc, 0, 0, 0);
        }
        // lane: query column `col`, chunk rows row0 + 4 * ks + i
        if (STORE) {
            if (col < B) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t chunk = row0 + 4 * ks + i;
                    if (chunk < c1) S0[(int64_t)col * ld_s0 + (chunk - c0)] = chunk < N ? acc[i] : -INFINITY;
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sc = acc[i];
            if (sc >= t_q) {                                             // rare: a fraction of a percent of the scores
                const int64_t chunk = row0 + 4 * ks + i;
                if (chunk < lim && (fd < 0 || (int)dir_id[chunk] == fd)) {
                    const uint32_t pos = atomicAdd(&cand_cnt[col], 1u);
                    if (pos < (uint32_t)cap) {
                        ErhCand c;
                        c.s = sc;
                        c.idx = (int32_t)chunk;
                        cand[(int64_t)col * cap + pos] = c;
                    } else {
                        atomicOr(overflow, 1u);
                    }
                }
            }
        }
    }
}

}  // namespace

namespace erh {

int dense_gemv_max_queries() { return 16; }

int g_gemv_kb = 32, g_gemv_wgs = 2;          // tuning knobs (option dense_gemv_kb / dense_gemv_wgs)

static hipError_t gemv_launch(bool store, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q,
                              int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                              uint32_t *cand_cnt, int cap, uint32_t *overflow, float *S0, int ld_s0, int n_cus,
                              hipStream_t st) {
    if (c1 <= c0) return hipSuccess;
    const size_t lds = (size_t)d * 32;                                   // d/32 steps x 1 KiB
    if (B > 16 || d % 32 != 0 || lds > 128 * 1024) return hipErrorInvalidValue;
    // resident grid: LDS allows 160 KiB / lds workgroups per CU; registers 2-3 (KB 32) or 4-5 (KB 16) waves per SIMD
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > g_gemv_wgs) per_cu = g_gemv_wgs;
    if (per_cu < 1) per_cu = 1;
    const int64_t n_groups = (c1 - c0 + 15) / 16;
    int64_t grid = (int64_t)n_cus * per_cu;
    if (grid * 4 > n_groups) grid = (n_groups + 3) / 4;
#define ERH_GV_LAUNCH(ST, KB)                                                                                          \
    hipLaunchKernelGGL((dense_gemv_kernel<ST, KB>), dim3((unsigned)grid), dim3(kGvThreads), lds, st, X, N, d, c0, c1, Q, B, \
                       tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, S0, ld_s0)
    if (g_gemv_kb == 16) { if (store) ERH_GV_LAUNCH(true, 16); else ERH_GV_LAUNCH(false, 16); }
    else { if (store) ERH_GV_LAUNCH(true, 32); else ERH_GV_LAUNCH(false, 32); }
#undef ERH_GV_LAUNCH
    return hipGetLastError();
}

// dynamic-LDS limit of the kernels on the CURRENT device (erh_create calls it for every handle: attributes are per device)
hipError_t dense_gemv_init() {
    const void *fns[] = {(const void *)dense_gemv_kernel<false, 32>, (const void *)dense_gemv_kernel<true, 32>,
                         (const void *)dense_gemv_kernel<false, 16>, (const void *)dense_gemv_kernel<true, 16>};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void dense_gemv_tune(int kb, int wgs) {
    if (kb == 16 || kb == 32) g_gemv_kb = kb;
    if (wgs >= 1 && wgs <= 5) g_gemv_wgs = wgs;
}

// hipErrorInvalidValue when the shape does not qualify (more than 16 queries, d not a multiple of 32, query
// fragments larger than LDS): the caller then uses the padded MFMA scan.
hipError_t launch_dense_gemv_append(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q, int B,
                                    const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                                    uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, hipStream_t st) {
    return gemv_launch(false, X, N, d, c0, c1, Q, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, nullptr, 0,
                       n_cus, st);
}

// Seed prefix of the same small batch: S0[q][chunk - c0] for chunks [c0, c0 + nc), q < B.
hipError_t launch_dense_gemv_store(const _Float16 *X, int64_t N, int d, int64_t c0, int nc, const _Float16 *Q, int B,
                                   float *S0, int ld_s0, int n_cus, hipStream_t st) {
    return gemv_launch(true, X, N, d, c0, c0 + nc, Q, B, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, S0, ld_s0,
                       n_cus, st);
}

}  // namespace erh
