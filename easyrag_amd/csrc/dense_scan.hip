// Dense route, kernel K1: batched query x chunk-matrix inner products on the gfx950 matrix cores with
// a fused threshold epilogue.  Replaces the Qdrant COSINE scan behind QdrantRetriever
// (/root/reference/src/easyrag/custom/retrievers.py:37-52; collection ingestion.py:178-183).
//
// Shape: S = Rows_A . Rows_B^T over K = d, both operands fp16 row-major with d contiguous, fp32
// accumulation in v_mfma_f32_32x32x16_f16.  A workgroup of 8 waves owns a BM x BN score tile:
//   - both operand tiles are staged HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B per lane, no
//     VGPR round trip), BK = 64 halves = one 128-byte line per row per K-step, double buffered;
//   - the LDS image is lane-linear (the DMA requires it), so the 16-byte slot index inside each
//     128-byte row is XOR-swizzled on the *source* address with ((row>>1)&7) and un-swizzled on
//     the ds_read_b128 side: the 16 lanes of a b128 lane group then hit 16 distinct slots of the
//     256-byte bank row (conflict-free), and the 8 lanes of a row still fetch one whole line;
//   - the fragment k-mapping (lane>>5 picks the 8-half slot inside a 16-wide k-substep) is the same
//     for both operands, which is all a dot product needs.
// The score matrix is never written.  Epilogues:
//   STORE  (A = queries, B = chunks): lane = chunk column -> coalesced rows of S0[q][chunk] for the
//          threshold-seeding prefix of the corpus;
//   APPEND (A = chunks, B = queries): lane = query column; a score survives if >= tau[q] (the
//          pruning threshold from the previous stage, already lowered by the fp32 error margin),
//          passes the optional dir filter, and is appended to the query's candidate list.
#include "common.h"
#include "kernels.h"

namespace {

template <int BM_, int BN_, int WGM_, int WGN_>
struct ScanCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_;
    static constexpr int NW = WGM * WGN;
    static constexpr int NT = NW * 64;
    static constexpr int WM = BM / WGM, WN = BN / WGN;
    static constexpr int MT = WM / 32, NTL = WN / 32;
    static constexpr int BK = 64;                       // halves per K-step = 128 bytes per row
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "staging must divide evenly");
};

// Per-lane source pointers of one operand tile (ROWS rows x 128 bytes per K-step), computed once per tile:
// piece = 16-byte unit, 8 per row; instruction `it` of wave `w` moves pieces [(it*NW + w)*64, +64).
template <int ROWS, int NW>
struct TileSrc {
    static constexpr int ITERS = (ROWS * 8) / (NW * 64);
    const _Float16 *src[ITERS];
    __device__ __forceinline__ void init(const _Float16 *__restrict__ base, int64_t row0, int64_t rows_total,
                                         int d, int wave, int lane) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int piece = (it * NW + wave) * 64 + lane;
            const int r = piece >> 3;                   // row inside the tile
            const int p = piece & 7;                    // physical 16-byte slot inside the 128-byte row
            int64_t grow = row0 + r;
            if (grow > rows_total - 1) grow = rows_total - 1;   // clamp: rows past the end are masked later
            const int ls = p ^ ((r >> 1) & 7);          // logical slot stored at physical slot p
            src[it] = base + grow * (int64_t)d + ls * 8;
        }
    }
    // Issue the LDS-DMA loads for K-step kt into lds_tile (lane-linear image).
    __device__ __forceinline__ void issue(int kt, char *lds_tile, int wave) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int piece0 = (it * NW + wave) * 64;   // wave-uniform
            __builtin_amdgcn_global_load_lds((const void *)(src[it] + (int64_t)kt * 64),
                                             ERH_LDS_PTR(lds_tile + piece0 * 16), 16, 0, 0);
        }
    }
};

// The K-loop: on return acc[mt][nt] holds the 32x32 fp32 tiles of this wave
// (rows = A rows wave_m*WM + mt*32 + .., cols = B rows wave_n*WN + nt*32 + ..).
template <class C>
__device__ __forceinline__ void gemm_tile(const _Float16 *__restrict__ A, int64_t a_row0, int64_t a_rows,
                                          const _Float16 *__restrict__ B, int64_t b_row0, int64_t b_rows,
                                          int d, char *lds, f32x16 (&acc)[C::MT][C::NTL],
                                          int wave, int lane, int wave_m, int wave_n) {
    const int nk = d / C::BK;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // per-lane fragment addressing: row = .. + (lane & 31); logical slot for k-substep j = 2j + (lane >> 5)
    const int l31 = lane & 31, h = lane >> 5;
    const int sw = (l31 >> 1) & 7;                      // == ((row >> 1) & 7): tile/wave/mt offsets are multiples of 16
    int soff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) soff[j] = (((2 * j + h) ^ sw) << 4);
    const int a_lane_off = (wave_m * C::WM + l31) * 128;
    const int b_lane_off = C::A_BYTES + (wave_n * C::WN + l31) * 128;

    TileSrc<C::BM, C::NW> ta;
    TileSrc<C::BN, C::NW> tb;
    ta.init(A, a_row0, a_rows, d, wave, lane);
    tb.init(B, b_row0, b_rows, d, wave, lane);
    ta.issue(0, lds, wave);
    tb.issue(0, lds + C::A_BYTES, wave);

    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed for this wave's DMAs; the barrier extends that to all waves and also
        // guarantees every wave is done reading the other buffer (its MFMAs consumed the ds_reads).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        char *cur = lds + (kt & 1) * C::STAGE_BYTES;
        if (kt + 1 < nk) {
            char *nxt = lds + ((kt + 1) & 1) * C::STAGE_BYTES;
            ta.issue(kt + 1, nxt, wave);
            tb.issue(kt + 1, nxt + C::A_BYTES, wave);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half8 af[C::MT], bf[C::NTL];
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
                af[mt] = *reinterpret_cast<const half8 *>(cur + a_lane_off + mt * 32 * 128 + soff[j]);
#pragma unroll
            for (int nt = 0; nt < C::NTL; ++nt)
                bf[nt] = *reinterpret_cast<const half8 *>(cur + b_lane_off + nt * 32 * 128 + soff[j]);
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < C::NTL; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
    }
}

// 32x32 MFMA C/D layout (dtype independent on gfx950): lane holds column (lane & 31) and rows
// (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), r = 0..15.
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------
// STORE epilogue: S0[q][chunk - c0] for chunks [c0, c0 + nc), all Bpad query rows.
template <class C>
__global__ __launch_bounds__(C::NT) void dense_scan_store_kernel(
    const _Float16 *__restrict__ Q, int Bpad, const _Float16 *__restrict__ X, int64_t N, int d,
    int64_t c0, int nc, float *__restrict__ S0, int ld_s0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_m = wave / C::WGN, wave_n = wave % C::WGN;
    const int n_ctiles = (nc + C::BN - 1) / C::BN;
    const int ct = blockIdx.x % n_ctiles, qt = blockIdx.x / n_ctiles;
    const int64_t q_row0 = (int64_t)qt * C::BM;
    const int64_t c_row0 = c0 + (int64_t)ct * C::BN;

    f32x16 acc[C::MT][C::NTL];
    gemm_tile<C>(Q, q_row0, Bpad, X, c_row0, N, d, lds, acc, wave, lane, wave_m, wave_n);

#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt) {
        const int64_t chunk = c_row0 + wave_n * C::WN + nt * 32 + (lane & 31);
        const int col = (int)(chunk - c0);
        const bool col_ok = col < nc;
        const bool live = chunk < N;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (int)q_row0 + wave_m * C::WM + mt * 32 + mfma_row(r, lane);
                if (col_ok && q < Bpad) S0[(int64_t)q * ld_s0 + col] = live ? acc[mt][nt][r] : -INFINITY;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// APPEND epilogue: chunks [c0, c1) against Bpad queries; survivors (>= tau[q]) go to cand[q][..].
template <class C>
__global__ __launch_bounds__(C::NT) void dense_scan_append_kernel(
    const _Float16 *__restrict__ X, int64_t N, int d, int64_t c0, int64_t c1,
    const _Float16 *__restrict__ Q, int Bpad, int B,
    const float *__restrict__ tau, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap, uint32_t *__restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_m = wave / C::WGN, wave_n = wave % C::WGN;

    // XCD-aware mapping: block b runs on XCD b % 8.  The n_qt query tiles that share one chunk tile get
    // consecutive slots of the same XCD so the re-read of the chunk tile is served by that XCD's L2.
    const int n_qt = Bpad / C::BN;
    const int64_t n_ct = (c1 - c0 + C::BM - 1) / C::BM;
    const int x = blockIdx.x & 7;
    const int64_t jx = blockIdx.x >> 3;
    const int qt = (int)(jx % n_qt);
    const int64_t ct = (jx / n_qt) * 8 + x;
    if (ct >= n_ct) return;
    const int64_t c_row0 = c0 + ct * C::BM;
    const int64_t q_row0 = (int64_t)qt * C::BN;

    f32x16 acc[C::MT][C::NTL];
    gemm_tile<C>(X, c_row0, N, Q, q_row0, Bpad, d, lds, acc, wave, lane, wave_m, wave_n);

    const int64_t lim = (c1 < N) ? c1 : N;
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt) {
        const int q = (int)q_row0 + wave_n * C::WN + nt * 32 + (lane & 31);
        const float t = (q < B) ? tau[q] : INFINITY;
        float m = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mt][nt][r]);
        if (m >= t) {
            const int fd = filter_dir ? (int)filter_dir[q] : -1;
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = acc[mt][nt][r];
                    if (s >= t) {
                        const int64_t chunk = c_row0 + wave_m * C::WM + mt * 32 + mfma_row(r, lane);
                        if (chunk < lim && (fd < 0 || (int)dir_id[chunk] == fd)) {
                            const uint32_t pos = atomicAdd(&cand_cnt[q], 1u);
                            if (pos < (uint32_t)cap) {
                                ErhCand c;
                                c.s = s;
                                c.idx = (int32_t)chunk;
                                cand[(int64_t)q * cap + pos] = c;
                            } else {
                                atomicOr(overflow, 1u);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Plain VALU reference on the device (debug / layout triangulation only): one thread per score.
__global__ void dense_naive_kernel(const _Float16 *__restrict__ Q, int B, const _Float16 *__restrict__ X,
                                   int64_t row0, int rows, int d, float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * rows) return;
    const int q = (int)(t / rows), r = (int)(t % rows);
    const _Float16 *x = X + (row0 + r) * (int64_t)d;
    const _Float16 *qq = Q + (int64_t)q * d;
    float s = 0.f;
    for (int k = 0; k < d; ++k) s = fmaf((float)x[k], (float)qq[k], s);
    out[t] = s;
}

using CfgMain = ScanCfg<256, 256, 2, 4>;   // 256 chunks x 256 queries, wave tile 128 x 64

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------
namespace erh {

int dense_scan_lds_bytes() { return CfgMain::LDS_BYTES; }
int dense_scan_q_tile() { return CfgMain::BN; }

hipError_t dense_scan_init() {
    hipError_t e = hipFuncSetAttribute((const void *)dense_scan_store_kernel<CfgMain>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, CfgMain::LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)dense_scan_append_kernel<CfgMain>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, CfgMain::LDS_BYTES);
}

hipError_t launch_dense_scan_store(const _Float16 *Q, int Bpad, const _Float16 *X, int64_t N, int d,
                                   int64_t c0, int nc, float *S0, int ld_s0, hipStream_t st) {
    if (nc <= 0) return hipSuccess;
    const int n_ctiles = (nc + CfgMain::BN - 1) / CfgMain::BN;
    const int n_qtiles = Bpad / CfgMain::BM;
    dim3 grid(n_ctiles * n_qtiles), block(CfgMain::NT);
    hipLaunchKernelGGL(dense_scan_store_kernel<CfgMain>, grid, block, CfgMain::LDS_BYTES, st,
                       Q, Bpad, X, N, d, c0, nc, S0, ld_s0);
    return hipGetLastError();
}

hipError_t launch_dense_scan_append(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1,
                                    const _Float16 *Q, int Bpad, int B, const float *tau,
                                    const int16_t *filter_dir, const int16_t *dir_id,
                                    ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, hipStream_t st) {
    if (c1 <= c0) return hipSuccess;
    const int n_qt = Bpad / CfgMain::BN;
    const int64_t n_ct = (c1 - c0 + CfgMain::BM - 1) / CfgMain::BM;
    const int64_t n_ct8 = (n_ct + 7) / 8 * 8;
    dim3 grid((unsigned)(n_ct8 * n_qt)), block(CfgMain::NT);
    hipLaunchKernelGGL(dense_scan_append_kernel<CfgMain>, grid, block, CfgMain::LDS_BYTES, st,
                       X, N, d, c0, c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow);
    return hipGetLastError();
}

hipError_t launch_dense_naive(const _Float16 *Q, int B, const _Float16 *X, int64_t row0, int rows, int d,
                              float *out, hipStream_t st) {
    const int64_t total = (int64_t)B * rows;
    if (total <= 0) return hipSuccess;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL(dense_naive_kernel, grid, block, 0, st, Q, B, X, row0, rows, d, out);
    return hipGetLastError();
}

}  // namespace erh
