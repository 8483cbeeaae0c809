// Dense route, kernel K1: batched query x chunk-matrix inner products on the gfx950 matrix cores with
// a fused threshold epilogue.  Replaces the Qdrant COSINE scan behind QdrantRetriever
// (/root/reference/src/easyrag/custom/retrievers.py:37-52; collection ingestion.py:178-183).
//
// Shape: S = Rows_A . Rows_B^T over K = d, both operands fp16 row-major with d contiguous, fp32
// accumulation in v_mfma_f32_32x32x16_f16.  A workgroup owns a BM x BN score tile:
//   - both operand tiles are staged HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR
//     round trip), BK halves per K-step, in per-operand rings (A_STAGES / B_STAGES deep) so that several
//     K-steps of loads stay in flight across the (raw) barrier; completion is tracked with a counted
//     s_waitcnt vmcnt(N), never a drain, except in the last steps of a tile;
//   - the LDS image is lane-linear (the DMA requires it), so the 16-byte slot index inside each row is
//     XOR-swizzled on the *source* address and un-swizzled on the ds_read_b128 side: the 16 lanes of a
//     b128 lane group hit 16 distinct slots of the 256-byte bank row (conflict-free) while the lanes of a
//     row still fetch one contiguous run of a cache line;
//   - the fragment k-mapping (lane>>5 picks the 8-half slot inside a 16-wide k-substep) is the same for
//     both operands, which is all a dot product needs.
// The score matrix is never written.  Epilogues:
//   STORE  (A = queries, B = chunks): lane = chunk column -> coalesced rows of S0[q][chunk] for the
//          threshold-seeding prefix of the corpus;
//   APPEND (A = chunks, B = queries): lane = query column; a score survives if >= tau[q] (the
//          pruning threshold from the previous stage, already lowered by the fp32 error margin),
//          passes the optional dir filter, and is appended to the query's candidate list.
// Two tile configurations are built (option "dense_cfg"):
//   0: 256 x 256, BK 64, 8 waves, A ring 3 / B ring 2, 160 KiB LDS  -> one workgroup per CU
//   1: 128 x 256, BK 32, 4 waves, A ring 4 / B ring 3,  80 KiB LDS  -> two independent workgroups per CU,
//      whose barrier / issue bubbles overlap each other's MFMA phases
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace {

template <int BM_, int BN_, int WGM_, int WGN_, int BK_, int AST_, int BST_>
struct ScanCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, BK = BK_;
    static constexpr int NW = WGM * WGN;
    static constexpr int NT = NW * 64;
    static constexpr int WM = BM / WGM, WN = BN / WGN;
    static constexpr int MT = WM / 32, NTL = WN / 32;
    static constexpr int RB = BK * 2;                   // bytes per row per K-step (64 or 128)
    static constexpr int PR = RB / 16;                  // 16-byte pieces per row (4 or 8)
    static constexpr int KS = BK / 16;                  // k-substeps (one MFMA K) per K-step
    static constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;
    static constexpr int A_STAGES = AST_, B_STAGES = BST_;
    static constexpr int DA = AST_ - 1, DB = BST_ - 1;  // prefetch distances in K-steps
    static constexpr int B_BASE = A_STAGES * A_BYTES;
    static constexpr int LDS_BYTES = A_STAGES * A_BYTES + B_STAGES * B_BYTES;
    static constexpr int A_ITERS = (BM * PR) / NT, B_ITERS = (BN * PR) / NT;   // LDS-DMA instructions per wave per stage
    // loads that may still be in flight at the top of a K-step: everything issued after B(kt)
    static constexpr int WAIT_N = (AST_ > BST_) ? A_ITERS * DB + B_ITERS * (DB - 1) : (DB - 1) * (A_ITERS + B_ITERS);
    // read-ahead variant (needs DA == DB >= 2): stage kt+1 must have landed too, so only the loads issued after
    // A(kt+1) -- the later (DB - 2) steps -- may stay in flight
    static constexpr bool CAN_RA = (AST_ == BST_) && (BST_ >= 3);
#ifdef ERH_MEASURE
    static constexpr bool MEASURE = (BK_ == 64) && (AST_ == 3);   // ablation variants are built for the default lock-step config only
#else
    static constexpr bool MEASURE = false;                        // product build: no measurement instantiations
#endif
    static constexpr int WAIT_RA = (DB - 2) * (A_ITERS + B_ITERS);
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
    static_assert((BM * PR) % NT == 0 && (BN * PR) % NT == 0, "staging must divide evenly");
    static_assert(DA >= DB && DB >= 1, "ring depths");
    static_assert(MT * 16 <= 64, "survivor mask is 64 bits");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int PR>
__device__ __forceinline__ int row_swizzle(int r) { return PR == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

// Per-lane source pointers of one operand tile (ROWS rows x RB bytes per K-step), computed once per tile:
// piece = 16-byte unit, PR per row; instruction `it` of wave `w` moves pieces [(it*NW + w)*64, +64).
template <int ROWS, int NW, int PR>
struct TileSrc {
    static constexpr int ITERS = (ROWS * PR) / (NW * 64);
    const _Float16 *src[ITERS];
    __device__ __forceinline__ void init(const _Float16 *__restrict__ base, int64_t row0, int64_t rows_total,
                                         int d, int wave, int lane) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int piece = (it * NW + wave) * 64 + lane;
            const int r = piece / PR;                   // row inside the tile
            const int p = piece % PR;                   // physical 16-byte slot inside the row
            int64_t grow = row0 + r;
            if (grow > rows_total - 1) grow = rows_total - 1;   // clamp: rows past the end are masked later
            const int ls = p ^ row_swizzle<PR>(r);      // logical slot stored at physical slot p
            src[it] = base + grow * (int64_t)d + ls * 8;
        }
    }
    // Issue instructions [it0, it1) of the K-step's DMA set (used to spread the issue cost between MFMA groups).
    __device__ __forceinline__ void issue_part(int kt, int bk, char *lds_tile, int wave, int it0, int it1) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (it < it0 || it >= it1) continue;
            const int piece0 = (it * NW + wave) * 64;   // wave-uniform
            __builtin_amdgcn_global_load_lds((const void *)(src[it] + (int64_t)kt * bk),
                                             ERH_LDS_PTR(lds_tile + piece0 * 16), 16, 0, 0);
        }
    }
    // Issue the LDS-DMA loads for K-step kt (bk halves each) into lds_tile (lane-linear image).
    __device__ __forceinline__ void issue(int kt, int bk, char *lds_tile, int wave) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int piece0 = (it * NW + wave) * 64;   // wave-uniform
            __builtin_amdgcn_global_load_lds((const void *)(src[it] + (int64_t)kt * bk),
                                             ERH_LDS_PTR(lds_tile + piece0 * 16), 16, 0, 0);
        }
    }
};

// The K-loop: on return acc[mt][nt] holds the 32x32 fp32 tiles of this wave
// (rows = A rows wave_m*WM + mt*32 + .., cols = B rows wave_n*WN + nt*32 + ..).
// ABL (measurement builds only, option "dense_ablate"; results are garbage for ABL 2..4):
//   0 full kernel   1 no epilogue   2 no MFMA (LDS reads kept alive)   3 no LDS fragment reads   4 no LDS-DMA
template <class C, int ABL>
__device__ __forceinline__ void gemm_tile(const _Float16 *__restrict__ A, int64_t a_row0, int64_t a_rows,
                                          const _Float16 *__restrict__ B, int64_t b_row0, int64_t b_rows,
                                          int d, char *lds, f32x16 (&acc)[C::MT][C::NTL],
                                          int wave, int lane, int wave_m, int wave_n, long long *tsec = nullptr) {
    const int nk = d / C::BK;
    long long t_mark = (ABL == 5) ? clock64() : 0;
#define ERH_SEC(I) do { if (ABL == 5) { const long long n_ = clock64(); tsec[I] += n_ - t_mark; t_mark = n_; } } while (0)
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // per-lane fragment addressing: row = .. + (lane & 31); logical slot for k-substep j = 2j + (lane >> 5)
    const int l31 = lane & 31, h = lane >> 5;
    const int sw = row_swizzle<C::PR>(l31);             // tile / wave / mt offsets are multiples of 32
    int soff[C::KS];
#pragma unroll
    for (int j = 0; j < C::KS; ++j) soff[j] = (((2 * j + h) ^ sw) << 4);
    const int a_lane_off = (wave_m * C::WM + l31) * C::RB;
    const int b_lane_off = C::B_BASE + (wave_n * C::WN + l31) * C::RB;

    TileSrc<C::BM, C::NW, C::PR> ta;
    TileSrc<C::BN, C::NW, C::PR> tb;
    ta.init(A, a_row0, a_rows, d, wave, lane);
    tb.init(B, b_row0, b_rows, d, wave, lane);
    // Software pipeline.  At K-step s (after its barrier) every wave issues B(s + DB) and then A(s + DA); the
    // prologue replays the "virtual" steps -DA .. -1.  vmcnt retires in order, so at the top of step kt everything
    // issued after B(kt) -- A(kt-DB+DA .. kt-1+DA) and B(kt+1 .. kt-1+DB), WAIT_N instructions -- may stay in
    // flight while A(kt) and B(kt) are guaranteed landed.  The barrier (a) extends "landed" to every wave's pieces
    // and (b) proves all waves finished reading the slots overwritten by the loads issued right after it (their
    // MFMAs consumed those ds_reads).  Raw s_barrier: __syncthreads() would drain vmcnt to 0.
    if (ABL != 4) {
#pragma unroll
        for (int s = -C::DA; s < 0; ++s) {
            if (s + C::DB >= 0 && s + C::DB < nk)
                tb.issue(s + C::DB, C::BK, lds + C::B_BASE + ((s + C::DB) % C::B_STAGES) * C::B_BYTES, wave);
            if (s + C::DA < nk) ta.issue(s + C::DA, C::BK, lds + ((s + C::DA) % C::A_STAGES) * C::A_BYTES, wave);
        }
    }
    half8 fz;
#pragma unroll
    for (int u = 0; u < 8; ++u) fz[u] = (_Float16)(0.001f * (float)(lane + u));

    ERH_SEC(4);
    int a_slot = 0, b_slot = 0;                         // kt % A_STAGES, kt % B_STAGES
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + C::DA <= nk)                           // steady state: every load counted in WAIT_N exists
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(C::WAIT_N) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        ERH_SEC(0);
        const char *cur_a = lds + a_slot * C::A_BYTES;
        const char *cur_b = lds + b_slot * C::B_BYTES;
        if (ABL != 4) {
            if (kt + C::DB < nk) {
                int sb = b_slot + C::DB;                // (kt + DB) % B_STAGES
                if (sb >= C::B_STAGES) sb -= C::B_STAGES;
                tb.issue(kt + C::DB, C::BK, lds + C::B_BASE + sb * C::B_BYTES, wave);
            }
            if (kt + C::DA < nk) {
                int sa = a_slot + C::DA;                // (kt + DA) % A_STAGES
                if (sa >= C::A_STAGES) sa -= C::A_STAGES;
                ta.issue(kt + C::DA, C::BK, lds + sa * C::A_BYTES, wave);
            }
        }
        ERH_SEC(1);
#pragma unroll
        for (int j = 0; j < C::KS; ++j) {
            half8 af[C::MT], bf[C::NTL];
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
                af[mt] = (ABL == 3) ? fz : *reinterpret_cast<const half8 *>(cur_a + a_lane_off + mt * 32 * C::RB + soff[j]);
#pragma unroll
            for (int nt = 0; nt < C::NTL; ++nt)
                bf[nt] = (ABL == 3) ? fz : *reinterpret_cast<const half8 *>(cur_b + b_lane_off + nt * 32 * C::RB + soff[j]);
            if (ABL == 2) {
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) asm volatile("" ::"v"(af[mt]));
#pragma unroll
                for (int nt = 0; nt < C::NTL; ++nt) asm volatile("" ::"v"(bf[nt]));
            } else {
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < C::NTL; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        if (ABL == 5) asm volatile("s_nop 0" ::: "memory");   // MFMA issue only; the pipe drains in the next wait
        ERH_SEC(2);
        a_slot = (a_slot + 1 == C::A_STAGES) ? 0 : a_slot + 1;
        b_slot = (b_slot + 1 == C::B_STAGES) ? 0 : b_slot + 1;
    }
#undef ERH_SEC
}

// 32x32 MFMA C/D layout (dtype independent on gfx950): lane holds column (lane & 31) and rows
// (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), r = 0..15.
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------
// STORE epilogue: S0[q][chunk - c0] for chunks [c0, c0 + nc), all Bpad query rows.
// GROUPED (round 6; kernels.h: ErhDenseView): the 256 query rows of a tile score the seed prefix of THEIR OWN matrix -- views[q_row0 / 256]
// replaces (X, N), its prefix length replaces nc (chunk tiles past it return at once), c0 = 0.
template <class C, bool GROUPED = false>
__global__ __launch_bounds__(C::NT) void dense_scan_store_kernel(
    const _Float16 *__restrict__ Q, int Bpad, const _Float16 *__restrict__ X, int64_t N, int d,
    int64_t c0, int nc, float *__restrict__ S0, int ld_s0, const erh::ErhDenseView *__restrict__ views) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_m = wave / C::WGN, wave_n = wave % C::WGN;
    const int n_ctiles = (nc + C::BN - 1) / C::BN;
    const int ct = blockIdx.x % n_ctiles, qt = blockIdx.x / n_ctiles;
    const int64_t q_row0 = (int64_t)qt * C::BM;
    const int64_t c_row0 = c0 + (int64_t)ct * C::BN;
    if constexpr (GROUPED) {
        const erh::ErhDenseView &v = views[q_row0 >> 8];
        if (c_row0 >= v.n0) return;                      // whole workgroup, before any barrier
        X = v.X; N = v.N; nc = v.n0;
    }

    f32x16 acc[C::MT][C::NTL];
    gemm_tile<C, 0>(Q, q_row0, Bpad, X, c_row0, N, d, lds, acc, wave, lane, wave_m, wave_n);

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
template <class C, int ABL>
__global__ __launch_bounds__(C::NT) void dense_scan_append_kernel(
    const _Float16 *__restrict__ X, int64_t N, int d, int64_t c0, int64_t c1,
    const _Float16 *__restrict__ Q, int Bpad, int B,
    const float *__restrict__ tau, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap, uint32_t *__restrict__ overflow,
    unsigned long long *__restrict__ dbg /* ABL == 5 only: section clock sums of lane 0 of wave 0 */) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_m = wave / C::WGN, wave_n = wave % C::WGN;
    long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = (ABL == 5) ? clock64() : 0;

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
    gemm_tile<C, ABL>(X, c_row0, N, Q, q_row0, Bpad, d, lds, acc, wave, lane, wave_m, wave_n, tsec);
    const long long t_loop_end = (ABL == 5) ? clock64() : 0;
    if (ABL >= 1 && ABL <= 4) {
        // measurement builds: keep the accumulators alive with a never-true store, skip the epilogue
        float keep = 0.f;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[mt][nt][r];
        if (keep == 1.2345e-30f) *overflow = 7u;
        return;
    }

    // Epilogue.  Lane = query column; its MT*16 accumulators are chunk rows.  The threshold admits well under 1 %
    // of the scores, so everything is organised around "no lane of the wave survives":
    //   - per 32x32 tile one max over the lane's 16 scores and ONE wave-uniform test; only tiles with a hit look
    //     at individual registers, again with a wave-uniform test per register;
    //   - a survivor goes to the lane's own staging slots in LDS ([slot][lane], plain non-returning ds_writes, the
    //     slot counter lives in a register): no atomic and no wait inside the scan.  The staging area is the
    //     B-ring slot the last K-step did not read, so no barrier is needed;
    //   - the flush issues all staged records' global atomics back to back (one L2 round trip for the lot), then
    //     writes the records.  A lane with more than kSlots survivors appends the surplus directly.
    constexpr int kStageBytes = C::B_BYTES / C::NW;                  // per-wave share of one B slot
    constexpr int kSlots = kStageBytes / (64 * 12);                  // records per lane
    static_assert(C::B_STAGES >= 2 && kSlots >= 2, "staging area");
    const int nk_ = d / C::BK;
    char *stage = lds + C::B_BASE + (nk_ % C::B_STAGES) * C::B_BYTES + wave * kStageBytes;   // slot of step nk: unread
    float *w_s = reinterpret_cast<float *>(stage);
    int32_t *w_doc = reinterpret_cast<int32_t *>(stage + kSlots * 64 * 4);
    int32_t *w_q = reinterpret_cast<int32_t *>(stage + kSlots * 64 * 8);
    const int64_t lim = (c1 < N) ? c1 : N;
    const int64_t row_base = c_row0 + wave_m * C::WM + 4 * (lane >> 5);
    int n_mine = 0;                                                  // this lane's staged records
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt) {
        const int q = (int)q_row0 + wave_n * C::WN + nt * 32 + (lane & 31);
        const float t = (q < B) ? tau[q] : INFINITY;
        const int fd = (filter_dir && q < B) ? (int)filter_dir[q] : -1;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            float m = acc[mt][nt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[mt][nt][r]);
            if (__builtin_amdgcn_ballot_w64(m >= t) == 0) continue;   // wave-uniform: nothing in this tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = acc[mt][nt][r];
                const bool hit = sc >= t;
                if (__builtin_amdgcn_ballot_w64(hit)) {              // wave-uniform
                    if (hit) {
                        const int64_t chunk = row_base + mt * 32 + (r & 3) + 8 * (r >> 2);
                        if (chunk < lim && (fd < 0 || (int)dir_id[chunk] == fd)) {
                            if (n_mine < kSlots) {
                                const int at = n_mine * 64 + lane;
                                w_s[at] = sc;
                                w_doc[at] = (int32_t)chunk;
                                w_q[at] = q;
                                ++n_mine;
                            } else {                                 // staging full: straight to the list
                                const uint32_t pos = atomicAdd(&cand_cnt[q], 1u);
                                if (pos < (uint32_t)cap) {
                                    ErhCand c;
                                    c.s = sc;
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
    if (__builtin_amdgcn_ballot_w64(n_mine > 0)) {
        uint32_t pos[kSlots];
        int qj[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {                           // all atomics in flight before any is consumed
            qj[j] = (j < n_mine) ? w_q[j * 64 + lane] : 0;
            pos[j] = (j < n_mine) ? atomicAdd(&cand_cnt[qj[j]], 1u) : 0u;
        }
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            if (j < n_mine) {
                if (pos[j] < (uint32_t)cap) {
                    ErhCand c;
                    c.s = w_s[j * 64 + lane];
                    c.idx = w_doc[j * 64 + lane];
                    cand[(int64_t)qj[j] * cap + pos[j]] = c;
                } else {
                    atomicOr(overflow, 1u);
                }
            }
        }
    }
    if (ABL == 5 && dbg && threadIdx.x == 0) {
        const long long t_end = clock64();
        tsec[3] = t_end - t_loop_end;
        tsec[5] = t_end - t_begin;
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(&dbg[8 + i], (unsigned long long)tsec[i]);
    }
}

#ifdef ERH_MEASURE   // the lock-step persistent scan (round 1; superseded by the ping-pong scan): csrc/measure/dense_scan_persist.inc
#include "measure/dense_scan_persist.inc"
#endif  // ERH_MEASURE

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

// ---------------------------------------------------------------------------------------------
// Ping-pong persistent append scan (option "dense_pp", default on).  Same tiles, streams and results as the
// persistent kernel above; what changes is who does what when.  The 8 waves form two groups (waves 0-3 / 4-7:
// one wave of each group per SIMD, group = chunk-row half of the tile).  A K-step is BK = 32 halves ("stage").
// Per stage g there is ONE raw barrier; between barrier g and barrier g+1
//     group 0:  M(g)  then the 16 MFMAs of stage g+1          group 1:  the 16 MFMAs of stage g  then M(g)
// so each SIMD's matrix pipe has one wave feeding it from registers while its partner wave does
// the LDS/DMA work (MI355X_MICROARCH "two waves per SIMD").  Memory phase M(h): four LDS-DMA instructions per wave --
// the A stage pair (h+4, h+5) when h is even, the B pair (h+3, h+4) when h is odd, the two 64-byte halves of each
// 128-byte line back to back -- then ds_read_b128 the 12 fragments of stage h+1.
// Rings: A 5 stages, B 4 stages (16 KiB each).  Every read of stage g precedes barrier g (it happens in M(g-1)),
// so M(g) may overwrite the slots of stages g-1 and g.  Before barrier g each wave waits until everything it
// issued up to B(g+1) has landed: with M(g-1) the youngest issue, the 8 instructions of M(g-2) M(g-1) may stay in
// flight for even g, the 4 of M(g-1) for odd g -- and the barrier publishes stage g+1 to the reads of M(g).
// Epilogue (one extra phase per tile, both groups together): scores >= tau[q] become records
// {score, query-in-wave-tile | row-in-tile << 6 | tile << 14} (two 4-byte arrays) in the wave's LDS record area (ballot + mbcnt, no
// per-hit branches, no global traffic).  Records go to the global candidate lists only when a buffer is half
// full (decided for the whole workgroup one tile ahead so that all waves drain together), at the last tile,
// or -- pass repeated with a shifted window -- when one tile alone overflows the buffer.  The flush ends with
// vmcnt(0): stores and loads retire out of order with respect to each other, so no store may be outstanding
// when the counted waits resume.
namespace pp {
constexpr int BM = 256, BN = 256, BK = 32, NW = 8, NT = 512, RB = 64, PR = 4;
constexpr int AST = 5, BST = 4;
constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;
constexpr int B_BASE = AST * A_BYTES;
constexpr int REC_BASE = B_BASE + BST * B_BYTES;
constexpr int REC_BYTES = 2048;                       // per wave: 256 score words, then 256 packed-location words
constexpr int CAPW = 254;                             // records per wave; the last 16 bytes of wave 0's area hold the flush flags
constexpr int FLAG_OFF = REC_BASE + CAPW * 4;           // wave 0: score words 254, 255
constexpr int LDS_BYTES = REC_BASE + NW * REC_BYTES;
constexpr int TILE_BITS = 18;
static_assert(LDS_BYTES == 160 * 1024, "pp LDS");
}  // namespace pp

// flush flags of the ping-pong kernels: plain LDS words (a generic volatile pointer would become flat accesses, which
// count on vmcnt and drain the LDS-DMA queue at every use)
#define ERH_PP_FLAG(IDX) \
    (*reinterpret_cast<volatile __attribute__((address_space(3))) int *>(ERH_LDS_PTR(lds + pp::FLAG_OFF + 4 * (IDX))))

// maximum of four accumulator values in two instructions (fmaxf would add a canonicalising v_max per operand: MFMA
// results are never signalling NaNs, which the compiler cannot know)
__device__ __forceinline__ float erh_max4(float a, float b, float c, float d) {
    float m;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(m) : "v"(a), "v"(b), "v"(c), "v"(d));
    return m;
}

#define ERH_PP_BARRIER()                                   \
    do {                                                   \
        asm volatile("s_barrier" ::: "memory");            \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

// Epilogue of tile i for this wave (acc final); shared by both ping-pong kernels (it uses their local names).  See the
// header comment of the ping-pong scan above.
// the survivors of four accumulator registers (one 32 x 32 block row group): ballot + mbcnt compaction into the wave's records
#define ERH_PP_EPI_QUAD(MT, NT, R4) ERH_PP_EPI_QUAD_C(MT, NT, R4, pp::CAPW)
#define ERH_PP_EPI_QUAD_C(MT, NT, R4, CAPW_)                                                          \
    do {                                                                                              \
        _Pragma("unroll") for (int r = (R4); r < (R4) + 4; ++r) {                                     \
            const float sc_ = acc[MT][NT][r];                                                         \
            const bool hit_ = sc_ >= t_;                                                              \
            const unsigned long long m_ = __builtin_amdgcn_ballot_w64(hit_);                          \
            if (m_) {                                                                                 \
                const int pos_ = cnt_ - shift_ +                                                      \
                    (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32),                              \
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u));      \
                if (hit_ && (unsigned)pos_ < (unsigned)(CAPW_)) {                                     \
                    *reinterpret_cast<float *>(rec + pos_ * 4) = sc_;                                 \
                    *reinterpret_cast<uint32_t *>(rec + 1024 + pos_ * 4) =                            \
                        pk_l_ + ((uint32_t)((MT) * 32 + (r & 3) + 8 * (r >> 2)) << 6);                \
                }                                                                                     \
                cnt_ += __builtin_popcountll(m_);                                                     \
            }                                                                                         \
            __builtin_amdgcn_sched_barrier(0);   /* keeps one ballot mask live at a time */           \
        }                                                                                             \
    } while (0)
#define ERH_PP_EPILOGUE() ERH_PP_EPILOGUE_V(false, 0, 2)
// EPI2: the sixteen group tests of a query half are evaluated FIRST, branch-free, into sixteen wave masks (v_max3 + v_max +
// v_cmp with a scalar destination each: no VALU -> SALU round trip between them); the branches then run on finished masks
#define ERH_PP_EPILOGUE_V(EPI2, QMAP, NTL_)                                                                     \
    do {                                                                                              \
        if (PABL & kPpNoEpi) {                                                        \
            float keep_ = 0.f;                                                                        \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                          \
                _Pragma("unroll") for (int nt = 0; nt < (NTL_); ++nt)                                 \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) keep_ += acc[mt][nt][r];           \
            if (keep_ == 1.2345e-30f) *overflow = 7u;                                                 \
            break;                                                                                    \
        }                                                                                             \
        if (threadIdx.x == 0) ERH_PP_FLAG((i + 1) & 1) = 0;                                           \
        const bool last_ = (i + 1 == n_tiles);                                                        \
        const int fill0_ = fill;                                                                      \
        for (int shift_ = 0;; shift_ += pp::CAPW) {                                                   \
            int cnt_ = fill0_;                                                                        \
            float tt_[2] = {t_q[0], t_q[1]};                                                          \
            asm volatile("" : "+v"(tt_[0]), "+v"(tt_[1]));   /* opaque per pass: nothing of the pass is hoisted out of the loop */ \
            _Pragma("unroll") for (int nt = 0; nt < (NTL_); ++nt) {                                   \
                const float t_ = tt_[nt];                                                             \
                uint32_t pk_l_ = (uint32_t)(nt * 32 + l31) | ((uint32_t)(grp * 128 + 4 * hh) << 6) |     \
                                 ((uint32_t)i << 14);                                                 \
                asm volatile("" : "+v"(pk_l_));                                                       \
                if (EPI2) {                                                                           \
                    unsigned long long qm_[16];                                                       \
                    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                  \
                        _Pragma("unroll") for (int r4 = 0; r4 < 16; r4 += 4)                          \
                            qm_[mt * 4 + (r4 >> 2)] = __builtin_amdgcn_ballot_w64(                    \
                                erh_max4(acc[mt][nt][r4], acc[mt][nt][r4 + 1], acc[mt][nt][r4 + 2], acc[mt][nt][r4 + 3]) >= t_); \
                    unsigned long long or_ = 0ull;                                                    \
                    _Pragma("unroll") for (int j = 0; j < 16; ++j) or_ |= qm_[j];                     \
                    if (or_) {                                                                        \
                        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                            \
                            _Pragma("unroll") for (int r4 = 0; r4 < 16; r4 += 4) {                    \
                                if (qm_[mt * 4 + (r4 >> 2)]) ERH_PP_EPI_QUAD(mt, nt, r4);             \
                            }                                                                         \
                        }                                                                             \
                    }                                                                                 \
                } else {                                                                              \
                _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                    \
                    /* four accumulator registers per wave-uniform test: their maximum (v_max3 + v_max), one    \
                       compare and one branch in the common no-survivor case */                                 \
                    _Pragma("unroll") for (int r4 = 0; r4 < 16; r4 += 4) {                            \
                        const bool any_ = erh_max4(acc[mt][nt][r4], acc[mt][nt][r4 + 1], acc[mt][nt][r4 + 2],   \
                                                   acc[mt][nt][r4 + 3]) >= t_;                                 \
                        if (__builtin_amdgcn_ballot_w64(any_)) ERH_PP_EPI_QUAD(mt, nt, r4);           \
                        __builtin_amdgcn_sched_barrier(0);                                            \
                    }                                                                                 \
                }                                                                                     \
                }                                                                                     \
            }                                                                                         \
            const int avail_ = cnt_ - shift_;                                                         \
            const bool over_ = avail_ > pp::CAPW;                                                     \
            const int nrec_ = over_ ? pp::CAPW : avail_;                                              \
            if (over_ || flush_now || last_) {                                                        \
                for (int base_ = 0; base_ < nrec_; base_ += 64) {                                     \
                    const int j_ = base_ + lane;                                                      \
                    if (j_ < nrec_) {                                                                 \
                        uint2 rc_;                                                                    \
                        rc_.x = *reinterpret_cast<const uint32_t *>(rec + j_ * 4);                    \
                        rc_.y = *reinterpret_cast<const uint32_t *>(rec + 1024 + j_ * 4);             \
                        const int q_ = (QMAP) ? (int)q_row0 + (int)((rc_.y >> 5) & 1u) * 128 + wave_n * 32 + (int)(rc_.y & 31u) \
                                              : (int)q_row0 + wave_n * 64 + (int)(rc_.y & 63u);               \
                        const int64_t chunk_ = c0 + ((int64_t)stream + (int64_t)(rc_.y >> 14) * n_streams) * pp::BM + \
                                               (int64_t)((rc_.y >> 6) & 255u);                        \
                        bool ok_ = chunk_ < lim;                                                      \
                        if (ok_ && filter_dir) {                                                      \
                            const int fd_ = (int)filter_dir[q_];                                      \
                            ok_ = fd_ < 0 || (int)dir_id[chunk_] == fd_;                              \
                        }                                                                             \
                        if (ok_) {                                                                    \
                            const uint32_t p_ = atomicAdd(&cand_cnt[q_], 1u);                         \
                            if (p_ < (uint32_t)cap) {                                                 \
                                ErhCand c_;                                                           \
                                c_.s = __uint_as_float(rc_.x);                                        \
                                c_.idx = (int32_t)chunk_;                                             \
                                cand[(int64_t)q_ * cap + p_] = c_;                                    \
                            } else {                                                                  \
                                atomicOr(overflow, 1u);                                               \
                            }                                                                         \
                        }                                                                             \
                    }                                                                                 \
                }                                                                                     \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
                fill = 0;                                                                             \
            } else {                                                                                  \
                fill = nrec_;                                                                         \
            }                                                                                         \
            if (!over_) break;                                                                        \
        }                                                                                             \
        if (fill > pp::CAPW / 2 && lane == 0) ERH_PP_FLAG(i & 1) = 1;                                 \
    } while (0)


// The row-major memory segment of the ping-pong kernels (local names of the kernel that expands them): the stage pair
// (s, s+1) of one operand = the two 64-byte halves of the same 128-byte lines, issued back to back so that the second half
// hits in L1.  Used by dense_scan_pp3_kernel (and by the lean kernel of the measurement builds).
#define ERH_PP2_GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const void *)(SRC), ERH_LDS_PTR(DST), 16, 0, 0)
// the chunk side: kPpAuxX is a constant of the kernel that expands the macro -- 2 = the non-temporal hint (dense_scan_pp3_kernel VAR bit 6:
// a launch in which every chunk row is read by ONE workgroup, i.e. one query tile per matrix), 0 = the default policy
#define ERH_PP2_GLDS_X(SRC, DST) __builtin_amdgcn_global_load_lds((const void *)(SRC), ERH_LDS_PTR(DST), 16, 0, kPpAuxX)
#define ERH_PP2_ISSUE_A()                                                                             \
    do {                                                                                              \
        if (a_left > 0) {                                                                             \
            if (!(PABL & kPpNoDmaA)) {                                                                \
                int d1_ = a_dst + pp::A_BYTES;                                                        \
                if (d1_ == kABytes) d1_ = 0;                                                          \
                ERH_PP2_GLDS_X(pa[0], my_dst + a_dst);                                                  \
                ERH_PP2_GLDS_X(pa[0] + 32, my_dst + d1_);                                               \
                ERH_PP2_GLDS_X(pa[1], my_dst + a_dst + 8192);                                           \
                ERH_PP2_GLDS_X(pa[1] + 32, my_dst + d1_ + 8192);                                        \
            }                                                                                         \
            ka += 2;                                                                                  \
            int64_t inc_ = 64;                                                                        \
            if (ka == nk) { ka = 0; inc_ = 64 - (int64_t)d; }       /* wrap to column 0 of the same rows */ \
            if (ka == k0) inc_ += a_jump;                           /* tile complete: same column, next tile */ \
            pa[0] += inc_; pa[1] += inc_;                                                             \
            a_dst += 2 * pp::A_BYTES;                                                                 \
            if (a_dst >= kABytes) a_dst -= kABytes;                                                   \
            --a_left;                                                                                 \
        }                                                                                             \
    } while (0)
#define ERH_PP2_ISSUE_B()                                                                             \
    do {                                                                                              \
        if (b_left > 0) {                                                                             \
            if (!(PABL & kPpNoDmaB)) {                                                                \
                const int d0_ = pp::B_BASE + b_dst, d1_ = pp::B_BASE + ((b_dst + pp::B_BYTES) & (kBBytes - 1)); \
                ERH_PP2_GLDS(pb[0], my_dst + d0_);                                                    \
                ERH_PP2_GLDS(pb[0] + 32, my_dst + d1_);                                               \
                ERH_PP2_GLDS(pb[1], my_dst + d0_ + 8192);                                             \
                ERH_PP2_GLDS(pb[1] + 32, my_dst + d1_ + 8192);                                        \
            }                                                                                         \
            kb += 2;                                                                                  \
            int64_t inc_ = 64;                                                                        \
            if (kb == nk) { kb = 0; inc_ = 64 - (int64_t)d; }       /* same query rows for every tile */ \
            pb[0] += inc_; pb[1] += inc_;                                                             \
            b_dst = (b_dst + 2 * pp::B_BYTES) & (kBBytes - 1);                                        \
            --b_left;                                                                                 \
        }                                                                                             \
    } while (0)

// PABL (measurement builds only, -DERH_MEASURE): bit mask -- 1 no epilogue, 2 thresholds forced to +inf, 4 no MFMA,
// 8 no chunk-side DMA, 16 no query-side DMA, 32 no fragment reads, 64 phase clocks.  Anything but 0 and 64 gives
// invalid results.  The option "dense_ablate" keeps its round-1 codes (pp_mask_of below maps them).
#ifdef ERH_MEASURE
constexpr int kPp3LockStep = 1;    // VAR bit 0 of dense_scan_pp3_kernel (DMA inside the matrix segment, one barrier per stage)
#else
constexpr int kPp3LockStep = 0;    // ... a measured dead end: not in the product build
#endif
constexpr int kPpNoEpi = 1, kPpTauInf = 2, kPpNoMfma = 4, kPpNoDmaA = 8, kPpNoDmaB = 16, kPpNoFrag = 32, kPpClocks = 64;
#ifdef ERH_MEASURE   // the first two generations of the ping-pong scan (dense_scan_pp_kernel, dense_scan_pp2_kernel): csrc/measure/dense_scan_pp12.inc
#include "measure/dense_scan_pp12.inc"
#endif  // ERH_MEASURE

// ---------------------------------------------------------------------------------------------
// Tiled copy of the chunk matrix for the ping-pong scan (option "dense_tiled").  For the 256-row tile T and the
// stage pair kp the 32 KiB block at ((T * pairs + kp) * 32 KiB) holds the two 16 KiB LDS stage images back to back,
// already swizzled: piece p (16 bytes) of the image of stage 2 kp + s is row p >> 2 of the tile, logical 16-byte slot
// (p & 3) ^ ((p >> 4) & 3) of that row's 64 bytes of the stage.  Every LDS-DMA instruction of the scan then moves 1 KiB
// of consecutive bytes (eight full 128-byte lines instead of sixteen half lines with a 2 KiB stride) and a workgroup
// streams a tile as 512 KiB of consecutive addresses.  Rows past N come from the zero padding behind the matrix.
__global__ __launch_bounds__(256) void dense_tile_rows_kernel(const _Float16 *__restrict__ X, int d, int64_t n_tiles,
                                                              int4 *__restrict__ Xt) {
    const int pairs = d / 64;
    const int64_t total = n_tiles * pairs * 2048;                      // 16-byte pieces
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(o & 1023), sidx = (int)((o >> 10) & 1);
        const int64_t blk = o >> 11;
        const int kp = (int)(blk % pairs);
        const int64_t T = blk / pairs;
        const int r = p >> 2, ls = (p & 3) ^ row_swizzle<pp::PR>(r);
        Xt[o] = *reinterpret_cast<const int4 *>(X + (T * 256 + r) * (int64_t)d + (2 * kp + sidx) * 32 + ls * 8);
    }
}

// ---------------------------------------------------------------------------------------------
// Ping-pong scan, strict alternation (option "dense_pp" = 3).  Same tiles, streams, rings, DMA order, epilogue and
// results as dense_scan_pp2_kernel; what changes is what a segment contains.  Measured on the lean kernel
// (profiles/r02c_kbench_mm.log): with matrix segments and barriers ONLY a stage takes ~1300 cycles against 1024 of MFMA
// issue -- both groups' matrix segments fall between the same two barriers, the older group wins the pipe, and the
// younger group's memory segment plus the barrier are exposed once per stage; with the LDS-DMA in, a memory segment
// (12 fragment reads + 4 DMA instructions that block while the memory path is backed up) takes 720-940 cycles
// against ~500 of a matrix segment, and the stage becomes the SUM of the two memory segments.  Here
//   - there are two barriers per stage and the groups alternate strictly: between two barriers one group is in its
//     matrix segment and the other in its memory segment, so the matrix pipe changes hands without draining;
//   - the fragment reads move INTO the matrix segment: the two K sub-steps of a stage use separate fragment registers,
//     and as soon as the MFMAs that read a fragment register of stage g have been issued, the same register is
//     re-loaded with its stage g+1 contents (one ds_read_b128 behind every second MFMA), a whole segment before it
//     is used -- the matrix segment stays MFMA-paced and nothing waits for LDS;
//   - the memory segment is only the 4 DMA instructions and the counted wait, i.e. exactly the part that blocks on
//     the memory path, and it runs beside the partner's matrix segment.
// Order per group and stage g (M_h = DMA of A stages (h+4, h+5) for even h, of B stages (h+3, h+4) for odd h):
//     group 0:  C(g)  |A|  M_g, wait  |B|        group 1:  M_g  |A|  C(g), wait  |B|
// Reads of stage g+1 happen in C(g), i.e. after barrier |B| of stage g-1: before that barrier every wave has waited
// for its pieces of stage g+1 (after M_h everything but the youngest 4 (h even) / 8 (h odd) instructions has landed),
// and M_h overwrites the ring slots of stages h-1 and h, whose last reads (C(h-1)) retired before |B| of stage h-1
// (lgkmcnt(0) ahead of every |B|).
// GROUPED (VAR bit 5; round 6): ONE launch serves several matrices.  Every 256-row query tile has its own matrix (a dir's block copy,
// kernels.h: ErhDenseView) and its own set of chunk streams: workgroup b belongs to query tile gio.wg_view[b] and is stream
// b - wg0 of that tile's nwg streams, scanning rows [n0, N) of the tile's matrix.  Everything behind the prologue -- rings, segments,
// epilogue, records -- is the ordinary kernel's; the (X, N, c0, c1) arguments and the block -> (query tile, stream) map are what the table
// replaces.  Padding rows of a tile carry tau = +inf, so B = Bpad.
template <int PABL, int VAR>
__global__ __launch_bounds__(pp::NT) void dense_scan_pp3_kernel(
    const _Float16 *__restrict__ Xg, int64_t Ng, int d, int64_t c0g, int64_t c1g,
    const _Float16 *__restrict__ Q, int Bpad, int B,
    const float *__restrict__ tau, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap, uint32_t *__restrict__ overflow,
    unsigned long long *__restrict__ dbg /* kPpClocks only */, int rot_stages,
    uint32_t *__restrict__ stream_sync /* one zeroed word per stream, or null: see ERH_PP3_STREAM_SYNC */,
    const erh::ErhSeedIo sio /* VAR bit 4: this launch is the sample pass (kernels.h) */,
    const erh::ErhGroupIo gio /* VAR bit 5 */) {
    constexpr bool GROUPED = (VAR & 32) != 0;
    constexpr int kPpAuxX = (VAR & 64) ? 2 : 0;                        // chunk-side LDS-DMA: non-temporal when each chunk row has one reader
    const int g_qt = GROUPED ? gio.wg_view[blockIdx.x] : 0;
    const erh::ErhDenseView *const gv = GROUPED ? gio.views + g_qt : nullptr;
    constexpr bool GSEED = GROUPED && (VAR & 16) != 0;                 // the grouped sample pass: rows [0, seed_rows) of the tile's view
    const _Float16 *const X = GROUPED ? gv->X : Xg;
    const int64_t N = GROUPED ? gv->N : Ng;
    const int64_t c0 = GROUPED ? (GSEED ? (int64_t)0 : (int64_t)gv->n0) : c0g;
    const int64_t c1 = GROUPED ? (GSEED ? (int64_t)gv->seed_rows : gv->N) : c1g;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wave_n = wave & 3;
    long long tph[6] = {0, 0, 0, 0, 0, 0};
    long long t_mark = (PABL & kPpClocks) ? clock64() : 0;
#define ERH_PH(I) do { if (PABL & kPpClocks) { const long long n_ = clock64(); tph[I] += n_ - t_mark; t_mark = n_; } } while (0)
    const int nk = d / pp::BK;
    constexpr bool SEED = (VAR & 16) != 0;                             // sample pass: score the tiles, keep the two best of every cell, no thresholds

    const int n_qt = Bpad / pp::BN;
    const int64_t n_ct = (c1 - c0 + pp::BM - 1) / pp::BM;
    const int xcd = blockIdx.x & 7;
    const int jx = blockIdx.x >> 3;
    const int qt = GROUPED ? g_qt : jx % n_qt;
    const int stream = GROUPED ? (int)blockIdx.x - gv->wg0 : (jx / n_qt) * 8 + xcd;
    const int n_streams = GROUPED ? gv->nwg : (gridDim.x / (8 * n_qt)) * 8;
    if (stream >= n_ct) return;                                        // whole workgroup, before any barrier
    const int n_tiles = (int)((n_ct - stream + n_streams - 1) / n_streams);
    const int total = n_tiles * nk;                                    // flattened (tile, stage) sequence
    const int64_t q_row0 = (int64_t)qt * pp::BN;
    const int64_t lim = (c1 < N) ? c1 : N;

    const int l31 = lane & 31, hh = lane >> 5;
    const int k0 = ((qt * rot_stages) % nk) & ~1;                      // first K stage of every tile for this query tile
    // Query -> wave mapping: wave column wave_n holds queries wave_n * 32 + l31 (nt = 0) and 128 + wave_n * 32 + l31 (nt = 1) of
    // the tile, so a batch of at most 128 queries occupies nt = 0 of EVERY wave.  HALFQ (VAR bit 3; batches of 65 ... 128
    // queries): the nt = 1 half of the tile is not computed at all -- half the MFMAs, no fragment reads and no DMA for query
    // rows 128 ... 255 (8 KiB instead of 16 KiB per query stage) -- same chunk stream, barriers and epilogue.
    constexpr bool HALFQ = (VAR & 8) != 0;
    constexpr int NTL = HALFQ ? 1 : 2;
    float t_q[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int q = (int)q_row0 + nt * 128 + wave_n * 32 + l31;
        t_q[nt] = (!SEED && nt < NTL && q < B && !(PABL & kPpTauInf)) ? tau[q] : INFINITY;
    }
    asm volatile("" ::"v"(t_q[0]), "v"(t_q[1]));                      // loaded before the DMA stream starts (keeps vmcnt countable)

    constexpr bool TILED = (VAR & 2) != 0;                             // X is the tiled copy (dense_tile_rows_kernel), c0 % 256 == 0
    const _Float16 *pa[2], *pb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int piece = (it * pp::NW + wave) * 64 + lane;
        const int r = piece >> 2, p4 = piece & 3;
        const int ls = p4 ^ row_swizzle<pp::PR>(r);                    // logical 16-byte slot stored at physical slot p4
        pa[it] = X + (c0 + (int64_t)stream * pp::BM + r) * (int64_t)d + ls * 8 + k0 * pp::BK;
        pb[it] = Q + (q_row0 + r) * (int64_t)d + ls * 8 + k0 * pp::BK;
    }
    if (TILED)                                                         // one pointer: this wave's 1 KiB of the pair's first stage image
        pa[0] = X + ((c0 / pp::BM + stream) * (int64_t)(d / 64) + k0 / 2) * 16384 + wave * 512 + lane * 8;
    // a row -> the same row of the stream's next tile (tiled: a stage pair -> the same pair of the next tile), in halves
    const int64_t a_jump = (int64_t)n_streams * pp::BM * (int64_t)d;
    const int sw = row_swizzle<pp::PR>(l31);
    const int a_rd0 = (grp * 128 + l31) * pp::RB + ((hh ^ sw) << 4);
    const int a_rd1 = (grp * 128 + l31) * pp::RB + (((2 + hh) ^ sw) << 4);
    const int b_rd0 = pp::B_BASE + (wave_n * 32 + l31) * pp::RB + ((hh ^ sw) << 4);      // + nt * 128 rows
    const int b_rd1 = pp::B_BASE + (wave_n * 32 + l31) * pp::RB + (((2 + hh) ^ sw) << 4);
    char *const rec = lds + pp::REC_BASE + wave * pp::REC_BYTES;
    if (threadIdx.x == 0) { ERH_PP_FLAG(0) = 0; ERH_PP_FLAG(1) = 0; }

    constexpr int kABytes = pp::AST * pp::A_BYTES, kBBytes = pp::BST * pp::B_BYTES;
    int a_dst = 0, b_dst = 0, ka = k0, kb = k0;
    int a_left = total >> 1, b_left = total >> 1;                      // (nk is even: pairs never straddle a tile)
    int fa_off = 0, fb_off = 0;                                        // ring offsets of the next stage to read
    half8 fa[4][2], fb[2][2];
    char *const my_dst = lds + wave * 1024;                            // + it * 8192 + ring offset

// the chunk-side stage pair from the tiled copy: four instructions of 1 KiB of consecutive bytes each
#define ERH_PP3_ISSUE_AT()                                                                            \
    do {                                                                                              \
        if (a_left > 0) {                                                                             \
            if (!(PABL & kPpNoDmaA)) {                                                                \
                int d1_ = a_dst + pp::A_BYTES;                                                        \
                if (d1_ == kABytes) d1_ = 0;                                                          \
                ERH_PP2_GLDS_X(pa[0], my_dst + a_dst);                                                  \
                ERH_PP2_GLDS_X(pa[0] + 4096, my_dst + a_dst + 8192);                                    \
                ERH_PP2_GLDS_X(pa[0] + 8192, my_dst + d1_);                                             \
                ERH_PP2_GLDS_X(pa[0] + 12288, my_dst + d1_ + 8192);                                     \
            }                                                                                         \
            ka += 2;                                                                                  \
            int64_t inc_ = 16384;                                          /* halves: the next pair's block */ \
            if (ka == nk) { ka = 0; inc_ = 16384 - (int64_t)pp::BM * d; }  /* wrap to pair 0 of the same tile */ \
            if (ka == k0) inc_ += a_jump;                                                             \
            pa[0] += inc_;                                                                            \
            a_dst += 2 * pp::A_BYTES;                                                                 \
            if (a_dst >= kABytes) a_dst -= kABytes;                                                   \
            --a_left;                                                                                 \
        }                                                                                             \
    } while (0)
#define ERH_PP3_ISSUE_A() do { if (TILED) ERH_PP3_ISSUE_AT(); else ERH_PP2_ISSUE_A(); } while (0)
// the query-side stage pair; HALFQ: query rows 0 ... 127 only (this wave's pieces of them are the two instructions of pb[0])
#define ERH_PP3_ISSUE_B()                                                                             \
    do {                                                                                              \
        if (!HALFQ) {                                                                                 \
            ERH_PP2_ISSUE_B();                                                                        \
        } else if (b_left > 0) {                                                                      \
            if (!(PABL & kPpNoDmaB)) {                                                                \
                const int d0_ = pp::B_BASE + b_dst, d1_ = pp::B_BASE + ((b_dst + pp::B_BYTES) & (kBBytes - 1)); \
                ERH_PP2_GLDS(pb[0], my_dst + d0_);                                                    \
                ERH_PP2_GLDS(pb[0] + 32, my_dst + d1_);                                               \
            }                                                                                         \
            kb += 2;                                                                                  \
            int64_t inc_ = 64;                                                                        \
            if (kb == nk) { kb = 0; inc_ = 64 - (int64_t)d; }                                         \
            pb[0] += inc_;                                                                            \
            b_dst = (b_dst + 2 * pp::B_BYTES) & (kBBytes - 1);                                        \
            --b_left;                                                                                 \
        }                                                                                             \
    } while (0)
// one instruction (PART 0..3) of the A / B stage pair, then the pair's bookkeeping (VAR bit 0: issued from inside the
// matrix segment)
#define ERH_PP3_PART_A(PART)                                                                          \
    do {                                                                                              \
        if (a_left > 0) {                                                                             \
            if (!(PABL & kPpNoDmaA)) {                                                                \
                int d1_ = a_dst + pp::A_BYTES;                                                        \
                if (d1_ == kABytes) d1_ = 0;                                                          \
                if ((PART) == 0) ERH_PP2_GLDS_X(pa[0], my_dst + a_dst);                                 \
                if ((PART) == 1) ERH_PP2_GLDS_X(pa[0] + 32, my_dst + d1_);                              \
                if ((PART) == 2) ERH_PP2_GLDS_X(pa[1], my_dst + a_dst + 8192);                          \
                if ((PART) == 3) ERH_PP2_GLDS_X(pa[1] + 32, my_dst + d1_ + 8192);                       \
            }                                                                                         \
            if ((PART) == 3) {                                                                        \
                ka += 2;                                                                              \
                int64_t inc_ = 64;                                                                    \
                if (ka == nk) { ka = 0; inc_ = 64 - (int64_t)d; }                                     \
                if (ka == k0) inc_ += a_jump;                                                         \
                pa[0] += inc_; pa[1] += inc_;                                                         \
                a_dst += 2 * pp::A_BYTES;                                                             \
                if (a_dst >= kABytes) a_dst -= kABytes;                                               \
                --a_left;                                                                             \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#define ERH_PP3_PART_B(PART)                                                                          \
    do {                                                                                              \
        if (b_left > 0) {                                                                             \
            if (!(PABL & kPpNoDmaB)) {                                                                \
                const int d0_ = pp::B_BASE + b_dst, d1_ = pp::B_BASE + ((b_dst + pp::B_BYTES) & (kBBytes - 1)); \
                if ((PART) == 0) ERH_PP2_GLDS(pb[0], my_dst + d0_);                                   \
                if ((PART) == 1) ERH_PP2_GLDS(pb[0] + 32, my_dst + d1_);                              \
                if ((PART) == 2) ERH_PP2_GLDS(pb[1], my_dst + d0_ + 8192);                            \
                if ((PART) == 3) ERH_PP2_GLDS(pb[1] + 32, my_dst + d1_ + 8192);                       \
            }                                                                                         \
            if ((PART) == 3) {                                                                        \
                kb += 2;                                                                              \
                int64_t inc_ = 64;                                                                    \
                if (kb == nk) { kb = 0; inc_ = 64 - (int64_t)d; }                                     \
                pb[0] += inc_; pb[1] += inc_;                                                         \
                b_dst = (b_dst + 2 * pp::B_BYTES) & (kBBytes - 1);                                    \
                --b_left;                                                                             \
            }                                                                                         \
        }                                                                                             \
    } while (0)
// the twelve fragments of the stage at (fa_off, fb_off), all at once (prologue only)
#define ERH_PP3_READ_ALL()                                                                            \
    do {                                                                                              \
        const char *pa0_ = lds + (a_rd0 + fa_off), *pa1_ = lds + (a_rd1 + fa_off);                    \
        const char *pb0_ = lds + (b_rd0 + fb_off), *pb1_ = lds + (b_rd1 + fb_off);                    \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                            \
            fa[mt][0] = *reinterpret_cast<const half8 *>(pa0_ + mt * 32 * pp::RB);                    \
            fa[mt][1] = *reinterpret_cast<const half8 *>(pa1_ + mt * 32 * pp::RB);                    \
        }                                                                                             \
        _Pragma("unroll") for (int nt = 0; nt < NTL; ++nt) {                                          \
            fb[nt][0] = *reinterpret_cast<const half8 *>(pb0_ + nt * 128 * pp::RB);                   \
            fb[nt][1] = *reinterpret_cast<const half8 *>(pb1_ + nt * 128 * pp::RB);                   \
        }                                                                                             \
        fa_off += pp::A_BYTES;                                                                        \
        if (fa_off == kABytes) fa_off = 0;                                                            \
        fb_off = (fb_off + pp::B_BYTES) & (kBBytes - 1);                                              \
    } while (0)
// matrix segment of the current stage; each fragment register is re-loaded with the NEXT stage's contents right behind
// the last MFMA that reads it (past the end of the stream the reads fetch stale ring bytes that nothing uses)
#define ERH_PP3_HALF(J, PA_, PB_, FIRST, DMA)                                                            \
    do {                                                                                              \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                            \
            if (!(PABL & kPpNoMfma)) {                                                                \
                if (FIRST) {                                                                          \
                    const f32x16 z_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[0][J], z_, 0, 0, 0);  \
                    if (!HALFQ) acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[1][J], z_, 0, 0, 0);  \
                } else {                                                                              \
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[0][J], acc[mt][0], 0, 0, 0); \
                    if (!HALFQ) acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][J], fb[1][J], acc[mt][1], 0, 0, 0); \
                }                                                                                     \
            } else {                                                                                  \
                asm volatile("" ::"v"(fa[mt][J]), "v"(fb[0][J]));                                     \
                if (!HALFQ) asm volatile("" ::"v"(fb[1][J]));                                         \
                if (FIRST) { _Pragma("unroll") for (int r = 0; r < 16; ++r) { acc[mt][0][r] = 0.f; if (!HALFQ) acc[mt][1][r] = 0.f; } } \
            }                                                                                         \
            if (!(PABL & kPpNoFrag)) fa[mt][J] = *reinterpret_cast<const half8 *>(PA_ + mt * 32 * pp::RB); \
            if (DMA == 1) ERH_PP3_PART_A(mt);                                                         \
            if (DMA == 2) ERH_PP3_PART_B(mt);                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
        if (!(PABL & kPpNoFrag)) {                                                                    \
            fb[0][J] = *reinterpret_cast<const half8 *>(PB_);                                         \
            if (!HALFQ) fb[1][J] = *reinterpret_cast<const half8 *>(PB_ + 128 * pp::RB);              \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
// DMA0 / DMA1: which DMA pair (0 none, 1 the A pair, 2 the B pair) is issued, one instruction behind each MFMA pair,
// inside the first / second K sub-step of the segment (VAR bit 0 only)
#define ERH_PP3_COMPUTE(FIRST, DMA0, DMA1)                                                            \
    do {                                                                                              \
        const char *pa0_ = lds + (a_rd0 + fa_off), *pa1_ = lds + (a_rd1 + fa_off);                    \
        const char *pb0_ = lds + (b_rd0 + fb_off), *pb1_ = lds + (b_rd1 + fb_off);                    \
        ERH_PP3_HALF(0, pa0_, pb0_, FIRST, DMA0);                                                     \
        ERH_PP3_HALF(1, pa1_, pb1_, false, DMA1);                                                     \
        fa_off += pp::A_BYTES;                                                                        \
        if (fa_off == kABytes) fa_off = 0;                                                            \
        fb_off = (fb_off + pp::B_BYTES) & (kBBytes - 1);                                              \
    } while (0)
// after M_h: everything but the youngest 4 (h even: the A pair just issued) / 8 (h odd) instructions has landed; the
// fragment reads of this wave's last matrix segment have retired (ring slots may be overwritten after the barrier)
#define ERH_PP3_WAIT(H, ODD)                                                                          \
    do {                                                                                              \
        if ((H) + 6 >= total) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             \
        else if ((ODD) && HALFQ) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   /* the B pair is two instructions */ \
        else if (ODD) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                     \
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                              \
    } while (0)

// The query-tile workgroups of a stream read the same chunk tiles; left alone they drift apart (different survivor
// counts), lines leave the XCD's 4 MiB L2 between the first and the last reader, and HBM delivers 1.5x the matrix at
// B = 1024.  Every kSyncTiles tiles the workgroups of a stream meet at a counter (thread 0: one relaxed agent-scope
// add, then a bounded poll; no data changes hands, so no fences) -- pacing them to the slowest costs nothing, the
// launch ends with the slowest anyway.  Together with the K-rotation (each query tile starts a chunk tile at another
// K offset) the workgroups then miss on DIFFERENT lines of the SAME tile.
#define ERH_PP3_STREAM_SYNC()                                                                         \
    do {                                                                                              \
        if (stream_sync && n_qt > 1 && (i + 1) % kSyncTiles == 0 && i + 1 < n_tiles) {                \
            if (threadIdx.x == 0) {                                                                   \
                __hip_atomic_fetch_add(stream_sync + stream, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                const uint32_t want_ = (uint32_t)n_qt * (uint32_t)((i + 1) / kSyncTiles);             \
                for (int spin_ = 0; spin_ < 8192; ++spin_) {                                          \
                    if (__hip_atomic_load(stream_sync + stream, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want_) break; \
                    __builtin_amdgcn_s_sleep(8);                                                      \
                }                                                                                     \
            }                                                                                         \
            ERH_PP_BARRIER();                                                                         \
        }                                                                                             \
    } while (0)
    constexpr int kSyncTiles = 4;

// Sample pass (VAR bit 4): of the tile's scores only the two best of every cell -- this lane's 64 rows of one query column --
// leave the registers.  Three VALU instructions per score, written as inline asm: left to the compiler the chain is
// re-associated across the 64 accumulators and spills into the main loop.
#define ERH_PP3_SEED_EPILOGUE()                                                                       \
    do {                                                                                              \
        _Pragma("unroll") for (int nt = 0; nt < NTL; ++nt) {                                          \
            float t0_ = -INFINITY, t1_ = -INFINITY;                                                   \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                        \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                      \
                    float a_;                                                                         \
                    asm volatile("v_min_f32 %2, %0, %3\n\tv_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %2" \
                                 : "+v"(t0_), "+v"(t1_), "=&v"(a_)                                    \
                                 : "v"(acc[mt][nt][r]));                                              \
                }                                                                                     \
            }                                                                                         \
            const int q_ = (int)q_row0 + nt * 128 + wave_n * 32 + l31;                                \
            const int cell_ = (i * n_streams + stream) * 4 + grp * 2 + hh;                            \
            if (q_ < B) {                                                                             \
                float2 top_;                                                                          \
                top_.x = t0_;                                                                         \
                top_.y = t1_;                                                                         \
                *reinterpret_cast<float2 *>(sio.seed_top + ((int64_t)q_ * sio.n_cells + cell_) * 2) = top_; \
            }                                                                                         \
        }                                                                                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     /* stores and loads retire out of order: none outstanding when the counted waits resume */ \
    } while (0)
#define ERH_PP3_EPILOGUE()                                                                            \
    do {                                                                                              \
        if constexpr (SEED) { ERH_PP3_SEED_EPILOGUE(); }                                              \
        else { ERH_PP_EPILOGUE_V(true, 1, NTL); }     /* mask-first group tests: profiles/r04k_kbench_epi2.log */ \
    } while (0)

    // prologue: A(0,1) B(0,1) A(2,3) B(2,3); stages 0 and 1 complete = the last 8 instructions may stay in flight
    ERH_PP3_ISSUE_A();
    ERH_PP3_ISSUE_B();
    ERH_PP3_ISSUE_A();
    ERH_PP3_ISSUE_B();
    if (HALFQ) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    ERH_PP_BARRIER();
    ERH_PP3_READ_ALL();                                                // stage 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ERH_PP_BARRIER();                                                  // M_0 overwrites the slots of stage 0

    f32x16 acc[4][2];
    int fill = 0;                                                      // records buffered in this wave's area
    int flush_now = 0;                                                 // workgroup-uniform, decided one tile ahead
    int g = 0;

    if (VAR & 1) {
        // no memory segments at all: every wave issues the stage's DMA pair from inside its matrix segment (the older
        // group behind the MFMA pairs of the first K sub-step, the younger group behind those of the second, so the two
        // waves of a SIMD block on the memory path at different times) and there is one barrier per stage
#define ERH_PP3_TILE_LOOP(D0A, D1A, D0B, D1B)                                                         \
        for (int i = 0; i < n_tiles; ++i) {                                                           \
            for (int kt = 0; kt < nk; kt += 2, g += 2) {                                              \
                ERH_PP3_COMPUTE(kt == 0, D0A, D1A);                    /* C(g) with M_g inside */     \
                ERH_PH(0);                                                                            \
                ERH_PP3_WAIT(g, false);                                                               \
                ERH_PH(1);                                                                            \
                ERH_PP_BARRIER();                                                                     \
                ERH_PH(2);                                                                            \
                ERH_PP3_COMPUTE(false, D0B, D1B);                      /* C(g+1) with M_{g+1} inside */ \
                ERH_PH(0);                                                                            \
                ERH_PP3_WAIT(g + 1, true);                                                            \
                ERH_PH(1);                                                                            \
                ERH_PP_BARRIER();                                                                     \
                ERH_PH(2);                                                                            \
            }                                                                                         \
            ERH_PP3_EPILOGUE();                                                                        \
            ERH_PH(4);                                                                                \
            ERH_PP_BARRIER();                                                                         \
            ERH_PH(5);                                                                                \
            if (!(PABL & kPpNoEpi)) flush_now = __builtin_amdgcn_readfirstlane(ERH_PP_FLAG(i & 1));   \
            ERH_PP3_STREAM_SYNC();                                                                    \
        }
        if (grp == 0) { ERH_PP3_TILE_LOOP(1, 0, 2, 0) } else { ERH_PP3_TILE_LOOP(0, 1, 0, 2) }
#undef ERH_PP3_TILE_LOOP
    } else if (grp == 0) {
        for (int i = 0; i < n_tiles; ++i) {
            for (int kt = 0; kt < nk; kt += 2, g += 2) {
                ERH_PP3_COMPUTE(kt == 0, 0, 0);                              // C(g)
                ERH_PH(0);
                ERH_PP_BARRIER();                                      // |A|
                ERH_PH(2);
                ERH_PP3_ISSUE_A();                                     // M_g
                __builtin_amdgcn_sched_barrier(0);
                ERH_PH(3);
                ERH_PP3_WAIT(g, false);
                ERH_PH(1);
                ERH_PP_BARRIER();                                      // |B|
                ERH_PH(2);
                ERH_PP3_COMPUTE(false, 0, 0);                                // C(g+1)
                ERH_PH(0);
                ERH_PP_BARRIER();
                ERH_PH(2);
                ERH_PP3_ISSUE_B();                                     // M_{g+1}
                __builtin_amdgcn_sched_barrier(0);
                ERH_PH(3);
                ERH_PP3_WAIT(g + 1, true);
                ERH_PH(1);
                ERH_PP_BARRIER();
                ERH_PH(2);
            }
            ERH_PP3_EPILOGUE();
            ERH_PH(4);
            ERH_PP_BARRIER();
            ERH_PH(5);
            if (!(PABL & kPpNoEpi)) flush_now = __builtin_amdgcn_readfirstlane(ERH_PP_FLAG(i & 1));
            ERH_PP3_STREAM_SYNC();
        }
    } else {
        for (int i = 0; i < n_tiles; ++i) {
            for (int kt = 0; kt < nk; kt += 2, g += 2) {
                ERH_PP3_ISSUE_A();                                     // M_g
                __builtin_amdgcn_sched_barrier(0);
                ERH_PH(3);
                ERH_PP_BARRIER();                                      // |A|
                ERH_PH(2);
                ERH_PP3_COMPUTE(kt == 0, 0, 0);                              // C(g)
                ERH_PH(0);
                ERH_PP3_WAIT(g, false);
                ERH_PH(1);
                ERH_PP_BARRIER();                                      // |B|
                ERH_PH(2);
                ERH_PP3_ISSUE_B();                                     // M_{g+1}
                __builtin_amdgcn_sched_barrier(0);
                ERH_PH(3);
                ERH_PP_BARRIER();
                ERH_PH(2);
                ERH_PP3_COMPUTE(false, 0, 0);                                // C(g+1)
                ERH_PH(0);
                ERH_PP3_WAIT(g + 1, true);
                ERH_PH(1);
                ERH_PP_BARRIER();
                ERH_PH(2);
            }
            ERH_PP3_EPILOGUE();
            ERH_PH(4);
            ERH_PP_BARRIER();
            ERH_PH(5);
            if (!(PABL & kPpNoEpi)) flush_now = __builtin_amdgcn_readfirstlane(ERH_PP_FLAG(i & 1));
            ERH_PP3_STREAM_SYNC();
        }
    }
    if ((PABL & kPpClocks) && dbg && lane == 0 && (wave == 0 || wave == 4)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(&dbg[grp * 8 + i], (unsigned long long)tph[i]);
        atomicAdd(&dbg[grp * 8 + 6], (unsigned long long)total);
    }
#undef ERH_PH
#undef ERH_PP3_READ_ALL
#undef ERH_PP3_STREAM_SYNC
#undef ERH_PP3_ISSUE_A
#undef ERH_PP3_ISSUE_AT
#undef ERH_PP3_SEED_EPILOGUE
#undef ERH_PP3_EPILOGUE
#undef ERH_PP3_ISSUE_B
#undef ERH_PP3_PART_A
#undef ERH_PP3_PART_B
#undef ERH_PP3_HALF
#undef ERH_PP3_COMPUTE
#undef ERH_PP3_WAIT
}
#undef ERH_PP2_GLDS
#undef ERH_PP2_GLDS_X
#undef ERH_PP2_ISSUE_A
#undef ERH_PP2_ISSUE_B

#ifdef ERH_MEASURE   // the tiled-operand ping-pong scan of round 4 (dense_scan_pp4_kernel; measured equal to pp3: profiles/r04e_kbench_pp4.log): csrc/measure/dense_scan_pp4.inc
#include "measure/dense_scan_pp4.inc"
#endif  // ERH_MEASURE

// ---------------------------------------------------------------------------------------------
// Ping-pong scan on a 384 x 256 tile over TILED operands (batches padded to >= 512 queries; round 4).  The loop of
// scripts/ubench/scan_tile384.hip as a product kernel: same waves, strict alternation (two barriers per 32-half stage),
// epilogue, records and results as dense_scan_pp3_kernel, but a workgroup holds 384 chunk rows against its 256 queries --
// 17 % fewer LDS fill bytes per MAC -- which is worth 7-8 % of the bare loop where the chunk side hits in L2, i.e. where two
// or more query-tile workgroups share a chunk stream (profiles/r04v_ubench_scan_tile384.log; at one query tile per stream
// the shallower rings lose 6 %, so 256-query batches stay on pp3).  What it takes to fit 256 VGPRs and 160 KiB:
//   - 6 x 2 accumulator tiles per wave (192 VGPRs) and fragment registers for HALF a stage (32): each is re-loaded behind its
//     last MFMA with the next half-stage's contents (the second K sub-step of the same stage, then the first of the next);
//   - rings of 4 x 24 KiB (chunk side, three stages ahead) + 3 x 16 KiB (query side, two ahead) -- the record area and the
//     flush flags stay where the 256-row kernels have them; the query side of stage g + 2 is issued FIRST in memory segment g
//     and waited for there (vmcnt(3): only the three chunk-side instructions of stage g + 3 stay in flight), because the
//     next matrix segment re-loads from stage g + 2;
//   - both operands from tiled copies (one LDS stage image per tile and stage, swizzle included: launch_dense_tile_rows_n
//     with 384 rows for the chunk matrix -- built once per erh_set_dense, on first use -- and 256 rows for the query block
//     of the call): every DMA instruction moves 1 KiB of consecutive bytes, addresses are a wave-uniform pointer + lane * 16;
//   - a record carries nine row bits (tile index: 17 bits).
namespace pp5 {
constexpr int BM = 384, MT = 6, GROWS = 192, ROW_BITS = 9, AST = 4, BST = 3;
constexpr int A_BYTES = BM * pp::RB;
constexpr int B_BASE = AST * A_BYTES;
constexpr int TILE_BITS = 32 - 6 - ROW_BITS;
// Records per wave: 190 of the 256 slots; the last 64 words of the wave's LOCATION array hold its 64 pruning thresholds (round 5).
// Kept in registers across the main loop they were spilled to scratch and re-loaded at the top of every epilogue -- a scratch
// load counts on vmcnt, so each tile waited for the three LDS-DMA instructions in flight (the NEXT tile's first chunk stages)
// before its first compare.  From LDS the wait is lgkmcnt only.  (The flush flags stay in wave 0's score words 254 / 255.)
constexpr int CAPW = 190;
constexpr int TAU_OFF = 1024 + 192 * 4;          // byte offset in the wave's record area
static_assert(CAPW <= 192 && TAU_OFF + 64 * 4 == pp::REC_BYTES && pp::CAPW >= CAPW, "pp5 record area layout");
static_assert(B_BASE + BST * pp::B_BYTES == pp::REC_BASE, "records and flush flags sit where the 256-row kernels have them");
}  // namespace pp5

__global__ __launch_bounds__(pp::NT) void dense_scan_pp5_kernel(
    const _Float16 *__restrict__ Xt, int64_t N, int d, int64_t c0, int64_t c1,
    const _Float16 *__restrict__ Qt, int Bpad, int B,
    const float *__restrict__ tau, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap, uint32_t *__restrict__ overflow, int rot_stages) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wave_n = wave & 3;
    const int nk = d / pp::BK;
    const int n_qt = Bpad / pp::BN;
    const int64_t n_ct = (c1 - c0 + pp5::BM - 1) / pp5::BM;
    const int xcd = blockIdx.x & 7;
    const int jx = blockIdx.x >> 3;
    const int qt = jx % n_qt;
    const int stream = (jx / n_qt) * 8 + xcd;
    const int n_streams = (gridDim.x / (8 * n_qt)) * 8;
    if (stream >= n_ct) return;                                        // whole workgroup, before any barrier
    const int n_tiles = (int)((n_ct - stream + n_streams - 1) / n_streams);
    const int total = n_tiles * nk;                                    // flattened (tile, stage) sequence
    const int64_t q_row0 = (int64_t)qt * pp::BN;
    const int64_t lim = (c1 < N) ? c1 : N;
    const int l31 = lane & 31, hh = lane >> 5;
    const int k0 = (qt * rot_stages) % nk;
    {   // this wave's 64 thresholds -> LDS (loaded and parked before the DMA stream starts: vmcnt stays countable)
        float t_q[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int q = (int)q_row0 + wave_n * 64 + nt * 32 + l31;
            t_q[nt] = q < B ? tau[q] : INFINITY;
        }
        char *const tq = lds + pp::REC_BASE + wave * pp::REC_BYTES + pp5::TAU_OFF + l31 * 4;   // (both halves of the wave write the same values)
        *reinterpret_cast<float *>(tq) = t_q[0];
        *reinterpret_cast<float *>(tq + 128) = t_q[1];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }

    // wave-uniform byte pointers (this wave's 1 KiB slice of a stage image; the lane's 16 bytes are lane_off) -- the loads take
    // the scalar-base form, no 64-bit address registers
    const int64_t tile_bytes_a = (int64_t)nk * pp5::A_BYTES, tile_bytes_b = (int64_t)nk * pp::B_BYTES;
    const char *xa = reinterpret_cast<const char *>(Xt) + ((c0 / pp5::BM + stream) * (int64_t)nk + k0) * pp5::A_BYTES + wave * 1024;
    const char *const qb = reinterpret_cast<const char *>(Qt) + (int64_t)qt * tile_bytes_b + wave * 1024;
    const int64_t a_jump = (int64_t)n_streams * tile_bytes_a;         // a stage -> the same stage of the stream's next tile
    // fragment read addresses of the first K sub-step; the second one's 16-byte slot has bit 1 flipped: address ^ 32,
    // re-computed where it is needed (by an asm statement the compiler cannot hoist) instead of held in registers
    // ONE per-lane address (row l31 of a 32-row block, this lane's slot); what distinguishes the operands and the waves is
    // wave-uniform and rides on the scalar ring offsets (multiples of 64, so the ^ 32 commutes with them)
    // ... and even that one is re-derived from the lane index in every matrix segment (eight VALU instructions beside 24 MFMAs)
    // instead of held: row (lane & 31) * 64 + ((lane >> 5) ^ ((lane >> 2) & 3)) * 16
    static_assert(pp::PR == 4 && pp::RB == 64, "ERH_PP5_FADR spells out the swizzle of 64-byte rows");
#define ERH_PP5_FADR(F)                                                                               \
    do {                                                                                              \
        int t_, u_;                                                                                   \
        asm volatile("v_mbcnt_lo_u32_b32 %1, -1, 0\n\tv_mbcnt_hi_u32_b32 %1, -1, %1\n\tv_and_b32 %0, 31, %1\n\t"   \
                     "v_lshrrev_b32 %1, 5, %1\n\tv_bfe_u32 %2, %0, 2, 2\n\tv_xor_b32 %1, %1, %2\n\t"                \
                     "v_lshlrev_b32 %0, 6, %0\n\tv_lshl_or_b32 %0, %1, 4, %0"                                       \
                     : "=&v"(F), "=&v"(t_), "=&v"(u_));                                               \
    } while (0)
    const int a_wave = __builtin_amdgcn_readfirstlane(grp * pp5::GROWS * pp::RB);
    const int b_wave = __builtin_amdgcn_readfirstlane(pp5::B_BASE + wave_n * 64 * pp::RB);
    char *const rec = lds + pp::REC_BASE + wave * pp::REC_BYTES;
    if (threadIdx.x == 0) { ERH_PP_FLAG(0) = 0; ERH_PP_FLAG(1) = 0; }

    constexpr int kABytes = pp5::AST * pp5::A_BYTES, kBBytes = pp5::BST * pp::B_BYTES;
    int a_dst = 0, b_dst = 0, ka = k0, kb = k0;
    int a_left = total, b_left = total;
    int fa_off = 0, fb_off = 0;                                        // ring offsets of the stage whose FIRST half is read next
    int cur_a = 0, cur_b = 0;                                          // ... of the stage whose SECOND half is read next
    half8 fa[pp5::MT], fb[2];
    // LDS-DMA by inline assembly in the scalar-base form: wave-uniform 64-bit source in SGPRs + this lane's 32-bit byte offset,
    // destination = M0 (wave-uniform LDS byte address; lane i lands at + 16 i).  Through the builtin the compiler keeps a 64-bit
    // per-lane pointer per operand, which this kernel has no registers for.
    const uint32_t lds0 = (uint32_t)(uintptr_t)ERH_LDS_PTR(lds);       // (0: the kernel has no static LDS)
    const uint32_t my_dst = lds0 + (uint32_t)wave * 1024u;
#define ERH_PP5_GLDS(SRC, DST)                                                                        \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                    \
                 :: "s"((uint32_t)(DST)), "v"(lo_), "s"((const char *)(SRC)) : "memory", "m0")
// this lane's byte offset inside its wave's 1 KiB slice, re-derived from the lane index (two VALU instructions per issue
// instead of a register held across the main loop)
#define ERH_PP5_LANE_OFF(V) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(V))
#define ERH_PP5_ISSUE_A()                                                                             \
    do {                                                                                              \
        if (a_left > 0) {                                                                             \
            int lo_;                                                                                  \
            ERH_PP5_LANE_OFF(lo_);                                                                    \
            ERH_PP5_GLDS(xa, my_dst + a_dst);                                                         \
            ERH_PP5_GLDS(xa + 8192, my_dst + a_dst + 8192);                                           \
            ERH_PP5_GLDS(xa + 16384, my_dst + a_dst + 16384);                                         \
            xa += pp5::A_BYTES;                                                                       \
            if (++ka == nk) { ka = 0; xa -= tile_bytes_a; }                /* wrap to stage 0 of the same tile */ \
            if (ka == k0) xa += a_jump;                                    /* tile complete: the stream's next tile */ \
            a_dst += pp5::A_BYTES;                                                                    \
            if (a_dst == kABytes) a_dst = 0;                                                          \
            --a_left;                                                                                 \
        }                                                                                             \
    } while (0)
#define ERH_PP5_ISSUE_B()                                                                             \
    do {                                                                                              \
        if (b_left > 0) {                                                                             \
            int lo_;                                                                                  \
            ERH_PP5_LANE_OFF(lo_);                                                                    \
            const char *q_ = qb + (int64_t)kb * pp::B_BYTES;                                          \
            ERH_PP5_GLDS(q_, my_dst + pp5::B_BASE + b_dst);                                           \
            ERH_PP5_GLDS(q_ + 8192, my_dst + pp5::B_BASE + b_dst + 8192);                             \
            if (++kb == nk) kb = 0;                                                                   \
            b_dst += pp::B_BYTES;                                                                     \
            if (b_dst == kBBytes) b_dst = 0;                                                          \
            --b_left;                                                                                 \
        }                                                                                             \
    } while (0)
// one K sub-step: 6 x 2 MFMAs; every fragment register is re-loaded from LDS address PA_ / PB_ (VGPR byte address + immediate
// offset) right behind the last MFMA that reads it.  The re-loads are inline assembly so that they land IN the register they
// replace (left to the compiler the loaded values get registers of their own until the old ones die: sixteen more than this
// kernel has); whatever the previous sub-step re-loaded is complete behind the lgkmcnt(0) at the top.  (Past the end of the
// stream the reads fetch stale ring bytes that nothing uses.)
#define ERH_PP5_LD(DST, ADR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(DST) : "v"(ADR), "n"(OFF) : "memory")   /* "+": the register it replaces */
#define ERH_PP5_HALF(PA_, PB_, FIRST)                                                                 \
    do {                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        _Pragma("unroll") for (int mt = 0; mt < pp5::MT; ++mt) {                                      \
            if (FIRST) {                                                                              \
                const f32x16 z_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt], fb[0], z_, 0, 0, 0);      \
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt], fb[1], z_, 0, 0, 0);      \
            } else {                                                                                  \
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt], fb[0], acc[mt][0], 0, 0, 0); \
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt], fb[1], acc[mt][1], 0, 0, 0); \
            }                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            ERH_PP5_LD(fa[mt], PA_, mt * 32 * pp::RB);                                                \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
        ERH_PP5_LD(fb[0], PB_, 0);                                                                    \
        ERH_PP5_LD(fb[1], PB_, 32 * pp::RB);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
// matrix segment of stage g: the first sub-step re-loads with the second half of stage g (cur_*), the second one with the
// first half of stage g + 1 (f*_off)
#define ERH_PP5_COMPUTE(FIRST)                                                                        \
    do {                                                                                              \
        /* LDS byte addresses (the dynamic LDS block starts at 0: this kernel has no static __shared__) */ \
        { int pa_, pb_, f_;                                                                           \
          ERH_PP5_FADR(f_);                                                                           \
          asm volatile("v_xor_b32 %0, 32, %2\n\tv_add_u32 %1, %4, %0\n\tv_add_u32 %0, %3, %0"         \
                       : "=&v"(pa_), "=&v"(pb_) : "v"(f_), "s"(a_wave + cur_a), "s"(b_wave + cur_b)); \
          ERH_PP5_HALF(pa_, pb_, FIRST); }                                                            \
        { int pa_, pb_, f_;                                                                           \
          ERH_PP5_FADR(f_);                                                                           \
          asm volatile("v_add_u32 %0, %3, %2\n\tv_add_u32 %1, %4, %2"                                 \
                       : "=&v"(pa_), "=&v"(pb_) : "v"(f_), "s"(a_wave + fa_off), "s"(b_wave + fb_off)); \
          ERH_PP5_HALF(pa_, pb_, false); }                                                            \
        cur_a = __builtin_amdgcn_readfirstlane(fa_off);                                               \
        cur_b = __builtin_amdgcn_readfirstlane(fb_off);                                               \
        fa_off += pp5::A_BYTES;                                                                       \
        if (fa_off == kABytes) fa_off = 0;                                                            \
        fb_off += pp::B_BYTES;                                                                        \
        if (fb_off == kBBytes) fb_off = 0;                                                            \
    } while (0)
// after M_g (query stage g+2, then chunk stage g+3 issued): everything but the three chunk-side instructions has landed, i.e.
// stage g+2; the fragment reads of this wave's last matrix segment have retired (ring slots may be overwritten after the barrier)
#define ERH_PP5_WAIT(G)                                                                               \
    do {                                                                                              \
        if ((G) + 4 >= total) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             \
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                              \
    } while (0)
// Epilogue of tile i for this wave: ERH_PP_EPILOGUE's group tests over six block rows, nine row bits in the record.  The thresholds
// come from the wave's LDS slots (pp5::TAU_OFF), not from registers held -- i.e. spilled -- across the main loop: -0.7 % scan time,
// no scratch access inside the tile loop.  (Measured and NOT kept: pp3's mask-first group tests in two halves of three block rows,
// +-0 here; with no epilogue at all the scan class is 5.6 % faster -- profiles/r05e_ab_pp5_epilogue.log.)
#define ERH_PP5_EPILOGUE()                                                                            \
    do {                                                                                              \
        int lane, l31, hh;      /* re-derived here (opaque to the compiler): not held in registers across the main loop */ \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_and_b32 %1, 31, %0\n\tv_lshrrev_b32 %2, 5, %0" \
                     : "=&v"(lane), "=&v"(l31), "=&v"(hh));                                           \
        if (wave == 0 && lane == 0) ERH_PP_FLAG((i + 1) & 1) = 0;                                     \
        const bool last_ = (i + 1 == n_tiles);                                                        \
        const int fill0_ = fill;                                                                      \
        for (int shift_ = 0;; shift_ += pp5::CAPW) {                                                   \
            int cnt_ = fill0_;                                                                        \
            float tt_[2];                                                                             \
            {                                                                                         \
                const uint32_t ta_ = lds0 + (uint32_t)(pp::REC_BASE + pp5::TAU_OFF) + (uint32_t)wave * (uint32_t)pp::REC_BYTES + \
                                     (uint32_t)l31 * 4u;                                              \
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)"   \
                             : "=&v"(tt_[0]), "=&v"(tt_[1]) : "v"(ta_) : "memory");                  \
            }                                                                                         \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                        \
                const float t_ = tt_[nt];                                                             \
                uint32_t pk_l_ = (uint32_t)(nt * 32 + l31) | ((uint32_t)(grp * pp5::GROWS + 4 * hh) << 6) | \
                                 ((uint32_t)i << (6 + pp5::ROW_BITS));                                \
                asm volatile("" : "+v"(pk_l_));                                                       \
                _Pragma("unroll") for (int mt = 0; mt < pp5::MT; ++mt) {                              \
                    _Pragma("unroll") for (int r4 = 0; r4 < 16; r4 += 4) {                            \
                        const bool any_ = erh_max4(acc[mt][nt][r4], acc[mt][nt][r4 + 1], acc[mt][nt][r4 + 2],   \
                                                   acc[mt][nt][r4 + 3]) >= t_;                                 \
                        if (__builtin_amdgcn_ballot_w64(any_)) ERH_PP_EPI_QUAD_C(mt, nt, r4, pp5::CAPW);           \
                        __builtin_amdgcn_sched_barrier(0);                                            \
                    }                                                                                 \
                }                                                                                     \
            }                                                                                         \
            const int avail_ = cnt_ - shift_;                                                         \
            const bool over_ = avail_ > pp5::CAPW;                                                     \
            const int nrec_ = over_ ? pp5::CAPW : avail_;                                              \
            if (over_ || flush_now || last_) {                                                        \
                for (int base_ = 0; base_ < nrec_; base_ += 64) {                                     \
                    const int j_ = base_ + lane;                                                      \
                    if (j_ < nrec_) {                                                                 \
                        uint2 rc_;                                                                    \
                        rc_.x = *reinterpret_cast<const uint32_t *>(rec + j_ * 4);                    \
                        rc_.y = *reinterpret_cast<const uint32_t *>(rec + 1024 + j_ * 4);             \
                        const int q_ = (int)q_row0 + wave_n * 64 + (int)(rc_.y & 63u);                \
                        const int64_t chunk_ = c0 + ((int64_t)stream + (int64_t)(rc_.y >> (6 + pp5::ROW_BITS)) * n_streams) * pp5::BM + \
                                               (int64_t)((rc_.y >> 6) & ((1u << pp5::ROW_BITS) - 1u)); \
                        bool ok_ = chunk_ < lim;                                                      \
                        if (ok_ && filter_dir) {                                                      \
                            const int fd_ = (int)filter_dir[q_];                                      \
                            ok_ = fd_ < 0 || (int)dir_id[chunk_] == fd_;                              \
                        }                                                                             \
                        if (ok_) {                                                                    \
                            const uint32_t p_ = atomicAdd(&cand_cnt[q_], 1u);                         \
                            if (p_ < (uint32_t)cap) {                                                 \
                                ErhCand c_;                                                           \
                                c_.s = __uint_as_float(rc_.x);                                        \
                                c_.idx = (int32_t)chunk_;                                             \
                                cand[(int64_t)q_ * cap + p_] = c_;                                    \
                            } else {                                                                  \
                                atomicOr(overflow, 1u);                                               \
                            }                                                                         \
                        }                                                                             \
                    }                                                                                 \
                }                                                                                     \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
                fill = 0;                                                                             \
            } else {                                                                                  \
                fill = nrec_;                                                                         \
            }                                                                                         \
            if (!over_) break;                                                                        \
        }                                                                                             \
        if (fill > pp5::CAPW / 2 && lane == 0) ERH_PP_FLAG(i & 1) = 1;                                 \
    } while (0)

    // prologue: A0 B0 A1 B1 A2; stages 0 and 1 complete = the last 3 instructions may stay in flight
    ERH_PP5_ISSUE_A(); ERH_PP5_ISSUE_B(); ERH_PP5_ISSUE_A(); ERH_PP5_ISSUE_B(); ERH_PP5_ISSUE_A();
    if (total >= 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ERH_PP_BARRIER();
    {
        int f_adr0;
        ERH_PP5_FADR(f_adr0);
#pragma unroll
        for (int mt = 0; mt < pp5::MT; ++mt) fa[mt] = *reinterpret_cast<const half8 *>(lds + f_adr0 + a_wave + mt * 32 * pp::RB);
        fb[0] = *reinterpret_cast<const half8 *>(lds + f_adr0 + b_wave);
        fb[1] = *reinterpret_cast<const half8 *>(lds + f_adr0 + b_wave + 32 * pp::RB);
    }
    fa_off = pp5::A_BYTES;                                             // (stage 0's first half is in registers, its second half is cur_* = 0)
    fb_off = pp::B_BYTES;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ERH_PP_BARRIER();

    f32x16 acc[pp5::MT][2];
    int fill = 0;                                                      // records buffered in this wave's area
    int flush_now = 0;                                                 // workgroup-uniform, decided one tile ahead
    int g = 0;
    if (grp == 0) {
        for (int i = 0; i < n_tiles; ++i) {
            for (int kt = 0; kt < nk; ++kt, ++g) {
                ERH_PP5_COMPUTE(kt == 0);                              // C(g)
                ERH_PP_BARRIER();                                      // |A|
                ERH_PP5_ISSUE_B();                                     // M_g: the query side first (it is waited for)
                ERH_PP5_ISSUE_A();
                __builtin_amdgcn_sched_barrier(0);
                ERH_PP5_WAIT(g);
                ERH_PP_BARRIER();                                      // |B|
            }
            ERH_PP5_EPILOGUE();
            ERH_PP_BARRIER();
            flush_now = __builtin_amdgcn_readfirstlane(ERH_PP_FLAG(i & 1));
        }
    } else {
        for (int i = 0; i < n_tiles; ++i) {
            for (int kt = 0; kt < nk; ++kt, ++g) {
                ERH_PP5_ISSUE_B();                                     // M_g
                ERH_PP5_ISSUE_A();
                __builtin_amdgcn_sched_barrier(0);
                ERH_PP_BARRIER();                                      // |A|
                ERH_PP5_COMPUTE(kt == 0);                              // C(g)
                ERH_PP5_WAIT(g);
                ERH_PP_BARRIER();                                      // |B|
            }
            ERH_PP5_EPILOGUE();
            ERH_PP_BARRIER();
            flush_now = __builtin_amdgcn_readfirstlane(ERH_PP_FLAG(i & 1));
        }
    }
#undef ERH_PP5_GLDS
#undef ERH_PP5_ISSUE_A
#undef ERH_PP5_ISSUE_B
#undef ERH_PP5_HALF
#undef ERH_PP5_LD
#undef ERH_PP5_FADR
#undef ERH_PP5_LANE_OFF
#undef ERH_PP5_COMPUTE
#undef ERH_PP5_WAIT
#undef ERH_PP5_EPILOGUE
}

// tiled copy with `rows` rows per tile (384: the chunk matrix for dense_scan_pp5_kernel): per tile and 32-half stage one block
// of rows * 64 bytes that IS the LDS stage image -- piece p (16 bytes) = row p >> 2 of the tile, logical slot (p & 3) ^ swizzle
__global__ __launch_bounds__(256) void dense_tile_rows_n_kernel(const _Float16 *__restrict__ X, int d, int rows, int64_t n_tiles,
                                                                int4 *__restrict__ Xt) {
    const int nk = d / 32, per = rows * 4;
    const int64_t total = n_tiles * nk * per;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(o % per);
        const int64_t blk = o / per;
        const int s = (int)(blk % nk);
        const int64_t T = blk / nk;
        const int r = p >> 2, ls = (p & 3) ^ row_swizzle<pp::PR>(r);
        Xt[o] = *reinterpret_cast<const int4 *>(X + (T * rows + r) * (int64_t)d + s * 32 + ls * 8);
    }
}

using Cfg0 = ScanCfg<256, 256, 2, 4, 64, 3, 2>;   // one 8-wave workgroup per CU, 160 KiB LDS
using Cfg1 = ScanCfg<128, 256, 1, 4, 32, 4, 3>;   // two 4-wave workgroups per CU, 80 KiB LDS each
using Cfg2 = ScanCfg<256, 256, 2, 4, 32, 5, 5>;   // one workgroup per CU, BK 32, both operands 4 half-steps ahead
static_assert(Cfg2::LDS_BYTES == 160 * 1024, "cfg2");
static_assert(Cfg0::WAIT_N == 4 && Cfg0::LDS_BYTES == 160 * 1024, "cfg0");
static_assert(Cfg1::LDS_BYTES == 80 * 1024, "cfg1");

template <class C>
hipError_t set_attrs() {
    hipError_t e = hipFuncSetAttribute((const void *)dense_scan_store_kernel<C, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)dense_scan_store_kernel<C, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
#define ERH_SET_ABL(A)                                                                                     \
    e = hipFuncSetAttribute((const void *)dense_scan_append_kernel<C, A>,                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);                      \
    if (e != hipSuccess) return e;
    ERH_SET_ABL(0)
    if constexpr (C::MEASURE) { ERH_SET_ABL(1) ERH_SET_ABL(2) ERH_SET_ABL(3) ERH_SET_ABL(4) ERH_SET_ABL(5) }
#undef ERH_SET_ABL
#define ERH_SET_P(A)                                                                                       \
    e = hipFuncSetAttribute((const void *)dense_scan_persist_kernel<C, A, false>,                          \
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);                      \
    if (e != hipSuccess) return e;                                                                         \
    if (C::CAN_RA) {                                                                                       \
        e = hipFuncSetAttribute((const void *)dense_scan_persist_kernel<C, A, C::CAN_RA>,                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);                  \
        if (e != hipSuccess) return e;                                                                     \
    }
#ifdef ERH_MEASURE
    ERH_SET_P(0)
    if constexpr (C::MEASURE) { ERH_SET_P(7) ERH_SET_P(8) ERH_SET_P(10) }
#endif
#undef ERH_SET_P
    return hipSuccess;
}

#ifdef ERH_MEASURE
// Persistent launch: `ctas` = workgroups that are co-resident (CUs x workgroups per CU for this configuration).
template <class C>
hipError_t launch_persist(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q, int Bpad,
                          int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                          uint32_t *cand_cnt, int cap, uint32_t *overflow, int ctas, int pabl, bool ra, hipStream_t st) {
    const int n_qt = Bpad / C::BN;
    int grid_n = ctas / (8 * n_qt) * (8 * n_qt);
    if (grid_n <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)grid_n), block(C::NT);
#define ERH_LAUNCH_P(A)                                                                                    \
    do {                                                                                                   \
        if (ra && C::CAN_RA)                                                                               \
            hipLaunchKernelGGL((dense_scan_persist_kernel<C, A, C::CAN_RA>), grid, block, C::LDS_BYTES, st, X, N, d, c0, \
                               c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow);     \
        else                                                                                               \
            hipLaunchKernelGGL((dense_scan_persist_kernel<C, A, false>), grid, block, C::LDS_BYTES, st, X, N, d, c0, c1, \
                               Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow);         \
    } while (0)
    if constexpr (C::MEASURE) {
        switch (pabl) {
            case 7: ERH_LAUNCH_P(7); break;
            case 8: ERH_LAUNCH_P(8); break;
            case 10: ERH_LAUNCH_P(10); break;
            default: ERH_LAUNCH_P(0); break;
        }
    } else {
        ERH_LAUNCH_P(0);
    }
#undef ERH_LAUNCH_P
    return hipGetLastError();
}

#endif  // ERH_MEASURE

template <class C>
hipError_t launch_store(const _Float16 *Q, int Bpad, const _Float16 *X, int64_t N, int d, int64_t c0, int nc,
                        float *S0, int ld_s0, hipStream_t st) {
    const int n_ctiles = (nc + C::BN - 1) / C::BN;
    const int n_qtiles = Bpad / C::BM;
    dim3 grid(n_ctiles * n_qtiles), block(C::NT);
    hipLaunchKernelGGL((dense_scan_store_kernel<C, false>), grid, block, C::LDS_BYTES, st, Q, Bpad, X, N, d, c0, nc, S0, ld_s0,
                       (const erh::ErhDenseView *)nullptr);
    return hipGetLastError();
}

template <class C>
hipError_t launch_store_grouped(const erh::ErhDenseView *views, int n0_max, const _Float16 *Q, int Bpad, int d, float *S0, int ld_s0,
                                hipStream_t st) {
    static_assert(256 % C::BM == 0, "a query tile of the store kernel lies inside one 256-row tile of the view table");
    const int n_ctiles = (n0_max + C::BN - 1) / C::BN;
    const int n_qtiles = Bpad / C::BM;
    dim3 grid(n_ctiles * n_qtiles), block(C::NT);
    hipLaunchKernelGGL((dense_scan_store_kernel<C, true>), grid, block, C::LDS_BYTES, st, Q, Bpad, (const _Float16 *)nullptr, (int64_t)0, d,
                       (int64_t)0, n0_max, S0, ld_s0, views);
    return hipGetLastError();
}

template <class C>
hipError_t launch_append(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q, int Bpad,
                         int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                         uint32_t *cand_cnt, int cap, uint32_t *overflow, int ablate, unsigned long long *dbg,
                         hipStream_t st) {
    const int n_qt = Bpad / C::BN;
    const int64_t n_ct = (c1 - c0 + C::BM - 1) / C::BM;
    const int64_t n_ct8 = (n_ct + 7) / 8 * 8;
    dim3 grid((unsigned)(n_ct8 * n_qt)), block(C::NT);
#define ERH_LAUNCH_ABL(A)                                                                                  \
    hipLaunchKernelGGL((dense_scan_append_kernel<C, A>), grid, block, C::LDS_BYTES, st, X, N, d, c0, c1, Q, Bpad, B, \
                       tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg)
    if constexpr (C::MEASURE) {
        switch (ablate) {
            case 1: ERH_LAUNCH_ABL(1); break;
            case 2: ERH_LAUNCH_ABL(2); break;
            case 3: ERH_LAUNCH_ABL(3); break;
            case 4: ERH_LAUNCH_ABL(4); break;
            case 5: ERH_LAUNCH_ABL(5); break;
            default: ERH_LAUNCH_ABL(0); break;
        }
    } else {
        ERH_LAUNCH_ABL(0);
    }
#undef ERH_LAUNCH_ABL
    return hipGetLastError();
}

#ifdef ERH_MEASURE
// option code -> ablation mask of the ping-pong kernel (round-1 codes kept; 16..18, 21 new)
constexpr int pp_mask_of(int code) {
    switch (code) {
        case 7: return kPpNoEpi;
        case 8: return kPpTauInf;
        case 11: return kPpNoEpi | kPpNoMfma;
        case 12: return kPpNoEpi | kPpNoDmaA | kPpNoDmaB;
        case 13: return kPpNoEpi | kPpNoDmaA;
        case 14: return kPpNoEpi | kPpNoFrag;
        case 15: return kPpNoEpi | kPpNoDmaB;
        case 16: return kPpNoEpi | kPpNoMfma | kPpNoFrag;                          // LDS-DMA + barriers only
        case 17: return kPpNoEpi | kPpNoMfma | kPpNoDmaA | kPpNoDmaB;             // fragment reads + barriers only
        case 18: return kPpNoEpi | kPpNoMfma | kPpNoDmaA | kPpNoDmaB | kPpNoFrag; // barriers only
        case 20: return kPpNoEpi | kPpClocks;
        case 21: return kPpClocks;
        case 22: return kPpNoEpi | kPpNoDmaA | kPpNoDmaB | kPpClocks;             // matrix + fragment reads, phase clocks
        case 23: return kPpNoEpi | kPpNoDmaA | kPpNoDmaB | kPpNoFrag;             // matrix segments + barriers only
        case 24: return kPpNoEpi | kPpNoDmaA | kPpNoDmaB | kPpNoFrag | kPpClocks;
        default: return 0;
    }
}
#define ERH_PP_MASKS(X) X(1) X(2) X(5) X(25) X(9) X(33) X(17) X(37) X(29) X(61) X(65) X(64) X(89) X(57) X(121)
#endif

// the half-query-tile mode (VAR bit 3) is instantiated for the full kernel and, in measurement builds, for "no epilogue"
constexpr bool pp3_halfq_ok(int mask) { return mask == 0 || mask == 1; }
constexpr int pp3_halfq_mask(int mask) { return pp3_halfq_ok(mask) ? mask : 0; }

hipError_t launch_pp(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q, int Bpad, int B,
                     const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                     uint32_t *cand_cnt, int cap, uint32_t *overflow, int ctas, int pabl, unsigned long long *dbg,
                     int lean, uint32_t *stream_sync, const erh::ErhSeedIo *sio, hipStream_t st) {
    const int n_qt = Bpad / pp::BN;
    const int grid_n = ctas / (8 * n_qt) * (8 * n_qt);
    if (grid_n <= 0) return hipErrorInvalidValue;
    erh::ErhSeedIo sio_v{};
    if (sio) sio_v = *sio;
    const bool seed = sio_v.mode == 1;
    if (sio_v.mode != 0 && !(lean & 8)) return hipErrorInvalidValue;    // the sample pass exists for the strict ping-pong kernel only
    const int64_t n_ct = (c1 - c0 + pp::BM - 1) / pp::BM;
    if (n_ct / (grid_n / n_qt) + 1 >= (1ll << pp::TILE_BITS)) return hipErrorInvalidValue;   // tile index must fit the record
    dim3 grid((unsigned)grid_n), block(pp::NT);
    // lean: bit 0 = lean-issue kernel, bits 1-2 = its VAR, bits 8.. = rot_stages (see dense_scan_pp2_kernel)
    const int var = (lean >> 1) & 3, rot = lean >> 8;
    const bool halfq = (lean & 8) && B <= pp::BN / 2 && Bpad == pp::BN;
    // lean bit 4: chunk-side LDS-DMA with the non-temporal hint -- only where every chunk row has ONE reader (one query tile), on the
    // row-major plain variant of the strict ping-pong kernel
    const bool nt_x = (lean & 16) && (lean & 8) && n_qt == 1 && var == 0 && !seed && pabl == 0;
#define ERH_LAUNCH_PP2(A, V)                                                                               \
    hipLaunchKernelGGL((dense_scan_pp2_kernel<A, V>), grid, block, pp::LDS_BYTES, st, X, N, d, c0, c1, Q, Bpad, B, \
                       tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg, rot)
#define ERH_LAUNCH_PP3V(A, V)                                                                              \
    hipLaunchKernelGGL((dense_scan_pp3_kernel<A, V>), grid, block, pp::LDS_BYTES, st, X, N, d, c0, c1, Q, Bpad, B, tau, \
                       filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg, rot, stream_sync, sio_v, erh::ErhGroupIo{})
#define ERH_LAUNCH_PP3(A)                                                                                  \
    do {                                                                                                   \
        if (seed) {                                        /* the sample pass: row-major operands, full kernel only */ \
            if (halfq) ERH_LAUNCH_PP3V(0, 24); else ERH_LAUNCH_PP3V(0, 16);                                \
        } else if (nt_x) {                                                                                 \
            if (halfq) ERH_LAUNCH_PP3V(0, 72); else ERH_LAUNCH_PP3V(0, 64);                                \
        } else if (halfq && pp3_halfq_ok(A)) {             /* 65 ... 128 queries: the nt = 1 half of the tile is not computed */ \
            if (var & 2) ERH_LAUNCH_PP3V(pp3_halfq_mask(A), 10); else ERH_LAUNCH_PP3V(pp3_halfq_mask(A), 8); \
        } else if (var & 2)                                                                                \
            hipLaunchKernelGGL((dense_scan_pp3_kernel<A, 2>), grid, block, pp::LDS_BYTES, st, X, N, d, c0, c1, Q, Bpad, \
                               B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg, rot, stream_sync, sio_v, erh::ErhGroupIo{}); \
        else if ((var & 1) && kPp3LockStep)                                                                \
            hipLaunchKernelGGL((dense_scan_pp3_kernel<A, kPp3LockStep>), grid, block, pp::LDS_BYTES, st, X, N, d, c0, c1, Q, \
                               Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg, rot, stream_sync, sio_v, erh::ErhGroupIo{}); \
        else                                                                                               \
            hipLaunchKernelGGL((dense_scan_pp3_kernel<A, 0>), grid, block, pp::LDS_BYTES, st, X, N, d, c0, c1, Q, Bpad, \
                               B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg, rot, stream_sync, sio_v, erh::ErhGroupIo{}); \
    } while (0)
#ifdef ERH_MEASURE
#define ERH_LAUNCH_PP(A)                                                                                   \
    do {                                                                                                   \
        if (lean & 8)                                                                                      \
            ERH_LAUNCH_PP3(A);                                                                             \
        else if (lean & 1)                                                                                 \
            ERH_LAUNCH_PP2(A, 0);                                                                          \
        else                                                                                               \
            hipLaunchKernelGGL((dense_scan_pp_kernel<0>), grid, block, pp::LDS_BYTES, st, X, N, d, c0, c1, Q, Bpad, B, \
                               tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg);                \
    } while (0)
#endif
#define ERH_LAUNCH_PP2V(A)                                                                                 \
    do {                                                                                                   \
        switch (var) {                                                                                     \
            case 1: ERH_LAUNCH_PP2(A, 1); break;                                                           \
            case 2: ERH_LAUNCH_PP2(A, 2); break;                                                           \
            case 3: ERH_LAUNCH_PP2(A, 3); break;                                                           \
            default: ERH_LAUNCH_PP2(A, 0); break;                                                          \
        }                                                                                                  \
    } while (0)
#ifdef ERH_MEASURE
    const int mask = pp_mask_of(pabl);
    if ((lean & 1) && !(lean & 8) && var != 0 && (mask == 0 || mask == 1 || mask == 64 || mask == 65)) {
        switch (mask) {
            case 1: ERH_LAUNCH_PP2V(1); break;
            case 64: ERH_LAUNCH_PP2V(64); break;
            case 65: ERH_LAUNCH_PP2V(65); break;
            default: ERH_LAUNCH_PP2V(0); break;
        }
        return hipGetLastError();
    }
    switch (mask) {
#define ERH_PP_CASE(M) case M: ERH_LAUNCH_PP(M); break;
        ERH_PP_MASKS(ERH_PP_CASE)
#undef ERH_PP_CASE
        default: ERH_LAUNCH_PP(0); break;
    }
#else
    (void)pabl;
    ERH_LAUNCH_PP3(0);                      // the product build carries the strict ping-pong kernel only (dense_pp 1 ... 3)
#endif
#undef ERH_LAUNCH_PP2V
#undef ERH_LAUNCH_PP2
#undef ERH_LAUNCH_PP3
#undef ERH_LAUNCH_PP3V
#undef ERH_LAUNCH_PP
    return hipGetLastError();
}

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------
namespace erh {

int dense_scan_q_tile() { return 256; }   // both configurations tile the query block by 256 (or a divisor)

hipError_t dense_scan_init() {
    hipError_t e = set_attrs<Cfg0>();
    if (e != hipSuccess) return e;
    e = set_attrs<Cfg1>();
    if (e != hipSuccess) return e;
    e = set_attrs<Cfg2>();
    if (e != hipSuccess) return e;
#define ERH_SET_PP3(A, V)                                                                                  \
    e = hipFuncSetAttribute((const void *)dense_scan_pp3_kernel<A, V>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            pp::LDS_BYTES);                                                                \
    if (e != hipSuccess) return e;
#define ERH_SET_PP(A) ERH_SET_PP3(A, 0) ERH_SET_PP3(A, 2)
    ERH_SET_PP(0)
    ERH_SET_PP3(0, 8) ERH_SET_PP3(0, 10) ERH_SET_PP3(0, 16) ERH_SET_PP3(0, 24)
    ERH_SET_PP3(0, 32) ERH_SET_PP3(0, 40)              // the grouped launch (several matrices, one per query tile), whole / half query tile
    ERH_SET_PP3(0, 48) ERH_SET_PP3(0, 56)              // ... and its sample pass
    ERH_SET_PP3(0, 64) ERH_SET_PP3(0, 72) ERH_SET_PP3(0, 96) ERH_SET_PP3(0, 104)   // chunk-side loads with the non-temporal hint (one query tile per matrix)
    e = hipFuncSetAttribute((const void *)dense_scan_pp5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_BYTES);
    if (e != hipSuccess) return e;
#ifdef ERH_MEASURE
    e = hipFuncSetAttribute((const void *)dense_scan_pp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            pp::LDS_BYTES);
    if (e != hipSuccess) return e;
#define ERH_SET_PP2V(A, V)                                                                                 \
    e = hipFuncSetAttribute((const void *)dense_scan_pp2_kernel<A, V>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            pp::LDS_BYTES);                                                                \
    if (e != hipSuccess) return e;
#define ERH_SET_PP4(A)                                                                                     \
    e = hipFuncSetAttribute((const void *)dense_scan_pp4_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            pp::LDS_BYTES);                                                                \
    if (e != hipSuccess) return e;
#define ERH_SET_MEASURE(A) ERH_SET_PP(A) ERH_SET_PP3(A, 1) ERH_SET_PP2V(A, 0) ERH_SET_PP4(A)
    ERH_SET_PP3(0, 1) ERH_SET_PP2V(0, 0) ERH_SET_PP4(0) ERH_SET_PP3(1, 8) ERH_SET_PP3(1, 10)
    ERH_PP_MASKS(ERH_SET_MEASURE)
    ERH_SET_PP2V(0, 1) ERH_SET_PP2V(0, 2) ERH_SET_PP2V(0, 3)
    ERH_SET_PP2V(1, 1) ERH_SET_PP2V(1, 2) ERH_SET_PP2V(1, 3)
    ERH_SET_PP2V(64, 1) ERH_SET_PP2V(64, 2) ERH_SET_PP2V(64, 3)
    ERH_SET_PP2V(65, 1) ERH_SET_PP2V(65, 2) ERH_SET_PP2V(65, 3)
#undef ERH_SET_MEASURE
#undef ERH_SET_PP2V
#undef ERH_SET_PP4
#endif
#undef ERH_SET_PP
#undef ERH_SET_PP3
    return hipSuccess;
}

// Ping-pong persistent append scan; hipErrorInvalidValue when the shape does not qualify (fewer than 8 stages of
// 32 halves, query tiles that do not divide the resident grid, tile index too wide for the record).
hipError_t launch_dense_scan_pp(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q,
                                int Bpad, int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id,
                                ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int pabl,
                                unsigned long long *dbg, int lean, uint32_t *stream_sync, const ErhSeedIo *sio,
                                hipStream_t st) {
    if (c1 <= c0) return hipSuccess;
    if (d % (2 * pp::BK) != 0 || d / pp::BK < 8) return hipErrorInvalidValue;   // stage pairs never straddle a tile
    return launch_pp(X, N, d, c0, c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, n_cus, pabl,
                     dbg, lean, stream_sync, sio, st);
}

// The grouped launches (kernels.h: ErhDenseView / ErhGroupIo).  Store: every query tile against the seed prefix of its own matrix;
// the 128 x 256 configuration when the 256 x 256 grid would leave CUs idle.  Scan: `grid` workgroups = the sum of the tiles' streams.
hipError_t launch_dense_scan_store_grouped(const ErhGroupIo &gio, int n_qt, int n0_max, int n_cus, const _Float16 *Q, int Bpad, int d,
                                           float *S0, int ld_s0, hipStream_t st) {
    if (n0_max <= 0 || n_qt <= 0) return hipSuccess;
    if ((int64_t)((n0_max + 255) / 256) * n_qt < n_cus) return launch_store_grouped<Cfg1>(gio.views, n0_max, Q, Bpad, d, S0, ld_s0, st);
    return launch_store_grouped<Cfg0>(gio.views, n0_max, Q, Bpad, d, S0, ld_s0, st);
}

hipError_t launch_dense_scan_pp_grouped(const ErhGroupIo &gio, int grid, int d, const _Float16 *Q, int Bpad, const float *tau,
                                        ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int halfq, hipStream_t st,
                                        const ErhSeedIo *sio, int nt) {
    if (grid <= 0) return hipSuccess;
    if (d % (2 * pp::BK) != 0 || d / pp::BK < 8) return hipErrorInvalidValue;
    erh::ErhSeedIo sio_v{};
    if (sio) sio_v = *sio;
#define ERH_LAUNCH_PP3G(V)                                                                                 \
    hipLaunchKernelGGL((dense_scan_pp3_kernel<0, V>), dim3((unsigned)grid), dim3(pp::NT), pp::LDS_BYTES, st, (const _Float16 *)nullptr, \
                       (int64_t)0, d, (int64_t)0, (int64_t)0, Q, Bpad, Bpad, tau, (const int16_t *)nullptr, (const int16_t *)nullptr, cand, \
                       cand_cnt, cap, overflow, (unsigned long long *)nullptr, 0, (uint32_t *)nullptr, sio_v, gio)
    if (sio_v.mode == 1) { if (halfq) ERH_LAUNCH_PP3G(56); else ERH_LAUNCH_PP3G(48); }
    else if (nt) { if (halfq) ERH_LAUNCH_PP3G(104); else ERH_LAUNCH_PP3G(96); }
    else if (halfq) ERH_LAUNCH_PP3G(40);
    else ERH_LAUNCH_PP3G(32);
#undef ERH_LAUNCH_PP3G
    return hipGetLastError();
}

int dense_scan_pp_streams(int n_cus, int Bpad) {
    const int n_qt = Bpad / pp::BN;
    if (n_qt < 1 || Bpad % pp::BN != 0) return 0;
    const int grid_n = n_cus / (8 * n_qt) * (8 * n_qt);
    return grid_n > 0 ? (grid_n / (8 * n_qt)) * 8 : 0;
}

// The tiled-operand ping-pong scan (dense_scan_pp4_kernel): Xt / Qt are the tiled copies (launch_dense_tile_rows) of the
// chunk matrix and of the query block; c0 must be a multiple of 256.  hipErrorInvalidValue when the shape does not qualify.
hipError_t launch_dense_scan_pp4(const _Float16 *Xt, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Qt,
                                 int Bpad, int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id,
                                 ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int pabl,
                                 unsigned long long *dbg, int rot_stages, hipStream_t st) {
#ifndef ERH_MEASURE
    return hipErrorInvalidValue;            // measurement builds only: the caller falls through to the strict ping-pong scan
#else
    if (c1 <= c0) return hipSuccess;
    if (d % pp::BK != 0 || d / pp::BK < 8 || c0 % pp::BM != 0) return hipErrorInvalidValue;
    const int n_qt = Bpad / pp::BN;
    const int grid_n = n_cus / (8 * n_qt) * (8 * n_qt);
    if (grid_n <= 0) return hipErrorInvalidValue;
    const int64_t n_ct = (c1 - c0 + pp::BM - 1) / pp::BM;
    if (n_ct / (grid_n / n_qt) + 1 >= (1ll << pp::TILE_BITS)) return hipErrorInvalidValue;   // tile index must fit the record
    dim3 grid((unsigned)grid_n), block(pp::NT);
#define ERH_LAUNCH_PP4(A)                                                                                  \
    hipLaunchKernelGGL((dense_scan_pp4_kernel<A>), grid, block, pp::LDS_BYTES, st, Xt, N, d, c0, c1, Qt, Bpad, B, tau, \
                       filter_dir, dir_id, cand, cand_cnt, cap, overflow, dbg, rot_stages)
    switch (pp_mask_of(pabl)) {
#define ERH_PP_CASE(M) case M: ERH_LAUNCH_PP4(M); break;
        ERH_PP_MASKS(ERH_PP_CASE)
#undef ERH_PP_CASE
        default: ERH_LAUNCH_PP4(0); break;
    }
#undef ERH_LAUNCH_PP4
    return hipGetLastError();
#endif
}

hipError_t launch_dense_tile_rows(const _Float16 *X, int64_t N, int d, void *Xt, hipStream_t st) {
    if (d % 64 != 0) return hipErrorInvalidValue;
    const int64_t n_tiles = (N + pp::BM - 1) / pp::BM;
    const int64_t pieces = n_tiles * (d / 64) * 2048;                 // 16-byte pieces to move
    const unsigned blocks = (unsigned)std::min<int64_t>(8192, (pieces + 255) / 256);
    hipLaunchKernelGGL(dense_tile_rows_kernel, dim3(blocks), dim3(256), 0, st, X, d, n_tiles, reinterpret_cast<int4 *>(Xt));
    return hipGetLastError();
}

int dense_scan_pp5_rows() { return pp5::BM; }

hipError_t launch_dense_tile_rows_n(const _Float16 *X, int64_t N, int d, int rows, void *Xt, hipStream_t st) {
    if (d % 32 != 0 || rows % 32 != 0 || rows <= 0) return hipErrorInvalidValue;
    const int64_t n_tiles = (N + rows - 1) / rows;
    const int64_t pieces = n_tiles * (d / 32) * rows * 4;             // 16-byte pieces to move
    const unsigned blocks = (unsigned)std::min<int64_t>(8192, (pieces + 255) / 256);
    hipLaunchKernelGGL(dense_tile_rows_n_kernel, dim3(blocks), dim3(256), 0, st, X, d, rows, n_tiles, reinterpret_cast<int4 *>(Xt));
    return hipGetLastError();
}

// The 384 x 256 ping-pong scan (dense_scan_pp5_kernel): Xt = launch_dense_tile_rows_n(X, N, d, 384), Qt = launch_dense_tile_rows of the
// query block; c0 must be a multiple of 384, Bpad >= 512.  hipErrorInvalidValue when the shape does not qualify.
hipError_t launch_dense_scan_pp5(const _Float16 *Xt, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Qt,
                                 int Bpad, int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id,
                                 ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int rot_stages,
                                 hipStream_t st) {
    if (c1 <= c0) return hipSuccess;
    if (d % 64 != 0 || d / pp::BK < 8 || c0 % pp5::BM != 0 || Bpad % pp::BN != 0 || Bpad < 2 * pp::BN) return hipErrorInvalidValue;
    const int n_qt = Bpad / pp::BN;
    const int grid_n = n_cus / (8 * n_qt) * (8 * n_qt);
    if (grid_n <= 0) return hipErrorInvalidValue;
    const int64_t n_ct = (c1 - c0 + pp5::BM - 1) / pp5::BM;
    if (n_ct / (grid_n / n_qt) + 1 >= (1ll << pp5::TILE_BITS)) return hipErrorInvalidValue;   // tile index must fit the record
    hipLaunchKernelGGL(dense_scan_pp5_kernel, dim3((unsigned)grid_n), dim3(pp::NT), pp::LDS_BYTES, st, Xt, N, d, c0, c1, Qt,
                       Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, rot_stages);
    return hipGetLastError();
}

hipError_t launch_dense_scan_store(int cfg, const _Float16 *Q, int Bpad, const _Float16 *X, int64_t N, int d,
                                   int64_t c0, int nc, float *S0, int ld_s0, hipStream_t st) {
    if (nc <= 0) return hipSuccess;
    if (cfg == 1) return launch_store<Cfg1>(Q, Bpad, X, N, d, c0, nc, S0, ld_s0, st);
    if (cfg == 2) return launch_store<Cfg2>(Q, Bpad, X, N, d, c0, nc, S0, ld_s0, st);
    return launch_store<Cfg0>(Q, Bpad, X, N, d, c0, nc, S0, ld_s0, st);
}

hipError_t launch_dense_scan_append(int cfg, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1,
                                    const _Float16 *Q, int Bpad, int B, const float *tau,
                                    const int16_t *filter_dir, const int16_t *dir_id,
                                    ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int ablate,
                                    unsigned long long *dbg, hipStream_t st) {
    if (c1 <= c0) return hipSuccess;
    if (cfg == 1)
        return launch_append<Cfg1>(X, N, d, c0, c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow,
                                   ablate, dbg, st);
    if (cfg == 2)
        return launch_append<Cfg2>(X, N, d, c0, c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow,
                                   ablate, dbg, st);
    return launch_append<Cfg0>(X, N, d, c0, c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow,
                               ablate, dbg, st);
}

// Persistent variant of the append scan; returns hipErrorInvalidValue when the shape does not qualify
// (query tiles do not divide the resident grid, or too few K-steps) so that the caller uses the plain launch.
hipError_t launch_dense_scan_persist(int cfg, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1,
                                     const _Float16 *Q, int Bpad, int B, const float *tau,
                                     const int16_t *filter_dir, const int16_t *dir_id,
                                     ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus,
                                     int pabl, int readahead, hipStream_t st) {
#ifndef ERH_MEASURE
    return hipErrorInvalidValue;                 // the lock-step persistent kernel exists in measurement builds only
#else
    if (c1 <= c0) return hipSuccess;
    if (cfg == 1) return hipErrorInvalidValue;   // the two-workgroup configuration keeps the per-tile launch
    if (cfg == 2) {
        if (d / Cfg2::BK <= Cfg2::DA) return hipErrorInvalidValue;
        return launch_persist<Cfg2>(X, N, d, c0, c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow,
                                    n_cus, pabl, readahead != 0, st);
    }
    if (d / Cfg0::BK <= Cfg0::DA) return hipErrorInvalidValue;
    return launch_persist<Cfg0>(X, N, d, c0, c1, Q, Bpad, B, tau, filter_dir, dir_id, cand, cand_cnt, cap, overflow, n_cus,
                                pabl, false, st);
#endif
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
