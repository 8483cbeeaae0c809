// Launcher prototypes shared between the kernel translation units and the host side (api.hip, pipeline_dense.hip, pipeline_bm25.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace erh {

// ---- dense_scan.hip --------------------------------------------------------------------------
int dense_scan_q_tile();   // queries are padded to a multiple of this
// Sample pass of the strict ping-pong scan (round 4: the scan kernel draws the sample its pruning threshold is selected from).
// The first `seed_tiles` tiles of every chunk stream are scored WITHOUT a threshold and nothing but the two best scores of
// every "cell" -- the 64 chunk rows one lane holds of one query column -- leaves the registers: `seed_top`, the sample of
// seed_cells_select_kernel.  The main launch then scans ALL rows, the sampled ones included, against that threshold.
struct ErhSeedIo {
    float *seed_top;       // [Bpad][n_cells][2]
    int seed_tiles;        // tiles per chunk stream in the sample pass
    int n_cells;           // seed_tiles * streams * 4
    int mode;              // 0 none, 1 sample pass (no thresholds, no candidates)
};
// Grouped dense call (round 6; pipeline_dense.hip: dense_topk_grouped): the batch's queries, grouped by their `dir` filter, are laid out group
// after group, each group padded to whole 256-row QUERY TILES, and every tile scans ITS OWN matrix -- the dir's block copy -- in the
// same launch as all the others.  One table entry per query tile; every stage of the pipeline (query preparation, seed-prefix store
// kernel, seed select, persistent scan, final kernel) reads it instead of taking one matrix per launch.
struct ErhDenseView {
    const _Float16 *X;     // first row of the tile's matrix (a block of Xb)
    int64_t N;             // its rows
    int64_t mul, inv;      // row placement inside it: document r of the block is stored at (r * mul) mod N, inv the inverse
    int32_t n0;            // rows [0, n0) are scored densely (store kernel -> S0 -> threshold), rows [n0, N) scanned against it
    int32_t rank;          // rank of the prefix score that seeds the threshold (erh_dense_seed_rank; k: guaranteed bound)
    int32_t id_lo;         // block row r is the caller's document id_map[id_lo + r]
    int32_t wg0, nwg;      // persistent scan: workgroups [wg0, wg0 + nwg) walk this tile's chunk tiles (nwg chunk streams)
    int32_t nq;            // queries of the tile = its first nq rows; the rest is padding (zero rows, threshold +inf)
    int32_t seed_rows;     // sample-pass mode (n0 == 0): rows [0, seed_rows) = seed_tiles x nwg x 256 are what the sample pass scores ...
    int32_t n_cells;       // ... into this many 64-row cells (two best scores each) of the tile's queries; 0: the store-kernel mode
};
static_assert(sizeof(ErhDenseView) == 64, "ErhDenseView layout");
struct ErhGroupIo {
    const ErhDenseView *views;   // [query tiles]; null = an ordinary launch (one matrix, kernel arguments)
    const int32_t *wg_view;      // [grid of the persistent scan]: workgroup -> query tile
    const int32_t *q_src;        // [Bpad]: padded row -> the caller's query row, -1 for padding rows
    const int32_t *id_map;       // block row -> document id (the handle's blk_ids)
    // one view for the whole launch (views == null; a routed group run as a pipeline of its own): the final kernel writes query q to the
    // caller's row row_map[q] (null: q) and maps ids through id_map[single_lo + id] (null: unchanged) -- no scatter launch behind it
    const int32_t *row_map;
    int32_t single_lo;
};
hipError_t launch_dense_scan_store_grouped(const ErhGroupIo &gio, int n_qt, int n0_max, int n_cus, const _Float16 *Q, int Bpad, int d,
                                           float *S0, int ld_s0, hipStream_t st);
hipError_t launch_dense_scan_pp_grouped(const ErhGroupIo &gio, int grid, int d, const _Float16 *Q, int Bpad, const float *tau,
                                        ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow,
                                        int halfq /* every tile holds at most 128 queries: the other half of the tile is not computed */,
                                        hipStream_t st,
                                        const ErhSeedIo *sio = nullptr /* the grouped SAMPLE PASS: rows [0, seed_rows) of every view, the cells' two best
                                                                          scores to sio->seed_top[q][sio->n_cells][2] (n_cells: the largest of the views) */,
                                        int nt = 0 /* main scan: chunk-side loads with the non-temporal hint (every view's rows have one reader) */);
// chunk streams (co-resident workgroups per query tile) of the ping-pong scan on n_cus CUs, 0 if the shape does not qualify
int dense_scan_pp_streams(int n_cus, int Bpad);
hipError_t dense_scan_init();
hipError_t launch_dense_scan_store(int cfg, const _Float16 *Q, int Bpad, const _Float16 *X, int64_t N, int d,
                                   int64_t c0, int nc, float *S0, int ld_s0, hipStream_t st);
hipError_t launch_dense_scan_append(int cfg, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1,
                                    const _Float16 *Q, int Bpad, int B, const float *tau,
                                    const int16_t *filter_dir, const int16_t *dir_id,
                                    ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int ablate,
                                    unsigned long long *dbg, hipStream_t st);
hipError_t launch_dense_scan_pp(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q,
                                int Bpad, int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id,
                                ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int pabl,
                                unsigned long long *dbg, int lean /* the lean-issue kernel: X must be padded by kDensePadRows zero rows */,
                                uint32_t *stream_sync /* dense_pp 3: 256 zeroed words (one per stream) or null */,
                                const ErhSeedIo *sio /* null, or the sample pass (dense_pp 3 only) */,
                                hipStream_t st);
// 384 x 256 ping-pong scan over tiled operands (dense_scan_pp5_kernel; batches padded to >= 512 queries): Xt from
// launch_dense_tile_rows_n(X, N, d, dense_scan_pp5_rows(), ...) -- ceil(N / 384) * 384 * d halves --, Qt from launch_dense_tile_rows of
// the query block; c0 a multiple of 384.  hipErrorInvalidValue when the shape does not qualify (the caller falls through).
int dense_scan_pp5_rows();
hipError_t launch_dense_tile_rows_n(const _Float16 *X, int64_t N, int d, int rows, void *Xt, hipStream_t st);
hipError_t launch_dense_scan_pp5(const _Float16 *Xt, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Qt,
                                 int Bpad, int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id,
                                 ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int rot_stages,
                                 hipStream_t st);
// tiled copy of the chunk matrix for the ping-pong scan: ceil(N / 256) * 256 * d halves (see dense_tile_rows_kernel)
hipError_t launch_dense_tile_rows(const _Float16 *X, int64_t N, int d, void *Xt, hipStream_t st);
hipError_t launch_dense_scan_pp4(const _Float16 *Xt, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Qt,
                                 int Bpad, int B, const float *tau, const int16_t *filter_dir, const int16_t *dir_id,
                                 ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int pabl,
                                 unsigned long long *dbg, int rot_stages, hipStream_t st);
constexpr int kDensePadRows = 384;   // zero rows erh_set_dense keeps behind the matrix (tiles past N read them: up to 383 for the 384-row tile)
hipError_t launch_dense_scan_persist(int cfg, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1,
                                     const _Float16 *Q, int Bpad, int B, const float *tau,
                                     const int16_t *filter_dir, const int16_t *dir_id,
                                     ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus,
                                     int pabl, int readahead, hipStream_t st);
hipError_t launch_dense_naive(const _Float16 *Q, int B, const _Float16 *X, int64_t row0, int rows, int d,
                              float *out, hipStream_t st);

// ---- dense_gemv.hip: append scan for batches of at most dense_gemv_max_queries() queries ----------------------------
void dense_finalize_set_wgs(int v);
int dense_gemv_max_queries();
hipError_t dense_gemv_init();
hipError_t launch_dense_gemv_append(const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q, int B,
                                    const float *tau, const int16_t *filter_dir, const int16_t *dir_id, ErhCand *cand,
                                    uint32_t *cand_cnt, int cap, uint32_t *overflow, int n_cus, int kb, int wgs,
                                    int pipe, hipStream_t st, int nt = 0 /* chunk loads with the non-temporal hint (option dense_gemv_nt) */);
hipError_t launch_dense_gemv_store(const _Float16 *X, int64_t N, int d, int64_t c0, int nc, const _Float16 *Q, int B,
                                   float *S0, int ld_s0, int n_cus, int kb, int wgs, int pipe, hipStream_t st);

// ---- select.hip ------------------------------------------------------------------------------
constexpr int kDenseN0Max = 32768;    // longest seed prefix (one fp32 score row per query in S0; until round 6 it also had to fit LDS for the full-sort fall-back)
constexpr int kDenseCapMax = 16384;   // candidates per query that the LDS sort can hold (64-bit keys)
constexpr int kDenseRescoreMax = 1024;
hipError_t select_init();
// fp32/fp16 -> fp16 query block [Bpad x d] (+ fp32 norm of the fp16 row); rows >= B are zeroed.
hipError_t launch_prep_queries(const void *q, int q_dtype, int normalize, int B, int Bpad, int d,
                               _Float16 *Q16, float *qnorm, uint32_t *zero_bad /* null, or B words cleared by the kernel */,
                               uint32_t *zero_flags /* null, or 16 words cleared by the kernel */, hipStream_t st,
                               const int32_t *q_src = nullptr /* grouped call: row r of the block is the caller's row q_src[r] (-1: a zero
                                                                  padding row); B == Bpad then */,
                               uint32_t *zero_extra = nullptr, int n_extra = 0 /* further words the call wants cleared (<= 16: the final
                                                                                  kernel's sync words -- a faulted launch must not leave a ticket behind) */);
// rows fp32 -> fp16 (optionally L2-normalised) for erh_set_dense
hipError_t launch_convert_rows(const float *x, int64_t n, int d, int normalize, _Float16 *out_base, int64_t r0,
                               int64_t mul, int64_t N, hipStream_t st);
hipError_t launch_permute_rows(const _Float16 *x, int64_t n, int d, _Float16 *out_base, int64_t r0, int64_t mul,
                               int64_t N, hipStream_t st);
hipError_t launch_gather_rows(const _Float16 *X, const int32_t *rows /* null: original rows row0 ..; else rows[row0 ..] */, int64_t row0,
                              int64_t n, int d, int64_t mul, int64_t N, _Float16 *out, hipStream_t st);
hipError_t launch_permute_dir(const int16_t *dir_id, int64_t N, int64_t inv, int16_t *out, hipStream_t st);
// max L2 norm over fp16 rows -> *out (float, device)
hipError_t launch_row_norm_max(const _Float16 *x, int64_t n, int d, float *out, hipStream_t st);
// Seed stage: k-th best of S0[q][0..n0) (filter applied) -> tau[q] = kth - margin(q); candidates >= tau
// are written to cand[q] and cand_cnt[q] is (re)initialised.
// rank <= k: position in the prefix that seeds the threshold (k: guaranteed bound, < k: speculative, verified by
// launch_dense_finalize when it is given tau_verify)
bool seed_cells_select_fits(int n_vals);   // the select sorts pow2(n_vals) floats in 48 KiB of LDS
hipError_t launch_seed_cells_select(const float *seed_top, int n_vals, int B, int rank, const float *qnorm, float xnorm_max, int d,
                                    float *tau, uint32_t *cand_cnt, hipStream_t st,
                                    const ErhDenseView *views = nullptr /* grouped call: cells and rank per query tile, n_vals = the row stride */);
hipError_t launch_seed_select(const float *S0, int ld_s0, int n0, int64_t c0, int B, int k, int rank,
                              const float *qnorm, float xnorm_max, int d,
                              const int16_t *filter_dir, const int16_t *dir_id,
                              float *tau, ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *bad, uint32_t *need_full,
                              hipStream_t st, const ErhDenseView *views = nullptr /* grouped call: n0 / rank per query tile, B = Bpad */);
// Refine: k-th best over the current candidates -> tighter tau; candidates below it are dropped.
hipError_t launch_cand_refine(int B, int k, const float *qnorm, float xnorm_max, int d,
                              float *tau, ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *bad, hipStream_t st);
// Final: sort candidates, (EXACT) re-score the margin set in pinned fp64, rank, write top-k.
hipError_t launch_dense_finalize(int B, int k, int mode, const float *qnorm, float xnorm_max, int d,
                                 const _Float16 *X, const _Float16 *Q16,
                                 const ErhCand *cand, const uint32_t *cand_cnt, int cap,
                                 int32_t *out_ids, double *out_scores, int32_t *out_len,
                                 float *diag_maxerr, uint32_t *diag_uncert, uint32_t *bad, int64_t N, int64_t pos_mul,
                                 int64_t pos_inv, const float *tau_verify, int n_cus,
                                 double *ws_s64 /* null, or dense_finalize_split_max() x kDenseRescoreMax doubles */,
                                 uint32_t *ws_sync /* null, or 2 x dense_finalize_split_max() words, zero between calls */, hipStream_t st,
                                 const ErhGroupIo *gio = nullptr /* grouped call: matrix and placement per query tile, results written to the
                                                                    caller's row q_src[q] with block rows mapped to document ids */);
int dense_finalize_split_max();
// dense calls routed by dir block: gather the rows idx[0..n) of a query block (row_bytes % 16 == 0), scatter a group's results back
hipError_t launch_gather_query_rows(const void *q, const int32_t *idx, int n, int row_bytes, void *out, hipStream_t st);
hipError_t launch_scatter_topk_rows(const int32_t *ids, const double *sc, const int32_t *len, const int32_t *idx, int n, int k,
                                    int32_t id_offset, const int32_t *id_map /* null: id + id_offset; else id_map[id_offset + id] */,
                                    int32_t *out_ids, double *out_sc, int32_t *out_len, hipStream_t st);
// Exhaustive path for the queries flagged in bad[] (select.hip): exact fp64 scores of every chunk + streaming top-k.
int dense_exhaustive_max();
size_t dense_exhaustive_bytes(int64_t N);
hipError_t launch_dense_exhaustive(const uint32_t *bad, int B, int skip, int k, const _Float16 *X, int64_t N, int d,
                                   const _Float16 *Q16, const int16_t *filter_dir, const int16_t *dir_id,
                                   int64_t pos_inv, void *ws, uint32_t *flags, int n_cus,
                                   int32_t *out_ids, double *out_scores, int32_t *out_len,
                                   unsigned long long *stats /* null, or the handle's device counters: [0] += flagged queries */,
                                   int collect_only /* 1: count the flagged queries (flag words, counter) and stop: what a call enqueues;
                                                       0: one answering round (collect + the two exact kernels) from skip on */,
                                   hipStream_t st);

// ---- bm25.hip --------------------------------------------------------------------------------
constexpr int kBm25TileF32 = 32768;   // documents per LDS accumulator tile (fp32 sums)
constexpr int kBm25TileF64 = 16384;   // (fp64 sums)
hipError_t bm25_init();
hipError_t launch_bm25_tile_off(const int64_t *indptr, const int32_t *doc_ids, int64_t V, int tile_docs,
                                int n_tiles, int32_t *tile_off, hipStream_t st);
hipError_t launch_bm25_payload(int variant, int64_t V, int64_t nnz, const int64_t *indptr, const int32_t *doc_ids,
                               const int32_t *tf, const int32_t *doc_len, const void *idf, double avgdl,
                               double k1, double b, void *payload, hipStream_t st);
// segs partial lists per query; partial_* are [B][segs][k]
hipError_t launch_bm25_scan(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                            const int32_t *tile_off, int n_tiles, int64_t N,
                            const int32_t *q_indptr, const int32_t *q_tok,
                            const int32_t *q_order /* workgroup -> query (heaviest first) or null */, int B, int k, int segs,
                            const int16_t *filter_dir, const int16_t *dir_id,
                            double *part_scores, int32_t *part_ids, int32_t *part_len,
                            const uint32_t *only /* null, or [B][segs]: scan only the flagged (query, segment) pairs */,
                            int cut_tiles, int cut_shift /* segment cuts at (cut_tiles * seg / segs) << cut_shift tiles; 0: n_tiles */, int ablate,
                            unsigned long long *dbg, hipStream_t st);
// fixed-point scan + exact re-score (bm25.hip: bm25_ascan_kernel): integer LDS atomics in any order, exact scores of the
// final list by binary search.  tile_off: n_tab + 1 entries per term at 32768 >> tshift documents; post: interleaved
// {document, fixed-point payload} postings built by launch_bm25_post (nnz + 1 entries, the last one a sentinel), qmax the
// largest fixed-point payload; redo: zeroed [B][segs] words, set where the list overflowed with near ties or the sums
// could overflow (those workgroups are then scanned by launch_bm25_scan(..., only = redo)).
int bm25_ascan_tile_docs(int small);   // 32768, or 16384 for the two-workgroups-per-CU shape
int bm25_ascan_small_max_k();
float bm25_post_scale(float pmax);
hipError_t launch_bm25_post(const int32_t *doc_ids, const float *pay32, int64_t nnz, float scale, void *post, hipStream_t st);
// packed shape for the queries of at most long_tokens tokens, 32-bit 16384-document shape (*_s) for the longer ones, one launch (bm25.hip)
hipError_t launch_bm25_ascan_mixed(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                                   const void *post16, int g16, uint32_t nnz, double qmax, const int32_t *tile_off, int n_tab, int tshift, int cut_mul,
                                   const void *post_s, const int32_t *tile_off_s, int n_tab_s, int tshift_s, int cut_mul_s, int long_tokens,
                                   int64_t N, const int32_t *q_indptr, const int32_t *q_tok, const int32_t *q_order, int B, int k, int segs,
                                   const int16_t *filter_dir, const int16_t *dir_id, double *part_scores, int32_t *part_ids, int32_t *part_len,
                                   uint32_t *redo, unsigned long long *stats, const int32_t *dir_rng, int dir_rng_n, unsigned long long *dbg,
                                   hipStream_t st,
                                   const int32_t *q_items = nullptr /* segs == 1 only: grid = n_items workgroups, item = query | segment << 24: a long query as
                                                                       segs_l items whose partial lists go to l_* [B][segs_l] (merged by the caller:
                                                                       launch_bm25_merge with the list of long queries), every other query as one item */,
                                   int n_items = 0, int segs_l = 1, double *l_scores = nullptr, int32_t *l_ids = nullptr, int32_t *l_len = nullptr,
                                   uint32_t *l_redo = nullptr);
hipError_t launch_bm25_ascan(int variant, int small /* 0: 1024 threads; 1: 512 threads, 16384-document tiles; 2: packed 16-bit sums */,
                             const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                             const void *post, const void *post16 /* shape 2: the 4-byte postings (launch_bm25_post16), or null */, int g16,
                             uint32_t nnz, double qmax, const int32_t *tile_off, int n_tab, int tshift,
                             int64_t N, const int32_t *q_indptr, const int32_t *q_tok, const int32_t *q_order, int B, int k,
                             int segs, int cut_mul /* segment cuts on multiples of this many tiles */,
                             const int16_t *filter_dir, const int16_t *dir_id,
                             double *part_scores, int32_t *part_ids, int32_t *part_len, uint32_t *redo,
                             unsigned long long *stats /* null, or the handle's device counters: [1] += (query, segment) pairs handed to the exact scan */,
                             const int32_t *dir_rng /* null, or int32[2 * dir_rng_n]: {first document, last + 1} of every dir class */, int dir_rng_n,
                             int ablate /* measurement builds only */, unsigned long long *dbg, hipStream_t st,
                             int32_t *fin_ids = nullptr /* split finish: [B * segs][bm25_ascan_fin_cap()] work space: the scan hands its final
                                                           lists over and ONE batch-wide kernel behind it re-scores and ranks them */,
                             int32_t *fin_cnt = nullptr /* ... [B * segs] */);
int bm25_ascan_fin_cap();
// 4-byte postings of the packed scan {document & 32767, (q >> g) + 1 in 16 bits}: nnz + 8 words (zeros behind the postings)
int bm25_post16_shift(double qmax);
hipError_t launch_bm25_post16(const void *post, int64_t nnz, int g, void *post16, hipStream_t st);
hipError_t launch_narrow_f64(const double *in, int64_t n, float *out, hipStream_t st);
hipError_t launch_bm25_payload_max(const float *pay32, int64_t nnz, uint32_t *bits, hipStream_t st);
// wave-owned scan (bm25.hip: bm25_wscan_kernel): fine_off = skip table at bm25_wscan_sub_docs() granularity
int bm25_wscan_max_tokens();
int bm25_wscan_sub_docs(int variant);
hipError_t launch_bm25_wscan(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                             const int32_t *fine_off, int n_fine, int n_tiles, int64_t N,
                             const int32_t *q_indptr, const int32_t *q_tok, const int32_t *q_order, int B, int k, int segs,
                             const int16_t *filter_dir, const int16_t *dir_id,
                             double *part_scores, int32_t *part_ids, int32_t *part_len,
                             int crossing /* every payload > 0: threshold crossings replace the sweep */,
                             unsigned long long *dbg, hipStream_t st);
// *flag |= 1 if any payload is <= 0 (flag must be zeroed by the caller)
hipError_t launch_bm25_payload_sign(int variant, const void *payload, int64_t nnz, uint32_t *flag, hipStream_t st);
hipError_t launch_bm25_merge(int B, int k, int segs, const double *part_scores, const int32_t *part_ids,
                             const int32_t *part_len, int32_t *out_ids, double *out_scores, int32_t *out_len,
                             hipStream_t st, const int32_t *q_list = nullptr /* workgroup -> query: B = its length */);
// scores[doc] += payload for one term (launched once per query token, in order) -> get_scores parity
hipError_t launch_bm25_add_term(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                                int32_t term, void *scores, hipStream_t st);
hipError_t launch_widen_f32(const float *in, int64_t n, double *out, hipStream_t st);

// ---- index_build.hip: token stream -> CSR postings on the device ----------------------------------------------------
hipError_t launch_csr_keys(const int32_t *tok, const int64_t *doc_off, int64_t T, int64_t N, int64_t V, uint64_t *keys,
                           unsigned long long *first_pos, uint32_t *bad_token, hipStream_t st);
hipError_t csr_sort_rle(const uint64_t *keys_in, uint64_t *keys_sorted, int64_t T, int key_bits, uint64_t *uniq,
                        int32_t *counts, int32_t *num_runs, void *temp, size_t *temp_bytes, hipStream_t st);
hipError_t launch_csr_split(const uint64_t *uniq, const int32_t *counts, int64_t nnz, int64_t V, int32_t *doc_ids,
                            int32_t *tf, unsigned long long *df, hipStream_t st);

// ---- fuse.hip --------------------------------------------------------------------------------
constexpr int kFuseMaxItems = 2048;   // depth_a + depth_b
hipError_t fuse_init();
hipError_t launch_rrf(const int32_t *ids_a, const int32_t *len_a, int depth_a,
                      const int32_t *ids_b, const int32_t *len_b, int depth_b,
                      const int32_t *content_id, int B, int K, int topk,
                      int32_t *out_ids, double *out_scores, int32_t *out_len, hipStream_t st);
hipError_t launch_fusion(const int32_t *ids_a, const double *sc_a, const int32_t *len_a, int depth_a,
                         const int32_t *ids_b, const double *sc_b, const int32_t *len_b, int depth_b,
                         const int32_t *content_id, int B, int topk,
                         int32_t *out_ids, double *out_scores, int32_t *out_len, hipStream_t st);

// packed top-k rows for the multi-GPU all-gather (fuse.hip)
int topk_row_bytes(int k);
hipError_t launch_pack_topk(const int32_t *ids, const double *sc, const int32_t *len, int b_local, int k, int m,
                            void *out, hipStream_t st);
hipError_t launch_unpack_topk(const void *gathered, int n_queries, int world, int k, int m, int32_t *ids, double *sc,
                              int32_t *len, hipStream_t st);

}  // namespace erh
