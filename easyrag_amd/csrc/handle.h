// Host-side state of libeasyrag_hip.so shared by its translation units (api.hip: the C ABI; pipeline_dense.hip / pipeline_bm25.hip:
// stage orchestration on the caller's HIP stream; comm.hip: the RCCL exchange): the handle, device buffers, event-based kernel timing.
// No kernel lives here and no CPU compute path.
#pragma once
#include "../../include/easyrag_hip.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <future>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <map>
#include <vector>
#include "kernels.h"


struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        // grow geometrically so that alternating batch sizes do not reallocate every call
        size_t want = std::max(bytes, cap + cap / 2);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct EvPair { hipEvent_t a, b; int cls; };

struct Bm25State {
    int variant = -1;
    int64_t V = 0, Nb = 0, nnz = 0;
    DevBuf indptr, doc_ids, payload, tile_off, fine_off;
    DevBuf tile_off16;                    // bm25s: skip table at 16384 documents for the two-workgroups-per-CU scan (Okapi: tile_off is that)
    int n_tiles16 = 0;
    DevBuf post;                          // fixed-point scan: interleaved {document, fixed-point payload} postings + one sentinel
    DevBuf post16;                        // ... and the 4-byte postings of its packed shape {document & 32767, (q >> g16) + 1}
    int g16 = 0;
    double qmax = 0;                      // largest fixed-point payload
    DevBuf tf;                            // kept by erh_build_bm25_index (what erh_get_bm25_csr returns)
    std::vector<double> idf_host;         // idem (float32 values widened exactly for the bm25s variant)
    double avgdl = 0, average_idf = 0;
    bool built_on_device = false;
    bool payload_positive = false;        // every payload > 0: the wave-owned scan may use threshold crossings instead of the sweep
    bool ascan_ok = false;                // ... and also as fp32: the fixed-point scan applies (bm25.hip: bm25_ascan_kernel)
    std::vector<int64_t> host_indptr;     // host copy: query validation + algorithmic-byte accounting
    int n_tiles = 0, tile_docs = 0;
    int n_fine = 0;                       // sub-ranges of the fine skip table (0 = not built: block scan only)
    void release() { indptr.release(); doc_ids.release(); payload.release(); tile_off.release(); fine_off.release(); tf.release(); post.release(); post16.release(); tile_off16.release(); }
};



// ncclCommInitRank runs on a helper thread (erh_comm_init); the state outlives a timed-out call
struct CommInitState { void *comm = nullptr; int rc = -1; std::atomic<int> finished{0}; };

struct erh_handle {
    int device = 0;
    std::string err;
    // dense state
    DevBuf X;
    int32_t *qorder = nullptr;               // BM25: workgroup -> query, heaviest posting volume first (bm25_lpt); a slice of qpack
    std::vector<int32_t> qorder_host;
    DevBuf qpack;                            // the call's query CSR + launch order, one upload: q_indptr | q_tok | order
    std::vector<char> qpack_host;
    int32_t *qptr = nullptr, *qtok = nullptr;
    bool qorder_valid = false;
    // a batch with queries longer than bm25_long_tokens (upload_bm25_queries): the launch items of the mixed scan -- query | segment << 24, a long
    // query as bm25_long_segs items, heaviest first -- and the list of the long queries; slices of qpack, valid for the call's query CSR only
    int32_t *qitems = nullptr, *qlong = nullptr;
    int n_qitems = 0, n_qlong = 0, qitems_segs = 0;
    DevBuf scan_sync;                        // one counter per chunk-tile stream of the ping-pong scan (dense_sync)
    DevBuf Xt;                               // tiled copy of X for the ping-pong scan (option dense_tiled), valid iff xt_valid
    DevBuf Qt;                               // tiled copy of the query block of the current call (dense_pp = 4)
    DevBuf seed_top;                         // sample pass of the ping-pong scan: the cells' two best scores (kernels.h: ErhSeedIo)
    bool xt_valid = false;
    DevBuf Xt384;                            // 384-row tiled copy of X for the 384 x 256 scan (dense_scan_pp5_kernel), built on first use
    bool xt384_valid = false;
    bool xt384_nomem = false;                // the copy did not fit beside X: the 256 x 256 scan serves every batch until the next erh_set_dense
    int64_t opt_tile384_max_mb = -1;         // test hook: refuse a 384-row copy above this many MiB as if the allocation had failed (-1: no limit)
    bool qt5_valid = false;                  // Qt holds the tiled copy of the CURRENT call's Q16 for that scan
    bool qt_valid = false;                   // Qt holds the tiled copy of the CURRENT call's Q16
    int64_t N = 0;
    int d = 0;
    float xnorm_max = 0.f;
    // Row placement: original row o is stored at position (o * pos_mul) mod N; pos_inv is the inverse multiplier
    // (position -> original).  With the golden-ratio inverse every prefix of the stored order is an evenly spread
    // sample of the caller's order, so the pruning thresholds seeded from a prefix are representative even when
    // the corpus is sorted by topic.  (1, 1) = stored as given (option dense_shuffle = 0).
    int64_t pos_mul = 1, pos_inv = 1;
    int opt_dense_shuffle = 1;
    // What a dense call scans: the whole matrix (global: X with its placement; the tiled copies belong to it) or, for queries
    // filtered on a dir whose documents are one block, that block's own copy with its own placement (round 5, DenseBlocks below).
    struct DenseView { const _Float16 *X = nullptr; int64_t N = 0, mul = 1, inv = 1; bool global = true; } view;
    void view_global() { view.X = X.as<_Float16>(); view.N = N; view.mul = pos_mul; view.inv = pos_inv; view.global = true; }
    DevBuf dir_pos;                         // dir id by stored position (built on demand)
    bool dir_pos_valid = false;
    // bm25 state: up to ERH_BM25_SLOTS independent indices (e.g. the content route and the know_path route of the
    // reference pipeline, pipeline.py:187-210); erh_bm25_select picks the one the set / query calls act on
    Bm25State bm[ERH_BM25_SLOTS];
    int cur = 0;
    int opt_bm25_lpt = 1;                 // launch the queries with the most postings first (shorter tail of the scan)
    int opt_bm25_segs = 0;                // document-range segments per query (0: enough to give the chip >= 512 workgroups)
    int opt_bm25_crossing = 1;            // wave-owned scan: threshold crossings instead of the accumulator sweep (indices with positive
                                          // payloads); 1 = fp32 sums only (the fp64 kernel runs out of registers with it: +12 % time), 2 = both
    int opt_hybrid_overlap = -1;          // erh_hybrid_topk: 1 = the sparse route on a side stream from the start, 2 = forked behind the dense scan,
                                          // 0 = one stream, -1 = by batch size (1 up to 256 queries: neither scan fills the chip; 0 above)
    hipStream_t side = nullptr;           // ... created at first use
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fork_after_scan = false;         // dense_topk_dev records ev_fork behind its last scan launch (hybrid_overlap 2)
    // a routed group run as a pipeline of its own writes its lists straight to the caller's rows, block rows mapped to document ids
    // (dense_finalize_kernel: ErhGroupIo::row_map / id_map / single_lo) -- set around that dense_topk_dev call only
    struct ViewOut { bool on = false; const int32_t *id_map = nullptr; int32_t id_lo = 0; const int32_t *row_map = nullptr; } view_out;
    int cand_rows = 0;                    // query rows of cand_cnt the last dense pipeline used (erh_get_stat: dense_candidates_last_call)
    bool rerun = false;                   // dense_topk_dev is re-running a group of a routed call at its check: its flagged queries were counted already
    int opt_bm25_small = 2;               // fixed-point scan, when k allows: 2 the 512-thread shape with packed 16-bit sums over 32768-document
                                          // tiles (two workgroups per CU), 1 the 512-thread shape over 16384-document tiles, 0 always 1024 threads
    int opt_bm25_post16 = 1;              // packed shape: read the 4-byte postings (built when an index is set; 0: the 8-byte ones)
    int opt_bm25_ascan = 1;               // approximate-order scan + exact re-score when the index qualifies (positive payloads)
    int opt_bm25_wscan = 0;               // otherwise: wave-owned scan when the batch qualifies (needs the fine skip table, built at the
                                          // next erh_set_bm25_*), else the block scan
    int64_t opt_bm25_fine_max_mb = 8192;  // largest fine skip table built for it
    // metadata
    int64_t Nmeta = 0;
    DevBuf content_id, dir_id;
    bool has_content = false, has_dir = false;
    // Dense route by dir block (round 5): every dir class of at least dense_dir_block_min_rows documents gets its OWN copy with its own
    // golden-ratio placement (Xb: block c at rows [lo_c, lo_c + n_c), its documents in ascending order -- one run of the caller's numbering
    // in the reference's layout, gathered from anywhere otherwise; built on the first filtered call), and the queries filtered on that dir
    // scan n_c rows instead of N -- through the same kernels, as a view.
    struct DenseBlocks {
        bool valid = false;
        std::vector<int64_t> lo, n, mul, inv;                      // per class; n = 0: not a block (scattered, empty or too small)
    } blocks;
    DevBuf Xb, blk_tmp, blk_ids;
    std::vector<int32_t> dir_lo_h, dir_hi_h, dir_cnt_h;           // per class, from erh_set_doc_meta
    std::vector<int32_t> dir_order_h;                              // the documents that carry a class, ordered by (class, document)
    std::vector<int64_t> dir_off_h;                                // class c: dir_order_h[dir_off_h[c] .. dir_off_h[c + 1])
    int opt_dense_dir_blocks = 1;
    int64_t opt_dir_block_min_rows = 4096;
    int64_t opt_route_ridge = 160;                                 // query columns below which a scan of R rows costs like R x ridge (HBM-bound): the route decision's only constant
    int opt_dense_group_sample = 1;                                // ... with thresholds from a sample pass of the scan kernel per view where every view qualifies (0: store kernel + seed select)
    int opt_dense_group_launch = 1;                                // two or more block groups of a batch run as ONE launch per stage (dense_topk_grouped); 0: one pipeline per group
    // One routed dense call (dense_topk_routed), kept until its synchronisation point (dense_check_flags) has read its flag words:
    // the batch's groups, where each group's queries lie, and where results go.  Nothing else of a routed call lives on the handle.
    struct RoutedGroup {
        int c;          // dir class whose block the group scans; -1: the ordinary call with the group's filter values, -2: ... without a filter column
        int at, n;      // the group's queries = r_idx[at .. at + n) (caller rows, ascending)
        int pad_at;     // grouped launch: first row of the group in the padded query block (a multiple of 256); -1: run as its own pipeline
        int flag_slot;  // which 16-byte record of r_flags holds the flag words of the pipeline that answered it
    };
    struct Routed {
        bool done = false;                                         // the last dense call ran routed
        bool pending = false;                                      // ... and its flag words (r_flags) have not been read yet
        std::vector<RoutedGroup> groups;
        int n_flag_slots = 0;
        int last_slot = -1;                                        // the call's last pipeline: its flag words are read from h->flags, not from r_flags
        int grouped_slot = -1;                                     // flag slot of the grouped launch, -1: none in this call
        int grouped_bpad = 0;                                      // its padded query rows
        int q_dtype = 0, normalize_q = 0, B = 0, k = 0, mode = 0;
        int32_t *d_ids = nullptr; double *d_sc = nullptr; int32_t *d_len = nullptr;
    } routed;
    DevBuf r_q, r_ids, r_sc, r_len, r_flags;                      // the batch in group order, a group's results, every pipeline's flag words
    DevBuf r_tab;                                                  // ONE upload per routed call: r_idx | r_filt | (grouped launch:) view table | workgroup map | padded-row map
    int32_t *r_idx = nullptr;                                      // ... slices of r_tab: the batch's caller rows in group order,
    int16_t *r_filt = nullptr;                                     // ... and their filter values
    DevBuf r_q16;                                                  // ... its fp16 query block, copied aside only when a group has to run again
    uint32_t *r_flags_host = nullptr;                              // pinned
    size_t r_flags_host_cap = 0;
    std::vector<int32_t> r_idx_host;
    std::vector<int16_t> r_filt_host;
    std::vector<uint32_t> r_bad_host;
    std::vector<char> r_tab_host;
    DevBuf dir_rng;                          // {first document, last + 1} of every dir class (erh_set_doc_meta): a filtered BM25 query walks those tiles only
    int dir_rng_n = 0;
    int opt_bm25_dir_range = 1;
    // work space
    DevBuf qin, Q16, qnorm, tau, S0, cand, cand_cnt, flags, filt, filt2, seed_need;
    DevBuf o_ids, o_sc, o_len;              // staging for host outputs
    DevBuf part_sc, part_ids, part_len;
    DevBuf bm_redo;                          // approximate-order scan: (query, segment) pairs that go to the exact block scan
    // Long queries (round 6): the packed shape's 16-bit sums leave a query of nq tokens (65535 / nq) payload levels and an error bound of
    // 3 nq units -- from ~30 tokens on the list of "documents that can still reach the top k" no longer shrinks below its capacity and the
    // query falls back to the exact block scan (1024 queries with the reference's question lengths, 4 ... 45 tokens: 5 such segments,
    // 0.5 -> 1.7 ms per batch).  A batch whose longest query has more than bm25_long_tokens tokens scans with 32-bit sums (the
    // 16384-document shape): 0.68 ms.  (Only the long queries on that shape, in a launch of their own beside the packed one -- built and
    // measured, both stream orders, both 32-bit shapes: 0.84 ... 1.02 ms.  The launches do not overlap usefully, and ONE 45-token query in
    // one workgroup takes 0.5 ms whatever runs beside it: profiles/r06d_bm25_long_queries.log.)
    //   bm25_mixed (later in round 6): ONE launch whose workgroups pick their body by the length of their query -- the 32-bit body for the
    // queries longer than bm25_long_tokens, the packed one for all others (bm25_ascan_mixed_kernel); 0 = the whole batch on the 32-bit shape.
    int opt_bm25_long_tokens = 28;
    int opt_bm25_mixed = 1;
    int opt_bm25_long_segs = 4;              // mixed launch at one segment per query (>= 512 queries): document ranges a LONG query is cut into (1: none)
    DevBuf lpart_sc, lpart_ids, lpart_len, l_redo;   // ... their partial lists [B][bm25_long_segs][k] and redo words
    DevBuf bm_fin_ids, bm_fin_cnt;           // ... its final lists, handed to the batch-wide finish kernel (bm25_split_finish)
    int opt_bm25_split_finish = 0;
    DevBuf hy_sids, hy_ssc, hy_slen, hy_dids, hy_dsc, hy_dlen;
    DevBuf fa_ids, fa_sc, fa_len, fb_ids, fb_sc, fb_len;
    DevBuf scores_tmp, scores_wide;
    DevBuf dbg;                              // 16 x u64 section counters (measurement only)
    int opt_debug_counters = 0;
    // erh_get_stat: which kernels answered the calls since erh_create / erh_reset_stats.  Host counters (launch decisions are
    // made on the host) + two device counters the kernels bump themselves, so that device-output pipelines need no round trip:
    // dstats[0] queries answered by the dense exhaustive path, dstats[1] BM25 (query, segment) pairs handed to the exact scan
    DevBuf dstats;
    struct Stats {
        int64_t dense_calls = 0, dense_scan_pp5 = 0, dense_scan_pp3 = 0, dense_scan_gemv = 0, dense_scan_tile = 0,
                dense_sample_passes = 0, dense_tile384_nomem = 0, bm25_calls = 0, hybrid_calls = 0, dense_block_groups = 0,
                dense_grouped_launches = 0, bm25_mixed_launches = 0;
    } stats;
    // multi-GPU exchange (erh_comm_* / erh_allgather_topk): RCCL communicator + packed send / receive rows
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    std::shared_ptr<CommInitState> comm_pending;   // an init that timed out: its communicator (if it ever arrives) is destroyed later
    int opt_comm_timeout_s = 120;            // bounded wait of erh_comm_init
    DevBuf gather_send, gather_recv;
    // options
    int64_t opt_n0 = 32768, opt_n1 = 131072;
    int opt_n1_auto = 1;                   // snap n1 to a whole number of persistent-scan rounds (performance only)
    int opt_dense_ablate = 0, opt_bm25_ablate = 0;   // measurement only (results invalid when non-zero)
    int opt_dense_cfg = 0;                 // dense scan tile configuration (dense_scan.hip)
    int opt_dense_readahead = 1;           // cfg 2 only: fragments of the next K-step are read before its barrier
    int opt_small_single = 1;              // small batches: skip the refinement boundaries when the lists can take it
    int opt_gemv_pipe = -1;                  // ... software-pipelined loads: -1 by batch (2 / 4 column groups), 0 off, 1 on
    // ... its chunk loads with the non-temporal hint: -1 by call (on up to 32 queries unless the sparse route of a fused call runs beside
    // the stream), 0 off, 1 on.  Interleaved A/B (profiles/r06r_ab_gemv_nt_*.log): dense calls of 1 / 2 / 4 / 8 / 12 / 24 queries -7 ... -11 % scan
    // (-6 ... -9 % wall; 0.353 ... 0.373 ms whatever the batch, where the plain loads swing between 0.357 and 0.417), 16 / 32: +2 / -2 %,
    // 48 / 64 (four column groups): +4 / +1 %; beside the BM25 scan of a fused single query: +2 ... +8 % (the side stream's postings and
    // the hinted stream fight over the same queues) -- there the plain loads stay.
    int opt_gemv_nt = -1;
    // 256 x 256 ping-pong scan at ONE query tile per matrix (<= 256 queries, the grouped launch): chunk-side LDS-DMA with the non-temporal hint
    // (dense_scan_pp3_kernel VAR bit 6).  Measured (profiles/r06v_ab_scan_nt_*.log): +2.2 % scan at 256 queries, +5.5 % at 128, +0.8 % grouped -- the
    // second 64-byte half of a line no longer finds the first one's fill in L1.  Off; kept as an arm.
    int opt_dense_scan_nt = 0;
    bool sparse_beside = false;              // erh_hybrid_topk, hybrid_overlap 1: the sparse route runs on the side stream beside this dense pipeline
    int opt_gemv_kb = 32, opt_gemv_wgs = 2;  // skinny-GEMM stream: steps whose loads are in flight together, workgroups per CU at most
    int opt_dense_gemv = 1;                // batches of <= 16 queries: skinny-GEMM stream instead of the padded 256-query scan
    int opt_n0_auto = 0;                   // seed prefix snapped down to a whole number of scan rounds (less seed work, more candidates: a wash at 1M chunks)
    int opt_dense_sync = 0;                // the query-tile workgroups of a stream meet at a counter every four tiles (measured: no gain)
    int opt_dense_selfseed = 1;            // the ping-pong scan draws its own threshold sample (sample pass + cell maxima) instead of store kernel + S0 + seed select
    int opt_dense_tile384 = 1;             // batches padded to >= 512 queries scan on a 384 x 256 tile over tiled operands (+ N * d * 2 bytes on first use)
    int opt_dense_tiled = 0;               // keep a tiled, pre-swizzled copy of the chunk matrix for the ping-pong scan (+ N * d * 2 bytes; no measurable gain: off)
    int opt_dense_speculate = 1;           // speculative (verified) first threshold + a single scan stage; 0: guaranteed bounds, refined in stages
    int opt_dense_var = 0;                 // dense_scan_pp2_kernel VAR (bit 0: two barriers per stage, bit 1: static priority)
    int opt_dense_rot = 0;                 // K-rotation between the query tiles of a stream, in stages per query tile (-1: nk / n_qt)
    int opt_dense_pp = 3;                  // ping-pong persistent append scan (falls back to the kernels below when it does not apply)
    int opt_dense_persist = 1;             // persistent append scan (falls back to the plain launch when it does not apply)
    int n_cus = 0;                         // compute units persistent grids are sized for (one workgroup per CU): the device's, or option n_cus
    int n_cus_dev = 0;                     // compute units of the device
    // profiling
    bool prof = false;
    std::vector<EvPair> pending;
    std::vector<EvPair> pool;
    double ms[ERH_K_COUNT] = {0};
    int64_t launches[ERH_K_COUNT] = {0};
    double work_bytes[ERH_K_COUNT] = {0};
    double work_flops[ERH_K_COUNT] = {0};
    // diag of the last dense call
    double diag_maxerr = 0, diag_margin = 0;
    int32_t diag_uncert = 0;
    int32_t diag_exhaustive = 0;            // queries of the last call answered by the exhaustive path
    // exhaustive path (select.hip): per-query "not certifiable from the candidate list" flags, work space, and what
    // the last dense call needs for further rounds from the host (more than dense_exhaustive_max() flagged queries)
    DevBuf bad, ex_ws;
    DevBuf fin_ws;                           // dense_finalize_kernel, several workgroups per query (batches of <= 64): sync words + exact scores
    int opt_dense_fin_split = 1;
    struct LastDense {
        bool valid = false, hybrid = false;
        int B = 0, k = 0;
        const _Float16 *X = nullptr; int64_t N = 0, pos_inv = 1;      // what the call scanned (a view)
        const int16_t *filter_dev = nullptr;
        int32_t *d_ids = nullptr; double *d_sc = nullptr; int32_t *d_len = nullptr;
        // hybrid: the fusion to redo after the dense lists changed
        int k_sparse = 0, K = 0, topk = 0;
        int32_t *f_ids = nullptr; double *f_sc = nullptr; int32_t *f_len = nullptr;
    } last;

    int fail(int code, const char *what, hipError_t e = hipSuccess) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof buf, "%s: %s (%s)", erh_status_str(code), what, hipGetErrorString(e));
        else
            snprintf(buf, sizeof buf, "%s: %s", erh_status_str(code), what);
        err = buf;
        return code;
    }
};

#define HIPCHK(h, call)                                                     \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return (h)->fail(e_ == hipErrorOutOfMemory ? ERH_ERR_NOMEM : ERH_ERR_HIP, #call, e_); \
    } while (0)



struct ProfScope {
    erh_handle *h;
    hipStream_t st;
    EvPair ev;
    bool on;
    ProfScope(erh_handle *h_, hipStream_t st_, int cls, double bytes, double flops) : h(h_), st(st_), on(h_->prof) {
        if (!on) return;
        if (!h->pool.empty()) { ev = h->pool.back(); h->pool.pop_back(); }
        else {
            if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) { on = false; return; }
        }
        ev.cls = cls;
        h->work_bytes[cls] += bytes;
        h->work_flops[cls] += flops;
        (void)hipEventRecord(ev.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(ev.b, st);
        h->pending.push_back(ev);
    }
};

inline void drain_events(erh_handle *h) {
    for (auto &ev : h->pending) {
        if (hipEventSynchronize(ev.b) == hipSuccess) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ev.a, ev.b) == hipSuccess) { h->ms[ev.cls] += t; h->launches[ev.cls] += 1; }
        }
        h->pool.push_back(ev);
    }
    h->pending.clear();
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Multiplier pair of the row placement (see erh_handle::pos_mul): inv ~ n / golden ratio, coprime with n; mul = inv^-1 mod n.
inline void choose_placement(int64_t n, int64_t *mul, int64_t *inv) {
    auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
    int64_t g = (int64_t)((double)n * 0.6180339887498949);
    if (g < 1) g = 1;
    while (gcd(g, n) != 1) g = (g + 1 < n) ? g + 1 : 1;
    // extended Euclid: x with g * x == 1 (mod n)
    int64_t r0 = n, r1 = g, t0 = 0, t1 = 1;
    while (r1) { const int64_t q = r0 / r1; int64_t t = r0 - q * r1; r0 = r1; r1 = t; t = t0 - q * t1; t0 = t1; t1 = t; }
    *inv = g;
    *mul = ((t0 % n) + n) % n;
}


// ---- stage orchestration shared between the translation units (pipeline_dense.hip, pipeline_bm25.hip) -------------------------
// Dense top-k on device buffers with the dir filter pushed down as a row range where that pays (dense_topk_routed -> dense_topk_dev /
// dense_topk_grouped), the synchronisation point that reads a call's flag words and finishes flagged queries (dense_check_flags), and
// the BM25 scan + merge (bm25_topk_dev) with its query upload.
int dense_topk_dev(erh_handle *h, const void *q_dev, int q_dtype, int normalize_q, int B, int k,
                   const int16_t *filter_dev, int mode, int32_t *d_ids, double *d_sc, int32_t *d_len, hipStream_t st);
int dense_topk_routed(erh_handle *h, const void *q_dev, int q_dtype, int normalize_q, int B, int k, const int16_t *filter_host,
                      const int16_t *filter_dev, int mode, int32_t *d_ids, double *d_sc, int32_t *d_len, hipStream_t st);
int dense_check_flags(erh_handle *h, hipStream_t st);
int bm25_topk_dev(erh_handle *h, const int32_t *qptr_dev, const int32_t *qtok_dev, int B, int k,
                  const int16_t *filter_dev, int32_t *d_ids, double *d_sc, int32_t *d_len, double postings_bytes,
                  int max_qlen, hipStream_t st);
int upload_bm25_queries(erh_handle *h, const int32_t *q_indptr, const int32_t *q_tok, int B, hipStream_t st,
                        const std::vector<int64_t> &host_indptr, double *bytes, int *max_qlen,
                        const int16_t *filt_a = nullptr, const int16_t **filt_a_dev = nullptr /* validated host filter column(s) that ride along */,
                        const int16_t *filt_b = nullptr, const int16_t **filt_b_dev = nullptr);

