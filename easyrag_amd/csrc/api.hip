// C ABI of libeasyrag_hip.so (see include/easyrag_hip.h): handle, device state, work space, stage
// orchestration on the caller's HIP stream, event-based kernel timing.  No CPU compute path lives here.
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

namespace {

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

}  // namespace

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
    int opt_bm25_long_tokens = 28;
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
                dense_grouped_launches = 0;
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

namespace {

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

void drain_events(erh_handle *h) {
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
void choose_placement(int64_t n, int64_t *mul, int64_t *inv) {
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

// K-rotation of the ping-pong scan in stages of 32 halves per query tile (see dense_scan_pp2_kernel)
static int dense_rot_stages(const erh_handle *h, int d, int Bpad) {
    const int nk = d / 32, n_qt = Bpad / erh::dense_scan_q_tile();
    if (h->opt_dense_rot >= 0) return h->opt_dense_rot % (nk > 0 ? nk : 1);
    return n_qt > 1 ? (nk / n_qt) & ~1 : 0;
}

// Append-stage scan: persistent kernel when enabled and applicable, else one workgroup per tile.
hipError_t scan_append(erh_handle *h, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q16,
                       int Bpad, int B, const float *tau, const int16_t *filt, const int16_t *dir, ErhCand *cand,
                       uint32_t *cnt, int cap, uint32_t *flags, hipStream_t st) {
    // small batches: the skinny-GEMM stream (dense_gemv.hip) instead of a 256-query tile that is mostly padding
    if (h->opt_dense_gemv && h->opt_dense_ablate == 0 && B <= erh::dense_gemv_max_queries()) {
        hipError_t e = erh::launch_dense_gemv_append(X, N, d, c0, c1, Q16, B, tau, filt, dir, cand, cnt, cap, flags,
                                                     h->n_cus, h->opt_gemv_kb, h->opt_gemv_wgs, h->opt_gemv_pipe, st);
        if (e != hipErrorInvalidValue) { h->stats.dense_scan_gemv += (c1 > c0); return e; }
        (void)hipGetLastError();
    }
    const int abl = h->opt_dense_ablate;
    const bool pp_code = abl == 0 || abl == 7 || abl == 8 || (abl >= 11 && abl <= 18) || (abl >= 20 && abl <= 24);
    const bool own_x = X == h->view.X;                        // the view's matrix itself (padded; not a gathered block of debug rows)
    const bool global_x = own_x && h->view.global;            // ... and the handle's whole matrix: its tiled copies apply
    if (h->qt5_valid && h->xt384_valid && abl == 0 && global_x) {
        // 384 x 256 tile over the tiled copies (the caller checked the options and built the copies)
        hipError_t e = erh::launch_dense_scan_pp5(h->Xt384.as<_Float16>(), N, d, c0, c1, h->Qt.as<_Float16>(), Bpad, B, tau, filt,
                                                  dir, cand, cnt, cap, flags, h->n_cus,
                                                  h->opt_dense_rot >= 0 ? h->opt_dense_rot : 0 /* the workgroups of a stream on the same stage: measured best here (profiles/r04w_kbench_tile384.log) */, st);
        if (e != hipErrorInvalidValue) { h->stats.dense_scan_pp5 += (c1 > c0); return e; }
        (void)hipGetLastError();
    }
    if (h->opt_dense_pp >= 4 && pp_code && global_x && h->xt_valid && h->qt_valid) {
        // both operands from their tiled copies (dense_scan_pp4_kernel)
        hipError_t e = erh::launch_dense_scan_pp4(h->Xt.as<_Float16>(), N, d, c0, c1, h->Qt.as<_Float16>(), Bpad, B, tau, filt,
                                                  dir, cand, cnt, cap, flags, h->n_cus, h->opt_dense_ablate,
                                                  h->opt_debug_counters ? h->dbg.as<unsigned long long>() : nullptr,
                                                  h->opt_dense_rot >= 0 ? h->opt_dense_rot : (Bpad > erh::dense_scan_q_tile() ? (d / 32) / (Bpad / erh::dense_scan_q_tile()) : 0),
                                                  st);
        if (e != hipErrorInvalidValue) return e;
        (void)hipGetLastError();
    }
    if (h->opt_dense_pp && pp_code) {
        const bool own = own_x;
        const int QT = erh::dense_scan_q_tile();
        // the strict ping-pong kernel streams the tiled copy when there is one and the stage starts on a tile boundary
        const bool tiled = global_x && h->opt_dense_pp >= 3 && h->opt_dense_tiled && h->xt_valid && h->opt_dense_var == 0 &&
                           c0 % QT == 0;
        const int var = tiled ? 2 : h->opt_dense_var;
        uint32_t *sync = nullptr;
        if (h->opt_dense_sync && h->opt_dense_pp >= 3 && own && Bpad > QT) {
            if (h->scan_sync.ensure(1024) == hipSuccess && hipMemsetAsync(h->scan_sync.p, 0, 1024, st) == hipSuccess)
                sync = h->scan_sync.as<uint32_t>();
        }
        hipError_t e = erh::launch_dense_scan_pp(tiled ? h->Xt.as<_Float16>() : X, N, d, c0, c1, Q16, Bpad, B, tau, filt, dir,
                                                 cand, cnt, cap, flags, h->n_cus, h->opt_dense_ablate,
                                                 h->opt_debug_counters ? h->dbg.as<unsigned long long>() : nullptr,
                                                 (h->opt_dense_pp >= 2 && own)
                                                     ? (1 | (var << 1) | (h->opt_dense_pp >= 3 ? 8 : 0) | (dense_rot_stages(h, d, Bpad) << 8)) : 0,
                                                 sync, nullptr, st);
        if (e != hipErrorInvalidValue) { h->stats.dense_scan_pp3 += (c1 > c0); return e; }
        (void)hipGetLastError();
    }
    if (h->opt_dense_persist && (h->opt_dense_ablate == 0 || h->opt_dense_ablate >= 6)) {
        hipError_t e = erh::launch_dense_scan_persist(h->opt_dense_cfg, X, N, d, c0, c1, Q16, Bpad, B, tau, filt, dir, cand,
                                                      cnt, cap, flags, h->n_cus, h->opt_dense_ablate, h->opt_dense_readahead, st);
        if (e != hipErrorInvalidValue) { h->stats.dense_scan_tile += (c1 > c0); return e; }
        (void)hipGetLastError();
    }
    h->stats.dense_scan_tile += (c1 > c0);
    return erh::launch_dense_scan_append(h->opt_dense_cfg, X, N, d, c0, c1, Q16, Bpad, B, tau, filt, dir, cand, cnt, cap,
                                         flags, h->opt_dense_ablate,
                                         h->opt_debug_counters ? h->dbg.as<unsigned long long>() : nullptr, st);
}

// ---- dense pipeline on device buffers ------------------------------------------------------------
int dense_topk_dev(erh_handle *h, const void *q_dev, int q_dtype, int normalize_q, int B, int k,
                   const int16_t *filter_dev, int mode, int32_t *d_ids, double *d_sc, int32_t *d_len, hipStream_t st) {
    const int QT = erh::dense_scan_q_tile();
    const int Bpad = round_up(B, QT);
    h->routed.done = false;
    const int d = h->d;
    const int64_t N = h->view.N;
    const int64_t pos_mul = h->view.mul, pos_inv = h->view.inv;
    const bool global_view = h->view.global;
    const int cap = erh::kDenseCapMax;
    HIPCHK(h, h->Q16.ensure((size_t)Bpad * d * 2));
    HIPCHK(h, h->qnorm.ensure((size_t)Bpad * 4));
    HIPCHK(h, h->tau.ensure((size_t)Bpad * 4));
    HIPCHK(h, h->cand.ensure((size_t)B * cap * sizeof(ErhCand)));
    HIPCHK(h, h->cand_cnt.ensure((size_t)B * 4));
    h->cand_rows = B;
    HIPCHK(h, h->flags.ensure(64));
    HIPCHK(h, h->seed_need.ensure((size_t)B * 4));
    HIPCHK(h, h->bad.ensure((size_t)B * 4));
    HIPCHK(h, h->ex_ws.ensure(erh::dense_exhaustive_bytes(N)));
    uint32_t *bad = h->bad.as<uint32_t>();                       // (cleared by the query-prep kernel, like the flag words)
    int64_t n0 = std::min<int64_t>(std::min<int64_t>(h->opt_n0, erh::kDenseN0Max), N);
    if (n0 < 1) n0 = 1;
    // The persistent scan walks ceil(tiles / streams) rounds of 256-chunk tiles.  With dense_n0_auto the seed prefix shrinks
    // (never below a quarter of the option, nor below 16 k) to where the rest of the corpus is a whole number of rounds:
    // the same number of rounds as with the full prefix, and less seed work (1M chunks: 32768 -> 16960).
    if (h->opt_n0_auto && h->opt_dense_speculate && N > n0 && B > erh::dense_gemv_max_queries()) {
        const int64_t streams = std::max<int64_t>(8, (std::max(h->n_cus, 8) / (8 * (Bpad / QT))) * 8);
        const int64_t step = streams * QT;
        const int64_t rounds = (N - n0 + step - 1) / step;
        const int64_t cand = N - rounds * step;
        if (cand >= std::max<int64_t>(n0 / 4, 16 * (int64_t)k) && cand < n0) n0 = cand;
    }
    // batches on the 384 x 256 tile whose threshold comes from the stored prefix (dir filters, deep ranks): the prefix ends on a tile
    // boundary of that kernel -- 32640 = 85 x 384 instead of 32768 -- so the append stage can start there (c0 % 384 == 0); otherwise
    // it would fall back to the 256 x 256 scan (filtered 1024-query batch: scan class -1.4 ... -2 %, profiles/r05j_ab_filtered.log)
    // (only when that kernel can run at all -- the same predicate that builds its operands below, the failed copy included: in every
    // fall-back a prefix of 32640 rows would take the tiled 256 x 256 path away from the append stage instead)
    const bool tile384_ok = h->opt_dense_tile384 && global_view && Bpad >= 2 * QT && h->opt_dense_pp == 3 && h->opt_dense_var == 0 &&
                            h->opt_dense_ablate == 0 && !h->opt_dense_sync && d % 64 == 0 && d / 32 >= 8 &&
                            N >= 2 * erh::dense_scan_pp5_rows() && !h->xt384_nomem &&
                            (h->xt384_valid || h->opt_tile384_max_mb < 0 ||
                             (size_t)((N + erh::dense_scan_pp5_rows() - 1) / erh::dense_scan_pp5_rows()) * erh::dense_scan_pp5_rows() * (size_t)d * 2 <=
                                 ((size_t)h->opt_tile384_max_mb << 20));
    if (tile384_ok && n0 < N && n0 >= 4 * erh::dense_scan_pp5_rows())
        n0 = n0 / erh::dense_scan_pp5_rows() * erh::dense_scan_pp5_rows();
    const int ld = round_up((int)n0, 256);
    HIPCHK(h, h->S0.ensure((size_t)Bpad * ld * 4));
    uint32_t *flags = h->flags.as<uint32_t>();   // [0] overflow, [1] maxerr (float bits), [2] uncertified
    // small batches: work space of the final kernel's several-workgroups-per-query mode (zeroed once; the kernel leaves it zero)
    double *fin_s64 = nullptr;
    uint32_t *fin_sync = nullptr;
    if (h->opt_dense_fin_split && B <= erh::dense_finalize_split_max()) {
        if (!h->fin_ws.p) {
            const size_t sync_bytes = (size_t)erh::dense_finalize_split_max() * 8;
            HIPCHK(h, h->fin_ws.ensure(sync_bytes + (size_t)erh::dense_finalize_split_max() * erh::kDenseRescoreMax * 8));
        }                                                   // (the sync words are cleared by every call's query preparation, below)
        fin_sync = h->fin_ws.as<uint32_t>();
        fin_s64 = reinterpret_cast<double *>(h->fin_ws.as<char>() + (size_t)erh::dense_finalize_split_max() * 8);
    }

    erh::ErhGroupIo vo_io{};
    const erh::ErhGroupIo *vo = nullptr;
    if (h->view_out.on) { vo_io.id_map = h->view_out.id_map; vo_io.single_lo = h->view_out.id_lo; vo_io.row_map = h->view_out.row_map; vo = &vo_io; }
    h->qt_valid = false;
    h->qt5_valid = false;
    { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
      HIPCHK(h, erh::launch_prep_queries(q_dev, q_dtype, normalize_q, B, Bpad, d, h->Q16.as<_Float16>(),
                                         h->qnorm.as<float>(), bad, flags, st, nullptr, fin_sync, fin_sync ? 2 * erh::dense_finalize_split_max() : 0));
      // the tiled-operand scan reads the query block as stage images too (512 KiB per 256 queries, once per call)
      if (h->opt_dense_pp >= 4 && global_view && h->xt_valid && d % 64 == 0 && B > erh::dense_gemv_max_queries()) {
          HIPCHK(h, h->Qt.ensure((size_t)Bpad * d * 2));
          HIPCHK(h, erh::launch_dense_tile_rows(h->Q16.as<_Float16>(), Bpad, d, h->Qt.p, st));
          h->qt_valid = true;
      }
      // the 384 x 256 scan of batches padded to >= 512 queries: the chunk matrix' 384-row tiled copy (once per erh_set_dense, here
      // on first use) and the query block as stage images (512 KiB per 256 queries, per call)
      if (h->opt_dense_tile384 && global_view && Bpad >= 2 * QT && h->opt_dense_pp == 3 && h->opt_dense_var == 0 && h->opt_dense_ablate == 0 &&
          !h->opt_dense_sync && d % 64 == 0 && d / 32 >= 8 && N >= 2 * erh::dense_scan_pp5_rows()) {      // (tile384_ok without its memory terms)
          if (!h->xt384_valid && !h->xt384_nomem) {
              const int rows = erh::dense_scan_pp5_rows();
              const int64_t n_tiles = (N + rows - 1) / rows;
              // The copy doubles the matrix.  A corpus that leaves no room for it (X above about half of HBM) keeps the 256 x 256
              // scan, which needs no copy: out-of-memory here is a fall-through, not an error, and is not retried until the
              // next erh_set_dense.
              const size_t want = (size_t)n_tiles * rows * (size_t)d * 2;
              const hipError_t ea = (h->opt_tile384_max_mb >= 0 && want > ((size_t)h->opt_tile384_max_mb << 20))
                                        ? hipErrorOutOfMemory : h->Xt384.ensure(want);
              if (ea == hipErrorOutOfMemory) {
                  (void)hipGetLastError();
                  h->xt384_nomem = true;
                  h->stats.dense_tile384_nomem += 1;
              } else {
                  HIPCHK(h, ea);
                  HIPCHK(h, erh::launch_dense_tile_rows_n(h->X.as<_Float16>(), N, d, rows, h->Xt384.p, st));
                  h->xt384_valid = true;
              }
          }
          if (h->xt384_valid) {
              if (!h->qt_valid) {
                  HIPCHK(h, h->Qt.ensure((size_t)Bpad * d * 2));
                  HIPCHK(h, erh::launch_dense_tile_rows(h->Q16.as<_Float16>(), Bpad, d, h->Qt.p, st));
              }
              h->qt5_valid = true;
          }
      } }

    const _Float16 *X = h->view.X;
    const _Float16 *Q16 = h->Q16.as<_Float16>();
    if (filter_dev && !global_view) return h->fail(ERH_ERR_INVALID, "dense block view with a filter");
    const int16_t *dir = nullptr;                 // dir id by stored position, only needed when a filter is present
    if (filter_dev && h->has_dir) {
        if (pos_mul == 1) {
            dir = h->dir_id.as<int16_t>();
        } else {
            if (!h->dir_pos_valid) {
                HIPCHK(h, h->dir_pos.ensure((size_t)N * 2));
                HIPCHK(h, erh::launch_permute_dir(h->dir_id.as<int16_t>(), N, pos_inv, h->dir_pos.as<int16_t>(), st));
                h->dir_pos_valid = true;
            }
            dir = h->dir_pos.as<int16_t>();
        }
    }
    auto scan_work = [&](int64_t rows, double *bytes, double *flops) {
        *bytes = (double)rows * d * 2.0 + (double)Bpad * d * 2.0;
        *flops = 2.0 * (double)rows * (double)Bpad * (double)d;
    };
    double wb, wf;
    const bool small = h->opt_dense_gemv && h->opt_dense_ablate == 0 && B <= erh::dense_gemv_max_queries();
    // ---- the ping-pong scan draws its own threshold sample (round 4) ----------------------------------------------------
    // Sample pass = the scan kernel over the first tile(s) of every chunk stream, without thresholds: the two best scores of
    // every 64-row cell are all that leaves the registers, and the speculative threshold is the rank-th largest of them
    // (seed_cells_select_kernel: 4 KiB per query instead of a 128 KiB row of S0 read twice).  The main launch then scans ALL
    // rows -- the sampled ones again -- so there is no store kernel, no S0 and no candidate hand-over: what the sampled rows
    // cost twice (1.6 % of the scan at 1024 queries) is less than storing and selecting from their scores.
    // Not with a dir filter (the sample would have to be filtered per query), not for the skinny-GEMM batches, not when the
    // rank is so deep that cells with three or more of the sample's best would be the rule (the threshold would still be
    // valid, only loose): those take the stages below.
    {
        const int n_streams = erh::dense_scan_pp_streams(h->n_cus, Bpad);
        // one tile per chunk stream, more only if that samples fewer than 16384 rows (1024 queries: 64 streams -> 16384 rows, 512
        // queries: 128 streams -> 32768; the pass takes a tile time whatever the number of streams)
        const int seed_tiles = n_streams > 0 ? (int)std::max<int64_t>(1, std::min<int64_t>(h->opt_n0, 16384) / ((int64_t)n_streams * QT)) : 0;
        // (round 6) with MORE streams than the sample needs tiles -- one query tile: 256 streams -- only the first ceil(16384 / 256) streams
        // take a tile: the pass still lasts one tile time, but the main launch re-scans 16384 rows instead of 65536
        const bool partial = h->opt_dense_selfseed >= 2 && seed_tiles == 1 && (int64_t)n_streams * QT > std::min<int64_t>(h->opt_n0, 16384);
        const int64_t rows_seed = partial ? (std::min<int64_t>(h->opt_n0, 16384) + QT - 1) / QT * QT : (int64_t)seed_tiles * n_streams * QT;
        const bool tiled_run = h->opt_dense_tiled && h->xt_valid && global_view;
        const int n_cells = partial ? (int)(rows_seed / QT) * 4 : seed_tiles * n_streams * 4;
        const int rank = (rows_seed > 0 && rows_seed <= N) ? erh_dense_seed_rank(k, rows_seed, N) : k;
        // From 512 queries on: below, the sampled rows scanned twice (one tile per stream = 65536 rows at 256 queries) cost more
        // than the store kernel and the select they replace (profiles/r04s_kbench_sample_pass.log).
        const bool ok = h->opt_dense_selfseed && (Bpad >= 2 * QT || partial) && h->opt_dense_speculate && h->opt_dense_pp == 3 && h->opt_dense_var == 0 &&
                        h->opt_dense_ablate == 0 && !h->opt_dense_sync && !tiled_run && !small && !filter_dev && n_streams > 0 &&
                        d % 64 == 0 && d / 32 >= 8 && rows_seed > 0 && N >= 2 * rows_seed && rank < k && 4 * rank <= n_cells &&
                        erh::seed_cells_select_fits(n_cells * 2);   // (its LDS sort: out of reach with the device's CU count, checked anyway)
        if (ok) {
            erh::ErhSeedIo sio{};
            sio.seed_tiles = seed_tiles;
            sio.n_cells = n_cells;
            sio.mode = 1;
            const int n_vals = n_cells * 2;
            HIPCHK(h, h->seed_top.ensure((size_t)Bpad * n_vals * 4));
            sio.seed_top = h->seed_top.as<float>();
            h->stats.dense_sample_passes += 1;
            const int lean = 1 | 8 | (dense_rot_stages(h, d, Bpad) << 8);
            // (the pass books no work: the sampled rows are scanned again below, and N rows are what the algorithm needs; its own class)
            { ProfScope ps(h, st, ERH_K_DENSE_SAMPLE, 0, 0);
              HIPCHK(h, erh::launch_dense_scan_pp(X, N, d, 0, rows_seed, Q16, Bpad, B, h->tau.as<float>(), nullptr, nullptr,
                                                  h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, flags, h->n_cus, 0,
                                                  nullptr, lean, nullptr, &sio, st)); }
            { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
              HIPCHK(h, erh::launch_seed_cells_select(sio.seed_top, n_vals, B, rank, h->qnorm.as<float>(), h->xnorm_max, d,
                                                      h->tau.as<float>(), h->cand_cnt.as<uint32_t>(), st)); }
            scan_work(N, &wb, &wf);
            int rc_scan = ERH_OK;
            { ProfScope ps(h, st, ERH_K_DENSE_SCAN, wb, wf);
              hipError_t e = scan_append(h, X, N, d, 0, N, Q16, Bpad, B, h->tau.as<float>(), nullptr, nullptr,
                                         h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, flags, st);
              if (e != hipSuccess) rc_scan = h->fail(ERH_ERR_HIP, "dense scan behind the sample pass", e); }
            if (rc_scan != ERH_OK) return rc_scan;
            if (h->fork_after_scan) HIPCHK(h, hipEventRecord(h->ev_fork, st));
            { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
              HIPCHK(h, erh::launch_dense_finalize(B, k, mode, h->qnorm.as<float>(), h->xnorm_max, d, X, Q16,
                                                   h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, d_ids, d_sc, d_len,
                                                   reinterpret_cast<float *>(flags + 1), flags + 2, bad, N, pos_mul, pos_inv,
                                                   h->tau.as<float>(), h->n_cus, fin_s64, fin_sync, st, vo));
              HIPCHK(h, erh::launch_dense_exhaustive(bad, B, 0, k, X, N, d, Q16, filter_dev,
                                                     (filter_dev && h->has_dir) ? h->dir_id.as<int16_t>() : nullptr, pos_inv, h->ex_ws.p,
                                                     flags, h->n_cus, d_ids, d_sc, d_len, h->rerun ? nullptr : h->dstats.as<unsigned long long>(), 1 /* count only */, st)); }
            h->last = erh_handle::LastDense();
            h->last.valid = true;
            h->last.B = B; h->last.k = k; h->last.filter_dev = filter_dev; h->last.X = X; h->last.N = N; h->last.pos_inv = pos_inv;
            h->last.d_ids = d_ids; h->last.d_sc = d_sc; h->last.d_len = d_len;
            return ERH_OK;
        }
    }
    // stage A: score the seed prefix densely, k-th best -> pruning threshold
    scan_work(n0, &wb, &wf);
    { ProfScope ps(h, st, ERH_K_DENSE_SCAN, wb, wf);
      hipError_t e = hipErrorInvalidValue;
      if (small) e = erh::launch_dense_gemv_store(X, N, d, 0, (int)n0, Q16, B, h->S0.as<float>(), ld, h->n_cus,
                                                  h->opt_gemv_kb, h->opt_gemv_wgs, h->opt_gemv_pipe, st);
      if (e == hipErrorInvalidValue) {
          (void)hipGetLastError();
          // one 256 x 256 tile per workgroup leaves CUs idle when the seed grid is small (B = 256: 128 tiles on 256 CUs);
          // the 128 x 256 configuration (4 waves, two workgroups per CU) halves the tile and fills the chip
          int store_cfg = h->opt_dense_cfg;
          if (store_cfg == 0 && ((n0 + QT - 1) / QT) * (Bpad / QT) < h->n_cus) store_cfg = 1;
          e = erh::launch_dense_scan_store(store_cfg, Q16, Bpad, X, N, d, 0, (int)n0, h->S0.as<float>(), ld, st);
      }
      HIPCHK(h, e); }
    // Rank of the prefix score that seeds the threshold.  Guaranteed: k.  Speculative: the prefix is an even sample of
    // the corpus (erh_set_dense's row placement), so the number of true top-k members inside it is ~Poisson(mu),
    // mu = k * n0 / N; the rank mu + 6.5 sqrt(mu) + 3 is reached with probability < 1e-7, i.e. the rank-th prefix score
    // is below the corpus' k-th best -- which dense_finalize_kernel verifies for every query (exhaustive path if not).
    int rank = k;
    bool speculate = false;
    if (h->opt_dense_speculate && N > n0) {
        rank = erh_dense_seed_rank(k, n0, N);
        speculate = rank < k;
    }
    { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
      HIPCHK(h, erh::launch_seed_select(h->S0.as<float>(), ld, (int)n0, 0, B, k, rank, h->qnorm.as<float>(), h->xnorm_max, d,
                                        filter_dev, dir, h->tau.as<float>(), h->cand.as<ErhCand>(),
                                        h->cand_cnt.as<uint32_t>(), cap, bad, h->seed_need.as<uint32_t>(), st)); }
    if (N > n0) {
        // Stage boundaries n0 < b1 < b2 < ... < N: the threshold is refined (and the candidate list cut back to what
        // still matters) at every boundary, so a stage adds about k * (b_next - b) / b candidates however large N is.
        // b1 = option dense_n1 (0: no refinement at all), then x4 while at least twice that much corpus remains.
        // Results do not depend on the boundaries; the number of tile rounds of the persistent scans does: each
        // stream (n_cus / query-tiles of them) walks ceil(tiles / streams) tiles, so with dense_n1_auto a boundary
        // moves (within -25 % .. +50 %) to where the stage is a whole number of rounds and the rest wastes least.
        const int64_t streams = std::max<int64_t>(8, (std::max(h->n_cus, 8) / (8 * (Bpad / QT))) * 8);
        const int64_t step = streams * QT;                              // chunks per round
        auto snap = [&](int64_t from, int64_t want) -> int64_t {
            if (!h->opt_n1_auto) return want;
            int64_t best = want, best_rounds = -1;
            for (int64_t r1 = 1; from + r1 * step < N; ++r1) {
                const int64_t c = from + r1 * step;
                if (c < want - want / 4) continue;
                if (c > want + want / 2) break;
                const int64_t rest_tiles = (N - c + QT - 1) / QT;
                const int64_t rounds = r1 + (rest_tiles + streams - 1) / streams;
                if (best_rounds < 0 || rounds < best_rounds) { best_rounds = rounds; best = c; }
            }
            return best;
        };
        int64_t cur = n0;
        int64_t want = h->opt_n1;
        if (want <= n0) want = 0;
        if (speculate) want = 0;                 // already tighter than any refinement of a guaranteed bound: one stage
        // small batches: one stage when the candidates a seed-only threshold admits (about k * N / n0 per query) fit
        // the lists comfortably -- the refinement launches cost more than they save when there are 16 lists to cut
        if (small && h->opt_small_single && (double)k * (double)N / (double)n0 <= 0.5 * cap) want = 0;
        while (cur < N) {
            int64_t next = N;
            if (want > cur && want < N) {
                const int64_t b = snap(cur, want);
                if (b > cur && b < N) next = b;
            }
            scan_work(next - cur, &wb, &wf);
            { ProfScope ps(h, st, ERH_K_DENSE_SCAN, wb, wf);
              HIPCHK(h, scan_append(h, X, N, d, cur, next, Q16, Bpad, B, h->tau.as<float>(), filter_dev, dir,
                                     h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, flags, st)); }
            if (next < N) {
                ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
                HIPCHK(h, erh::launch_cand_refine(B, k, h->qnorm.as<float>(), h->xnorm_max, d, h->tau.as<float>(),
                                                  h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, bad, st));
                want = (N >= 8 * next) ? 4 * next : 0;                  // another boundary only if plenty of corpus remains
            }
            cur = next;
        }
    }
    if (h->fork_after_scan) HIPCHK(h, hipEventRecord(h->ev_fork, st));   // the sparse route may start beside the selection kernels
    { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
      HIPCHK(h, erh::launch_dense_finalize(B, k, mode, h->qnorm.as<float>(), h->xnorm_max, d, X, Q16,
                                           h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, d_ids, d_sc, d_len,
                                           reinterpret_cast<float *>(flags + 1), flags + 2, bad, N, pos_mul, pos_inv,
                                           speculate ? h->tau.as<float>() : nullptr, h->n_cus, fin_s64, fin_sync, st, vo));
      // queries the candidate budgets could not certify get their exact answer from the exhaustive path: the call enqueues the
      // COUNT only (one workgroup; it settles the "unanswered" word and the flagged count), dense_check_flags -- the synchronisation
      // point every caller passes before it reads results -- runs the exact rounds when, and only when, the count is not zero
      HIPCHK(h, erh::launch_dense_exhaustive(bad, B, 0, k, X, N, d, Q16, filter_dev,
                                             (filter_dev && h->has_dir) ? h->dir_id.as<int16_t>() : nullptr, pos_inv, h->ex_ws.p,
                                             flags, h->n_cus, d_ids, d_sc, d_len, h->rerun ? nullptr : h->dstats.as<unsigned long long>(), 1 /* count only */, st)); }
    h->last = erh_handle::LastDense();
    h->last.valid = true;
    h->last.B = B; h->last.k = k; h->last.filter_dev = filter_dev; h->last.X = X; h->last.N = N; h->last.pos_inv = pos_inv;
    h->last.d_ids = d_ids; h->last.d_sc = d_sc; h->last.d_len = d_len;
    return ERH_OK;
}

// Read the flag words of the last dense call (synchronises the stream).  If queries were flagged for the exhaustive path, its
// rounds run here, dense_exhaustive_max() queries at a time (and a fused call's RRF is redone over the corrected dense lists),
// so the caller always gets an answer.
int routed_group_run(erh_handle *h, const erh_handle::RoutedGroup &g, const void *q_rows, int q_dtype, int normalize_q, hipStream_t st, bool direct = false);
int routed_group_scatter(erh_handle *h, const erh_handle::RoutedGroup &g, hipStream_t st);

int dense_check_flags(erh_handle *h, hipStream_t st) {
    if (h->routed.done) {
        erh_handle::Routed &R = h->routed;
        if (!R.pending) { HIPCHK(h, hipStreamSynchronize(st)); return ERH_OK; }
        // A routed call: every pipeline (a group run on its own, or the grouped launch over all block groups) left its flag words in
        // r_flags.  A group with flagged queries is run again on its own, to the end (its exhaustive rounds included), and scattered
        // over its first answer; a fused call's RRF is redone then.
        const erh_handle::LastDense saved = h->last;
        // (slots are handed out in pipeline order, so the last pipeline's slot is the highest: its words are still in h->flags)
        if (R.n_flag_slots > 1)
            HIPCHK(h, hipMemcpyAsync(h->r_flags_host, h->r_flags.p, (size_t)(R.n_flag_slots - 1) * 16, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipMemcpyAsync(h->r_flags_host + 4 * (R.n_flag_slots - 1), h->flags.p, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        R.pending = false;
        double maxerr = 0;
        int uncert = 0, exhaustive = 0;
        bool redone = false;
        auto slot_words = [&](int slot, uint32_t *f) { memcpy(f, h->r_flags_host + 4 * slot, 16); };
        for (int sl = 0; sl < R.n_flag_slots; ++sl) {                  // what the pipelines without a flagged query certify
            uint32_t f[4];
            slot_words(sl, f);
            if (f[0]) continue;
            float me;
            memcpy(&me, &f[1], 4);
            maxerr = std::max(maxerr, (double)me); uncert += (int32_t)f[2];
        }
        // The grouped launch flags QUERIES (bad[] over its padded rows), not groups: read them, and keep its fp16 query block -- the
        // re-runs below reuse the work space it lives in.  (The grouped launch is the last pipeline of its call, so both are intact.)
        const _Float16 *q16_copy = nullptr;
        if (R.grouped_slot >= 0) {
            uint32_t f[4];
            slot_words(R.grouped_slot, f);
            if (f[0]) {
                h->r_bad_host.resize((size_t)R.grouped_bpad);
                HIPCHK(h, hipMemcpyAsync(h->r_bad_host.data(), h->bad.p, (size_t)R.grouped_bpad * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(h, h->r_q16.ensure((size_t)R.grouped_bpad * h->d * 2));
                HIPCHK(h, hipMemcpyAsync(h->r_q16.p, h->Q16.p, (size_t)R.grouped_bpad * h->d * 2, hipMemcpyDeviceToDevice, st));
                HIPCHK(h, hipStreamSynchronize(st));
                q16_copy = h->r_q16.as<_Float16>();
            }
        }
        const size_t row_bytes = (size_t)h->d * (R.q_dtype == ERH_F16 ? 2 : 4);
        for (size_t gi = 0; gi < R.groups.size(); ++gi) {
            const erh_handle::RoutedGroup g = R.groups[gi];
            uint32_t f[4];
            slot_words(g.flag_slot, f);
            if (!f[0]) continue;
            const void *q_rows;
            int dt = R.q_dtype, nq = R.normalize_q;
            if (g.pad_at >= 0) {                                       // a group of the grouped launch: flagged iff one of its queries is
                bool any = false;
                for (int i = 0; i < g.n; ++i) any = any || h->r_bad_host[(size_t)g.pad_at + i] != 0u;
                if (!any) continue;
                q_rows = q16_copy + (size_t)g.pad_at * h->d;           // already unit fp16: the same values the first run scored
                dt = ERH_F16; nq = 0;
            } else {
                q_rows = h->r_q.as<char>() + (size_t)g.at * row_bytes;
            }
            h->rerun = true;
            int rc = routed_group_run(h, g, q_rows, dt, nq, st);       // (clears routed.done: the check below is the ordinary one)
            if (rc == ERH_OK) rc = dense_check_flags(h, st);
            if (rc == ERH_OK) rc = routed_group_scatter(h, g, st);
            h->rerun = false;
            if (rc != ERH_OK) return rc;
            maxerr = std::max(maxerr, h->diag_maxerr); uncert += h->diag_uncert; exhaustive += h->diag_exhaustive;
            redone = true;
        }
        if (redone && saved.hybrid) {
            const int32_t *cid = h->has_content ? h->content_id.as<int32_t>() : nullptr;
            HIPCHK(h, erh::launch_rrf(h->hy_sids.as<int32_t>(), h->hy_slen.as<int32_t>(), saved.k_sparse, saved.d_ids, saved.d_len,
                                      saved.k, cid, saved.B, saved.K, saved.topk, saved.f_ids, saved.f_sc, saved.f_len, st));
        }
        if (redone) HIPCHK(h, hipStreamSynchronize(st));
        h->last = erh_handle::LastDense();
        R.done = true;
        h->diag_maxerr = maxerr; h->diag_uncert = uncert; h->diag_exhaustive = exhaustive;
        h->diag_margin = 2.0 * (double)h->d * 1.1920929e-7 * (double)h->xnorm_max;
        return ERH_OK;
    }
    uint32_t f[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(f, h->flags.p, sizeof f, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (f[0] && h->last.valid) {
        const int total = (int)f[3], per = erh::dense_exhaustive_max();
        const erh_handle::LastDense &L = h->last;
        for (int skip = 0; skip < total; skip += per)          // (the call enqueued the count only: every answering round runs here)
            HIPCHK(h, erh::launch_dense_exhaustive(h->bad.as<uint32_t>(), L.B, skip, L.k, L.X, L.N, h->d,
                                                   h->Q16.as<_Float16>(), L.filter_dev,
                                                   (L.filter_dev && h->has_dir) ? h->dir_id.as<int16_t>() : nullptr, L.pos_inv,
                                                   h->ex_ws.p, h->flags.as<uint32_t>(), h->n_cus, L.d_ids, L.d_sc,
                                                   L.d_len, h->dstats.as<unsigned long long>(), 0, st));
        if (L.hybrid) {
            const int32_t *cid = h->has_content ? h->content_id.as<int32_t>() : nullptr;
            HIPCHK(h, erh::launch_rrf(h->hy_sids.as<int32_t>(), h->hy_slen.as<int32_t>(), L.k_sparse, L.d_ids, L.d_len,
                                      L.k, cid, L.B, L.K, L.topk, L.f_ids, L.f_sc, L.f_len, st));
        }
        HIPCHK(h, hipMemcpyAsync(f, h->flags.p, sizeof f, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
    }
    float me;
    memcpy(&me, &f[1], 4);
    h->diag_maxerr = me;
    h->diag_uncert = (int32_t)f[2];
    h->diag_exhaustive = (int32_t)f[3];
    h->diag_margin = 2.0 * (double)h->d * 1.1920929e-7 * (double)h->xnorm_max;   // for a unit-norm query
    if (f[0]) return h->fail(ERH_ERR_OVERFLOW, "dense candidate list overflowed and the exhaustive path could not finish");
    return ERH_OK;
}

// One group of a routed dense call as a pipeline of its own: its queries (`q_rows`, n of them) against its dir's block as a view, or
// the ordinary call with the group's filter values; results in r_ids / r_sc / r_len (routed_group_scatter puts them into the caller's rows).
int routed_group_run(erh_handle *h, const erh_handle::RoutedGroup &g, const void *q_rows, int q_dtype, int normalize_q, hipStream_t st, bool direct) {
    const erh_handle::Routed &R = h->routed;
    const int16_t *sub_filter = nullptr;
    if (g.c >= 0) {
        h->view.X = h->Xb.as<_Float16>() + (size_t)h->blocks.lo[g.c] * h->d;
        h->view.N = h->blocks.n[g.c]; h->view.mul = h->blocks.mul[g.c]; h->view.inv = h->blocks.inv[g.c]; h->view.global = false;
    } else if (g.c == -1) {
        sub_filter = h->r_filt + g.at;
    }
    // direct: the final kernel writes the group's lists to the caller's rows with block rows mapped to document ids (the call's first
    // pass: nothing but the final kernel writes results then).  A re-run at the check goes through r_ids + the scatter kernel, because
    // its exhaustive rounds write unmapped ids.
    if (direct) {
        h->view_out.on = true;
        h->view_out.row_map = h->r_idx + g.at;
        h->view_out.id_map = g.c >= 0 ? h->blk_ids.as<int32_t>() : nullptr;
        h->view_out.id_lo = g.c >= 0 ? (int32_t)h->blocks.lo[g.c] : 0;
    }
    const int rc = dense_topk_dev(h, q_rows, q_dtype, normalize_q, g.n, R.k, sub_filter, R.mode,
                                  direct ? R.d_ids : h->r_ids.as<int32_t>(), direct ? R.d_sc : h->r_sc.as<double>(),
                                  direct ? R.d_len : h->r_len.as<int32_t>(), st);
    h->view_out = erh_handle::ViewOut();
    h->view_global();
    return rc;
}

int routed_group_scatter(erh_handle *h, const erh_handle::RoutedGroup &g, hipStream_t st) {
    const erh_handle::Routed &R = h->routed;
    HIPCHK(h, erh::launch_scatter_topk_rows(h->r_ids.as<int32_t>(), h->r_sc.as<double>(), h->r_len.as<int32_t>(), h->r_idx + g.at,
                                            g.n, R.k, g.c >= 0 ? (int32_t)h->blocks.lo[g.c] : 0, g.c >= 0 ? h->blk_ids.as<int32_t>() : nullptr,
                                            R.d_ids, R.d_sc, R.d_len, st));
    return ERH_OK;
}

// The per-dir copies of the chunk matrix (see erh_handle::DenseBlocks): built on the first filtered call after erh_set_dense /
// erh_set_doc_meta; the blocks' rows come back in the caller's order through the gather kernel and are placed by their own multiplier.
// Xb holds the BLOCK classes only, one after the other (a corpus with one large dir and a long tail of small ones pays
// for the large one, not for a second copy of everything), + kDensePadRows zero rows.
int ensure_dense_blocks(erh_handle *h, hipStream_t st) {
    if (h->blocks.valid) return ERH_OK;
    const int nc = (int)h->dir_cnt_h.size();
    const int d = h->d;
    h->blocks.lo.assign(nc, 0); h->blocks.n.assign(nc, 0); h->blocks.mul.assign(nc, 1); h->blocks.inv.assign(nc, 1);
    int64_t rows = 0;
    const int64_t total = (int64_t)h->dir_order_h.size();              // documents that carry a class, in (class, document) order
    for (int c = 0; c < nc && total <= h->N; ++c) {
        const int64_t cnt = h->dir_cnt_h[c];
        if (cnt >= h->opt_dir_block_min_rows) { h->blocks.lo[c] = rows; h->blocks.n[c] = cnt; rows += cnt; }
    }
    if (rows > 0) {
        // The block copies are (at most) a second chunk matrix.  Like the 384-row copy: a corpus that leaves no room for it keeps the filter
        // column (no blocks until the next erh_set_dense / erh_set_doc_meta); dense_tile384_max_mb bounds both copies (test hook).
        const size_t want = (size_t)(rows + erh::kDensePadRows) * d * 2;
        const hipError_t ea = (h->opt_tile384_max_mb >= 0 && want > ((size_t)h->opt_tile384_max_mb << 20)) ? hipErrorOutOfMemory : h->Xb.ensure(want);
        if (ea == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            h->blocks.n.assign(nc, 0);
            h->blocks.valid = true;
            return ERH_OK;
        }
        HIPCHK(h, ea);
        HIPCHK(h, hipMemsetAsync(h->Xb.as<char>() + (size_t)rows * d * 2, 0, (size_t)erh::kDensePadRows * d * 2, st));
        // row r of block c is the caller's document blk_ids[lo_c + r]: the class' documents in ascending order (ties keep their order),
        // wherever they lie in the caller's numbering -- one run when the corpus was loaded dir by dir, scattered otherwise
        std::vector<int32_t> ids((size_t)rows);
        for (int c = 0; c < nc; ++c)
            if (h->blocks.n[c])
                memcpy(ids.data() + h->blocks.lo[c], h->dir_order_h.data() + h->dir_off_h[c], (size_t)h->blocks.n[c] * 4);
        HIPCHK(h, h->blk_ids.ensure((size_t)rows * 4));
        HIPCHK(h, hipMemcpyAsync(h->blk_ids.p, ids.data(), (size_t)rows * 4, hipMemcpyHostToDevice, st));
        for (int c = 0; c < nc; ++c) {
            const int64_t cnt = h->blocks.n[c];
            if (!cnt) continue;
            int64_t mul = 1, inv = 1;
            if (h->opt_dense_shuffle && cnt > 2) choose_placement(cnt, &mul, &inv);
            h->blocks.mul[c] = mul; h->blocks.inv[c] = inv;
            HIPCHK(h, h->blk_tmp.ensure((size_t)cnt * d * 2));
            HIPCHK(h, erh::launch_gather_rows(h->X.as<_Float16>(), h->blk_ids.as<int32_t>(), h->blocks.lo[c], cnt, d, h->pos_mul, h->N,
                                              h->blk_tmp.as<_Float16>(), st));
            HIPCHK(h, erh::launch_permute_rows(h->blk_tmp.as<_Float16>(), cnt, d, h->Xb.as<_Float16>() + (size_t)h->blocks.lo[c] * d, 0, mul, cnt, st));
        }
        HIPCHK(h, hipStreamSynchronize(st));                           // (`ids` is pageable host memory of this scope)
        h->blk_tmp.release();
    }
    h->blocks.valid = true;
    return ERH_OK;
}

// ---- the grouped launch (round 6): every block group of a batch in ONE launch per stage ----------------------------------------
// The batch's block groups are laid out one after the other in a padded query block, each group in whole 256-row query tiles; a table
// with one entry per query tile (kernels.h: ErhDenseView -- the dir's block copy, its placement, its seed prefix and rank, the chunk
// streams of the persistent scan that belong to it) is read by every stage instead of one (X, N) pair per launch:
//   query preparation (rows gathered through q_src, padding rows zeroed) -> seed prefix of every tile's own block scored densely
//   (dense_scan_store_kernel<.., GROUPED>) -> rank-th best per query = threshold (seed_select_kernel with the table) -> ONE persistent
//   scan over the rest of all blocks (dense_scan_pp3_kernel<0, 32 | 40>: n_cus workgroups dealt to the tiles in proportion to their
//   chunk tiles, so the launch takes max over tiles of ceil(chunk tiles / streams) rounds -- four blocks of 250 k rows: 14 rounds
//   instead of 4 x 4) -> final kernel (pinned fp64 re-score out of the tile's block, results written to the caller's rows with block
//   rows mapped to document ids: no scatter launch) -> the count of flagged queries.  Seven launches and one 16-byte flag record
//   whatever the number of groups.  Queries the budgets cannot certify are flagged as always; their GROUPS are then run again as
//   pipelines of their own at the synchronisation point (dense_check_flags) -- rare, and the code that ran every group before round 6.
struct GroupedPlan {
    std::vector<erh::ErhDenseView> views;
    std::vector<int32_t> wg_view, q_src;
    int grid = 0, n0_max = 0, bpad = 0;
    int64_t n_max = 0;
    bool halfq = true;
    erh::ErhGroupIo gio{};      // device pointers into the routed call's upload (r_tab)
    bool sample = false;        // thresholds from a sample pass of the scan kernel over every view (views[].seed_rows / n_cells) instead of store kernel + S0 + seed select
    int cells_max = 0;
};

// chunk streams per query tile: the smallest number of rounds R with sum ceil(tiles_v / R) <= n_cus, then ceil(tiles_v / R) streams each
static bool plan_streams(std::vector<erh::ErhDenseView> &views, const std::vector<int64_t> &tiles, int n_cus, int *grid) {
    int64_t with_work = 0, t_max = 0;
    for (int64_t t : tiles) { with_work += t > 0; t_max = std::max(t_max, t); }
    if (with_work > n_cus) return false;
    int64_t lo = 1, hi = std::max<int64_t>(t_max, 1);
    auto need = [&](int64_t r) { int64_t s = 0; for (int64_t t : tiles) s += (t + r - 1) / r; return s; };
    while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (need(mid) <= n_cus) hi = mid; else lo = mid + 1; }
    int at = 0;
    for (size_t v = 0; v < views.size(); ++v) {
        const int nwg = (int)((tiles[v] + lo - 1) / lo);
        views[v].wg0 = at; views[v].nwg = nwg;
        at += nwg;
    }
    *grid = at;
    return true;
}

int dense_topk_grouped(erh_handle *h, const void *q_dev, int q_dtype, int normalize_q, int k, int mode, const GroupedPlan &P,
                       int32_t *d_ids, double *d_sc, int32_t *d_len, hipStream_t st) {
    const int d = h->d, Bpad = P.bpad, n_qt = (int)P.views.size();
    const int cap = erh::kDenseCapMax;
    const int ld = round_up(std::max(P.n0_max, 1), 256);
    h->routed.done = false;
    HIPCHK(h, h->Q16.ensure((size_t)Bpad * d * 2));
    HIPCHK(h, h->qnorm.ensure((size_t)Bpad * 4));
    HIPCHK(h, h->tau.ensure((size_t)Bpad * 4));
    HIPCHK(h, h->cand.ensure((size_t)Bpad * cap * sizeof(ErhCand)));
    HIPCHK(h, h->cand_cnt.ensure((size_t)Bpad * 4));
    h->cand_rows = Bpad;
    HIPCHK(h, h->flags.ensure(64));
    HIPCHK(h, h->seed_need.ensure((size_t)Bpad * 4));
    HIPCHK(h, h->bad.ensure((size_t)Bpad * 4));
    HIPCHK(h, h->ex_ws.ensure(erh::dense_exhaustive_bytes(P.n_max)));
    if (P.sample) HIPCHK(h, h->seed_top.ensure((size_t)Bpad * P.cells_max * 2 * 4));
    else HIPCHK(h, h->S0.ensure((size_t)Bpad * ld * 4));
    const erh::ErhGroupIo &gio = P.gio;                               // (the tables went up with the routed call's one upload)
    uint32_t *flags = h->flags.as<uint32_t>(), *bad = h->bad.as<uint32_t>();
    h->qt_valid = false;
    h->qt5_valid = false;
    { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
      HIPCHK(h, erh::launch_prep_queries(q_dev, q_dtype, normalize_q, Bpad, Bpad, d, h->Q16.as<_Float16>(), h->qnorm.as<float>(), bad, flags, st,
                                         gio.q_src)); }
    // booked work: what the algorithm needs -- every block row once per query tile that scans it
    double seed_rows = 0, scan_rows = 0;
    for (const erh::ErhDenseView &v : P.views) { seed_rows += v.n0; scan_rows += (double)(v.N - v.n0); }
    if (P.sample) {
        // thresholds from the scan kernel's own sample: rows [0, seed_rows) of every view without thresholds, the two best scores of every
        // 64-row cell -> the rank-th largest of them per query (the unfiltered path's scheme, per view); the main launch scans ALL rows.
        // (The pass books no work: its rows are scanned again, and N rows per view are what the algorithm needs.)
        erh::ErhSeedIo sio{};
        sio.seed_top = h->seed_top.as<float>();
        sio.n_cells = P.cells_max;
        sio.mode = 1;
        h->stats.dense_sample_passes += 1;
        { ProfScope ps(h, st, ERH_K_DENSE_SAMPLE, 0, 0);
          HIPCHK(h, erh::launch_dense_scan_pp_grouped(gio, P.grid, d, h->Q16.as<_Float16>(), Bpad, h->tau.as<float>(), h->cand.as<ErhCand>(),
                                                      h->cand_cnt.as<uint32_t>(), cap, flags, P.halfq ? 1 : 0, st, &sio)); }
        { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
          HIPCHK(h, erh::launch_seed_cells_select(sio.seed_top, P.cells_max * 2, Bpad, k, h->qnorm.as<float>(), h->xnorm_max, d, h->tau.as<float>(),
                                                  h->cand_cnt.as<uint32_t>(), st, gio.views)); }
    } else {
        { ProfScope ps(h, st, ERH_K_DENSE_SCAN, seed_rows * d * 2.0 + (double)Bpad * d * 2.0, 2.0 * seed_rows * 256.0 * d);
          HIPCHK(h, erh::launch_dense_scan_store_grouped(gio, n_qt, P.n0_max, h->n_cus, h->Q16.as<_Float16>(), Bpad, d, h->S0.as<float>(), ld, st)); }
        { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
          HIPCHK(h, erh::launch_seed_select(h->S0.as<float>(), ld, P.n0_max, 0, Bpad, k, k, h->qnorm.as<float>(), h->xnorm_max, d, nullptr, nullptr,
                                            h->tau.as<float>(), h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, bad,
                                            h->seed_need.as<uint32_t>(), st, gio.views)); }
    }
    if (P.grid > 0) {
        ProfScope ps(h, st, ERH_K_DENSE_SCAN, scan_rows * d * 2.0 + (double)Bpad * d * 2.0, 2.0 * scan_rows * 256.0 * d);
        HIPCHK(h, erh::launch_dense_scan_pp_grouped(gio, P.grid, d, h->Q16.as<_Float16>(), Bpad, h->tau.as<float>(), h->cand.as<ErhCand>(),
                                                    h->cand_cnt.as<uint32_t>(), cap, flags, P.halfq ? 1 : 0, st));
        h->stats.dense_scan_pp3 += 1;
    }
    if (h->fork_after_scan) HIPCHK(h, hipEventRecord(h->ev_fork, st));
    { ProfScope ps(h, st, ERH_K_DENSE_SELECT, 0, 0);
      HIPCHK(h, erh::launch_dense_finalize(Bpad, k, mode, h->qnorm.as<float>(), h->xnorm_max, d, nullptr, h->Q16.as<_Float16>(),
                                           h->cand.as<ErhCand>(), h->cand_cnt.as<uint32_t>(), cap, d_ids, d_sc, d_len,
                                           reinterpret_cast<float *>(flags + 1), flags + 2, bad, 0, 1, 1, h->tau.as<float>(), h->n_cus,
                                           nullptr, nullptr, st, &gio));
      HIPCHK(h, erh::launch_dense_exhaustive(bad, Bpad, 0, k, nullptr, P.n_max, d, h->Q16.as<_Float16>(), nullptr, nullptr, 1, h->ex_ws.p,
                                             flags, h->n_cus, d_ids, d_sc, d_len, h->dstats.as<unsigned long long>(), 1 /* count only */, st)); }
    h->stats.dense_grouped_launches += 1;
    return ERH_OK;
}

// Dense top-k with the dir filter pushed down as a ROW RANGE: the batch's queries are grouped by filter class; a class with a block
// copy scans that copy (n_c rows, no filter, block rows mapped back to the caller's document ids), everything else -- unfiltered
// queries, small or unknown classes -- runs the ordinary call with its filter column.  Two or more block groups run as ONE launch per
// stage (dense_topk_grouped); a single block group (the reference's one filtered query per call) and the ordinary group are pipelines
// of their own.  filter_host: the caller's host column.
int dense_topk_routed(erh_handle *h, const void *q_dev, int q_dtype, int normalize_q, int B, int k, const int16_t *filter_host,
                      const int16_t *filter_dev, int mode, int32_t *d_ids, double *d_sc, int32_t *d_len, hipStream_t st) {
    const int nc = (int)h->dir_cnt_h.size();
    erh_handle::Routed &R = h->routed;
    R.pending = false;
    bool route = h->opt_dense_dir_blocks && filter_host && filter_dev && h->has_dir && nc > 0 && h->Nmeta == h->N && h->opt_dense_ablate == 0;
    std::map<int, std::vector<int32_t>> groups;
    int n_block_groups = 0;
    if (route) {
        if (!h->blocks.valid) { int rc = ensure_dense_blocks(h, st); if (rc != ERH_OK) return rc; }
        for (int b = 0; b < B; ++b) {
            const int f = filter_host[b];
            const bool blk = f >= 0 && f < nc && h->blocks.n[f] > 0;
            groups[blk ? f : -1].push_back(b);
        }
        n_block_groups = (int)groups.size() - (groups.count(-1) ? 1 : 0);
        route = n_block_groups > 0;
    }
    const int QT = erh::dense_scan_q_tile();
    const bool grouped = route && n_block_groups >= 2 && h->opt_dense_group_launch && h->opt_dense_speculate && h->opt_dense_pp >= 1 &&
                         h->d % 64 == 0 && h->d / 32 >= 8;
    // Route or not: compare the WORK of the two ways, in row x query-column units.  A scan of R rows against n queries costs
    // R x max(columns(n), ridge): `columns` is the width the kernel that would run it computes (16-column groups of the skinny-GEMM
    // stream up to 64 queries, half a query tile up to 128, whole 256-row tiles above), `ridge` (option dense_route_ridge, 160) the
    // width below which the scan is bound by the matrix bytes and the columns are free -- a property of the chip (HBM bytes per
    // MFMA flop), not a timing of one box.  dense_dir_blocks = 2 always routes (the parity tests).
    if (route && h->opt_dense_dir_blocks == 1) {
        const double ridge = (double)std::max<int64_t>(h->opt_route_ridge, 1);
        auto cols_plain = [&](int n) { return (double)(n <= 64 ? round_up(n, 16) : n <= 128 ? 128 : round_up(n, QT)); };
        auto cols_group = [&](int n) { return (double)(n <= 128 ? 128 : round_up(n, QT)); };
        double routed_work = 0;
        for (auto &g : groups) {
            const int n = (int)g.second.size();
            if (g.first < 0) routed_work += (double)h->N * std::max(cols_plain(n), ridge);
            else routed_work += (double)h->blocks.n[g.first] * std::max(grouped ? cols_group(n) : cols_plain(n), ridge);
        }
        route = routed_work < (double)h->N * std::max(cols_plain(B), ridge);
    }
    if (!route) return dense_topk_dev(h, q_dev, q_dtype, normalize_q, B, k, filter_dev, mode, d_ids, d_sc, d_len, st);

    // ---- the plan: group order, flag slots, and for the grouped launch its tables --------------------------------------------------
    R.groups.clear();
    R.q_dtype = q_dtype; R.normalize_q = normalize_q; R.B = B; R.k = k; R.mode = mode;
    R.d_ids = d_ids; R.d_sc = d_sc; R.d_len = d_len;
    R.grouped_slot = -1; R.grouped_bpad = 0;
    h->r_idx_host.clear();
    h->r_filt_host.clear();
    GroupedPlan P;
    std::vector<int64_t> tiles;
    int n_slots = 0;
    for (auto &g : groups) {                                            // (the ordinary group, key -1, comes first)
        bool any_filter = false;
        for (int32_t b : g.second) any_filter = any_filter || filter_host[b] >= 0;
        erh_handle::RoutedGroup rg{g.first >= 0 ? g.first : (any_filter ? -1 : -2), (int)h->r_idx_host.size(), (int)g.second.size(), -1, 0};
        if (g.first >= 0 && grouped) {
            const int c = g.first;
            const int64_t Nv = h->blocks.n[c];
            rg.pad_at = P.bpad;
            int64_t n0 = std::min<int64_t>(std::min<int64_t>(h->opt_n0, erh::kDenseN0Max), Nv);
            if (n0 < 1) n0 = 1;
            const int rank = Nv > n0 ? erh_dense_seed_rank(k, n0, Nv) : k;
            for (int t0 = 0; t0 < rg.n; t0 += QT) {
                erh::ErhDenseView v{};
                v.X = h->Xb.as<_Float16>() + (size_t)h->blocks.lo[c] * h->d;
                v.N = Nv; v.mul = h->blocks.mul[c]; v.inv = h->blocks.inv[c];
                v.n0 = (int32_t)n0; v.rank = rank; v.id_lo = (int32_t)h->blocks.lo[c];
                v.nq = std::min(QT, rg.n - t0);
                P.views.push_back(v);
                tiles.push_back((Nv - n0 + QT - 1) / QT);
                for (int i = 0; i < QT; ++i) P.q_src.push_back(i < v.nq ? g.second[(size_t)t0 + i] : -1);
                P.halfq = P.halfq && v.nq <= QT / 2;
                P.n0_max = std::max(P.n0_max, (int)n0);
                P.n_max = std::max(P.n_max, Nv);
            }
            P.bpad += round_up(rg.n, QT);
        } else {
            rg.flag_slot = n_slots++;
        }
        R.groups.push_back(rg);
        for (int32_t b : g.second) { h->r_idx_host.push_back(b); h->r_filt_host.push_back(filter_host[b]); }
    }
    bool run_grouped = grouped && !P.views.empty();
    if (run_grouped) {
        // more query tiles with work than compute units, or a padded block beyond what the work space should grow to: every group on its own
        if (P.bpad > 16384 || !plan_streams(P.views, tiles, h->n_cus, &P.grid)) {
            run_grouped = false;
            for (auto &rg : R.groups) if (rg.pad_at >= 0) { rg.pad_at = -1; rg.flag_slot = n_slots++; }
        } else {
            // Thresholds from a sample pass (the unfiltered path's scheme, per view) when every view can give one: rows of its first
            // seed_tiles x streams x 256 positions (>= min(16384, N / 4), at most half of the view), a speculative rank below k, and cells
            // enough that the rank-th largest of the cells' two best is close to the sample's (2 rank <= cells: a cell with three of the
            // sample's best hides one: a few ranks of looseness, verified like every speculative threshold).  The scan then covers all rows.
            if (h->opt_dense_selfseed && h->opt_dense_group_sample) {
                std::vector<erh::ErhDenseView> vs = P.views;
                std::vector<int64_t> t_all(vs.size());
                for (size_t v = 0; v < vs.size(); ++v) t_all[v] = (vs[v].N + QT - 1) / QT;
                int grid2 = 0, cells_max = 0;
                bool ok = plan_streams(vs, t_all, h->n_cus, &grid2);
                for (size_t v = 0; ok && v < vs.size(); ++v) {
                    const int64_t per = (int64_t)vs[v].nwg * QT;
                    const int64_t want = std::max<int64_t>(per, std::min<int64_t>(std::min<int64_t>(h->opt_n0, 16384), vs[v].N / 4));
                    const int64_t seed_tiles = (want + per - 1) / per, rows = seed_tiles * per;
                    const int rank = erh_dense_seed_rank(k, rows, vs[v].N);
                    const int64_t cells = seed_tiles * vs[v].nwg * 4;
                    ok = per > 0 && rows * 2 <= vs[v].N && rank < k && 2 * (int64_t)rank <= cells && cells * 2 <= 12288;
                    vs[v].n0 = 0; vs[v].rank = rank; vs[v].seed_rows = (int32_t)rows; vs[v].n_cells = (int32_t)cells;
                    cells_max = std::max(cells_max, (int)cells);
                }
                if (ok && erh::seed_cells_select_fits(cells_max * 2)) { P.views = vs; P.grid = grid2; P.sample = true; P.cells_max = cells_max; P.n0_max = 0; }
            }
            P.wg_view.resize((size_t)P.grid);
            for (size_t v = 0; v < P.views.size(); ++v)
                for (int i = 0; i < P.views[v].nwg; ++i) P.wg_view[(size_t)P.views[v].wg0 + i] = (int32_t)v;
            R.grouped_slot = n_slots++;
            R.grouped_bpad = P.bpad;
            for (auto &rg : R.groups) if (rg.pad_at >= 0) rg.flag_slot = R.grouped_slot;
        }
    }
    R.n_flag_slots = n_slots;
    HIPCHK(h, h->r_flags.ensure((size_t)n_slots * 16));
    if (h->r_flags_host_cap < (size_t)n_slots * 16) {
        if (h->r_flags_host) (void)hipHostFree(h->r_flags_host);
        h->r_flags_host = nullptr; h->r_flags_host_cap = 0;
        const size_t want = std::max<size_t>((size_t)n_slots * 16, 1024);
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->r_flags_host), want, hipHostMallocDefault));
        h->r_flags_host_cap = want;
    }
    // ONE upload for everything the call's kernels read from the host (a pageable copy costs ~4 us of host time and ~7 us on the
    // stream whatever its size): caller rows in group order | their filter values | the grouped launch's tables
    {
        const size_t n_qt = run_grouped ? P.views.size() : 0;
        const size_t off_filt = (size_t)B * 4, off_views = (off_filt + (size_t)B * 2 + 63) / 64 * 64;
        const size_t off_wg = off_views + n_qt * sizeof(erh::ErhDenseView), off_src = off_wg + (run_grouped ? (size_t)P.grid * 4 : 0);
        const size_t bytes = off_src + (run_grouped ? (size_t)P.bpad * 4 : 0);
        h->r_tab_host.resize(bytes);
        memcpy(h->r_tab_host.data(), h->r_idx_host.data(), (size_t)B * 4);
        memcpy(h->r_tab_host.data() + off_filt, h->r_filt_host.data(), (size_t)B * 2);
        if (run_grouped) {
            memcpy(h->r_tab_host.data() + off_views, P.views.data(), n_qt * sizeof(erh::ErhDenseView));
            memcpy(h->r_tab_host.data() + off_wg, P.wg_view.data(), (size_t)P.grid * 4);
            memcpy(h->r_tab_host.data() + off_src, P.q_src.data(), (size_t)P.bpad * 4);
        }
        HIPCHK(h, h->r_tab.ensure(bytes));
        HIPCHK(h, hipMemcpyAsync(h->r_tab.p, h->r_tab_host.data(), bytes, hipMemcpyHostToDevice, st));
        char *base = h->r_tab.as<char>();
        h->r_idx = reinterpret_cast<int32_t *>(base);
        h->r_filt = reinterpret_cast<int16_t *>(base + off_filt);
        if (run_grouped) {
            P.gio.views = reinterpret_cast<const erh::ErhDenseView *>(base + off_views);
            P.gio.wg_view = reinterpret_cast<const int32_t *>(base + off_wg);
            P.gio.q_src = reinterpret_cast<const int32_t *>(base + off_src);
            P.gio.id_map = h->blk_ids.as<int32_t>();
        }
    }
    const size_t row_bytes = (size_t)h->d * (q_dtype == ERH_F16 ? 2 : 4);
    bool any_seq = false;
    for (const erh_handle::RoutedGroup &rg : R.groups) any_seq = any_seq || rg.pad_at < 0;
    HIPCHK(h, h->r_ids.ensure((size_t)B * k * 4));                     // (a re-run of a group of the grouped launch needs its result rows too: sized here, while nothing is in flight)
    HIPCHK(h, h->r_sc.ensure((size_t)B * k * 8));
    HIPCHK(h, h->r_len.ensure((size_t)B * 4));
    if (any_seq) {
        HIPCHK(h, h->r_q.ensure((size_t)B * row_bytes));
        // the batch in group order: a copy of the library's own, so a group can be run again at erh_dense_check time
        HIPCHK(h, erh::launch_gather_query_rows(q_dev, h->r_idx, B, (int)row_bytes, h->r_q.p, st));
    }
    // ---- groups that are pipelines of their own first, the grouped launch last (its work space must survive until the check) ----------
    // (the LAST pipeline's flag words stay where they are -- h->flags -- until the check reads them: no copy behind it)
    int last_slot = run_grouped ? R.grouped_slot : -1;
    if (!run_grouped) for (const erh_handle::RoutedGroup &rg : R.groups) last_slot = rg.flag_slot;
    R.last_slot = last_slot;
    for (const erh_handle::RoutedGroup &rg : R.groups) {
        if (rg.pad_at >= 0) continue;
        int rc = routed_group_run(h, rg, h->r_q.as<char>() + (size_t)rg.at * row_bytes, q_dtype, normalize_q, st, true);
        if (rc != ERH_OK) return rc;
        // the group's flag words, kept aside (the next pipeline's query preparation clears them): read all at once in dense_check_flags
        if (rg.flag_slot != last_slot)
            HIPCHK(h, hipMemcpyAsync(h->r_flags.as<char>() + (size_t)rg.flag_slot * 16, h->flags.p, 16, hipMemcpyDeviceToDevice, st));
        h->stats.dense_block_groups += (rg.c >= 0);
    }
    if (run_grouped) {
        int rc = dense_topk_grouped(h, q_dev, q_dtype, normalize_q, k, mode, P, d_ids, d_sc, d_len, st);
        if (rc != ERH_OK) return rc;
        for (const erh_handle::RoutedGroup &rg : R.groups) h->stats.dense_block_groups += (rg.pad_at >= 0);
    }
    h->last = erh_handle::LastDense();
    h->last.B = B; h->last.k = k; h->last.d_ids = d_ids; h->last.d_sc = d_sc; h->last.d_len = d_len;   // (a fused call's RRF redo reads these)
    R.done = true;
    R.pending = true;
    return ERH_OK;
}

int bm25_topk_dev(erh_handle *h, const int32_t *qptr_dev, const int32_t *qtok_dev, int B, int k,
                  const int16_t *filter_dev, int32_t *d_ids, double *d_sc, int32_t *d_len, double postings_bytes,
                  int max_qlen, hipStream_t st) {
    Bm25State &S = h->bm[h->cur];
    const int16_t *dir = h->has_dir ? h->dir_id.as<int16_t>() : nullptr;
    // approximate-order scan + exact re-score (default) when every payload is a positive normal number; otherwise the
    // wave-owned scan (no per-token workgroup barrier) when the index has its fine skip table and lane j can own token j
    // of every query; otherwise the block scan.  All three produce the same lists.
    const bool ascan = h->opt_bm25_ascan && S.ascan_ok;
    const bool wscan = !ascan && h->opt_bm25_wscan && S.n_fine > 0 && max_qlen <= erh::bm25_wscan_max_tokens() &&
                       h->opt_bm25_ablate == 0;
    // fixed-point scan: the 512-thread shape (two workgroups = two queries per CU) whenever its list holds k and a skip
    // table at its tile size exists
    const int small_docs = erh::bm25_ascan_tile_docs(1);
    const bool have16 = S.tile_docs == small_docs || S.n_tiles16 > 0;
    const bool small_k = ascan && k <= erh::bm25_ascan_small_max_k() && B >= 8;   // (a handful of queries: 16 waves per query finish sooner)
    int shape_ = !small_k ? 0 : h->opt_bm25_small == 2 ? 2 : (h->opt_bm25_small == 1 && have16) ? 1 : 0;
    // a batch with a query too long for 16-bit sums: the 32-bit shape for all of it (erh_handle::opt_bm25_long_tokens)
    if (shape_ == 2 && h->opt_bm25_long_tokens > 0 && max_qlen > h->opt_bm25_long_tokens && h->opt_bm25_ablate == 0) shape_ = have16 ? 1 : 0;
    const int shape = shape_;
    const bool small = shape != 0;                                        // two workgroups per CU
    const int as_docs = erh::bm25_ascan_tile_docs(shape);
    const int tiles = ascan ? (int)((S.Nb + as_docs - 1) / as_docs) : S.n_tiles;
    // segments per query: one resident round of workgroups -- 512 slots with two 512-thread workgroups per CU.  (The packed shape
    // walks half as many tiles per query as the 16384-document shape, whose best was two rounds: every segment pays a first
    // tile without a threshold and a re-score of its own list.  profiles/r04r_kbench_bm25_segs.log)
    int segs = h->opt_bm25_segs > 0 ? h->opt_bm25_segs : ((shape == 1 ? 1024 : 512) + B - 1) / B;
    segs = std::max(1, std::min(segs, ascan ? std::min(tiles, std::max(S.n_tiles, 1)) : tiles));
    // the merge sorts pow2(segs * k) padded slots in one workgroup: beyond 2048 it costs more than the extra segments save
    // (profiles/r03c_small_batch.log: one query, k = 192: 31 segments 0.038 + 0.106 ms, 10 segments 0.052 + 0.027 ms)
    while (segs > 1 && (int64_t)segs * k > 2048) --segs;
    unsigned long long *dbg = h->opt_debug_counters ? h->dbg.as<unsigned long long>() : nullptr;
    const int32_t *q_order = (h->qorder_valid && qptr_dev == h->qptr) ? h->qorder : nullptr;
    const bool split_fin = ascan && small && h->opt_bm25_split_finish && h->opt_bm25_ablate == 0;
    if (ascan) {
        HIPCHK(h, h->bm_redo.ensure((size_t)B * segs * 4));
        HIPCHK(h, hipMemsetAsync(h->bm_redo.p, 0, (size_t)B * segs * 4, st));
    }
    if (split_fin) {
        HIPCHK(h, h->bm_fin_ids.ensure((size_t)B * segs * erh::bm25_ascan_fin_cap() * 4));
        HIPCHK(h, h->bm_fin_cnt.ensure((size_t)B * segs * 4));
    }
    auto scan = [&](double *p_sc, int32_t *p_ids, int32_t *p_len) -> hipError_t {
        if (ascan) {
            // the skip table the scan walks: at its own tile size (tshift 0) or finer by one power of two (tshift 1)
            const bool use16 = shape == 1 && S.tile_docs != small_docs;
            const int32_t *tab = use16 ? S.tile_off16.as<int32_t>() : S.tile_off.as<int32_t>();
            const int n_tab = use16 ? S.n_tiles16 : S.n_tiles;
            const int tshift = (use16 || S.tile_docs == as_docs) ? 0 : 1;
            const int cut_mul = S.tile_docs > as_docs ? 2 : 1;                 // segment cuts on the exact scan's (larger) tiles
            hipError_t e = erh::launch_bm25_ascan(S.variant, shape, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
                                                  S.post.p, (h->opt_bm25_post16 && S.post16.p) ? S.post16.p : nullptr, S.g16,
                                                  (uint32_t)S.nnz, S.qmax,
                                                  tab, n_tab, tshift, S.Nb, qptr_dev, qtok_dev, q_order,
                                                  B, k, segs, cut_mul, filter_dev, dir, p_sc, p_ids, p_len, h->bm_redo.as<uint32_t>(),
                                                  h->dstats.as<unsigned long long>(),
                                                  (filter_dev && h->opt_bm25_dir_range && h->dir_rng_n > 0) ? h->dir_rng.as<int32_t>() : nullptr,
                                                  h->dir_rng_n, h->opt_bm25_ablate, dbg, st,
                                                  split_fin ? h->bm_fin_ids.as<int32_t>() : nullptr, split_fin ? h->bm_fin_cnt.as<int32_t>() : nullptr);
            if (e != hipSuccess) return e;
            // near-tie floods (rare): those workgroups are scanned again by the exact block scan (same document ranges per
            // segment: the cuts are expressed in the block scan's own tiles), the others exit at once
            int cut_tiles = tiles, cut_shift = 0;
            if (S.tile_docs < as_docs) cut_shift = 1;                          // block-scan tiles are half a scan tile
            else if (S.tile_docs > as_docs) cut_tiles = S.n_tiles;             // ... or two of them (cut_mul = 2 above)
            return erh::launch_bm25_scan(S.variant, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
                                         S.tile_off.as<int32_t>(), S.n_tiles, S.Nb, qptr_dev, qtok_dev, q_order, B, k, segs,
                                         filter_dev, dir, p_sc, p_ids, p_len, h->bm_redo.as<uint32_t>(), cut_tiles, cut_shift, 0,
                                         nullptr, st);
        }
        if (wscan)
            return erh::launch_bm25_wscan(S.variant, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
                                          S.fine_off.as<int32_t>(), S.n_fine, S.n_tiles, S.Nb, qptr_dev, qtok_dev, q_order, B, k,
                                          segs, filter_dev, dir, p_sc, p_ids, p_len,
                                          (S.payload_positive && (h->opt_bm25_crossing >= 2 ||
                                                                  (h->opt_bm25_crossing == 1 && S.variant != ERH_BM25_OKAPI))) ? 1 : 0,
                                          dbg, st);
        return erh::launch_bm25_scan(S.variant, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
                                     S.tile_off.as<int32_t>(), S.n_tiles, S.Nb, qptr_dev, qtok_dev, q_order, B, k, segs,
                                     filter_dev, dir, p_sc, p_ids, p_len, nullptr, 0, 0, h->opt_bm25_ablate, dbg, st);
    };
    if (segs == 1) {
        ProfScope ps(h, st, ERH_K_BM25_SCAN, postings_bytes, 0);
        HIPCHK(h, scan(d_sc, d_ids, d_len));
        return ERH_OK;
    }
    HIPCHK(h, h->part_sc.ensure((size_t)B * segs * k * 8));
    HIPCHK(h, h->part_ids.ensure((size_t)B * segs * k * 4));
    HIPCHK(h, h->part_len.ensure((size_t)B * segs * 4));
    { ProfScope ps(h, st, ERH_K_BM25_SCAN, postings_bytes, 0);
      HIPCHK(h, scan(h->part_sc.as<double>(), h->part_ids.as<int32_t>(), h->part_len.as<int32_t>())); }
    { ProfScope ps(h, st, ERH_K_BM25_MERGE, 0, 0);
      HIPCHK(h, erh::launch_bm25_merge(B, k, segs, h->part_sc.as<double>(), h->part_ids.as<int32_t>(),
                                       h->part_len.as<int32_t>(), d_ids, d_sc, d_len, st)); }
    return ERH_OK;
}

// Upload the query CSR; returns the algorithmic posting bytes of the batch in *bytes (0 if an id is bad -> error).
int upload_bm25_queries(erh_handle *h, const int32_t *q_indptr, const int32_t *q_tok, int B, hipStream_t st,
                        const std::vector<int64_t> &host_indptr, double *bytes, int *max_qlen) {
    if (q_indptr[0] != 0) return h->fail(ERH_ERR_INVALID, "q_indptr[0] must be 0");
    int longest = 0;
    for (int b = 0; b < B; ++b) {
        if (q_indptr[b + 1] < q_indptr[b]) return h->fail(ERH_ERR_INVALID, "q_indptr must be non-decreasing");
        longest = std::max(longest, q_indptr[b + 1] - q_indptr[b]);
    }
    *max_qlen = longest;
    const int nt = q_indptr[B];
    const size_t per = (h->bm[h->cur].variant == ERH_BM25_OKAPI) ? 12 : 8;
    double total = 0;
    for (int i = 0; i < nt; ++i) {
        const int32_t t = q_tok[i];
        if (t < 0 || t >= h->bm[h->cur].V) return h->fail(ERH_ERR_INVALID, "query term id out of range");
        total += (double)(host_indptr[t + 1] - host_indptr[t]) * per;
    }
    *bytes = total;
    // longest-processing-time-first order: a query's scan time follows its posting volume (60 k ... 300 k postings), the
    // dispatcher hands out workgroups in index order, and with four workgroups per CU the makespan is set by what starts last
    // ONE host-to-device copy for the call's query CSR and launch order (round 6: a pageable copy costs ~4 us of host time and ~7 us on the
    // stream whatever its size, and a single-query call was made of five of them): [q_indptr (B + 1) | q_tok (nt) | launch order (B)]
    h->qorder_valid = false;
    const bool lpt = h->opt_bm25_lpt && B > 1;
    const size_t off_tok = (size_t)(B + 1) * 4, off_ord = off_tok + (size_t)std::max(nt, 1) * 4, total_bytes = off_ord + (lpt ? (size_t)B * 4 : 0);
    h->qpack_host.resize(total_bytes);
    memcpy(h->qpack_host.data(), q_indptr, (size_t)(B + 1) * 4);
    if (nt) memcpy(h->qpack_host.data() + off_tok, q_tok, (size_t)nt * 4);
    if (lpt) {
        // longest-processing-time-first order: a query's scan time follows its posting volume (60 k ... 300 k postings), the
        // dispatcher hands out workgroups in index order, and with four workgroups per CU the makespan is set by what starts last
        std::vector<int64_t> cost((size_t)B, 0);
        for (int b = 0; b < B; ++b)
            for (int i = q_indptr[b]; i < q_indptr[b + 1]; ++i) cost[b] += host_indptr[q_tok[i] + 1] - host_indptr[q_tok[i]];
        h->qorder_host.resize((size_t)B);
        for (int b = 0; b < B; ++b) h->qorder_host[b] = b;
        std::stable_sort(h->qorder_host.begin(), h->qorder_host.end(), [&](int32_t x, int32_t y) { return cost[x] > cost[y]; });
        memcpy(h->qpack_host.data() + off_ord, h->qorder_host.data(), (size_t)B * 4);
    }
    HIPCHK(h, h->qpack.ensure(total_bytes));
    HIPCHK(h, hipMemcpyAsync(h->qpack.p, h->qpack_host.data(), total_bytes, hipMemcpyHostToDevice, st));
    h->qptr = h->qpack.as<int32_t>();
    h->qtok = reinterpret_cast<int32_t *>(h->qpack.as<char>() + off_tok);
    h->qorder = lpt ? reinterpret_cast<int32_t *>(h->qpack.as<char>() + off_ord) : nullptr;
    h->qorder_valid = lpt;
    return ERH_OK;
}

}  // namespace

extern "C" {

int erh_version(void) { return 200; }

const char *erh_status_str(int s) {
    switch (s) {
        case ERH_OK: return "ok";
        case ERH_ERR_INVALID: return "invalid argument";
        case ERH_ERR_NO_DEVICE: return "no usable gfx950 device";
        case ERH_ERR_HIP: return "HIP error";
        case ERH_ERR_STATE: return "state not set";
        case ERH_ERR_UNSUPPORTED: return "unsupported shape";
        case ERH_ERR_OVERFLOW: return "candidate overflow";
        case ERH_ERR_NOMEM: return "out of device memory";
        default: return "unknown status";
    }
}

int erh_comm_destroy(erh_handle *h);

int erh_create(int device, erh_handle **out) {
    if (!out) return ERH_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ERH_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ERH_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ERH_ERR_NO_DEVICE;   // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return ERH_ERR_NO_DEVICE;
    erh_handle *h = new (std::nothrow) erh_handle();
    if (!h) return ERH_ERR_NOMEM;
    h->device = device;
    h->n_cus = h->n_cus_dev = prop.multiProcessorCount;
    if (erh::dense_scan_init() != hipSuccess || erh::select_init() != hipSuccess || erh::bm25_init() != hipSuccess ||
        erh::fuse_init() != hipSuccess || erh::dense_gemv_init() != hipSuccess) {   // (function attributes are per device)
        delete h;
        return ERH_ERR_HIP;
    }
    if (h->dstats.ensure(64) != hipSuccess || hipMemset(h->dstats.p, 0, 64) != hipSuccess) {
        (void)erh_destroy(h);                              // (everything the handle owns so far, whatever that grows to)
        return ERH_ERR_NOMEM;
    }
    *out = h;
    return ERH_OK;
}

int erh_destroy(erh_handle *h) {
    if (!h) return ERH_ERR_INVALID;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    drain_events(h);
    for (auto &ev : h->pool) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    if (h->side) { (void)hipStreamDestroy(h->side); (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join); }
    DevBuf *bufs[] = {&h->X, &h->Xt, &h->Xt384, &h->Qt, &h->seed_top, &h->scan_sync, &h->content_id, &h->dir_id,
                      &h->qin, &h->Q16, &h->qnorm, &h->tau, &h->S0, &h->cand, &h->cand_cnt, &h->flags, &h->filt, &h->filt2,
                      &h->o_ids, &h->o_sc, &h->o_len, &h->qpack, &h->part_sc, &h->part_ids, &h->part_len,
                      &h->hy_sids, &h->hy_ssc, &h->hy_slen, &h->hy_dids, &h->hy_dsc, &h->hy_dlen,
                      &h->fa_ids, &h->fa_sc, &h->fa_len, &h->fb_ids, &h->fb_sc, &h->fb_len,
                      &h->scores_tmp, &h->scores_wide, &h->dbg, &h->dstats, &h->dir_pos, &h->seed_need, &h->bad, &h->ex_ws, &h->bm_redo, &h->fin_ws, &h->dir_rng, &h->Xb, &h->blk_tmp, &h->blk_ids, &h->r_q, &h->r_ids, &h->r_sc, &h->r_len, &h->r_flags, &h->r_tab, &h->r_q16, &h->bm_fin_ids, &h->bm_fin_cnt};
    for (DevBuf *b : bufs) b->release();
    if (h->r_flags_host) (void)hipHostFree(h->r_flags_host);
    for (auto &b : h->bm) b.release();
    if (h->comm || h->comm_pending) (void)erh_comm_destroy(h);
    h->gather_send.release();
    h->gather_recv.release();
    delete h;
    return ERH_OK;
}

const char *erh_last_error(erh_handle *h) { return h ? h->err.c_str() : "null handle"; }

int erh_sync(erh_handle *h, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    return ERH_OK;
}

int erh_set_option(erh_handle *h, const char *name, int64_t value) {
    if (!h || !name) return ERH_ERR_INVALID;
    if (!strcmp(name, "dense_n0")) { if (value < 1) return h->fail(ERH_ERR_INVALID, "dense_n0 < 1"); h->opt_n0 = value; return ERH_OK; }
    if (!strcmp(name, "dense_n1")) { if (value < 0) return h->fail(ERH_ERR_INVALID, "dense_n1 < 0"); h->opt_n1 = value; return ERH_OK; }
    if (!strcmp(name, "dense_cfg")) { if (value < 0 || value > 2) return h->fail(ERH_ERR_INVALID, "dense_cfg"); h->opt_dense_cfg = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_readahead")) { h->opt_dense_readahead = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_shuffle")) { h->opt_dense_shuffle = value != 0; return ERH_OK; }   // takes effect at the next erh_set_dense
    if (!strcmp(name, "dense_n1_auto")) { h->opt_n1_auto = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_pp")) { if (value < 0 || value > 4) return h->fail(ERH_ERR_INVALID, "dense_pp"); h->opt_dense_pp = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_n0_auto")) { h->opt_n0_auto = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_sync")) { h->opt_dense_sync = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_selfseed")) { if (value < 0 || value > 2) return h->fail(ERH_ERR_INVALID, "dense_selfseed"); h->opt_dense_selfseed = (int)value; return ERH_OK; }
    if (!strcmp(name, "n_cus")) {          // persistent grids: the CUs the caller's stream may use (a CU-masked stream); 0 = all of the device
        if (value < 0 || value > h->n_cus_dev) return h->fail(ERH_ERR_INVALID, "n_cus");
        h->n_cus = value == 0 ? h->n_cus_dev : (int)value;
        return ERH_OK;
    }
    if (!strcmp(name, "bm25_long_tokens")) { if (value < 0 || value > 4096) return h->fail(ERH_ERR_INVALID, "bm25_long_tokens"); h->opt_bm25_long_tokens = (int)value; return ERH_OK; }
    if (!strcmp(name, "bm25_split_finish")) { h->opt_bm25_split_finish = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_route_ridge")) { if (value < 1 || value > 4096) return h->fail(ERH_ERR_INVALID, "dense_route_ridge"); h->opt_route_ridge = value; return ERH_OK; }
    if (!strcmp(name, "dense_group_sample")) { h->opt_dense_group_sample = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_group_launch")) { h->opt_dense_group_launch = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_dir_blocks")) { if (value < 0 || value > 2) return h->fail(ERH_ERR_INVALID, "dense_dir_blocks"); h->opt_dense_dir_blocks = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_dir_block_min_rows")) { if (value < 1) return h->fail(ERH_ERR_INVALID, "dense_dir_block_min_rows"); h->opt_dir_block_min_rows = value; h->blocks.valid = false; return ERH_OK; }
    if (!strcmp(name, "bm25_dir_range")) { h->opt_bm25_dir_range = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_fin_split")) { h->opt_dense_fin_split = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_tile384")) { h->opt_dense_tile384 = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_tile384_max_mb")) { h->opt_tile384_max_mb = value; h->xt384_nomem = false; return ERH_OK; }
    if (!strcmp(name, "dense_tiled")) { h->opt_dense_tiled = value != 0; return ERH_OK; }   // building the copy: at the next erh_set_dense
    if (!strcmp(name, "dense_speculate")) { h->opt_dense_speculate = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_var")) { if (value < 0 || value > 3) return h->fail(ERH_ERR_INVALID, "dense_var"); h->opt_dense_var = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_rot")) { if (value < -1 || value > 4096) return h->fail(ERH_ERR_INVALID, "dense_rot"); h->opt_dense_rot = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_gemv")) { h->opt_dense_gemv = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_gemv_kb")) { if (value != 16 && value != 32) return h->fail(ERH_ERR_INVALID, "dense_gemv_kb"); h->opt_gemv_kb = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_gemv_pipe")) { if (value < -1 || value > 1) return h->fail(ERH_ERR_INVALID, "dense_gemv_pipe"); h->opt_gemv_pipe = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_gemv_wgs")) { if (value < 1 || value > 5) return h->fail(ERH_ERR_INVALID, "dense_gemv_wgs"); h->opt_gemv_wgs = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_small_single_stage")) { h->opt_small_single = value != 0; return ERH_OK; }
    if (!strcmp(name, "dense_persist")) { h->opt_dense_persist = value != 0; return ERH_OK; }
#ifdef ERH_MEASURE
    if (!strcmp(name, "dense_ablate")) { h->opt_dense_ablate = (int)value; return ERH_OK; }
    if (!strcmp(name, "bm25_ablate")) { h->opt_bm25_ablate = (int)value; return ERH_OK; }
#else
    if (!strcmp(name, "dense_ablate") || !strcmp(name, "bm25_ablate") || !strcmp(name, "debug_counters"))
        return value == 0 ? ERH_OK : h->fail(ERH_ERR_UNSUPPORTED, "measurement option: rebuild the library with ERH_MEASURE=1");
#endif
    if (!strcmp(name, "comm_timeout_s")) { if (value < 1 || value > 86400) return h->fail(ERH_ERR_INVALID, "comm_timeout_s"); h->opt_comm_timeout_s = (int)value; return ERH_OK; }
    if (!strcmp(name, "bm25_lpt")) { h->opt_bm25_lpt = value != 0; return ERH_OK; }
    if (!strcmp(name, "bm25_segs")) { if (value < 0 || value > 64) return h->fail(ERH_ERR_INVALID, "bm25_segs"); h->opt_bm25_segs = (int)value; return ERH_OK; }
    if (!strcmp(name, "bm25_crossing")) { if (value < 0 || value > 2) return h->fail(ERH_ERR_INVALID, "bm25_crossing"); h->opt_bm25_crossing = (int)value; return ERH_OK; }
    if (!strcmp(name, "bm25_ascan")) { h->opt_bm25_ascan = value != 0; return ERH_OK; }
    if (!strcmp(name, "bm25_post16")) { h->opt_bm25_post16 = value != 0; return ERH_OK; }
    if (!strcmp(name, "bm25_small")) { h->opt_bm25_small = value < 0 ? 0 : value > 2 ? 2 : (int)value; return ERH_OK; }
    if (!strcmp(name, "hybrid_overlap")) { if (value < -1 || value > 2) return h->fail(ERH_ERR_INVALID, "hybrid_overlap"); h->opt_hybrid_overlap = (int)value; return ERH_OK; }
    if (!strcmp(name, "bm25_wscan")) { h->opt_bm25_wscan = value != 0; return ERH_OK; }   // the fine table is built at the next erh_set_bm25_*
    if (!strcmp(name, "bm25_fine_max_mb")) { if (value < 0) return h->fail(ERH_ERR_INVALID, "bm25_fine_max_mb < 0"); h->opt_bm25_fine_max_mb = value; return ERH_OK; }
    if (!strcmp(name, "debug_counters")) {
        h->opt_debug_counters = value != 0;
        if (value) {
            HIPCHK(h, hipSetDevice(h->device));
            HIPCHK(h, h->dbg.ensure(16 * 8));
            HIPCHK(h, hipMemset(h->dbg.p, 0, 16 * 8));
        }
        return ERH_OK;
    }
    return h->fail(ERH_ERR_INVALID, "unknown option");
}

int erh_set_profiling(erh_handle *h, int enable) {
    if (!h) return ERH_ERR_INVALID;
    h->prof = enable != 0;
    return ERH_OK;
}

int erh_get_kernel_time(erh_handle *h, int cls, double *total_ms, int64_t *launches) {
    if (!h || cls < 0 || cls >= ERH_K_COUNT) return ERH_ERR_INVALID;
    (void)hipSetDevice(h->device);
    drain_events(h);
    if (total_ms) *total_ms = h->ms[cls];
    if (launches) *launches = h->launches[cls];
    return ERH_OK;
}

int erh_get_kernel_work(erh_handle *h, int cls, double *bytes, double *flops) {
    if (!h || cls < 0 || cls >= ERH_K_COUNT) return ERH_ERR_INVALID;
    if (bytes) *bytes = h->work_bytes[cls];
    if (flops) *flops = h->work_flops[cls];
    return ERH_OK;
}

int erh_reset_kernel_time(erh_handle *h) {
    if (!h) return ERH_ERR_INVALID;
    (void)hipSetDevice(h->device);
    drain_events(h);
    for (int i = 0; i < ERH_K_COUNT; ++i) { h->ms[i] = 0; h->launches[i] = 0; h->work_bytes[i] = 0; h->work_flops[i] = 0; }
    return ERH_OK;
}

int erh_dense_check(erh_handle *h, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!h->flags.p || (!h->last.valid && !h->routed.done)) return ERH_OK;           // no dense route has run on this handle
    HIPCHK(h, hipSetDevice(h->device));
    return dense_check_flags(h, (hipStream_t)stream);
}

int erh_debug_counters(erh_handle *h, uint64_t *out16) {
    if (!h || !out16) return ERH_ERR_INVALID;
    if (!h->dbg.p) { memset(out16, 0, 16 * 8); return ERH_OK; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out16, h->dbg.p, 16 * 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemset(h->dbg.p, 0, 16 * 8));
    return ERH_OK;
}

int erh_dense_diag(erh_handle *h, double *max_abs_err, double *margin, int32_t *uncertified) {
    if (!h) return ERH_ERR_INVALID;
    if (max_abs_err) *max_abs_err = h->diag_maxerr;
    if (margin) *margin = h->diag_margin;
    if (uncertified) *uncertified = h->diag_uncert;
    return ERH_OK;
}

int erh_dense_seed_rank(int k, int64_t n0, int64_t n) {
    if (k < 1) return k;
    if (n0 < 1 || n <= n0) return k;
    const double mu = (double)k * (double)n0 / (double)n;
    const double r = std::ceil(mu + 6.5 * std::sqrt(mu) + 3.0);
    return r < (double)k ? (int)r : k;
}

int erh_get_stat(erh_handle *h, const char *name, int64_t *value) {
    if (!h || !name || !value) return ERH_ERR_INVALID;
    const erh_handle::Stats &T = h->stats;
    const struct { const char *n; int64_t v; } host[] = {
        {"dense_calls", T.dense_calls}, {"dense_scan_pp5_launches", T.dense_scan_pp5}, {"dense_scan_pp3_launches", T.dense_scan_pp3},
        {"dense_scan_gemv_launches", T.dense_scan_gemv}, {"dense_scan_tile_launches", T.dense_scan_tile},
        {"dense_sample_passes", T.dense_sample_passes}, {"dense_tile384_nomem", T.dense_tile384_nomem},
        {"bm25_calls", T.bm25_calls}, {"hybrid_calls", T.hybrid_calls}, {"dense_block_groups", T.dense_block_groups},
        {"dense_grouped_launches", T.dense_grouped_launches}};
    for (const auto &e : host)
        if (!strcmp(name, e.n)) { *value = e.v; return ERH_OK; }
    if (!strcmp(name, "dense_candidates_last_call")) {
        // candidates the scan of the LAST dense pipeline handed to its final kernel, summed over its queries (a routed call: its last
        // pipeline -- the grouped launch when there was one): how selective the pruning threshold was on this data
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipDeviceSynchronize());
        const int rows = h->cand_rows;
        std::vector<uint32_t> c((size_t)std::max(rows, 0));
        if (rows > 0) HIPCHK(h, hipMemcpy(c.data(), h->cand_cnt.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
        int64_t sum = 0;
        for (uint32_t v : c) sum += std::min<uint32_t>(v, (uint32_t)erh::kDenseCapMax);
        *value = sum;
        return ERH_OK;
    }
    const int di = !strcmp(name, "dense_exhaustive_queries") ? 0 : !strcmp(name, "bm25_redo_segments") ? 1 : -1;
    if (di < 0) return h->fail(ERH_ERR_INVALID, "erh_get_stat: unknown counter");
    unsigned long long v[2] = {0, 0};
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());                       // the counters of everything enqueued so far
    HIPCHK(h, hipMemcpy(v, h->dstats.p, sizeof v, hipMemcpyDeviceToHost));
    *value = (int64_t)v[di];
    return ERH_OK;
}

int erh_reset_stats(erh_handle *h) {
    if (!h) return ERH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemset(h->dstats.p, 0, 64));
    h->stats = erh_handle::Stats();
    return ERH_OK;
}

int erh_dense_exhaustive_count(erh_handle *h, int32_t *count) {
    if (!h || !count) return ERH_ERR_INVALID;
    *count = h->diag_exhaustive;
    return ERH_OK;
}

// ---- corpus state ------------------------------------------------------------------------------------

int erh_set_dense(erh_handle *h, const void *x, int64_t n, int d, int dtype, int is_device_ptr, int normalize) {
    if (!h) return ERH_ERR_INVALID;
    if (!x || n <= 0 || d <= 0) return h->fail(ERH_ERR_INVALID, "erh_set_dense: null matrix or non-positive shape");
    if (d % 64 != 0) return h->fail(ERH_ERR_UNSUPPORTED, "erh_set_dense: d must be a multiple of 64");
    if (n > 2147483647LL) return h->fail(ERH_ERR_UNSUPPORTED, "erh_set_dense: n must fit int32 document ids");
    if (dtype != ERH_F16 && dtype != ERH_F32) return h->fail(ERH_ERR_INVALID, "erh_set_dense: dtype");
    if (dtype == ERH_F16 && normalize)
        return h->fail(ERH_ERR_UNSUPPORTED, "erh_set_dense: normalize=1 needs fp32 rows (fp16 rows are taken as stored)");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = nullptr;
    hipStream_t st_pad = nullptr;
    // the 384-row copy of the OLD matrix goes first (it is rebuilt on first use): it must not sit beside the old and the new X
    h->xt384_valid = false;
    h->xt384_nomem = false;
    h->Xt384.release();
    h->blocks.valid = false;
    h->Xb.release();
    h->blk_ids.release();
    HIPCHK(h, h->X.ensure((size_t)(n + erh::kDensePadRows) * d * 2));   // zero rows behind the matrix: tiles may run past N
    HIPCHK(h, hipMemsetAsync(h->X.as<char>() + (size_t)n * d * 2, 0, (size_t)erh::kDensePadRows * d * 2, st_pad));
    const hipMemcpyKind kind = is_device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    int64_t mul = 1, inv = 1;
    if (h->opt_dense_shuffle && n > 2) choose_placement(n, &mul, &inv);
    const size_t es = (dtype == ERH_F16) ? 2 : 4;
    if (dtype == ERH_F16 && mul == 1) {
        HIPCHK(h, hipMemcpyAsync(h->X.p, x, (size_t)n * d * 2, kind, st));
    } else if (dtype == ERH_F16 && is_device_ptr) {
        HIPCHK(h, erh::launch_permute_rows((const _Float16 *)x, n, d, h->X.as<_Float16>(), 0, mul, n, st));
    } else {
        // host rows (or fp32 rows to convert) go through the device in slabs, so that a 1M x 1024 fp32 host matrix
        // never needs 4 GB of staging; the slab kernels write every row at its stored position
        const int64_t slab = std::max<int64_t>(1, (int64_t)(256u << 20) / ((int64_t)d * (int64_t)es));
        const void *src = x;
        for (int64_t r0 = 0; r0 < n; r0 += slab) {
            const int64_t rows = std::min<int64_t>(slab, n - r0);
            const char *from = (const char *)x + (size_t)r0 * d * es;
            if (!is_device_ptr) {
                HIPCHK(h, h->qin.ensure((size_t)std::min<int64_t>(slab, n) * d * es));
                HIPCHK(h, hipMemcpyAsync(h->qin.p, from, (size_t)rows * d * es, kind, st));
                src = h->qin.p;
            } else {
                src = from;
            }
            if (dtype == ERH_F16)
                HIPCHK(h, erh::launch_permute_rows((const _Float16 *)src, rows, d, h->X.as<_Float16>(), r0, mul, n, st));
            else
                HIPCHK(h, erh::launch_convert_rows((const float *)src, rows, d, normalize, h->X.as<_Float16>(), r0, mul, n, st));
            HIPCHK(h, hipStreamSynchronize(st));
        }
    }
    h->pos_mul = mul;
    h->pos_inv = inv;
    h->dir_pos_valid = false;
    HIPCHK(h, h->flags.ensure(64));
    HIPCHK(h, erh::launch_row_norm_max(h->X.as<_Float16>(), n, d, reinterpret_cast<float *>(h->flags.p), st));
    float xn = 0.f;
    HIPCHK(h, hipMemcpyAsync(&xn, h->flags.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    h->xnorm_max = xn;
    h->N = n;
    h->d = d;
    h->view_global();
    // tiled copy for the ping-pong scan (dense_scan.hip: dense_tile_rows_kernel); d / 32 >= 8 stages as the kernel wants
    h->xt384_valid = false;                // (the 384-row copy is rebuilt on first use)
    h->xt_valid = false;
    if (h->opt_dense_tiled && d % 64 == 0 && d >= 256) {
        const int64_t n_tiles = (n + 255) / 256;
        HIPCHK(h, h->Xt.ensure((size_t)n_tiles * 256 * (size_t)d * 2));
        HIPCHK(h, erh::launch_dense_tile_rows(h->X.as<_Float16>(), n, d, h->Xt.p, st));
        HIPCHK(h, hipStreamSynchronize(st));
        h->xt_valid = true;
    } else {
        h->Xt.release();
    }
    return ERH_OK;
}

int erh_get_dense_rows(erh_handle *h, int64_t row0, int64_t rows, void *out_f16, int out_is_device) {
    if (!h) return ERH_ERR_INVALID;
    if (!h->X.p || h->N <= 0) return h->fail(ERH_ERR_STATE, "erh_get_dense_rows before erh_set_dense");
    if (!out_f16 || rows <= 0 || row0 < 0 || row0 + rows > h->N) return h->fail(ERH_ERR_INVALID, "erh_get_dense_rows: bad range");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = nullptr;
    const int d = h->d;
    // the caller's rows live at their golden-ratio positions: gather them back into the caller's order
    _Float16 *dst = reinterpret_cast<_Float16 *>(out_f16);
    if (!out_is_device) {
        HIPCHK(h, h->scores_tmp.ensure((size_t)rows * d * 2));
        dst = h->scores_tmp.as<_Float16>();
    }
    HIPCHK(h, erh::launch_gather_rows(h->X.as<_Float16>(), nullptr, row0, rows, d, h->pos_mul, h->N, dst, st));
    if (!out_is_device) HIPCHK(h, hipMemcpyAsync(out_f16, dst, (size_t)rows * d * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return ERH_OK;
}

// Skip tables over device-resident CSR postings (indptr / doc_ids of the selected slot): the 32768 / 16384-document
// tile table of the block scan and the fine table of the wave-owned scan.
static int bm25_finish_tables(erh_handle *h, int variant, int64_t V, int64_t N, hipStream_t st) {
    Bm25State &S = h->bm[h->cur];
    S.tile_docs = (variant == ERH_BM25_OKAPI) ? erh::kBm25TileF64 : erh::kBm25TileF32;
    S.n_tiles = (int)((N + S.tile_docs - 1) / S.tile_docs);
    HIPCHK(h, S.tile_off.ensure((size_t)V * (S.n_tiles + 1) * 4));
    HIPCHK(h, erh::launch_bm25_tile_off(S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), V, S.tile_docs, S.n_tiles,
                                        S.tile_off.as<int32_t>(), st));
    S.tile_off16.release();
    S.n_tiles16 = 0;
    if (S.tile_docs != erh::bm25_ascan_tile_docs(1) && h->opt_bm25_ascan && h->opt_bm25_small) {
        const int td = erh::bm25_ascan_tile_docs(1);
        S.n_tiles16 = (int)((N + td - 1) / td);
        HIPCHK(h, S.tile_off16.ensure((size_t)V * (S.n_tiles16 + 1) * 4));
        HIPCHK(h, erh::launch_bm25_tile_off(S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), V, td, S.n_tiles16,
                                            S.tile_off16.as<int32_t>(), st));
    }
    // fine skip table of the wave-owned scan: one int per (term, sub-range of tile_docs / 16 documents)
    S.n_fine = 0;
    const int sub = erh::bm25_wscan_sub_docs(variant);
    const int64_t nf = (N + sub - 1) / sub;
    const double mb = (double)V * (double)(nf + 1) * 4.0 / (1024.0 * 1024.0);
    if (h->opt_bm25_wscan && nf < (1 << 30) && mb <= (double)h->opt_bm25_fine_max_mb) {
        HIPCHK(h, S.fine_off.ensure((size_t)V * (size_t)(nf + 1) * 4));
        HIPCHK(h, erh::launch_bm25_tile_off(S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), V, sub, (int)nf,
                                            S.fine_off.as<int32_t>(), st));
        S.n_fine = (int)nf;
    } else {
        S.fine_off.release();
    }
    return ERH_OK;
}

// After the payload of the selected slot is in place: does every posting carry a payload > 0?
static int bm25_check_payload_sign(erh_handle *h, hipStream_t st) {
    Bm25State &S = h->bm[h->cur];
    S.payload_positive = false;
    if (S.nnz <= 0) return ERH_OK;
    HIPCHK(h, h->flags.ensure(64));
    uint32_t *w = h->flags.as<uint32_t>() + 8;
    HIPCHK(h, hipMemsetAsync(w, 0, 4, st));
    HIPCHK(h, erh::launch_bm25_payload_sign(S.variant, S.payload.p, S.nnz, w, st));
    uint32_t f = 1;
    HIPCHK(h, hipMemcpyAsync(&f, w, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    S.payload_positive = (f == 0);
    // fixed-point scan: an interleaved copy of the postings with the payload as trunc(p32 * 2^S) + 1.  bm25s payloads are
    // fp32 already; Okapi goes through an fp32 copy, which must be positive and normal as well (an fp64 payload below
    // 1.2e-38 would round to a subnormal or to zero).  8 bytes per posting on top of the index.
    S.ascan_ok = false;
    S.post.release();
    S.post16.release();
    // (a throw-away index of a handful of sentences -- BM25Retriever.get_scores(query, docs) -- is scanned by the block scan:
    // building the fixed-point copy would cost an allocation and three stream synchronisations per call)
    if (S.payload_positive && h->opt_bm25_ascan && S.nnz >= 2048 && S.nnz < (1LL << 28)) {         // (32-bit byte offsets into post[])
        DevBuf p32buf;
        const float *p32 = S.payload.as<float>();
        bool ok = true;
        hipError_t e = hipSuccess;
        if (S.variant == ERH_BM25_OKAPI) {
            e = p32buf.ensure((size_t)S.nnz * 4);
            if (e == hipSuccess) e = erh::launch_narrow_f64(S.payload.as<double>(), S.nnz, p32buf.as<float>(), st);
            if (e == hipSuccess) e = hipMemsetAsync(w, 0, 4, st);
            if (e == hipSuccess) e = erh::launch_bm25_payload_sign(ERH_BM25_BM25S, p32buf.p, S.nnz, w, st);
            f = 1;
            if (e == hipSuccess) e = hipMemcpyAsync(&f, w, 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            ok = (e == hipSuccess && f == 0);
            p32 = p32buf.as<float>();
        }
        float pmax = 0.f;
        if (ok) {
            e = hipMemsetAsync(w, 0, 4, st);
            if (e == hipSuccess) e = erh::launch_bm25_payload_max(p32, S.nnz, w, st);
            if (e == hipSuccess) e = hipMemcpyAsync(&f, w, 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            memcpy(&pmax, &f, 4);
            ok = e == hipSuccess && pmax > 0.f && std::isfinite(pmax);
        }
        if (ok) {
            const float scale = erh::bm25_post_scale(pmax);
            ok = scale >= 4096.f;                                                // a coarser grid than 2^-12 is not worth scanning
            if (ok) {
                e = S.post.ensure((size_t)(S.nnz + 2) * 8);            // + two sentinel postings (one 16-byte load)
                if (e == hipSuccess) e = erh::launch_bm25_post(S.doc_ids.as<int32_t>(), p32, S.nnz, scale, S.post.p, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                S.qmax = std::floor((double)pmax * (double)scale) + 1.0;
                ok = e == hipSuccess;
                if (ok && h->opt_bm25_small == 2 && h->opt_bm25_post16) {         // the packed shape's 4-byte postings
                    S.g16 = erh::bm25_post16_shift(S.qmax);
                    e = S.post16.ensure((size_t)(S.nnz + 8) * 4);
                    if (e == hipSuccess) e = erh::launch_bm25_post16(S.post.p, S.nnz, S.g16, S.post16.p, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    ok = e == hipSuccess;
                }
            }
        }
        p32buf.release();
        if (e != hipSuccess) { S.post.release(); S.post16.release(); return h->fail(e == hipErrorOutOfMemory ? ERH_ERR_NOMEM : ERH_ERR_HIP, "bm25 fixed-point postings", e); }
        S.ascan_ok = ok;
        if (!ok) { S.post.release(); S.post16.release(); }
    }
    return ERH_OK;
}

static int bm25_common_upload(erh_handle *h, int variant, int64_t V, int64_t N, int64_t nnz,
                              const int64_t *indptr, const int32_t *doc_ids) {
    if (variant != ERH_BM25_OKAPI && variant != ERH_BM25_BM25S) return h->fail(ERH_ERR_INVALID, "bm25 variant");
    if (V <= 0 || N <= 0 || nnz < 0 || !indptr || (nnz > 0 && !doc_ids)) return h->fail(ERH_ERR_INVALID, "bm25 csr: null or non-positive shape");
    if (N > 2147483647LL) return h->fail(ERH_ERR_UNSUPPORTED, "bm25 csr: N must fit int32 document ids");
    if (indptr[0] != 0 || indptr[V] != nnz) return h->fail(ERH_ERR_INVALID, "bm25 csr: indptr[0] != 0 or indptr[V] != nnz");
    for (int64_t t = 0; t < V; ++t) {
        if (indptr[t + 1] < indptr[t]) return h->fail(ERH_ERR_INVALID, "bm25 csr: indptr must be non-decreasing");
        for (int64_t p = indptr[t]; p < indptr[t + 1]; ++p) {
            const int32_t dd = doc_ids[p];
            if (dd < 0 || dd >= N) return h->fail(ERH_ERR_INVALID, "bm25 csr: document id out of range");
            if (p > indptr[t] && doc_ids[p - 1] >= dd)
                return h->fail(ERH_ERR_INVALID, "bm25 csr: document ids must be strictly ascending inside a term");
        }
    }
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = nullptr;
    HIPCHK(h, h->bm[h->cur].indptr.ensure((size_t)(V + 1) * 8));
    HIPCHK(h, h->bm[h->cur].doc_ids.ensure((size_t)(nnz + 1) * 4));          // + the sentinel posting of the fixed-point scan
    HIPCHK(h, hipMemcpyAsync(h->bm[h->cur].indptr.p, indptr, (size_t)(V + 1) * 8, hipMemcpyHostToDevice, st));
    if (nnz) HIPCHK(h, hipMemcpyAsync(h->bm[h->cur].doc_ids.p, doc_ids, (size_t)nnz * 4, hipMemcpyHostToDevice, st));
    int rc_t = bm25_finish_tables(h, variant, V, N, st);
    if (rc_t != ERH_OK) return rc_t;
    h->bm[h->cur].host_indptr.assign(indptr, indptr + V + 1);
    h->bm[h->cur].variant = variant;
    h->bm[h->cur].V = V;
    h->bm[h->cur].Nb = N;
    h->bm[h->cur].nnz = nnz;
    return ERH_OK;
}

int erh_set_bm25_csr(erh_handle *h, int variant, int64_t V, int64_t N, int64_t nnz,
                     const int64_t *indptr, const int32_t *doc_ids, const void *payload) {
    if (!h) return ERH_ERR_INVALID;
    if (nnz > 0 && !payload) return h->fail(ERH_ERR_INVALID, "bm25 csr: null payload");
    h->bm[h->cur].variant = -1;
    int rc = bm25_common_upload(h, variant, V, N, nnz, indptr, doc_ids);
    if (rc != ERH_OK) { h->bm[h->cur].variant = -1; return rc; }
    const size_t es = (variant == ERH_BM25_OKAPI) ? 8 : 4;
    HIPCHK(h, h->bm[h->cur].payload.ensure((size_t)(nnz + 1) * es));
    if (nnz) HIPCHK(h, hipMemcpyAsync(h->bm[h->cur].payload.p, payload, (size_t)nnz * es, hipMemcpyHostToDevice, nullptr));
    HIPCHK(h, hipStreamSynchronize(nullptr));
    return bm25_check_payload_sign(h, nullptr);
}

int erh_set_bm25_tf(erh_handle *h, int variant, int64_t V, int64_t N, int64_t nnz,
                    const int64_t *indptr, const int32_t *doc_ids, const int32_t *tf,
                    const int32_t *doc_len, const void *idf, double avgdl, double k1, double b) {
    if (!h) return ERH_ERR_INVALID;
    if (!tf || !doc_len || !idf || !(avgdl > 0)) return h->fail(ERH_ERR_INVALID, "bm25 tf: null input or avgdl <= 0");
    h->bm[h->cur].variant = -1;
    int rc = bm25_common_upload(h, variant, V, N, nnz, indptr, doc_ids);
    if (rc != ERH_OK) { h->bm[h->cur].variant = -1; return rc; }
    const size_t es = (variant == ERH_BM25_OKAPI) ? 8 : 4;
    hipStream_t st = nullptr;
    HIPCHK(h, h->bm[h->cur].payload.ensure((size_t)(nnz + 1) * es));
    DevBuf d_tf, d_dl, d_idf;
    auto cleanup = [&]() { d_tf.release(); d_dl.release(); d_idf.release(); };
    hipError_t e = d_tf.ensure((size_t)std::max<int64_t>(nnz, 1) * 4);
    if (e == hipSuccess) e = d_dl.ensure((size_t)N * 4);
    if (e == hipSuccess) e = d_idf.ensure((size_t)V * es);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(d_tf.p, tf, (size_t)nnz * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dl.p, doc_len, (size_t)N * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idf.p, idf, (size_t)V * es, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = erh::launch_bm25_payload(variant, V, nnz, h->bm[h->cur].indptr.as<int64_t>(), h->bm[h->cur].doc_ids.as<int32_t>(), d_tf.as<int32_t>(),
                                     d_dl.as<int32_t>(), d_idf.p, avgdl, k1, b, h->bm[h->cur].payload.p, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) { h->bm[h->cur].variant = -1; return h->fail(ERH_ERR_HIP, "erh_set_bm25_tf", e); }
    return bm25_check_payload_sign(h, st);
}


// ---- index build on the device (first half) --------------------------------------------------
int erh_build_bm25_index(erh_handle *h, int variant, int64_t V, int64_t N, int64_t n_tokens, const int32_t *token_ids,
                         const int32_t *doc_len, int is_device_ptr, double k1, double b, double epsilon,
                         int64_t *out_nnz) {
    if (!h) return ERH_ERR_INVALID;
    if (variant != ERH_BM25_OKAPI && variant != ERH_BM25_BM25S) return h->fail(ERH_ERR_INVALID, "bm25 variant");
    if (V <= 0 || N <= 0 || n_tokens < 0 || !doc_len || (n_tokens > 0 && !token_ids))
        return h->fail(ERH_ERR_INVALID, "erh_build_bm25_index: null pointer or non-positive shape");
    if (N > 2147483647LL || V > 2147483647LL || n_tokens > 2147483647LL)
        return h->fail(ERH_ERR_UNSUPPORTED, "erh_build_bm25_index: N, V and the token count must fit int32");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = nullptr;
    Bm25State &S = h->bm[h->cur];
    S.variant = -1;
    S.built_on_device = false;
    const int64_t T = n_tokens;
    // document lengths: host copy (offsets, avgdl) + device copy (payload kernel)
    std::vector<int32_t> dl((size_t)N);
    if (is_device_ptr) HIPCHK(h, hipMemcpy(dl.data(), doc_len, (size_t)N * 4, hipMemcpyDeviceToHost));
    else memcpy(dl.data(), doc_len, (size_t)N * 4);
    std::vector<int64_t> off((size_t)N + 1);
    off[0] = 0;
    for (int64_t i = 0; i < N; ++i) {
        if (dl[i] < 0) return h->fail(ERH_ERR_INVALID, "erh_build_bm25_index: negative document length");
        off[i + 1] = off[i] + dl[i];
    }
    if (off[N] != T) return h->fail(ERH_ERR_INVALID, "erh_build_bm25_index: document lengths do not sum to the token count");
    DevBuf d_tok, d_off, d_keys, d_sorted, d_uniq, d_cnt, d_first, d_misc, d_temp, d_df, d_dl, d_idf;
    auto cleanup = [&]() {
        for (DevBuf *x : {&d_tok, &d_off, &d_keys, &d_sorted, &d_uniq, &d_cnt, &d_first, &d_misc, &d_temp, &d_df, &d_dl, &d_idf})
            x->release();
    };
#define BUILD_CHK(call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) { cleanup(); return h->fail(e_ == hipErrorOutOfMemory ? ERH_ERR_NOMEM : ERH_ERR_HIP, #call, e_); } \
    } while (0)
    const size_t Tn = (size_t)std::max<int64_t>(T, 1);
    const int32_t *tok_dev = token_ids;
    if (!is_device_ptr) {
        BUILD_CHK(d_tok.ensure(Tn * 4));
        if (T) BUILD_CHK(hipMemcpyAsync(d_tok.p, token_ids, (size_t)T * 4, hipMemcpyHostToDevice, st));
        tok_dev = d_tok.as<int32_t>();
    }
    BUILD_CHK(d_off.ensure((size_t)(N + 1) * 8));
    BUILD_CHK(hipMemcpyAsync(d_off.p, off.data(), (size_t)(N + 1) * 8, hipMemcpyHostToDevice, st));
    BUILD_CHK(d_keys.ensure(Tn * 8));
    BUILD_CHK(d_sorted.ensure(Tn * 8));
    BUILD_CHK(d_uniq.ensure(Tn * 8));
    BUILD_CHK(d_cnt.ensure(Tn * 4));
    BUILD_CHK(d_first.ensure((size_t)V * 8));
    BUILD_CHK(d_misc.ensure(64));
    BUILD_CHK(hipMemsetAsync(d_misc.p, 0, 64, st));
    uint32_t *bad_tok = d_misc.as<uint32_t>();
    int32_t *num_runs = d_misc.as<int32_t>() + 4;
    BUILD_CHK(erh::launch_csr_keys(tok_dev, d_off.as<int64_t>(), T, N, V, d_keys.as<uint64_t>(),
                                   d_first.as<unsigned long long>(), bad_tok, st));
    int key_bits = 32;
    while (key_bits < 64 && (1ll << (key_bits - 32)) < V) ++key_bits;
    int64_t nnz = 0;
    if (T > 0) {
        size_t tb = 0;
        BUILD_CHK(erh::csr_sort_rle(d_keys.as<uint64_t>(), d_sorted.as<uint64_t>(), T, key_bits, d_uniq.as<uint64_t>(),
                                    d_cnt.as<int32_t>(), num_runs, nullptr, &tb, st));
        BUILD_CHK(d_temp.ensure(tb + 256));
        tb = d_temp.cap;
        BUILD_CHK(erh::csr_sort_rle(d_keys.as<uint64_t>(), d_sorted.as<uint64_t>(), T, key_bits, d_uniq.as<uint64_t>(),
                                    d_cnt.as<int32_t>(), num_runs, d_temp.p, &tb, st));
    }
    uint32_t misc[8] = {0};
    BUILD_CHK(hipMemcpyAsync(misc, d_misc.p, sizeof misc, hipMemcpyDeviceToHost, st));
    BUILD_CHK(hipStreamSynchronize(st));
    if (misc[0]) { cleanup(); return h->fail(ERH_ERR_INVALID, "erh_build_bm25_index: token id out of range"); }
    nnz = T > 0 ? (int64_t)(int32_t)misc[4] : 0;
    d_keys.release();
    d_sorted.release();
    d_temp.release();
    BUILD_CHK(S.doc_ids.ensure((size_t)(nnz + 1) * 4));
    BUILD_CHK(S.tf.ensure((size_t)std::max<int64_t>(nnz, 1) * 4));
    BUILD_CHK(d_df.ensure((size_t)V * 8));
    BUILD_CHK(erh::launch_csr_split(d_uniq.as<uint64_t>(), d_cnt.as<int32_t>(), nnz, V, S.doc_ids.as<int32_t>(),
                                    S.tf.as<int32_t>(), d_df.as<unsigned long long>(), st));
    std::vector<unsigned long long> df((size_t)V), first((size_t)V);
    BUILD_CHK(hipMemcpyAsync(df.data(), d_df.p, (size_t)V * 8, hipMemcpyDeviceToHost, st));
    BUILD_CHK(hipMemcpyAsync(first.data(), d_first.p, (size_t)V * 8, hipMemcpyDeviceToHost, st));
    BUILD_CHK(hipStreamSynchronize(st));
    // ---- V-sized host work: indptr, idf (+ epsilon floor), avgdl -- the libraries' arithmetic, libm's log --------
    std::vector<int64_t> indptr((size_t)V + 1);
    indptr[0] = 0;
    for (int64_t t = 0; t < V; ++t) indptr[t + 1] = indptr[t] + (int64_t)df[t];
    if (indptr[V] != nnz) { cleanup(); return h->fail(ERH_ERR_HIP, "erh_build_bm25_index: posting count mismatch"); }
    S.idf_host.assign((size_t)V, 0.0);
    S.average_idf = 0.0;
    const double total_len = (double)off[N];                               // < 2^53: exact
    S.avgdl = total_len / (double)N;                                       // rank_bm25: num_doc / corpus_size; bm25s: mean(len)
    if (variant == ERH_BM25_OKAPI) {
        for (int64_t t = 0; t < V; ++t)
            if (df[t]) S.idf_host[t] = std::log((double)(N - (int64_t)df[t]) + 0.5) - std::log((double)df[t] + 0.5);
        // average over the terms in first-appearance order (the order rank_bm25's `nd` dict was filled), sequentially
        std::vector<int64_t> order;
        order.reserve((size_t)V);
        for (int64_t t = 0; t < V; ++t) if (df[t]) order.push_back(t);
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t c) { return first[a] < first[c]; });
        double sum = 0.0;
        for (int64_t t : order) sum += S.idf_host[t];
        S.average_idf = sum / (double)std::max<size_t>(order.size(), 1);
        const double eps = epsilon * S.average_idf;
        for (int64_t t = 0; t < V; ++t) if (S.idf_host[t] < 0) S.idf_host[t] = eps;
    } else {
        for (int64_t t = 0; t < V; ++t)
            if (df[t]) S.idf_host[t] = (double)(float)std::log(1.0 + ((double)(N - (int64_t)df[t]) + 0.5) / ((double)df[t] + 0.5));
    }
    // ---- device: indptr, per-posting payload, skip tables ----------------------------------------------------------
    BUILD_CHK(S.indptr.ensure((size_t)(V + 1) * 8));
    BUILD_CHK(hipMemcpyAsync(S.indptr.p, indptr.data(), (size_t)(V + 1) * 8, hipMemcpyHostToDevice, st));
    const size_t es = (variant == ERH_BM25_OKAPI) ? 8 : 4;
    BUILD_CHK(S.payload.ensure((size_t)(nnz + 1) * es));
    BUILD_CHK(d_dl.ensure((size_t)N * 4));
    BUILD_CHK(hipMemcpyAsync(d_dl.p, dl.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
    BUILD_CHK(d_idf.ensure((size_t)V * es));
    std::vector<float> idf32;
    if (variant == ERH_BM25_OKAPI) {
        BUILD_CHK(hipMemcpyAsync(d_idf.p, S.idf_host.data(), (size_t)V * 8, hipMemcpyHostToDevice, st));
    } else {
        idf32.resize((size_t)V);
        for (int64_t t = 0; t < V; ++t) idf32[t] = (float)S.idf_host[t];
        BUILD_CHK(hipMemcpyAsync(d_idf.p, idf32.data(), (size_t)V * 4, hipMemcpyHostToDevice, st));
    }
    BUILD_CHK(erh::launch_bm25_payload(variant, V, nnz, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.tf.as<int32_t>(),
                                       d_dl.as<int32_t>(), d_idf.p, S.avgdl, k1, b, S.payload.p, st));
    int rc = bm25_finish_tables(h, variant, V, N, st);
    if (rc != ERH_OK) { cleanup(); return rc; }
    BUILD_CHK(hipStreamSynchronize(st));
#undef BUILD_CHK
    cleanup();
    S.host_indptr = std::move(indptr);
    S.variant = variant;
    S.V = V;
    S.Nb = N;
    S.nnz = nnz;
    S.built_on_device = true;
    if (out_nnz) *out_nnz = nnz;
    return bm25_check_payload_sign(h, st);
}

int erh_get_bm25_csr(erh_handle *h, int64_t *indptr, int32_t *doc_ids, int32_t *tf, double *idf, double *avgdl,
                     double *average_idf) {
    if (!h) return ERH_ERR_INVALID;
    Bm25State &S = h->bm[h->cur];
    if (S.variant < 0 || !S.built_on_device) return h->fail(ERH_ERR_STATE, "erh_get_bm25_csr: the selected slot was not built by erh_build_bm25_index");
    HIPCHK(h, hipSetDevice(h->device));
    if (indptr) memcpy(indptr, S.host_indptr.data(), (size_t)(S.V + 1) * 8);
    if (doc_ids && S.nnz) HIPCHK(h, hipMemcpy(doc_ids, S.doc_ids.p, (size_t)S.nnz * 4, hipMemcpyDeviceToHost));
    if (tf && S.nnz) HIPCHK(h, hipMemcpy(tf, S.tf.p, (size_t)S.nnz * 4, hipMemcpyDeviceToHost));
    if (idf) memcpy(idf, S.idf_host.data(), (size_t)S.V * 8);
    if (avgdl) *avgdl = S.avgdl;
    if (average_idf) *average_idf = S.average_idf;
    return ERH_OK;
}

int erh_bm25_select(erh_handle *h, int slot) {
    if (!h) return ERH_ERR_INVALID;
    if (slot < 0 || slot >= ERH_BM25_SLOTS) return h->fail(ERH_ERR_INVALID, "erh_bm25_select: slot out of range");
    h->cur = slot;
    return ERH_OK;
}

int erh_bm25_release(erh_handle *h, int slot) {
    if (!h) return ERH_ERR_INVALID;
    if (slot < 0 || slot >= ERH_BM25_SLOTS) return h->fail(ERH_ERR_INVALID, "erh_bm25_release: slot out of range");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());                            // nothing in flight may still read the slot
    Bm25State &S = h->bm[slot];
    S.release();
    S = Bm25State();
    return ERH_OK;
}

int erh_get_bm25_payload(erh_handle *h, void *out_payload) {
    if (!h || !out_payload) return ERH_ERR_INVALID;
    if (h->bm[h->cur].variant < 0) return h->fail(ERH_ERR_STATE, "bm25 index not set");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t es = (h->bm[h->cur].variant == ERH_BM25_OKAPI) ? 8 : 4;
    if (h->bm[h->cur].nnz) HIPCHK(h, hipMemcpy(out_payload, h->bm[h->cur].payload.p, (size_t)h->bm[h->cur].nnz * es, hipMemcpyDeviceToHost));
    return ERH_OK;
}

int erh_set_doc_meta(erh_handle *h, int64_t N, const int32_t *content_id, const int16_t *dir_id) {
    if (!h) return ERH_ERR_INVALID;
    if (N <= 0) return h->fail(ERH_ERR_INVALID, "erh_set_doc_meta: N <= 0");
    HIPCHK(h, hipSetDevice(h->device));
    if (content_id) {
        for (int64_t i = 0; i < N; ++i)
            if (content_id[i] < 0) return h->fail(ERH_ERR_INVALID, "erh_set_doc_meta: negative content id");
        HIPCHK(h, h->content_id.ensure((size_t)N * 4));
        HIPCHK(h, hipMemcpy(h->content_id.p, content_id, (size_t)N * 4, hipMemcpyHostToDevice));
    }
    h->dir_rng_n = 0;
    h->blocks.valid = false;
    h->dir_lo_h.clear(); h->dir_hi_h.clear(); h->dir_cnt_h.clear(); h->dir_order_h.clear(); h->dir_off_h.clear();
    if (dir_id) {
        HIPCHK(h, h->dir_id.ensure((size_t)N * 2));
        HIPCHK(h, hipMemcpy(h->dir_id.p, dir_id, (size_t)N * 2, hipMemcpyHostToDevice));
        // document range of every class: where its documents are one block (the reference's dirs are: its loader walks the
        // directories one after the other) a filtered BM25 query skips every tile outside it
        int maxc = -1;
        for (int64_t i = 0; i < N; ++i) maxc = dir_id[i] > maxc ? dir_id[i] : maxc;
        if (maxc >= 0 && N <= 2147483647LL) {
            std::vector<int32_t> rng((size_t)(maxc + 1) * 2);
            for (int c = 0; c <= maxc; ++c) { rng[2 * c] = 2147483647; rng[2 * c + 1] = 0; }
            for (int64_t i = 0; i < N; ++i) {
                const int c = dir_id[i];
                if (c < 0) continue;
                if ((int32_t)i < rng[2 * c]) rng[2 * c] = (int32_t)i;
                rng[2 * c + 1] = (int32_t)i + 1;
            }
            for (int c = 0; c <= maxc; ++c) if (rng[2 * c + 1] == 0) rng[2 * c] = 0;
            h->dir_lo_h.assign((size_t)maxc + 1, 0); h->dir_hi_h.assign((size_t)maxc + 1, 0); h->dir_cnt_h.assign((size_t)maxc + 1, 0);
            for (int c = 0; c <= maxc; ++c) { h->dir_lo_h[c] = rng[2 * c]; h->dir_hi_h[c] = rng[2 * c + 1]; }
            for (int64_t i = 0; i < N; ++i) if (dir_id[i] >= 0) h->dir_cnt_h[dir_id[i]] += 1;
            h->dir_off_h.assign((size_t)maxc + 2, 0);
            for (int c = 0; c <= maxc; ++c) h->dir_off_h[c + 1] = h->dir_off_h[c] + h->dir_cnt_h[c];
            h->dir_order_h.resize((size_t)h->dir_off_h[maxc + 1]);
            { std::vector<int64_t> at(h->dir_off_h.begin(), h->dir_off_h.end() - 1);
              for (int64_t i = 0; i < N; ++i) if (dir_id[i] >= 0) h->dir_order_h[(size_t)at[dir_id[i]]++] = (int32_t)i; }
            HIPCHK(h, h->dir_rng.ensure(rng.size() * 4));
            HIPCHK(h, hipMemcpy(h->dir_rng.p, rng.data(), rng.size() * 4, hipMemcpyHostToDevice));
            h->dir_rng_n = maxc + 1;
        }
    }
    h->has_content = content_id != nullptr;
    h->has_dir = dir_id != nullptr;
    h->Nmeta = N;
    h->dir_pos_valid = false;
    return ERH_OK;
}

// ---- queries ---------------------------------------------------------------------------------------

static int stage_filter(erh_handle *h, const int16_t *filter_dir, int B, int64_t n_docs, hipStream_t st, const int16_t **dev,
                        DevBuf *buf = nullptr) {
    if (!buf) buf = &h->filt;
    *dev = nullptr;
    if (!filter_dir) return ERH_OK;
    bool any = false;
    for (int b = 0; b < B; ++b) any = any || filter_dir[b] >= 0;
    if (!any) return ERH_OK;
    if (!h->has_dir || h->Nmeta < n_docs) return h->fail(ERH_ERR_STATE, "filter given but erh_set_doc_meta(dir_id) not set for all documents");
    HIPCHK(h, buf->ensure((size_t)B * 2));
    HIPCHK(h, hipMemcpyAsync(buf->p, filter_dir, (size_t)B * 2, hipMemcpyHostToDevice, st));
    *dev = buf->as<int16_t>();
    return ERH_OK;
}

static int stage_query_block(erh_handle *h, const void *q, int q_dtype, int q_is_device, int B, hipStream_t st, const void **dev) {
    if (q_is_device) { *dev = q; return ERH_OK; }
    const size_t bytes = (size_t)B * h->d * (q_dtype == ERH_F16 ? 2 : 4);
    HIPCHK(h, h->qin.ensure(bytes));
    HIPCHK(h, hipMemcpyAsync(h->qin.p, q, bytes, hipMemcpyHostToDevice, st));
    *dev = h->qin.p;
    return ERH_OK;
}

static int copy_out(erh_handle *h, int B, int k, const int32_t *d_ids, const double *d_sc, const int32_t *d_len,
                    int32_t *out_ids, double *out_scores, int32_t *out_len, hipStream_t st) {
    HIPCHK(h, hipMemcpyAsync(out_ids, d_ids, (size_t)B * k * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(out_scores, d_sc, (size_t)B * k * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(out_len, d_len, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return ERH_OK;
}

int erh_dense_topk(erh_handle *h, const void *q, int q_dtype, int q_is_device, int normalize_q,
                   int B, int k, const int16_t *filter_dir, int mode,
                   int32_t *out_ids, double *out_scores, int32_t *out_len, int out_is_device, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!h->X.p || h->N <= 0) return h->fail(ERH_ERR_STATE, "erh_dense_topk before erh_set_dense");
    if (!q || !out_ids || !out_scores || !out_len || B <= 0 || k <= 0) return h->fail(ERH_ERR_INVALID, "erh_dense_topk: null pointer or non-positive B/k");
    if (q_dtype != ERH_F16 && q_dtype != ERH_F32) return h->fail(ERH_ERR_INVALID, "erh_dense_topk: q_dtype");
    if (mode != ERH_DENSE_EXACT && mode != ERH_DENSE_FAST) return h->fail(ERH_ERR_INVALID, "erh_dense_topk: mode");
    if (k > 768) return h->fail(ERH_ERR_UNSUPPORTED, "erh_dense_topk: k > 768");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int16_t *filt = nullptr;
    int rc = stage_filter(h, filter_dir, B, h->N, st, &filt);
    if (rc != ERH_OK) return rc;
    const void *qd = nullptr;
    rc = stage_query_block(h, q, q_dtype, q_is_device, B, st, &qd);
    if (rc != ERH_OK) return rc;
    int32_t *d_ids = out_ids; double *d_sc = out_scores; int32_t *d_len = out_len;
    if (!out_is_device) {
        HIPCHK(h, h->o_ids.ensure((size_t)B * k * 4));
        HIPCHK(h, h->o_sc.ensure((size_t)B * k * 8));
        HIPCHK(h, h->o_len.ensure((size_t)B * 4));
        d_ids = h->o_ids.as<int32_t>(); d_sc = h->o_sc.as<double>(); d_len = h->o_len.as<int32_t>();
    }
    h->stats.dense_calls += 1;
    rc = dense_topk_routed(h, qd, q_dtype, normalize_q, B, k, filter_dir, filt, mode, d_ids, d_sc, d_len, st);
    if (rc != ERH_OK) return rc;
    if (!out_is_device) {
        rc = dense_check_flags(h, st);                  // (may run further exhaustive rounds before the copy)
        if (rc != ERH_OK) return rc;
        return copy_out(h, B, k, d_ids, d_sc, d_len, out_ids, out_scores, out_len, st);
    }
    return ERH_OK;
}

int erh_bm25_topk(erh_handle *h, const int32_t *q_indptr, const int32_t *q_tok, int B, int k,
                  const int16_t *filter_dir,
                  int32_t *out_ids, double *out_scores, int32_t *out_len, int out_is_device, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (h->bm[h->cur].variant < 0) return h->fail(ERH_ERR_STATE, "erh_bm25_topk before erh_set_bm25_*");
    if (!q_indptr || !out_ids || !out_scores || !out_len || B <= 0 || k <= 0) return h->fail(ERH_ERR_INVALID, "erh_bm25_topk: null pointer or non-positive B/k");
    if (q_indptr[B] > 0 && !q_tok) return h->fail(ERH_ERR_INVALID, "erh_bm25_topk: null q_tok");
    if (k > 1024) return h->fail(ERH_ERR_UNSUPPORTED, "erh_bm25_topk: k > 1024");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int16_t *filt = nullptr;
    int rc = stage_filter(h, filter_dir, B, h->bm[h->cur].Nb, st, &filt);
    if (rc != ERH_OK) return rc;
    double bytes = 0;
    int max_qlen = 0;
    rc = upload_bm25_queries(h, q_indptr, q_tok, B, st, h->bm[h->cur].host_indptr, &bytes, &max_qlen);
    if (rc != ERH_OK) return rc;
    int32_t *d_ids = out_ids; double *d_sc = out_scores; int32_t *d_len = out_len;
    if (!out_is_device) {
        HIPCHK(h, h->o_ids.ensure((size_t)B * k * 4));
        HIPCHK(h, h->o_sc.ensure((size_t)B * k * 8));
        HIPCHK(h, h->o_len.ensure((size_t)B * 4));
        d_ids = h->o_ids.as<int32_t>(); d_sc = h->o_sc.as<double>(); d_len = h->o_len.as<int32_t>();
    }
    h->stats.bm25_calls += 1;
    rc = bm25_topk_dev(h, h->qptr, h->qtok, B, k, filt, d_ids, d_sc, d_len, bytes, max_qlen, st);
    if (rc != ERH_OK) return rc;
    if (!out_is_device) return copy_out(h, B, k, d_ids, d_sc, d_len, out_ids, out_scores, out_len, st);
    return ERH_OK;
}

int erh_bm25_scores(erh_handle *h, const int32_t *q_tok, int n_tok, double *out_scores) {
    if (!h) return ERH_ERR_INVALID;
    if (h->bm[h->cur].variant < 0) return h->fail(ERH_ERR_STATE, "erh_bm25_scores before erh_set_bm25_*");
    if (!out_scores || n_tok < 0 || (n_tok > 0 && !q_tok)) return h->fail(ERH_ERR_INVALID, "erh_bm25_scores: null pointer");
    for (int i = 0; i < n_tok; ++i)
        if (q_tok[i] < 0 || q_tok[i] >= h->bm[h->cur].V) return h->fail(ERH_ERR_INVALID, "erh_bm25_scores: term id out of range");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = nullptr;
    const size_t es = (h->bm[h->cur].variant == ERH_BM25_OKAPI) ? 8 : 4;
    HIPCHK(h, h->scores_tmp.ensure((size_t)h->bm[h->cur].Nb * es));
    HIPCHK(h, hipMemsetAsync(h->scores_tmp.p, 0, (size_t)h->bm[h->cur].Nb * es, st));
    for (int i = 0; i < n_tok; ++i)
        HIPCHK(h, erh::launch_bm25_add_term(h->bm[h->cur].variant, h->bm[h->cur].indptr.as<int64_t>(), h->bm[h->cur].doc_ids.as<int32_t>(), h->bm[h->cur].payload.p,
                                            q_tok[i], h->scores_tmp.p, st));
    const double *src = h->scores_tmp.as<double>();
    if (h->bm[h->cur].variant == ERH_BM25_BM25S) {
        HIPCHK(h, h->scores_wide.ensure((size_t)h->bm[h->cur].Nb * 8));
        HIPCHK(h, erh::launch_widen_f32(h->scores_tmp.as<float>(), h->bm[h->cur].Nb, h->scores_wide.as<double>(), st));
        src = h->scores_wide.as<double>();
    }
    HIPCHK(h, hipMemcpyAsync(out_scores, src, (size_t)h->bm[h->cur].Nb * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return ERH_OK;
}

static int fuse_common(erh_handle *h, bool rrf, const int32_t *ids_a, const double *sc_a, const int32_t *len_a, int depth_a,
                       const int32_t *ids_b, const double *sc_b, const int32_t *len_b, int depth_b, int B, int K, int topk,
                       int32_t *out_ids, double *out_scores, int32_t *out_len, int io_is_device, hipStream_t st) {
    if (!ids_a || !ids_b || !out_ids || !out_scores || !out_len || B <= 0 || topk <= 0 || depth_a < 0 || depth_b < 0)
        return h->fail(ERH_ERR_INVALID, "fusion: null pointer or non-positive B/topk");
    if (!rrf && (!sc_a || !sc_b)) return h->fail(ERH_ERR_INVALID, "fusion: null scores");
    if (depth_a + depth_b > erh::kFuseMaxItems) return h->fail(ERH_ERR_UNSUPPORTED, "fusion: depth_a + depth_b > 2048");
    if (depth_a + depth_b == 0) return h->fail(ERH_ERR_INVALID, "fusion: both lists empty by construction");
    HIPCHK(h, hipSetDevice(h->device));
    const int32_t *cid = h->has_content ? h->content_id.as<int32_t>() : nullptr;
    const int32_t *da = ids_a, *db = ids_b, *dla = len_a, *dlb = len_b;
    const double *dsa = sc_a, *dsb = sc_b;
    int32_t *d_ids = out_ids; double *d_sc = out_scores; int32_t *d_len = out_len;
    if (!io_is_device) {
        HIPCHK(h, h->fa_ids.ensure((size_t)B * std::max(depth_a, 1) * 4));
        HIPCHK(h, h->fb_ids.ensure((size_t)B * std::max(depth_b, 1) * 4));
        HIPCHK(h, hipMemcpyAsync(h->fa_ids.p, ids_a, (size_t)B * depth_a * 4, hipMemcpyHostToDevice, st));
        HIPCHK(h, hipMemcpyAsync(h->fb_ids.p, ids_b, (size_t)B * depth_b * 4, hipMemcpyHostToDevice, st));
        da = h->fa_ids.as<int32_t>(); db = h->fb_ids.as<int32_t>();
        if (len_a) { HIPCHK(h, h->fa_len.ensure((size_t)B * 4)); HIPCHK(h, hipMemcpyAsync(h->fa_len.p, len_a, (size_t)B * 4, hipMemcpyHostToDevice, st)); dla = h->fa_len.as<int32_t>(); }
        if (len_b) { HIPCHK(h, h->fb_len.ensure((size_t)B * 4)); HIPCHK(h, hipMemcpyAsync(h->fb_len.p, len_b, (size_t)B * 4, hipMemcpyHostToDevice, st)); dlb = h->fb_len.as<int32_t>(); }
        if (!rrf) {
            HIPCHK(h, h->fa_sc.ensure((size_t)B * std::max(depth_a, 1) * 8));
            HIPCHK(h, h->fb_sc.ensure((size_t)B * std::max(depth_b, 1) * 8));
            HIPCHK(h, hipMemcpyAsync(h->fa_sc.p, sc_a, (size_t)B * depth_a * 8, hipMemcpyHostToDevice, st));
            HIPCHK(h, hipMemcpyAsync(h->fb_sc.p, sc_b, (size_t)B * depth_b * 8, hipMemcpyHostToDevice, st));
            dsa = h->fa_sc.as<double>(); dsb = h->fb_sc.as<double>();
        }
        HIPCHK(h, h->o_ids.ensure((size_t)B * topk * 4));
        HIPCHK(h, h->o_sc.ensure((size_t)B * topk * 8));
        HIPCHK(h, h->o_len.ensure((size_t)B * 4));
        d_ids = h->o_ids.as<int32_t>(); d_sc = h->o_sc.as<double>(); d_len = h->o_len.as<int32_t>();
    }
    { ProfScope ps(h, st, ERH_K_FUSE, 0, 0);
      if (rrf) HIPCHK(h, erh::launch_rrf(da, dla, depth_a, db, dlb, depth_b, cid, B, K, topk, d_ids, d_sc, d_len, st));
      else HIPCHK(h, erh::launch_fusion(da, dsa, dla, depth_a, db, dsb, dlb, depth_b, cid, B, topk, d_ids, d_sc, d_len, st)); }
    if (!io_is_device) return copy_out(h, B, topk, d_ids, d_sc, d_len, out_ids, out_scores, out_len, st);
    return ERH_OK;
}

int erh_rrf(erh_handle *h, const int32_t *ids_a, const int32_t *len_a, int depth_a,
            const int32_t *ids_b, const int32_t *len_b, int depth_b, int B, int K, int topk,
            int32_t *out_ids, double *out_scores, int32_t *out_len, int io_is_device, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (K < 0) return h->fail(ERH_ERR_INVALID, "erh_rrf: K < 0");
    return fuse_common(h, true, ids_a, nullptr, len_a, depth_a, ids_b, nullptr, len_b, depth_b, B, K, topk,
                       out_ids, out_scores, out_len, io_is_device, (hipStream_t)stream);
}

int erh_fusion(erh_handle *h, const int32_t *ids_a, const double *scores_a, const int32_t *len_a, int depth_a,
               const int32_t *ids_b, const double *scores_b, const int32_t *len_b, int depth_b, int B, int topk,
               int32_t *out_ids, double *out_scores, int32_t *out_len, int io_is_device, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    return fuse_common(h, false, ids_a, scores_a, len_a, depth_a, ids_b, scores_b, len_b, depth_b, B, 0, topk,
                       out_ids, out_scores, out_len, io_is_device, (hipStream_t)stream);
}

int erh_hybrid_topk(erh_handle *h, const void *q, int q_dtype, int q_is_device, int normalize_q,
                    const int32_t *q_indptr, const int32_t *q_tok, int B,
                    int k_dense, int k_sparse, int K, int topk, const int16_t *filter_sparse,
                    const int16_t *filter_dense,
                    int32_t *out_ids, double *out_scores, int32_t *out_len, int out_is_device, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!h->X.p || h->N <= 0) return h->fail(ERH_ERR_STATE, "erh_hybrid_topk before erh_set_dense");
    if (h->bm[h->cur].variant < 0) return h->fail(ERH_ERR_STATE, "erh_hybrid_topk before erh_set_bm25_*");
    if (h->bm[h->cur].Nb != h->N) return h->fail(ERH_ERR_STATE, "erh_hybrid_topk: dense and bm25 corpora differ in size");
    if (!q || !q_indptr || !out_ids || !out_scores || !out_len || B <= 0 || k_dense <= 0 || k_sparse <= 0 || topk <= 0 || K < 0)
        return h->fail(ERH_ERR_INVALID, "erh_hybrid_topk: null pointer or non-positive size");
    if (q_dtype != ERH_F16 && q_dtype != ERH_F32) return h->fail(ERH_ERR_INVALID, "erh_hybrid_topk: q_dtype");
    if (k_dense > 768 || k_sparse > 1024 || k_dense + k_sparse > erh::kFuseMaxItems)
        return h->fail(ERH_ERR_UNSUPPORTED, "erh_hybrid_topk: k_dense > 768 or k_sparse > 1024");
    if (q_indptr[B] > 0 && !q_tok) return h->fail(ERH_ERR_INVALID, "erh_hybrid_topk: null q_tok");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    // the two routes are filtered independently, as the reference does (filter_dict -> sparse, filters -> dense;
    // retrievers.py:278,283); equal pointers / equal contents share one staged column
    const int16_t *filt = nullptr, *filt_d = nullptr;
    int rc = stage_filter(h, filter_sparse, B, h->N, st, &filt);
    if (rc != ERH_OK) return rc;
    if (filter_dense == filter_sparse || (filter_dense && filter_sparse && !memcmp(filter_dense, filter_sparse, (size_t)B * 2))) {
        filt_d = filt;
    } else {
        rc = stage_filter(h, filter_dense, B, h->N, st, &filt_d, &h->filt2);
        if (rc != ERH_OK) return rc;
    }
    double bytes = 0;
    int max_qlen = 0;
    rc = upload_bm25_queries(h, q_indptr, q_tok, B, st, h->bm[h->cur].host_indptr, &bytes, &max_qlen);
    if (rc != ERH_OK) return rc;
    const void *qd = nullptr;
    rc = stage_query_block(h, q, q_dtype, q_is_device, B, st, &qd);
    if (rc != ERH_OK) return rc;
    HIPCHK(h, h->hy_sids.ensure((size_t)B * k_sparse * 4));
    HIPCHK(h, h->hy_ssc.ensure((size_t)B * k_sparse * 8));
    HIPCHK(h, h->hy_slen.ensure((size_t)B * 4));
    HIPCHK(h, h->hy_dids.ensure((size_t)B * k_dense * 4));
    HIPCHK(h, h->hy_dsc.ensure((size_t)B * k_dense * 8));
    HIPCHK(h, h->hy_dlen.ensure((size_t)B * 4));
    // sparse route (list a), dense route (list b), fusion -- no host round trip.  The two routes do not depend on each
    // other: with hybrid_overlap the sparse route is enqueued on a side stream that forks from the caller's stream (its
    // inputs were staged there) and joins it again in front of the fusion; whatever the dense pipeline leaves idle --
    // the under-filled seed grid, the selection kernels, the tail of the persistent scan -- the other route can use.
    h->stats.hybrid_calls += 1;
    hipStream_t st_sparse = st;
    const int ov = h->opt_hybrid_overlap >= 0 ? h->opt_hybrid_overlap : (B <= 256 ? 1 : 0);
    if (ov) {
        if (!h->side) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        st_sparse = h->side;
    }
    auto sparse_route = [&]() -> int {
        int r = bm25_topk_dev(h, h->qptr, h->qtok, B, k_sparse, filt, h->hy_sids.as<int32_t>(),
                              h->hy_ssc.as<double>(), h->hy_slen.as<int32_t>(), bytes, max_qlen, st_sparse);
        if (r == ERH_OK && st_sparse != st) {
            hipError_t e_ = hipEventRecord(h->ev_join, st_sparse);
            if (e_ != hipSuccess) r = h->fail(ERH_ERR_HIP, "hipEventRecord(join)", e_);
        }
        return r;
    };
    if (ov == 1) {                                   // fork at once: both routes compete for the CUs from the start
        HIPCHK(h, hipEventRecord(h->ev_fork, st));
        HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    }
    if (ov != 2) {
        rc = sparse_route();
        if (rc != ERH_OK) { if (st_sparse != st) (void)hipStreamSynchronize(st_sparse); return rc; }
    }
    h->fork_after_scan = (ov == 2);
    rc = dense_topk_routed(h, qd, q_dtype, normalize_q, B, k_dense, filter_dense, filt_d, ERH_DENSE_EXACT, h->hy_dids.as<int32_t>(),
                           h->hy_dsc.as<double>(), h->hy_dlen.as<int32_t>(), st);
    h->fork_after_scan = false;
    if (ov == 2) {                                   // fork behind the dense scan: the sparse route runs beside the selection kernels
        if (rc != ERH_OK) return rc;
        HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
        rc = sparse_route();
        if (rc != ERH_OK) { (void)hipStreamSynchronize(st_sparse); return rc; }
    }
    if (st_sparse != st) {                           // join (also on the error path: nothing may outlive the call's buffers)
        hipError_t e_ = hipStreamWaitEvent(st, h->ev_join, 0);
        if (e_ != hipSuccess && rc == ERH_OK) rc = h->fail(ERH_ERR_HIP, "hipStreamWaitEvent(join)", e_);
    }
    if (rc != ERH_OK) return rc;
    int32_t *d_ids = out_ids; double *d_sc = out_scores; int32_t *d_len = out_len;
    if (!out_is_device) {
        HIPCHK(h, h->o_ids.ensure((size_t)B * topk * 4));
        HIPCHK(h, h->o_sc.ensure((size_t)B * topk * 8));
        HIPCHK(h, h->o_len.ensure((size_t)B * 4));
        d_ids = h->o_ids.as<int32_t>(); d_sc = h->o_sc.as<double>(); d_len = h->o_len.as<int32_t>();
    }
    const int32_t *cid = h->has_content ? h->content_id.as<int32_t>() : nullptr;
    { ProfScope ps(h, st, ERH_K_FUSE, 0, 0);
      HIPCHK(h, erh::launch_rrf(h->hy_sids.as<int32_t>(), h->hy_slen.as<int32_t>(), k_sparse,
                                h->hy_dids.as<int32_t>(), h->hy_dlen.as<int32_t>(), k_dense, cid, B, K, topk,
                                d_ids, d_sc, d_len, st)); }
    h->last.hybrid = true;
    h->last.k_sparse = k_sparse; h->last.K = K; h->last.topk = topk;
    h->last.f_ids = d_ids; h->last.f_sc = d_sc; h->last.f_len = d_len;
    if (!out_is_device) {
        rc = dense_check_flags(h, st);
        if (rc != ERH_OK) return rc;
        return copy_out(h, B, topk, d_ids, d_sc, d_len, out_ids, out_scores, out_len, st);
    }
    return ERH_OK;
}

int erh_debug_dense_scores(erh_handle *h, const void *q_f16_host, int B, int64_t row0, int rows, int use_mfma, float *out) {
    if (!h) return ERH_ERR_INVALID;
    if (!h->X.p) return h->fail(ERH_ERR_STATE, "erh_debug_dense_scores before erh_set_dense");
    if (!q_f16_host || !out || B <= 0 || rows <= 0 || row0 < 0 || row0 + rows > h->N) return h->fail(ERH_ERR_INVALID, "erh_debug_dense_scores: bad range");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = nullptr;
    const int QT = erh::dense_scan_q_tile();
    const int Bpad = round_up(B, QT);
    const int d = h->d;
    HIPCHK(h, h->qin.ensure((size_t)B * d * 2));
    HIPCHK(h, hipMemcpyAsync(h->qin.p, q_f16_host, (size_t)B * d * 2, hipMemcpyHostToDevice, st));
    HIPCHK(h, h->Q16.ensure((size_t)Bpad * d * 2));
    HIPCHK(h, h->qnorm.ensure((size_t)Bpad * 4));
    HIPCHK(h, erh::launch_prep_queries(h->qin.p, ERH_F16, 0, B, Bpad, d, h->Q16.as<_Float16>(), h->qnorm.as<float>(), nullptr, nullptr, st));
    // the requested ORIGINAL rows, gathered into a contiguous block
    HIPCHK(h, h->scores_tmp.ensure((size_t)rows * d * 2));
    HIPCHK(h, erh::launch_gather_rows(h->X.as<_Float16>(), nullptr, row0, rows, d, h->pos_mul, h->N, h->scores_tmp.as<_Float16>(), st));
    const _Float16 *Xg = h->scores_tmp.as<_Float16>();
    if (use_mfma) {
        const int ld = round_up(rows, 256);
        HIPCHK(h, h->S0.ensure((size_t)Bpad * ld * 4));
        HIPCHK(h, erh::launch_dense_scan_store(h->opt_dense_cfg, h->Q16.as<_Float16>(), Bpad, Xg, rows, d, 0, rows,
                                               h->S0.as<float>(), ld, st));
        HIPCHK(h, hipMemcpy2DAsync(out, (size_t)rows * 4, h->S0.p, (size_t)ld * 4, (size_t)rows * 4, B, hipMemcpyDeviceToHost, st));
    } else {
        HIPCHK(h, h->S0.ensure((size_t)B * rows * 4));
        HIPCHK(h, erh::launch_dense_naive(h->Q16.as<_Float16>(), B, Xg, 0, rows, d, h->S0.as<float>(), st));
        HIPCHK(h, hipMemcpyAsync(out, h->S0.p, (size_t)B * rows * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(h, hipStreamSynchronize(st));
    return ERH_OK;
}

}  // extern "C"

// ---- multi-GPU: all-gather of the fused top-k over RCCL ---------------------------------------------------------------
// The corpus is replicated and the query batch sharded contiguously over the ranks (north_star),
// so the only exchange is one all-gather of [B_local x k] (score, id, len) rows.  RCCL is bound at run time with
// dlopen("librccl.so.1") -- the instance torch already mapped when the caller uses torch, the ROCm one otherwise --
// so the library has no link-time dependency on it and single-GPU users never load it.
namespace {

typedef struct { char internal[128]; } rccl_unique_id;                 // = ncclUniqueId (rccl.h)
struct Rccl {
    void *dl = nullptr;
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(void **, int, rccl_unique_id, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    if (r.dl) return r;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.dl) break;
    }
    if (!r.dl) return r;
    r.GetUniqueId = (int (*)(rccl_unique_id *))dlsym(r.dl, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void **, int, rccl_unique_id, int))dlsym(r.dl, "ncclCommInitRank");
    r.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(r.dl, "ncclAllGather");
    r.CommDestroy = (int (*)(void *))dlsym(r.dl, "ncclCommDestroy");
    r.GetErrorString = (const char *(*)(int))dlsym(r.dl, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
    return r;
}

// a timed-out erh_comm_init whose helper has meanwhile returned: destroy the communicator nobody will use
void comm_reap_pending(erh_handle *h) {
    if (!h->comm_pending || !h->comm_pending->finished.load(std::memory_order_acquire)) return;
    if (h->comm_pending->rc == 0 && h->comm_pending->comm) {
        (void)hipSetDevice(h->device);
        (void)rccl().CommDestroy(h->comm_pending->comm);
    }
    h->comm_pending.reset();
}

int rccl_fail(erh_handle *h, const char *what, int rc) {
    char buf[256];
    Rccl &r = rccl();
    snprintf(buf, sizeof buf, "%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
    return h->fail(ERH_ERR_HIP, buf);
}

}  // namespace

extern "C" {

int erh_comm_unique_id(void *out128) {
    if (!out128) return ERH_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return ERH_ERR_UNSUPPORTED;
    rccl_unique_id id;
    if (r.GetUniqueId(&id) != 0) return ERH_ERR_HIP;
    memcpy(out128, &id, sizeof id);
    return ERH_OK;
}

int erh_comm_init(erh_handle *h, int rank, int world, const void *id128) {
    if (!h) return ERH_ERR_INVALID;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return h->fail(ERH_ERR_INVALID, "erh_comm_init: bad rank / world / id");
    if (h->comm) return h->fail(ERH_ERR_STATE, "erh_comm_init: communicator already initialised");
    Rccl &r = rccl();
    if (!r.ok) return h->fail(ERH_ERR_UNSUPPORTED, "erh_comm_init: librccl.so.1 not found or incomplete");
    HIPCHK(h, hipSetDevice(h->device));
    rccl_unique_id id;
    memcpy(&id, id128, sizeof id);
    // ncclCommInitRank blocks until every rank has joined.  A rank that never arrives (crashed, wrong id) must not hang
    // the others for ever: the call runs on a helper thread and this one waits at most comm_timeout_s seconds (option,
    // default 120).  After a timeout the helper is abandoned (it may still be blocked inside RCCL) and the handle stays
    // without a communicator -- callers fall back to the torch.distributed gather (easyrag_amd.dist.QueryShards).
    // A communicator that arrives late is destroyed by the next erh_comm_init / erh_comm_destroy / erh_destroy that finds the
    // helper finished; a helper still inside RCCL at process exit is the caller's problem -- after a timeout the process
    // should exit (the peers hold a communicator this rank never joined).
    comm_reap_pending(h);
    auto state = std::make_shared<CommInitState>();
    auto done = std::make_shared<std::promise<void>>();
    std::future<void> fut = done->get_future();
    const int dev = h->device;
    auto init_fn = r.CommInitRank;
    std::thread([state, done, init_fn, id, world, rank, dev]() {
        (void)hipSetDevice(dev);
        state->rc = init_fn(&state->comm, world, id, rank);
        state->finished.store(1, std::memory_order_release);
        done->set_value();
    }).detach();
    if (fut.wait_for(std::chrono::seconds(h->opt_comm_timeout_s)) != std::future_status::ready) {
        h->comm_pending = state;
        return h->fail(ERH_ERR_HIP, "erh_comm_init: ncclCommInitRank did not return within comm_timeout_s (a rank is missing?)");
    }
    const int rc = state->rc;
    void *c = state->comm;
    if (rc != 0) return rccl_fail(h, "ncclCommInitRank", rc);
    h->comm = c;
    h->comm_rank = rank;
    h->comm_world = world;
    return ERH_OK;
}

int erh_comm_destroy(erh_handle *h) {
    if (!h) return ERH_ERR_INVALID;
    comm_reap_pending(h);
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)rccl().CommDestroy(h->comm);
        h->comm = nullptr;
    }
    h->comm_rank = 0;
    h->comm_world = 1;
    return ERH_OK;
}

int erh_topk_row_bytes(int k) { return k > 0 ? erh::topk_row_bytes(k) : 0; }

int erh_pack_topk(erh_handle *h, const int32_t *ids, const double *scores, const int32_t *lens, int b_local, int k,
                  int rows, void *out_rows, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!ids || !scores || !lens || !out_rows || b_local < 0 || k <= 0 || rows < b_local)
        return h->fail(ERH_ERR_INVALID, "erh_pack_topk: null pointer or bad sizes");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, erh::launch_pack_topk(ids, scores, lens, b_local, k, rows, out_rows, (hipStream_t)stream));
    return ERH_OK;
}

int erh_unpack_topk(erh_handle *h, const void *gathered_rows, int n_queries, int world, int k,
                    int32_t *out_ids, double *out_scores, int32_t *out_len, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!gathered_rows || !out_ids || !out_scores || !out_len || n_queries <= 0 || world <= 0 || k <= 0)
        return h->fail(ERH_ERR_INVALID, "erh_unpack_topk: null pointer or bad sizes");
    HIPCHK(h, hipSetDevice(h->device));
    const int m = (n_queries + world - 1) / world;
    HIPCHK(h, erh::launch_unpack_topk(gathered_rows, n_queries, world, k, m, out_ids, out_scores, out_len,
                                      (hipStream_t)stream));
    return ERH_OK;
}

int erh_allgather_topk(erh_handle *h, const int32_t *ids, const double *scores, const int32_t *lens, int b_local, int k,
                       int n_queries, int32_t *out_ids, double *out_scores, int32_t *out_len, void *stream) {
    if (!h) return ERH_ERR_INVALID;
    if (!ids || !scores || !lens || !out_ids || !out_scores || !out_len || k <= 0 || n_queries <= 0 || b_local < 0)
        return h->fail(ERH_ERR_INVALID, "erh_allgather_topk: null pointer or bad sizes");
    const int world = h->comm_world, rank = h->comm_rank;
    const int base = n_queries / world, rem = n_queries % world;
    if (b_local != base + (rank < rem ? 1 : 0))
        return h->fail(ERH_ERR_INVALID, "erh_allgather_topk: b_local is not this rank's contiguous shard of n_queries");
    if (world > 1 && !h->comm) return h->fail(ERH_ERR_STATE, "erh_allgather_topk before erh_comm_init");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int m = (n_queries + world - 1) / world;
    const size_t row = (size_t)erh::topk_row_bytes(k);
    HIPCHK(h, h->gather_send.ensure(row * m));
    HIPCHK(h, h->gather_recv.ensure(row * m * world));
    HIPCHK(h, erh::launch_pack_topk(ids, scores, lens, b_local, k, m, h->gather_send.p, st));
    if (world > 1) {
        const int rc = rccl().AllGather(h->gather_send.p, h->gather_recv.p, row * m, /*ncclChar*/ 0, h->comm, st);
        if (rc != 0) return rccl_fail(h, "ncclAllGather", rc);
    } else {
        HIPCHK(h, hipMemcpyAsync(h->gather_recv.p, h->gather_send.p, row * m, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(h, erh::launch_unpack_topk(h->gather_recv.p, n_queries, world, k, m, out_ids, out_scores, out_len, st));
    return ERH_OK;
}

}  // extern "C"
