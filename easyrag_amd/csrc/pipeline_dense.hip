// Dense route, host side: stage orchestration on the caller's HIP stream -- query preparation, threshold seed (store kernel or sample pass),
// persistent scan, final kernel, the exhaustive path's bookkeeping -- for one matrix (dense_topk_dev), for a batch routed by its `dir` filter
// over per-dir block copies (dense_topk_routed: one launch per stage over all block groups, dense_topk_grouped), and the synchronisation
// point that reads a call's flag words (dense_check_flags).  Replaces the Qdrant COSINE search behind QdrantRetriever
// (/root/reference/src/easyrag/custom/retrievers.py:37-52; filter: ingestion.py:207-216).  No kernel lives here.
#include "handle.h"

// K-rotation of the ping-pong scan in stages of 32 halves per query tile (see dense_scan_pp2_kernel)
static int dense_rot_stages(const erh_handle *h, int d, int Bpad) {
    const int nk = d / 32, n_qt = Bpad / erh::dense_scan_q_tile();
    if (h->opt_dense_rot >= 0) return h->opt_dense_rot % (nk > 0 ? nk : 1);
    return n_qt > 1 ? (nk / n_qt) & ~1 : 0;
}

// Append-stage scan: persistent kernel when enabled and applicable, else one workgroup per tile.
static hipError_t scan_append(erh_handle *h, const _Float16 *X, int64_t N, int d, int64_t c0, int64_t c1, const _Float16 *Q16,
                       int Bpad, int B, const float *tau, const int16_t *filt, const int16_t *dir, ErhCand *cand,
                       uint32_t *cnt, int cap, uint32_t *flags, hipStream_t st) {
    // small batches: the skinny-GEMM stream (dense_gemv.hip) instead of a 256-query tile that is mostly padding
    if (h->opt_dense_gemv && h->opt_dense_ablate == 0 && B <= erh::dense_gemv_max_queries()) {
        hipError_t e = erh::launch_dense_gemv_append(X, N, d, c0, c1, Q16, B, tau, filt, dir, cand, cnt, cap, flags,
                                                     h->n_cus, h->opt_gemv_kb, h->opt_gemv_wgs, h->opt_gemv_pipe, st,
                                                     h->opt_gemv_nt >= 0 ? h->opt_gemv_nt : (!h->sparse_beside && B <= 32) ? 1 : 0);
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
                                                     ? (1 | (var << 1) | (h->opt_dense_pp >= 3 ? 8 : 0) | (h->opt_dense_scan_nt ? 16 : 0) | (dense_rot_stages(h, d, Bpad) << 8)) : 0,
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
static int routed_group_run(erh_handle *h, const erh_handle::RoutedGroup &g, const void *q_rows, int q_dtype, int normalize_q, hipStream_t st, bool direct = false);
static int routed_group_scatter(erh_handle *h, const erh_handle::RoutedGroup &g, hipStream_t st);

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
static int routed_group_run(erh_handle *h, const erh_handle::RoutedGroup &g, const void *q_rows, int q_dtype, int normalize_q, hipStream_t st, bool direct) {
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

static int routed_group_scatter(erh_handle *h, const erh_handle::RoutedGroup &g, hipStream_t st) {
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
static int ensure_dense_blocks(erh_handle *h, hipStream_t st) {
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
namespace {
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
}  // namespace

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

static int dense_topk_grouped(erh_handle *h, const void *q_dev, int q_dtype, int normalize_q, int k, int mode, const GroupedPlan &P,
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
                                                    h->cand_cnt.as<uint32_t>(), cap, flags, P.halfq ? 1 : 0, st, nullptr, h->opt_dense_scan_nt));
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
