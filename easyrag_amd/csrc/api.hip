// C ABI of libeasyrag_hip.so (see include/easyrag_hip.h): handle lifetime, options, corpus state (chunk matrix, BM25 indices, metadata),
// the query entry points and their staging.  Stage orchestration: pipeline_dense.hip / pipeline_bm25.hip; the RCCL exchange: comm.hip.
// No CPU compute path lives here.
#include "../../include/easyrag_hip.h"
#include "handle.h"

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
    if (!strcmp(name, "bm25_mixed")) { h->opt_bm25_mixed = value != 0; return ERH_OK; }
    if (!strcmp(name, "bm25_long_segs")) { if (value < 1 || value > 16) return h->fail(ERH_ERR_INVALID, "bm25_long_segs"); h->opt_bm25_long_segs = (int)value; return ERH_OK; }
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
    if (!strcmp(name, "dense_gemv_nt")) { if (value < -1 || value > 1) return h->fail(ERH_ERR_INVALID, "dense_gemv_nt"); h->opt_gemv_nt = (int)value; return ERH_OK; }
    if (!strcmp(name, "dense_fin_wgs")) { if (value < 2 || value > 4) return h->fail(ERH_ERR_INVALID, "dense_fin_wgs"); erh::dense_finalize_set_wgs((int)value); return ERH_OK; }
    if (!strcmp(name, "dense_scan_nt")) { h->opt_dense_scan_nt = value != 0; return ERH_OK; }
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
        {"dense_grouped_launches", T.dense_grouped_launches}, {"bm25_mixed_launches", T.bm25_mixed_launches}};
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

// A call's host filter column -> device.  `defer`: validate only and hand back the host column (or null when no query is filtered): the
// caller lets it ride along with the query upload (upload_bm25_queries) instead of paying a copy of its own.
static int stage_filter(erh_handle *h, const int16_t *filter_dir, int B, int64_t n_docs, hipStream_t st, const int16_t **dev,
                        DevBuf *buf = nullptr, const int16_t **defer = nullptr) {
    if (!buf) buf = &h->filt;
    *dev = nullptr;
    if (defer) *defer = nullptr;
    if (!filter_dir) return ERH_OK;
    bool any = false;
    for (int b = 0; b < B; ++b) any = any || filter_dir[b] >= 0;
    if (!any) return ERH_OK;
    if (!h->has_dir || h->Nmeta < n_docs) return h->fail(ERH_ERR_STATE, "filter given but erh_set_doc_meta(dir_id) not set for all documents");
    if (defer) { *defer = filter_dir; return ERH_OK; }
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
    const int16_t *filt_host = nullptr;
    int rc = stage_filter(h, filter_dir, B, h->bm[h->cur].Nb, st, &filt, nullptr, &filt_host);
    if (rc != ERH_OK) return rc;
    double bytes = 0;
    int max_qlen = 0;
    rc = upload_bm25_queries(h, q_indptr, q_tok, B, st, h->bm[h->cur].host_indptr, &bytes, &max_qlen, filt_host, &filt);
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
    const int16_t *fs_host = nullptr, *fd_host = nullptr;
    int rc = stage_filter(h, filter_sparse, B, h->N, st, &filt, nullptr, &fs_host);
    if (rc != ERH_OK) return rc;
    const bool same_col = filter_dense == filter_sparse || (filter_dense && filter_sparse && !memcmp(filter_dense, filter_sparse, (size_t)B * 2));
    if (!same_col) {
        rc = stage_filter(h, filter_dense, B, h->N, st, &filt_d, &h->filt2, &fd_host);
        if (rc != ERH_OK) return rc;
    }
    double bytes = 0;
    int max_qlen = 0;
    // (the filter columns ride along with the query CSR: one host-to-device copy for all of them)
    rc = upload_bm25_queries(h, q_indptr, q_tok, B, st, h->bm[h->cur].host_indptr, &bytes, &max_qlen, fs_host, &filt, fd_host, &filt_d);
    if (rc != ERH_OK) return rc;
    if (same_col) filt_d = filt;
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
    h->sparse_beside = (ov == 1);
    rc = dense_topk_routed(h, qd, q_dtype, normalize_q, B, k_dense, filter_dense, filt_d, ERH_DENSE_EXACT, h->hy_dids.as<int32_t>(),
                           h->hy_dsc.as<double>(), h->hy_dlen.as<int32_t>(), st);
    h->fork_after_scan = false;
    h->sparse_beside = false;
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
