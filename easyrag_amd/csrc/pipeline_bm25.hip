// Sparse route, host side: the BM25 scan + merge on the caller's HIP stream (fixed-point scan with its exact re-score, the order-keeping
// scans as fall-backs) and the upload of a call's query CSR.  Replaces BM25Retriever.get_scores + .filter
// (/root/reference/src/easyrag/custom/retrievers.py:128-151, 191-220).  No kernel lives here.
#include "handle.h"

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
    // a batch with a query too long for 16-bit sums (erh_handle::opt_bm25_long_tokens): the long queries' workgroups run the 32-bit body, all
    // others the packed one, in ONE launch (bm25_mixed; needs the 4-byte postings and a skip table at 16384 documents) -- or, without
    // that, the 32-bit shape for all of the batch
    bool mixed = false;
    if (shape_ == 2 && h->opt_bm25_long_tokens > 0 && max_qlen > h->opt_bm25_long_tokens && h->opt_bm25_ablate == 0) {
        mixed = h->opt_bm25_mixed && have16 && h->opt_bm25_post16 && S.post16.p && !h->opt_bm25_split_finish;
        if (!mixed) shape_ = have16 ? 1 : 0;
    }
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
    // mixed launch at one segment per query: the long queries cut into lsegs document ranges (items built by upload_bm25_queries for THIS query CSR)
    const int lsegs = (mixed && segs == 1 && qptr_dev == h->qptr && h->n_qitems > 0 && h->qitems_segs > 1 && (int64_t)h->qitems_segs * k <= 2048 &&
                       h->qitems_segs <= std::min(tiles, std::max(S.n_tiles, 1))) ? h->qitems_segs : 1;
    if (lsegs > 1) {
        HIPCHK(h, h->lpart_sc.ensure((size_t)B * lsegs * k * 8));
        HIPCHK(h, h->lpart_ids.ensure((size_t)B * lsegs * k * 4));
        HIPCHK(h, h->lpart_len.ensure((size_t)B * lsegs * 4));
        HIPCHK(h, h->l_redo.ensure((size_t)B * lsegs * 4));
        HIPCHK(h, hipMemsetAsync(h->l_redo.p, 0, (size_t)B * lsegs * 4, st));
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
            hipError_t e;
            if (mixed) {
                // the 32-bit body beside the packed one: its own skip table (16384-document entries), cuts at the packed shape's tiles
                const bool own16 = S.tile_docs != small_docs;                  // (bm25s: tile_off16; Okapi: tile_off IS the 16384-document table)
                e = erh::launch_bm25_ascan_mixed(S.variant, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
                                                 S.post16.p, S.g16, (uint32_t)S.nnz, S.qmax, tab, n_tab, tshift, cut_mul,
                                                 S.post.p, own16 ? S.tile_off16.as<int32_t>() : S.tile_off.as<int32_t>(), own16 ? S.n_tiles16 : S.n_tiles,
                                                 0, 2, h->opt_bm25_long_tokens,
                                                 S.Nb, qptr_dev, qtok_dev, q_order, B, k, segs, filter_dev, dir, p_sc, p_ids, p_len, h->bm_redo.as<uint32_t>(),
                                                 h->dstats.as<unsigned long long>(),
                                                 (filter_dev && h->opt_bm25_dir_range && h->dir_rng_n > 0) ? h->dir_rng.as<int32_t>() : nullptr,
                                                 h->dir_rng_n, dbg, st, lsegs > 1 ? h->qitems : nullptr, h->n_qitems, lsegs,
                                                 h->lpart_sc.as<double>(), h->lpart_ids.as<int32_t>(), h->lpart_len.as<int32_t>(), h->l_redo.as<uint32_t>());
                if (e == hipSuccess) h->stats.bm25_mixed_launches += 1;
            } else
            e = erh::launch_bm25_ascan(S.variant, shape, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
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
            e = erh::launch_bm25_scan(S.variant, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
                                      S.tile_off.as<int32_t>(), S.n_tiles, S.Nb, qptr_dev, qtok_dev, q_order, B, k, segs,
                                      filter_dev, dir, p_sc, p_ids, p_len, h->bm_redo.as<uint32_t>(), cut_tiles, cut_shift, 0,
                                      nullptr, st);
            if (e != hipSuccess || lsegs <= 1) return e;
            // the long queries of the mixed launch: their flagged segments through the exact scan (same cuts), then their lsegs lists merged
            // into the caller's rows (the launch above wrote nothing for them)
            e = erh::launch_bm25_scan(S.variant, S.indptr.as<int64_t>(), S.doc_ids.as<int32_t>(), S.payload.p,
                                      S.tile_off.as<int32_t>(), S.n_tiles, S.Nb, qptr_dev, qtok_dev, h->qlong, h->n_qlong, k, lsegs,
                                      filter_dev, dir, h->lpart_sc.as<double>(), h->lpart_ids.as<int32_t>(), h->lpart_len.as<int32_t>(),
                                      h->l_redo.as<uint32_t>(), cut_tiles, cut_shift, 0, nullptr, st);
            if (e != hipSuccess) return e;
            return erh::launch_bm25_merge(h->n_qlong, k, lsegs, h->lpart_sc.as<double>(), h->lpart_ids.as<int32_t>(), h->lpart_len.as<int32_t>(),
                                          p_ids, p_sc, p_len, st, h->qlong);
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
                        const std::vector<int64_t> &host_indptr, double *bytes, int *max_qlen,
                        const int16_t *filt_a, const int16_t **filt_a_dev, const int16_t *filt_b, const int16_t **filt_b_dev) {
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
    h->qitems = h->qlong = nullptr;
    h->n_qitems = h->n_qlong = h->qitems_segs = 0;
    const bool lpt = h->opt_bm25_lpt && B > 1;
    // launch items of the mixed scan (bm25_topk_dev decides whether it uses them): only where they can matter -- one workgroup per query
    // (>= 512 queries), a query longer than bm25_long_tokens in the batch, query ids that fit 24 bits
    const int lsegs = h->opt_bm25_long_segs;
    int n_long = 0;
    if (lpt && h->opt_bm25_mixed && h->opt_bm25_long_tokens > 0 && lsegs > 1 && B >= 512 && B < (1 << 24) && longest > h->opt_bm25_long_tokens)
        for (int b = 0; b < B; ++b) n_long += (q_indptr[b + 1] - q_indptr[b] > h->opt_bm25_long_tokens) ? 1 : 0;
    const int n_items = n_long ? B + n_long * (lsegs - 1) : 0;
    // ... | filter column(s) of the call (validated by the caller; int16 per query), so that a filtered call needs no copy of its own for them
    const size_t off_tok = (size_t)(B + 1) * 4, off_ord = off_tok + (size_t)std::max(nt, 1) * 4, off_fa = off_ord + (lpt ? (size_t)B * 4 : 0);
    const size_t off_fb = off_fa + (filt_a ? ((size_t)B * 2 + 3) / 4 * 4 : 0);
    const size_t off_items = off_fb + (filt_b ? ((size_t)B * 2 + 3) / 4 * 4 : 0), off_long = off_items + (size_t)n_items * 4;
    const size_t total_bytes = off_long + (size_t)n_long * 4;
    h->qpack_host.resize(total_bytes);
    if (filt_a) memcpy(h->qpack_host.data() + off_fa, filt_a, (size_t)B * 2);
    if (filt_b) memcpy(h->qpack_host.data() + off_fb, filt_b, (size_t)B * 2);
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
        if (n_items) {
            int32_t *items = reinterpret_cast<int32_t *>(h->qpack_host.data() + off_items), *lq = reinterpret_cast<int32_t *>(h->qpack_host.data() + off_long);
            int ni = 0, nl = 0;
            for (int i = 0; i < B; ++i) {                                     // heaviest first; a long query's segments side by side
                const int32_t b = h->qorder_host[i];
                const bool is_long = q_indptr[b + 1] - q_indptr[b] > h->opt_bm25_long_tokens;
                if (is_long) lq[nl++] = b;
                for (int s = 0; s < (is_long ? lsegs : 1); ++s) items[ni++] = b | (int32_t)((uint32_t)s << 24);
            }
        }
    }
    HIPCHK(h, h->qpack.ensure(total_bytes));
    HIPCHK(h, hipMemcpyAsync(h->qpack.p, h->qpack_host.data(), total_bytes, hipMemcpyHostToDevice, st));
    h->qptr = h->qpack.as<int32_t>();
    h->qtok = reinterpret_cast<int32_t *>(h->qpack.as<char>() + off_tok);
    h->qorder = lpt ? reinterpret_cast<int32_t *>(h->qpack.as<char>() + off_ord) : nullptr;
    h->qorder_valid = lpt;
    if (n_items) {
        h->qitems = reinterpret_cast<int32_t *>(h->qpack.as<char>() + off_items);
        h->qlong = reinterpret_cast<int32_t *>(h->qpack.as<char>() + off_long);
        h->n_qitems = n_items; h->n_qlong = n_long; h->qitems_segs = lsegs;
    }
    if (filt_a_dev) *filt_a_dev = filt_a ? reinterpret_cast<const int16_t *>(h->qpack.as<char>() + off_fa) : nullptr;
    if (filt_b_dev) *filt_b_dev = filt_b ? reinterpret_cast<const int16_t *>(h->qpack.as<char>() + off_fb) : nullptr;
    return ERH_OK;
}

