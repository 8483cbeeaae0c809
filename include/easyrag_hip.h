/*
 * easyrag_hip.h -- C ABI of libeasyrag_hip.so: the MI355X (gfx950) coarse-ranking hot path of
 * BUAADreamer/EasyRAG (reference @ 2024-12-20), i.e. what src/easyrag/custom/retrievers.py
 * computes through rank_bm25 / bm25s / qdrant on the CPU.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes (no torch / C++ types),
 * returns an int status (0 = ok, < 0 = error, see ERH_ERR_*), never throws and never aborts.
 * The library is HIP-only: without a gfx950 device erh_create() fails with ERH_ERR_NO_DEVICE;
 * there is no CPU fallback behind this ABI.
 *
 * Ownership: the caller owns every input and output buffer; the handle owns its device copies
 * (chunk matrix, postings, metadata, work space).  Output slots beyond out_len[b] are padded
 * with id = -1, score = 0.  A handle is thread-compatible (one handle per host thread).
 * All device work of a call is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 * default stream).  With host output buffers the call returns after the results have landed;
 * with device output buffers (out_is_device = 1) it returns after enqueueing; BM25 / fusion results are
 * complete after erh_sync() (or any stream synchronisation), dense and fused results after erh_dense_check()
 * -- a plain stream synchronisation is NOT enough there: queries that need more than one round of the
 * exhaustive path are finished by that call.
 *
 * Tie rule everywhere: score descending, then document index ascending ("canonical" order; the
 * reference's numpy argsort()[::-1] order among equal scores is implementation-defined).
 *
 * Reference interface each entry point replaces (paths relative to /root/reference):
 *   erh_set_dense        Qdrant collection of chunk embeddings, Distance.COSINE
 *                        (src/easyrag/pipeline/ingestion.py:155-191)
 *   erh_set_bm25_csr     BM25Retriever.__init__ index build (src/easyrag/custom/retrievers.py:94-118)
 *   erh_set_bm25_tf      same, with the per-posting IDF*TF/(TF + k1*lenNorm) evaluated on the GPU
 *   erh_set_doc_meta     node text identity (get_content(), retrievers.py:245,263) and the `dir`
 *                        metadata used by filter_dict / qdrant filters (retrievers.py:198-202,
 *                        ingestion.py:207-216, pipeline.py:301-312)
 *   erh_dense_topk       QdrantRetriever._aretrieve / _retrieve (retrievers.py:37-69)
 *   erh_bm25_scores      BM25Retriever.get_scores (retrievers.py:128-151)
 *   erh_bm25_topk        BM25Retriever._retrieve = get_scores + filter (retrievers.py:191-220)
 *   erh_rrf              HybridRetriever.reciprocal_rank_fusion (retrievers.py:256-274)
 *   erh_fusion           HybridRetriever.fusion (retrievers.py:239-253)
 *   erh_hybrid_topk      HybridRetriever._aretrieve, retrieval_type == 3 (retrievers.py:276-291)
 */
#ifndef EASYRAG_HIP_H
#define EASYRAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct erh_handle erh_handle;

/* status codes */
#define ERH_OK               0
#define ERH_ERR_INVALID     (-1)  /* bad argument */
#define ERH_ERR_NO_DEVICE   (-2)  /* no usable HIP device / wrong architecture */
#define ERH_ERR_HIP         (-3)  /* a HIP runtime call failed (see erh_last_error) */
#define ERH_ERR_STATE       (-4)  /* required data not set (e.g. dense search before erh_set_dense) */
#define ERH_ERR_UNSUPPORTED (-5)  /* shape outside the supported range */
#define ERH_ERR_OVERFLOW    (-6)  /* candidate work space overflowed and the exhaustive path could not finish */
#define ERH_ERR_NOMEM       (-7)  /* device allocation failed */

/* element types */
#define ERH_F16 0
#define ERH_F32 1

/* BM25 variants; numbering = `bm25_type` of the reference config (src/configs/easyrag.yaml:21) */
#define ERH_BM25_OKAPI 0   /* rank_bm25.BM25Okapi: float64 accumulation */
#define ERH_BM25_BM25S 1   /* bm25s lucene:        float32 accumulation */

/* dense result modes */
#define ERH_DENSE_EXACT 0  /* candidates from the fp32 MFMA scan, re-scored and ranked in pinned fp64 */
#define ERH_DENSE_FAST  1  /* ranked by the fp32 MFMA score itself */

int         erh_version(void);
const char *erh_status_str(int status);

int         erh_create(int device, erh_handle **out);
int         erh_destroy(erh_handle *h);
const char *erh_last_error(erh_handle *h);
int         erh_sync(erh_handle *h, void *stream);

/* ---- corpus state ------------------------------------------------------------------- */

/* Chunk-embedding matrix, row-major [n x d], d % 64 == 0.  dtype ERH_F16 rows are taken as they
 * are; ERH_F32 rows are converted to fp16 on the device.  normalize != 0 L2-normalises each row
 * (in fp32) before the fp16 rounding, which is what Qdrant does at insert for Distance.COSINE.
 * The handle keeps its own device copy. */
int erh_set_dense(erh_handle *h, const void *x, int64_t n, int d, int dtype, int is_device_ptr, int normalize);

/* Read rows [row0, row0 + rows) of the stored chunk matrix back in the CALLER's order (fp16, as stored): what
 * HipVectorStore.save() writes, so that a restarted process reloads the matrix instead of re-embedding the corpus -- the
 * reference skips ingestion when its Qdrant collection is already populated (src/easyrag/pipeline/pipeline.py:138-141). */
int erh_get_dense_rows(erh_handle *h, int64_t row0, int64_t rows, void *out_f16, int out_is_device);

/* Inverted postings with precomputed ("eager") per-posting scores, CSR by term:
 *   indptr  int64[V+1]; doc_ids int32[nnz] strictly ascending inside each term;
 *   payload float32[nnz] (ERH_BM25_BM25S) or float64[nnz] (ERH_BM25_OKAPI).
 * score(q, doc) = sum over the query's term ids, in query order, repeats included, of the
 * payload of (term, doc); accumulated in the payload's type.  All pointers are host pointers. */
int erh_set_bm25_csr(erh_handle *h, int variant, int64_t V, int64_t N, int64_t nnz,
                     const int64_t *indptr, const int32_t *doc_ids, const void *payload);

/* Same postings, payload evaluated on the GPU from term frequencies:
 *   BM25S : float32  idf[t] * ( tf / ( f32(k1*((1-b) + b*dl/avgdl)) + tf ) )
 *   OKAPI : float64  idf[t] * ( tf*(k1+1) / ( tf + k1*((1-b) + b*dl/avgdl) ) )
 * idf is float32[V] (BM25S) or float64[V] (OKAPI), doc_len int32[N].  Bit-identical to the
 * host evaluation (operation order as rank_bm25 / bm25s, no contraction). */
int erh_set_bm25_tf(erh_handle *h, int variant, int64_t V, int64_t N, int64_t nnz,
                    const int64_t *indptr, const int32_t *doc_ids, const int32_t *tf,
                    const int32_t *doc_len, const void *idf, double avgdl, double k1, double b);

/* Index build on the device (BM25Retriever.__init__, retrievers.py:94-118, minus the tokeniser): the corpus as one
 * flat stream of term ids, token_ids int32[n_tokens] in document order, with doc_len int32[N] tokens per document
 * (host or device pointers).  Builds the CSR postings of the selected slot -- 64-bit (term, doc) keys, radix sort,
 * run-length tf, df -- then idf / epsilon floor / avgdl exactly as the variant's library computes them and the
 * per-posting payload as erh_set_bm25_tf does.  *out_nnz = number of postings.  Bit-identical to the host builder
 * (easyrag_amd/index.py).  erh_get_bm25_csr copies the result back (any pointer may be NULL): indptr int64[V+1],
 * doc_ids / tf int32[nnz], idf float64[V] (float32 values widened for ERH_BM25_BM25S), avgdl, average_idf. */
int erh_build_bm25_index(erh_handle *h, int variant, int64_t V, int64_t N, int64_t n_tokens, const int32_t *token_ids,
                         const int32_t *doc_len, int is_device_ptr, double k1, double b, double epsilon,
                         int64_t *out_nnz);
int erh_get_bm25_csr(erh_handle *h, int64_t *indptr, int32_t *doc_ids, int32_t *tf, double *idf, double *avgdl,
                     double *average_idf);

/* A handle holds up to ERH_BM25_SLOTS independent BM25 indices (the reference pipeline builds two over the same
 * nodes: node text with embed_type 2 and know_path with embed_type 5, src/easyrag/pipeline/pipeline.py:187-210).
 * erh_bm25_select chooses the slot that erh_set_bm25_*, erh_bm25_topk, erh_bm25_scores, erh_get_bm25_payload and
 * erh_hybrid_topk act on (default 0).  Document metadata (erh_set_doc_meta) is shared by all slots. */
#define ERH_BM25_SLOTS 4
int erh_bm25_select(erh_handle *h, int slot);
/* Free the device copies (postings, skip tables, fixed-point copy) of one slot; the slot is empty afterwards.
 * (BM25Retriever's throw-away index of get_scores(query, docs), retrievers.py:131-147, lives in such a slot.) */
int erh_bm25_release(erh_handle *h, int slot);

/* Copy the payload the handle holds back to the host (float32 or float64 [nnz]); parity/debug. */
int erh_get_bm25_payload(erh_handle *h, void *out_payload);

/* Per-document metadata (host pointers, either may be NULL):
 *   content_id int32[N]: documents with equal text share an id (RRF / fusion key); NULL = identity.
 *   dir_id     int16[N]: class id for the equality filter; NULL = no filtering possible. */
int erh_set_doc_meta(erh_handle *h, int64_t N, const int32_t *content_id, const int16_t *dir_id);

/* ---- text side of the index build (host only; no device, no handle) --------------------------------------------
 * Replaces the Python side of BM25Retriever.__init__ / tokenize_and_remove_stopwords (retrievers.py:72-76, 94-118) and
 * the jieba.Tokenizer() the pipeline creates (pipeline.py:176-178).
 *
 * erh_vocab: token bytes -> term ids, ids in order of first appearance (what the Python dict loop of the shim gave:
 * the CSR built from the ids is bit-identical).
 *   erh_vocab_encode  n_docs documents; document i = bytes[doc_off[i], doc_off[i+1]), its tokens separated by the byte
 *                     `sep` (an empty range = no tokens).  add != 0: unseen tokens get the next id; add == 0: they
 *                     encode as -1 (query side: out-of-vocabulary).  Writes the ids of all documents back to back
 *                     (at most `cap`; ERH_ERR_OVERFLOW with *n_out = the number needed if it does not fit -- ids stay
 *                     assigned, call again) and the token count of each document.  out_ids may be NULL to count only.
 *   erh_vocab_token   bytes of token `id` (valid until the next erh_vocab_encode with add != 0).
 *
 * erh_cutter: sentence -> tokens with jieba 0.42.1's rules for Tokenizer.cut(sentence, cut_all=False, HMM=False): blocks
 * of [CJK 4E00-9FD5, ASCII letters / digits, + # & . _ % -] are cut along the maximum-log-probability route through the
 * dictionary DAG (runs of single ASCII letters / digits are glued back together), "\r\n" and single white-space
 * characters are tokens of their own (the reference drops ' ' afterwards, retrievers.py:74-75), every other character is
 * a token.  The dictionary is the caller's, in jieba's text format ("word freq [tag]" per line).  jieba's DEFAULT call
 * (HMM=True) additionally re-cuts runs of out-of-dictionary characters with an HMM whose tables ship with jieba: see
 * erh_cutter_set_hmm below.
 *   erh_cutter_cut    byte offsets of the token ENDS (token t = text[end[t-1], end[t]), end[-1] = 0), at most `cap`;
 *                     *n_tokens = how many there are (ERH_ERR_OVERFLOW if more than cap); out_ends may be NULL. */
typedef struct erh_vocab erh_vocab;
int     erh_vocab_create(erh_vocab **out);
int     erh_vocab_destroy(erh_vocab *v);
int64_t erh_vocab_size(const erh_vocab *v);
int     erh_vocab_encode(erh_vocab *v, const char *bytes, const int64_t *doc_off, int64_t n_docs, int sep, int add,
                         int32_t *out_ids, int64_t cap, int32_t *out_lens, int64_t *n_out);
int     erh_vocab_token(const erh_vocab *v, int32_t id, const char **bytes, int32_t *len);
typedef struct erh_cutter erh_cutter;
int     erh_cutter_create(const char *dict_text, int64_t n_bytes, erh_cutter **out);
int     erh_cutter_destroy(erh_cutter *c);
int     erh_cutter_cut(const erh_cutter *c, const char *text, int64_t n_bytes, int64_t *out_ends, int64_t cap,
                       int64_t *n_tokens);
/* jieba's default call, cut(sentence) = cut(sentence, HMM=True), regroups runs of out-of-dictionary single characters
 * with an HMM (jieba/finalseg: viterbi over the states B M E S).  Its model tables ship with jieba; the caller supplies
 * them as text, one entry per line -- "start <state> <log p>", "trans <from> <to> <log p>", "emit <state> <char> <log p>"
 * (anything absent counts as jieba's MIN_FLOAT, -3.14e100; INTEGRATION.md shows the three-line dump from an installed
 * jieba).  With a model set, erh_cutter_cut and erh_text_encode* cut as HMM=True; n_bytes == 0 removes it.
 * erh_cutter_cut_mode: hmm = 1 / 0 forces the mode for one call (1 without a model: ERH_ERR_STATE), -1 = the default. */
int     erh_cutter_set_hmm(erh_cutter *c, const char *model_text, int64_t n_bytes);
int     erh_cutter_has_hmm(const erh_cutter *c);
int     erh_cutter_cut_mode(const erh_cutter *c, const char *text, int64_t n_bytes, int hmm, int64_t *out_ends, int64_t cap,
                            int64_t *n_tokens);
/* tokenize_and_remove_stopwords + the id walk in one pass, nothing per token on the caller's side: text i =
 * bytes[text_off[i], text_off[i+1]) is cut, tokens equal to ' ' or present in `stop` (a vocabulary used as a set, may be
 * NULL) are dropped (retrievers.py:72-76), the rest is encoded through `v` as erh_vocab_encode does (add == 0: unknown
 * tokens are dropped -- the query side).  Same output convention as erh_vocab_encode. */
int     erh_text_encode(const erh_cutter *c, erh_vocab *v, const erh_vocab *stop, const char *bytes, const int64_t *text_off,
                        int64_t n_texts, int add, int32_t *out_ids, int64_t cap, int32_t *out_lens, int64_t *n_out);
/* the same on n_threads host threads (add != 0; otherwise, and for fewer than 2 n_threads texts, one thread): contiguous
 * chunks of texts, chunk vocabularies merged in order -- ids identical to the one-thread call.  c and stop are only read. */
int     erh_text_encode_mt(const erh_cutter *c, erh_vocab *v, const erh_vocab *stop, const char *bytes, const int64_t *text_off,
                           int64_t n_texts, int add, int n_threads, int32_t *out_ids, int64_t cap, int32_t *out_lens,
                           int64_t *n_out);

/* ---- queries ------------------------------------------------------------------------- */

/* Dense top-k for B queries.  q is [B x d] (ERH_F16 or ERH_F32, host or device); normalize_q != 0
 * L2-normalises each query.  filter_dir: host int16[B], -1 = unfiltered, or NULL.
 * out_ids int32[B*k], out_scores float64[B*k], out_len int32[B] (all host or all device). */
int erh_dense_topk(erh_handle *h, const void *q, int q_dtype, int q_is_device, int normalize_q,
                   int B, int k, const int16_t *filter_dir, int mode,
                   int32_t *out_ids, double *out_scores, int32_t *out_len, int out_is_device, void *stream);

/* BM25 top-k.  Queries as CSR of term ids (host): q_indptr int32[B+1], q_tok int32[q_indptr[B]].
 * Documents with score <= 0 are never returned (retrievers.py:195-196), so out_len[b] <= k. */
int erh_bm25_topk(erh_handle *h, const int32_t *q_indptr, const int32_t *q_tok, int B, int k,
                  const int16_t *filter_dir,
                  int32_t *out_ids, double *out_scores, int32_t *out_len, int out_is_device, void *stream);

/* Dense score vector of one query (host int32[n_tok]) over all N documents: float64[N] on the host
 * (float32 sums widened exactly for ERH_BM25_BM25S).  Mirrors BM25Retriever.get_scores. */
int erh_bm25_scores(erh_handle *h, const int32_t *q_tok, int n_tok, double *out_scores);

/* Reciprocal-rank fusion of two rank lists per query (list a first: the reference passes
 * [sparse, dense]).  ids_x int32[B*depth_x] (entries >= len_x[b] ignored).  Key = content_id;
 * score = sum over occurrences of 1/(rank + K), rank 1-based inside each list, float64, added in
 * list order; ties keep first-seen order; returned id = the LAST document seen for that content. */
int erh_rrf(erh_handle *h, const int32_t *ids_a, const int32_t *len_a, int depth_a,
            const int32_t *ids_b, const int32_t *len_b, int depth_b, int B, int K, int topk,
            int32_t *out_ids, double *out_scores, int32_t *out_len, int io_is_device, void *stream);

/* Simple-merge fusion: de-duplicate by content keeping the FIRST occurrence, stable sort by the
 * raw route score (float64) descending, first topk. */
int erh_fusion(erh_handle *h, const int32_t *ids_a, const double *scores_a, const int32_t *len_a, int depth_a,
               const int32_t *ids_b, const double *scores_b, const int32_t *len_b, int depth_b, int B, int topk,
               int32_t *out_ids, double *out_scores, int32_t *out_len, int io_is_device, void *stream);

/* Dual route fused on the device: BM25 top-k_sparse and dense top-k_dense (ERH_DENSE_EXACT),
 * then RRF([sparse, dense], K) -> topk, without a host round trip between the stages.
 * filter_sparse / filter_dense: host int16[B] class per query (-1 = unfiltered) or NULL, one per route -- the
 * reference pushes filter_dict to the sparse retriever and filters to the dense one independently
 * (retrievers.py:278, 283). */
int erh_hybrid_topk(erh_handle *h, const void *q, int q_dtype, int q_is_device, int normalize_q,
                    const int32_t *q_indptr, const int32_t *q_tok, int B,
                    int k_dense, int k_sparse, int K, int topk, const int16_t *filter_sparse,
                    const int16_t *filter_dense,
                    int32_t *out_ids, double *out_scores, int32_t *out_len, int out_is_device, void *stream);

/* ---- multi-GPU: query batch sharded over ranks, corpus replicated, one all-gather of the fused top-k ----------
 * (north_star / SURVEY.md section 8(e); the reference is single-process: src/main.py:48-52.)  One process and one
 * handle per GPU.  Rank r owns the contiguous shard [lo, hi) of the n_queries global queries, shards differing by at
 * most one query: base = n / world, rem = n % world, lo = r*base + min(r, rem), size = base + (r < rem).
 *
 * erh_comm_unique_id: rank 0 obtains the 128-byte RCCL id and the caller distributes it (any side channel);
 * erh_comm_init: every rank joins (collective call); RCCL is bound at run time (dlopen "librccl.so.1").
 * erh_allgather_topk: DEVICE buffers.  Packs this rank's [b_local x k] result into rows
 * [k x f64 score | k x i32 id | i32 len], runs ncclAllGather on `stream`, and unpacks the world's rows into the
 * global [n_queries x k] arrays in query order -- no host round trip, no allocation after the first call.
 * world == 1 (no erh_comm_init) degenerates to a device copy.
 * erh_pack_topk / erh_unpack_topk / erh_topk_row_bytes expose the two kernels for callers that run the
 * collective themselves (easyrag_amd/dist.py uses them around torch.distributed.all_gather_into_tensor). */
int erh_comm_unique_id(void *out128);
int erh_comm_init(erh_handle *h, int rank, int world, const void *id128);
int erh_comm_destroy(erh_handle *h);
int erh_allgather_topk(erh_handle *h, const int32_t *ids, const double *scores, const int32_t *lens, int b_local, int k,
                       int n_queries, int32_t *out_ids, double *out_scores, int32_t *out_len, void *stream);
int erh_topk_row_bytes(int k);
int erh_pack_topk(erh_handle *h, const int32_t *ids, const double *scores, const int32_t *lens, int b_local, int k,
                  int rows, void *out_rows, void *stream);
int erh_unpack_topk(erh_handle *h, const void *gathered_rows, int n_queries, int world, int k,
                    int32_t *out_ids, double *out_scores, int32_t *out_len, void *stream);

/* ---- measurement / diagnostics --------------------------------------------------------- */

/* Kernel timing with HIP events on the launch stream.  Kernel classes: */
#define ERH_K_DENSE_SCAN   0   /* MFMA scan of the chunk matrix (all stages of a call) */
#define ERH_K_DENSE_SELECT 1   /* threshold / candidate selection + fp64 re-score */
#define ERH_K_BM25_SCAN    2   /* posting scatter-add + running top-k */
#define ERH_K_BM25_MERGE   3
#define ERH_K_FUSE         4   /* RRF / fusion */
#define ERH_K_DENSE_SAMPLE 5   /* sample pass of the dense scan (batches padded to >= 512 queries): one tile per chunk stream scored WITHOUT
                                  thresholds to draw the threshold sample; the rows are scanned again by the main launch, so it books no
                                  algorithmic work -- its own class since round 5, so that ERH_K_DENSE_SCAN at those batch sizes is ONE kernel
                                  whose per-launch figures are what rocprofv3 lists for it */
#define ERH_K_COUNT        6
int erh_set_profiling(erh_handle *h, int enable);
/* Sum of event-measured milliseconds and number of launches since the last reset. */
int erh_get_kernel_time(erh_handle *h, int kernel_class, double *total_ms, int64_t *launches);
/* Algorithmic work booked for the same launches: bytes = what the kernel must read at least (chunk rows
 * x d x 2 + query block for the dense scan; 8 or 12 bytes per posting touched for BM25), flops = 2*rows*B*d. */
int erh_get_kernel_work(erh_handle *h, int kernel_class, double *bytes, double *flops);
int erh_reset_kernel_time(erh_handle *h);

/* Tuning knobs: name/value pairs.  None of them changes a result (except the measurement-only ones, which say so).
 *   dense_n0 (32768)      rows of the stored prefix scored densely to seed the pruning thresholds (max 32768)
 *   dense_speculate (1)   first threshold = the rank-r prefix score, r << k (an estimate of the corpus' k-th best from the
 *                         prefix being an even sample; verified per query by the final kernel, exhaustive path if wrong),
 *                         one scan stage; 0 = guaranteed bounds refined in stages (dense_n1 ...)
 *   dense_n1 (131072)     dense_speculate 0: first refinement boundary; 0 = never refine.  Further boundaries follow x4 while 8x fits.
 *   dense_n1_auto (1)     snap the boundaries to whole rounds of the persistent scan
 *   dense_n0_auto (0)     shrink the seed prefix to where the rest is a whole number of scan rounds
 *   dense_tile384 (1)     batches padded to >= 512 queries: scan on a 384 x 256 tile over tiled copies of both operands
 *                         (dense_scan_pp5_kernel; the 384-row copy of the chunk matrix, + N * d * 2 bytes, is built on first use);
 *                         0 = the 256 x 256 tile for every batch size
 *   dense_tile384_max_mb (-1)  test hook: a 384-row copy larger than this many MiB is refused as if its allocation had failed
 *                         (the 256 x 256 scan then serves every batch; -1 = no limit); the per-dir block copies (dense_dir_blocks)
 *                         obey the same bound and fall back the same way (to the filter column)
 *   dense_selfseed (1)    batches padded to >= 512 queries: the scan kernel draws the threshold sample itself (a pass without
 *                         thresholds over one tile per chunk stream, the two best scores of every 64-row cell) and then scans
 *                         all rows; 0 = store kernel + S0 + seed select for every batch size
 *   n_cus (0)             CUs the persistent grids are sized for; set it when the caller's stream is CU-masked
 *                         (hipExtStreamCreateWithCUMask), 0 = all CUs of the device
 *   dense_shuffle (1)     golden-ratio row placement of the chunk matrix (takes effect at the next erh_set_dense);
 *                         keep it on for corpora sorted by document or topic
 *   dense_pp (3)          ping-pong persistent append scan: 3 = strict alternation (fragment reads inside the matrix segment),
 *                         2 = lean-issue kernel, 1 = the round-1 kernel; 0 = lock-step kernels (dense_persist 1 / 0 = persistent /
 *                         one workgroup per tile, dense_cfg 0..2 = their tile configuration, dense_readahead)
 *   dense_tiled (0)       keep a tiled, pre-swizzled copy of the chunk matrix for the ping-pong scan (+ N * d * 2 bytes; takes
 *                         effect at the next erh_set_dense).  Off: within the run-to-run noise on this workload
 *   dense_var, dense_rot, dense_sync   schedule variants of the ping-pong kernels (measured, off: see DESIGN.md, dead ends)
 *   dense_gemv (1)        batches of at most 16 queries stream the chunk matrix through a 16x16x32 skinny-GEMM kernel
 *                         (the reference's one-query-at-a-time call pattern) instead of the padded 256-query scan
 *   dense_scan_nt (0)     256 x 256 ping-pong scan at one query tile per matrix (<= 256 queries, the grouped launch): chunk-side LDS-DMA with
 *                         the non-temporal hint.  Measured +2 % (256 queries), +5.5 % (128), +0.8 % (grouped): off; a parity arm
 *   dense_fin_wgs (4)     workgroups of the final kernel per CU (4: what its 64 VGPRs and 36 KiB of LDS allow; 3 / 2: A/B arms -- select time
 *                         per 1024 queries 0.177 / 0.194 / 0.200 ms at 4 / 3 / 2).  Process-wide.  Same results
 *   dense_gemv_nt (-1)    that stream's chunk loads with the non-temporal hint: -1 on for calls of up to 32 queries unless the sparse
 *                         route of a fused call runs beside it (hybrid_overlap 1), 0 off, 1 on.  Same results
 *   bm25_ascan (1)        fixed-point BM25 scan + exact re-score: the postings are scattered into integer LDS sums in any order
 *                         (one 16-byte load and two integer atomics per lane), documents whose sum can still reach the
 *                         running k-th best stay on a list, and the final list is scored exactly (binary search per
 *                         document and token, library summation order).  Needs an index whose payloads are all positive
 *                         normal numbers, else the kernels below run; keeps an interleaved copy of the postings (8
 *                         bytes each), built at the next erh_set_bm25_* / erh_build_bm25_index.  0 = off
 *   hybrid_overlap (-1)   erh_hybrid_topk: the sparse route on a side stream -- 1 = beside the whole dense pipeline, 2 = forked
 *                         behind the dense scan (beside the selection kernels) -- joined before the fusion; 0 = one stream;
 *                         -1 = by batch size: 1 up to 256 queries, where neither scan fills the chip (one query: 0.59 -> 0.54
 *                         ms per call, 64: 0.85 -> 0.77), 0 above (1024 queries: 3.25 / 3.31 against 2.95 ms -- the scans need
 *                         a whole CU's LDS each, so they time-slice instead of sharing)
 *   bm25_small (2)        shape of the fixed-point scan for batches of >= 8 queries and k <= 384: 2 = 512 threads, 32768-document
 *                         tiles, two documents per accumulator word (16-bit sums, payloads shifted per query); 1 = 512 threads,
 *                         16384-document tiles, 32-bit sums (both: 80 KiB of LDS, two workgroups = two queries per CU);
 *                         0 = always 1024 threads, 32768-document tiles.  Same results, bit for bit
 *   bm25_dir_range (1)    fixed-point scan with a dir filter: the query walks only the posting tiles that hold documents of its class
 *                         (erh_set_doc_meta records every class's first and last document; the reference's dirs are contiguous blocks
 *                         of its document order, so a filter on one of four dirs skips three quarters of the tile passes); 0 = all tiles
 *   dense_dir_blocks (1)  dense route with a dir filter: queries scan a copy of their dir's chunks (the dir's documents in ascending order
 *                         -- one run in the reference's dir-by-dir layout, gathered from anywhere otherwise --, own row placement, built on
 *                         the first filtered call, + 2 d bytes per chunk of a dir that gets a block) instead of the whole matrix with a
 *                         filter column.  The batch is grouped by dir on the host; two or more groups run as ONE launch per stage
 *                         (dense_group_launch: a table with one entry per 256-query tile -- block, placement, seed prefix, chunk streams --
 *                         read by every kernel of the pipeline; any number of dirs), a single group (one filtered query per call, the
 *                         reference's pattern) and the queries without a block as pipelines of their own; results come back in the
 *                         caller's order and numbering; the pipelines' flag words are read together by erh_dense_check / the host-output
 *                         copy (a group with a flagged query is run again to the end).
 *                         1 = where the routed work is smaller than the whole-matrix work, in rows x query columns with columns below
 *                         dense_route_ridge counted as the ridge (a scan that narrow is bound by the matrix bytes);
 *                         2 = whenever the batch has a dir with a block; 0 = always the filter column.  Same results
 *   dense_group_launch (1)  0: every block group of a batch as a pipeline of its own (the round-5 path; a parity arm)
 *   dense_group_sample (1)  grouped launch: thresholds from a sample pass of the scan kernel per view (where every view of the batch can give
 *                         one) instead of store kernel + seed scores + seed select; 0: always the latter.  Same results
 *   dense_route_ridge (160) query columns below which a dense scan is HBM-bound on this chip (the route decision's only constant)
 *   dense_dir_block_min_rows (4096)  smallest dir that gets a block of its own
 *   bm25_long_tokens (28) packed shape: a batch whose longest query has MORE tokens than this scans with 32-bit sums (the bm25_small = 1
 *                         shape) -- 16-bit sums leave a query of nq tokens 65535 / nq payload levels and an error bound of 3 nq units, and
 *                         from ~30 tokens on the list of documents that can still reach the top k stops shrinking (the query then falls
 *                         back to the exact block scan: 0.5 -> 1.7 ms per 1024 queries with the reference's question lengths, 4 ... 45
 *                         tokens; 0.68 ms on the 32-bit shape).  0 = always the packed shape.  Same results
 *   bm25_mixed (1)        such a batch runs as ONE launch whose workgroups pick their body by the length of their query: the 32-bit body for
 *                         the queries longer than bm25_long_tokens, the packed one for all others (needs bm25_post16 and a skip table at
 *                         16384 documents); 0 = the whole batch on the 32-bit shape.  Same results
 *   bm25_long_segs (4)    mixed launch of >= 512 queries (one workgroup per query): a LONG query's documents are cut into this many ranges, each
 *                         a workgroup of its own, merged afterwards (one 45-token question in one workgroup is the launch's tail); 1 = no cut
 *   bm25_split_finish (0) 1: the exact re-score + rank of the scan's final lists as ONE batch-wide kernel behind the scan instead of
 *                         each workgroup's tail (measured +15 %: the tail overlaps the CU's other workgroup; a parity arm)
 *   bm25_post16 (1)       packed shape: read 4-byte postings {15-bit document offset in the tile, 16-bit payload} (built when an
 *                         index is set while bm25_small = 2; + 4 bytes per posting); 0 = the 8-byte fixed-point postings
 *   bm25_crossing (1)     wave-owned scan: survivors from threshold crossings noted in the token loop instead of a sweep
 *                         over the accumulators (1 = fp32 sums only, 2 = fp64 too, 0 = always sweep); indices with a
 *                         non-positive payload always sweep
 *   bm25_lpt (1)          launch the queries of a batch in order of decreasing posting volume (shorter tail of the scan)
 *   bm25_segs (0)         document-range segments per query (0 = enough for >= 512 workgroups)
 *   bm25_wscan (0)        when bm25_ascan does not apply: wave-owned BM25 scan (no per-token workgroup barrier) for batches
 *                         whose queries have at most 64 tokens; 0 = block scan.  Needs a fine skip table (4 bytes per term
 *                         and per 2048 / 1024 documents: 0.5 GB at 1M documents x 262144 terms, growing with V * N), built
 *                         by the next erh_set_bm25_* unless it would exceed bm25_fine_max_mb (8192)
 *   comm_timeout_s (120)  bounded wait of erh_comm_init for the other ranks
 *   dense_ablate, bm25_ablate, debug_counters   MEASUREMENT BUILDS ONLY (library compiled with -DERH_MEASURE, i.e.
 *                         ERH_MEASURE=1 python -m easyrag_amd._build): variants with parts of a kernel removed /
 *                         section clocks; results are invalid while an ablate value is non-zero.  The product
 *                         build contains none of these variants and rejects non-zero values (ERH_ERR_UNSUPPORTED). */
int erh_set_option(erh_handle *h, const char *name, int64_t value);

/* REQUIRED after a dense / hybrid call with DEVICE outputs, before the rows are read or sent anywhere (erh_sync is not
 * a substitute): synchronise `stream` and read the call's flag words (host-output calls do this themselves).  Queries whose candidate
 * budgets overflowed are answered by the exhaustive path: the call itself only COUNTS them on the device (one small launch); the exact
 * rounds -- 16 queries at a time -- and, for a fused call, the RRF over the corrected lists run here, when and only when the count is
 * not zero, so that the results are complete when this returns.  (Until round 5 a call enqueued the first round itself and paid two
 * empty launches for it in the normal case.) */
int erh_dense_check(erh_handle *h, void *stream);

/* Measurement only: with option "debug_counters" = 1 the scan kernels add per-section shader-clock sums
 * (thread 0 of every workgroup) into 16 counters; this reads and clears them. */
int erh_debug_counters(erh_handle *h, uint64_t *out16);

/* Diagnostics of the last dense EXACT call: max |fp64 - fp32| over re-scored candidates, the
 * margin used, and the number of queries whose exactness certificate failed. */
int erh_dense_diag(erh_handle *h, double *max_abs_err, double *margin, int32_t *uncertified);

/* Queries of the last dense / hybrid call (as of its erh_dense_check or host-output return) that the pruned
 * pipeline could not certify -- candidate list, gather or re-score budget exhausted, e.g. tens of thousands of
 * near-duplicate chunks around the k-th score -- and that were answered by the exhaustive path instead (exact fp64
 * score of every chunk, streaming top-k; same results contract).  Normally 0. */
int erh_dense_exhaustive_count(erh_handle *h, int32_t *count);

/* Which kernels answered the calls since erh_create / erh_reset_stats (a record that proves its own path: tests assert that
 * the kernel under test ran, bench.py reports that no query of the timed steps left the pruned pipeline).  Counters:
 *   dense_calls / bm25_calls / hybrid_calls           entry-point calls
 *   dense_scan_pp5_launches                            dense scan launches on the 384 x 256 tile (dense_scan_pp5_kernel)
 *   dense_scan_pp3_launches                            ... on the 256 x 256 ping-pong tile (main scans; sample passes are counted apart)
 *   dense_scan_gemv_launches / dense_scan_tile_launches  ... on the skinny-GEMM stream / the per-tile fallback kernels
 *   dense_sample_passes                                threshold samples drawn by the scan kernel itself (dense_selfseed)
 *   dense_tile384_nomem                                times the 384-row copy of the chunk matrix did not fit and the 256 x 256 scan took over
 *   dense_exhaustive_queries                           queries answered by the exhaustive path (device counter; summed over calls)
 *   dense_block_groups                                 query groups answered from their dir's block (dense_dir_blocks)
 *   dense_grouped_launches                             calls whose block groups ran as one launch per stage (dense_group_launch)
 *   dense_candidates_last_call                         candidates the last dense pipeline's scan handed to its final kernel, summed over
 *                                                      its queries (read back from the device: synchronises; not reset by erh_reset_stats)
 *   bm25_mixed_launches                                BM25 scans launched with both bodies (bm25_mixed: a batch with queries longer than bm25_long_tokens)
 *   bm25_redo_segments                                 (query, segment) pairs the fixed-point scan handed to the exact block scan (device counter)
 * Reading a device counter synchronises the device.  Unknown name: ERH_ERR_INVALID. */
int erh_get_stat(erh_handle *h, const char *name, int64_t *value);
int erh_reset_stats(erh_handle *h);

/* Pure function (no handle, no device): the rank of the seed-prefix score that erh_dense_topk takes as its first pruning
 * threshold for top-k over n chunks with a prefix of n0 -- k itself (a guaranteed bound) or, with option dense_speculate,
 * ceil(mu + 6.5 sqrt(mu) + 3) < k with mu = k * n0 / n: the prefix is an even sample of the corpus, the number of true
 * top-k members inside it is ~Binomial(k, n0 / n), and `rank` or more of them land there with probability < 1e-7
 * (tests/test_speculation_rank.py); every query's threshold is verified on the device anyway. */
int erh_dense_seed_rank(int k, int64_t n0, int64_t n);

/* Debug: plain (non-MFMA) fp32 scores of B fp16 queries against rows [row0, row0+rows) of the stored
 * matrix, out float32[B*rows] on the host; and the MFMA scores of the same block. */
int erh_debug_dense_scores(erh_handle *h, const void *q_f16_host, int B, int64_t row0, int rows,
                           int use_mfma, float *out);

#ifdef __cplusplus
}
#endif
#endif /* EASYRAG_HIP_H */
