// Sparse route, kernels K2/K3: BM25 scored over CSR inverted postings.  Replaces BM25Retriever.get_scores + filter
// (/root/reference/src/easyrag/custom/retrievers.py:128-151, 191-210), whose arithmetic is
// rank_bm25.BM25Okapi.get_scores (float64) or bm25s.BM25.get_scores (float32) -- SURVEY.md A.1/A.2.
//
// Bit parity rule: a document's per-term contributions are added in query-token order, repeats included, in the library's
// accumulation type.  Postings carry the precomputed per-(term, doc) contribution ("eager" payload, as bm25s stores it; for
// Okapi the same thing in float64), built on the host or by bm25_payload_kernel below.  Two ways to honour the rule:
//   - DEFAULT (bm25_ascan_kernel, since round 3): the scan only GENERATES candidates -- postings are scattered into
//     fixed-point integer sums in LDS with ds_add_rtn_u32 in any order (integer sums do not depend on it), a document is
//     listed at the one add that takes its sum over the threshold, and the short final list (k + the near ties of the k-th)
//     is then scored exactly: per (document, token) one binary search in the posting list, payloads summed in token order in
//     the library's type.  ids and scores equal the order-keeping scans' bit for bit.
//   - FALLBACK / parity arms (bm25_scan_kernel, bm25_wscan_kernel): a workgroup owns one query and walks document tiles; a
//     tile's accumulators live in LDS (32768 fp32 / 16384 fp64 sums); query tokens are applied one after the other (block
//     scan: barrier in between; wave-owned scan: each wave owns a sub-range of the tile), so plain LDS read-add-write by the
//     thread that holds the posting is race free; the tile is swept once and entries that beat the running k-th best
//     (score desc, index asc; score > 0 only, retrievers.py:195-196; optional dir filter, retrievers.py:198-202) are
//     compacted into an LDS candidate list that is re-sorted and cut to k whenever it fills.  Indices with a non-positive
//     payload always take this path.
// Per-term tile boundaries come from a skip table tile_off[term][tile] built once per index, so a tile touches exactly its
// postings (algorithmic bytes = 8 or 12 per posting touched).
#include <algorithm>
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBmThreads = 1024;
constexpr int kBmCap = 2048;       // LDS candidate slots (k <= 1024 so that k + one sweep chunk always fits)
constexpr int kBmTokChunk = 192;   // query tokens whose tile ranges are staged at once

// ---- index-time kernels ---------------------------------------------------------------------------
__global__ void bm25_tile_off_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids,
                                     int64_t V, int tile_docs, int n_tiles, int32_t *__restrict__ tile_off) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = V * (n_tiles + 1);
    if (t >= total) return;
    const int64_t term = t / (n_tiles + 1);
    const int tile = (int)(t % (n_tiles + 1));
    const int64_t s = indptr[term], e = indptr[term + 1];
    const int64_t target = (int64_t)tile * tile_docs;    // first posting with doc >= target
    int64_t lo = s, hi = e;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)doc_ids[mid] < target) lo = mid + 1; else hi = mid;
    }
    tile_off[t] = (int32_t)(lo - s);
}

// payload[p] for posting p of term t, document doc:
//   BM25S  (float32, bm25s lucene):  idf32[t] * ( tf / ( f32( k1*((1-b) + b*dl/avgdl) ) + tf ) )
//   OKAPI  (float64, rank_bm25):     idf64[t] * ( tf*(k1+1) / ( tf + k1*((1-b) + b*dl/avgdl) ) )
// Same operation order as the libraries; IEEE divide; contraction disabled for this file.
template <bool OKAPI>
__global__ void bm25_payload_kernel(int64_t V, int64_t nnz, const int64_t *__restrict__ indptr,
                                    const int32_t *__restrict__ doc_ids, const int32_t *__restrict__ tf,
                                    const int32_t *__restrict__ doc_len, const void *__restrict__ idf_v,
                                    double avgdl, double k1, double b, void *__restrict__ payload_v) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    // term of posting p: last t with indptr[t] <= p
    int64_t lo = 0, hi = V;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (indptr[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int64_t term = lo;
    const double dl = (double)doc_len[doc_ids[p]];
    const double bracket = (1.0 - b) + (b * dl) / avgdl;
    if (OKAPI) {
        const double tfd = (double)tf[p];
        const double num = tfd * (k1 + 1.0);
        const double den = tfd + k1 * bracket;
        const double idf = reinterpret_cast<const double *>(idf_v)[term];
        reinterpret_cast<double *>(payload_v)[p] = idf * (num / den);
    } else {
        const float tff = (float)tf[p];
        const float br = (float)(k1 * bracket);
        const float tfc = tff / (br + tff);
        const float idf = reinterpret_cast<const float *>(idf_v)[term];
        reinterpret_cast<float *>(payload_v)[p] = idf * tfc;
    }
}

// ---- query-time scan -------------------------------------------------------------------------------
constexpr int kBmPre = 2;          // postings per thread per (token, tile) held in prefetch registers
constexpr int kBmAhead = 4;        // tokens fetched ahead of the one being applied (5 rotating register sets)

template <typename ST>
struct BmLds {
    static constexpr int TILE = (sizeof(ST) == 4) ? erh::kBm25TileF32 : erh::kBm25TileF64;
    // layout (bytes): [0,64) header | acc TILE*ST | cand_s CAP*ST | cand_i CAP*4 | lo 2*CHUNK*8 | hi 2*CHUNK*8
    static constexpr size_t OFF_ACC = 64;
    static constexpr size_t OFF_CS = OFF_ACC + (size_t)TILE * sizeof(ST);
    static constexpr size_t OFF_CI = OFF_CS + (size_t)kBmCap * sizeof(ST);
    static constexpr size_t OFF_LO = OFF_CI + (size_t)kBmCap * 4;
    static constexpr size_t OFF_HI = OFF_LO + (size_t)2 * kBmTokChunk * 8;   // lo / hi: two buffers (this tile, next tile)
    static constexpr size_t BYTES = OFF_HI + (size_t)2 * kBmTokChunk * 8;
    static_assert(BYTES <= 160 * 1024, "BM25 scan LDS layout exceeds one CU");
};

struct BmHdr {
    int ncand;      // live entries in the candidate list
    int total;      // scratch for block-wide counts
    int tau_idx;    // running k-th best: index part
    int pad;
    double tau_s;   // running k-th best: score part (0 => "score > 0" is the only condition)
    int full[3];    // wave-owned scan: "list was full during sweep pass p" (p mod 3), see bm25_wscan_kernel
    int want[3];    // wave-owned scan: "list grew past k + 512 during sweep pass p"
    float theta;    // approximate-order scan: k-th best approximate sum seen so far (tau_s holds theta * (1 - margin))
    int redo;       // approximate-order scan: near-tie flood, the query goes to the exact block scan
};
static_assert(sizeof(BmHdr) <= 64, "BmHdr must fit the 64-byte LDS header");

template <typename ST>
struct BmPre {
    int32_t d[kBmPre];
    ST v[kBmPre];
};

template <typename ST>
__device__ __forceinline__ void bm_prefetch(BmPre<ST> &P, const int32_t *__restrict__ doc_ids,
                                            const ST *__restrict__ payload, int64_t lo, int64_t hi, int tid) {
#pragma unroll
    for (int u = 0; u < kBmPre; ++u) {
        const int64_t p = lo + tid + (int64_t)u * kBmThreads;
        const bool ok = p < hi;
        P.d[u] = ok ? doc_ids[p] : -1;
        P.v[u] = ok ? payload[p] : (ST)0;
    }
}

// Apply one token to the tile: this thread's prefetched postings, then (lists longer than kBmPre*1024
// inside this tile) the rest straight from memory.  Each document occurs at most once per term, so the
// read-add-write needs no atomics; the caller's barrier orders token j before token j+1.
template <typename ST>
__device__ __forceinline__ void bm_apply(const BmPre<ST> &P, ST *acc, int64_t base_doc,
                                         const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
                                         int64_t lo, int64_t hi, int tid) {
#pragma unroll
    for (int u = 0; u < kBmPre; ++u) {
        if (P.d[u] >= 0) {
            const int slot = (int)((int64_t)P.d[u] - base_doc);
            acc[slot] = acc[slot] + P.v[u];
        }
    }
#pragma unroll 4
    for (int64_t p = lo + tid + (int64_t)kBmPre * kBmThreads; p < hi; p += kBmThreads) {
        const int slot = (int)((int64_t)doc_ids[p] - base_doc);
        acc[slot] = acc[slot] + payload[p];
    }
}

// Thresholds are compared in the accumulation type: every threshold value originates from a score (or a
// float lower bound) of that type, so the conversion from the header's double is exact.
template <typename ST>
__device__ __forceinline__ bool bm_pass(ST s, int64_t doc, ST tau_s, int tau_idx) {
    return (s > tau_s) || (s == tau_s && doc < (int64_t)tau_idx);
}

// Sort the candidate list (score desc, idx asc), cut to k, refresh the running threshold.  Uniform call.
// hdr->ncand may exceed kBmCap by the failed reservations of a full list; only the first kBmCap slots are real.
template <typename ST>
__device__ __forceinline__ void bm_shrink(BmHdr *hdr, ST *cs, int32_t *ci, int k) {
    __syncthreads();
    const int n = hdr->ncand < kBmCap ? hdr->ncand : kBmCap;
    const int np2 = erh_next_pow2(n < 2 ? 2 : n);                      // sort only what is there
    for (int i = n + (int)threadIdx.x; i < np2; i += kBmThreads) { cs[i] = (ST)-1; ci[i] = 0x7fffffff; }
    erh_bitonic_rec_desc<ST>(cs, ci, np2);
    if (threadIdx.x == 0) {
        hdr->ncand = n < k ? n : k;
        if (n >= k) {
            const double ts = (double)cs[k - 1];
            if (ts > hdr->tau_s || (ts == hdr->tau_s && ci[k - 1] < hdr->tau_idx)) { hdr->tau_s = ts; hdr->tau_idx = ci[k - 1]; }
        }
    }
    __syncthreads();
}

// Sweep pass over the tile accumulators (16-byte LDS accesses).  Entries that cannot reach the top k are cleared;
// survivors are moved to the candidate list straight away.  When the list is full the survivor stays in its
// accumulator and hdr->total is raised: the caller shrinks the list (which tightens the threshold) and sweeps the
// leftovers again.  The common case after the first tiles is "touched but nowhere near the threshold": one max,
// one compare and one zero store per 16-byte vector.
// The pass covers accumulators [i_begin, i_end) with `nthr` threads (ltid = this thread's rank among them): the whole
// tile with the 1024 threads of the block scan, or one wave's own sub-range in the wave-owned scan.  *full_flag is
// raised when the list is full; *want_flag (may be null) when it has grown past want_at entries.
template <typename ST, int CAPT = kBmCap>
__device__ __forceinline__ void bm_sweep(BmHdr *hdr, ST *acc, ST *cs, int32_t *ci, int i_begin, int i_end, int nthr,
                                         int ltid, int64_t base_doc, int64_t N, int fd,
                                         const int16_t *__restrict__ dir_id, ST tau_s, int tau_idx, int *full_flag,
                                         int *want_flag, int want_at) {
    constexpr int VEC = 16 / (int)sizeof(ST);
    typedef ST VT __attribute__((ext_vector_type(VEC)));
    typedef uint32_t UT __attribute__((ext_vector_type(4)));
    constexpr int UNR = 4;                                                  // vectors per thread per iteration (8 would spill at 128 VGPRs)
    // Common case per 16-byte vector: read, OR of the words (touched?), max, one compare, predicated zero store --
    // no jump.  Only a wave that holds a possible survivor (ballot) enters the per-element path.
    // All reads of an iteration are issued BEFORE the first store: a store to the accumulators between two reads
    // cannot be reordered by the compiler (same array), and a read-wait-store chain per vector costs one LDS latency
    // each -- eight per tile with sixteen waves queueing on the LDS.
    for (int i0 = i_begin + ltid * VEC; i0 < i_end; i0 += nthr * VEC * UNR) {
        VT v[UNR];
        bool cand[UNR], touched[UNR];
        bool any = false;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * nthr * VEC;
            if (i < i_end) v[u] = *reinterpret_cast<VT *>(acc + i);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * nthr * VEC;
            cand[u] = false;
            touched[u] = false;
            if (i < i_end) {
                const UT bits = *reinterpret_cast<const UT *>(&v[u]);
                touched[u] = (bits[0] | bits[1] | bits[2] | bits[3]) != 0u;
                ST m = v[u][0];
#pragma unroll
                for (int e = 1; e < VEC; ++e) m = v[u][e] > m ? v[u][e] : m;
                cand[u] = touched[u] && !(m < tau_s);                       // something here may reach the top k
                any |= cand[u];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (touched[u] && !cand[u]) {
                VT z;
#pragma unroll
                for (int e = 0; e < VEC; ++e) z[e] = (ST)0;
                *reinterpret_cast<VT *>(acc + i0 + u * nthr * VEC) = z;
            }
        }
        if (__builtin_amdgcn_ballot_w64(any) == 0) continue;                // wave-uniform
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (!cand[u]) continue;
            const int i = i0 + u * nthr * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const ST sv = v[u][e];
                if (sv != (ST)0) {
                    const int64_t doc = base_doc + i + e;
                    bool pass = bm_pass<ST>(sv, doc, tau_s, tau_idx);
                    if (pass && (doc >= N || (fd >= 0 && (int)dir_id[doc] != fd))) pass = false;
                    if (pass) {
                        const int pos = atomicAdd(&hdr->ncand, 1);
                        if (pos < CAPT) {
                            cs[pos] = sv;
                            ci[pos] = (int32_t)doc;
                            v[u][e] = (ST)0;
                            if (want_flag && pos >= want_at) *want_flag = 1;
                        } else {
                            *full_flag = 1;                                 // list full: keep it for the next sweep
                        }
                    } else {
                        v[u][e] = (ST)0;
                    }
                }
            }
            *reinterpret_cast<VT *>(acc + i) = v[u];
        }
    }
}

// grid = (segs, B), block = 1024.  Segment `seg` of query q walks tiles [n_tiles*seg/segs, n_tiles*(seg+1)/segs).
template <typename ST>
__global__ __launch_bounds__(kBmThreads) void bm25_scan_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
    const int32_t *__restrict__ tile_off, int n_tiles, int64_t N,
    const int32_t *__restrict__ q_indptr, const int32_t *__restrict__ q_tok,
    const int32_t *__restrict__ q_order /* workgroup y -> query: heaviest queries first, or null */, int k, int segs,
    const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    double *__restrict__ part_scores, int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len,
    const uint32_t *__restrict__ only /* null, or one word per (query, segment): scan it only if non-zero */,
    int cut_tiles, int cut_shift /* segment boundaries: (cut_tiles * seg / segs) << cut_shift (the approximate-order scan's cuts) */,
    int ablate /* measurement only: 1 no add, 2 no sweep, 4 no token barrier, 8 no posting loads */,
    unsigned long long *__restrict__ dbg /* measurement only: per-section shader-clock sums of thread 0, or null */) {
    using L = BmLds<ST>;
    if (only && !only[(int64_t)(q_order ? q_order[blockIdx.y] : (int)blockIdx.y) * segs + blockIdx.x]) return;   // workgroup-uniform
    long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_mark = dbg ? clock64() : 0;
#define ERH_SEC(I) do { if (dbg) { const long long n_ = clock64(); t_sec[I] += n_ - t_mark; t_mark = n_; } } while (0)
    constexpr int TILE = L::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BmHdr *hdr = reinterpret_cast<BmHdr *>(smem);
    ST *acc = reinterpret_cast<ST *>(smem + L::OFF_ACC);
    ST *cs = reinterpret_cast<ST *>(smem + L::OFF_CS);
    int32_t *ci = reinterpret_cast<int32_t *>(smem + L::OFF_CI);
    int64_t *s_lo = reinterpret_cast<int64_t *>(smem + L::OFF_LO);
    int64_t *s_hi = reinterpret_cast<int64_t *>(smem + L::OFF_HI);

    const int seg = blockIdx.x, q = q_order ? q_order[blockIdx.y] : (int)blockIdx.y, tid = threadIdx.x;
    const int qs = q_indptr[q], nq = q_indptr[q + 1] - qs;
    const int fd = filter_dir ? (int)filter_dir[q] : -1;
    const int t_begin = (int)((int64_t)cut_tiles * seg / segs) << cut_shift;
    int t_end = (int)((int64_t)cut_tiles * (seg + 1) / segs) << cut_shift;
    t_end = t_end < n_tiles ? t_end : n_tiles;
    const int64_t out_base = ((int64_t)q * segs + seg) * k;

    if (tid == 0) { hdr->ncand = 0; hdr->total = 0; hdr->tau_idx = -1; hdr->tau_s = 0.0; }
    for (int i = tid; i < TILE; i += kBmThreads) acc[i] = (ST)0;
    __syncthreads();

    if (nq > 0) {
        const bool single = nq <= kBmTokChunk;                        // one chunk of tokens: ranges and postings pipelined across tiles
        BmPre<ST> P0, P1, P2, P3, P4;
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int64_t base_doc = (int64_t)tile * TILE;
            ERH_SEC(7);
            // ---- scatter-add, one query token after the other, postings prefetched kBmAhead tokens ahead ----------
            // Five register sets rotate through the tokens of a tile.  When the whole query fits one chunk
            // (the normal case) the ranges of the NEXT tile are staged while this one is processed and its first
            // kBmAhead tokens are fetched before the sweep, so only the very first tile exposes the load latency.
            for (int c0 = 0; c0 < nq; c0 += kBmTokChunk) {
                const int nqc = (nq - c0 < kBmTokChunk) ? (nq - c0) : kBmTokChunk;
                const int rb = single ? (tile & 1) : 0;                   // range buffer of this tile
                int64_t *lo_c = s_lo + rb * kBmTokChunk, *hi_c = s_hi + rb * kBmTokChunk;
                const bool have_cur = single && tile > t_begin;           // staged (and prefetched) during the previous tile
                if (!have_cur) {
                    for (int j = tid; j < nqc; j += kBmThreads) {
                        const int64_t tok = q_tok[qs + c0 + j];
                        const int64_t ip = indptr[tok];
                        const int32_t *to = tile_off + tok * (n_tiles + 1) + tile;
                        lo_c[j] = ip + to[0];
                        hi_c[j] = ip + to[1];
                    }
                }
                const bool stage_next = single && tile + 1 < t_end;
                if (stage_next) {
                    int64_t *lo_n = s_lo + (rb ^ 1) * kBmTokChunk, *hi_n = s_hi + (rb ^ 1) * kBmTokChunk;
                    for (int j = tid; j < nqc; j += kBmThreads) {
                        const int64_t tok = q_tok[qs + j];
                        const int64_t ip = indptr[tok];
                        const int32_t *to = tile_off + tok * (n_tiles + 1) + tile + 1;
                        lo_n[j] = ip + to[0];
                        hi_n[j] = ip + to[1];
                    }
                }
                __syncthreads();
                ERH_SEC(0);
                if (!have_cur) {
#pragma unroll
                    for (int u = 0; u < kBmPre; ++u) {
                        P0.d[u] = P1.d[u] = P2.d[u] = P3.d[u] = P4.d[u] = -1;
                        P0.v[u] = P1.v[u] = P2.v[u] = P3.v[u] = P4.v[u] = (ST)0;
                    }
                    if (!(ablate & 8)) {
                        bm_prefetch<ST>(P0, doc_ids, payload, lo_c[0], hi_c[0], tid);
                        if (nqc > 1) bm_prefetch<ST>(P1, doc_ids, payload, lo_c[1], hi_c[1], tid);
                        if (nqc > 2) bm_prefetch<ST>(P2, doc_ids, payload, lo_c[2], hi_c[2], tid);
                        if (nqc > 3) bm_prefetch<ST>(P3, doc_ids, payload, lo_c[3], hi_c[3], tid);
                    }
                }
#define ERH_BM_STEP(CUR, NXT, J)                                                                         \
    do {                                                                                                 \
        const int j_ = (J);                                                                              \
        if (j_ + kBmAhead < nqc && !(ablate & 8))                                                        \
            bm_prefetch<ST>(NXT, doc_ids, payload, lo_c[j_ + kBmAhead], hi_c[j_ + kBmAhead], tid);       \
        const int64_t lo_ = lo_c[j_], hi_ = hi_c[j_]; /* block-uniform */                                \
        if (lo_ < hi_) {                                                                                 \
            if (!(ablate & 1)) bm_apply<ST>(CUR, acc, base_doc, doc_ids, payload, lo_, hi_, tid);        \
            if (!(ablate & 4)) __syncthreads(); /* token j complete before token j+1 */                  \
        }                                                                                                \
    } while (0)
                for (int j0 = 0; j0 < nqc; j0 += 5) {
                    ERH_BM_STEP(P0, P4, j0);
                    if (j0 + 1 >= nqc) break;
                    ERH_BM_STEP(P1, P0, j0 + 1);
                    if (j0 + 2 >= nqc) break;
                    ERH_BM_STEP(P2, P1, j0 + 2);
                    if (j0 + 3 >= nqc) break;
                    ERH_BM_STEP(P3, P2, j0 + 3);
                    if (j0 + 4 >= nqc) break;
                    ERH_BM_STEP(P4, P3, j0 + 4);
                }
#undef ERH_BM_STEP
                __syncthreads();                                      // ranges of this chunk are free; next tile's are visible
                if (stage_next && !(ablate & 8)) {                    // head of the next tile: lands during the sweep
                    const int64_t *lo_n = s_lo + (rb ^ 1) * kBmTokChunk, *hi_n = s_hi + (rb ^ 1) * kBmTokChunk;
                    bm_prefetch<ST>(P0, doc_ids, payload, lo_n[0], hi_n[0], tid);
                    if (nqc > 1) bm_prefetch<ST>(P1, doc_ids, payload, lo_n[1], hi_n[1], tid);
                    if (nqc > 2) bm_prefetch<ST>(P2, doc_ids, payload, lo_n[2], hi_n[2], tid);
                    if (nqc > 3) bm_prefetch<ST>(P3, doc_ids, payload, lo_n[3], hi_n[3], tid);
                }
                ERH_SEC(1);
            }
            if (ablate & 2) continue;
            // ---- first tile: seed the threshold.  The k-th largest of the 1024 per-thread maxima is a lower bound of
            // the tile's k-th best score (k distinct documents reach it), so the first sweep admits about k entries
            // instead of every touched document (which costs several fill-sort-resweep rounds).
            if (tile == t_begin && k <= kBmThreads && fd < 0 && !(ablate & 16)) {   // (a dir filter would need the maxima of passing documents only)
                constexpr int VEC = 16 / (int)sizeof(ST);
                typedef ST VT __attribute__((ext_vector_type(VEC)));
                ST mx = (ST)0;
                for (int i = tid * VEC; i < TILE; i += kBmThreads * VEC) {
                    const VT v = *reinterpret_cast<const VT *>(acc + i);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) mx = v[e] > mx ? v[e] : mx;
                }
                cs[tid] = mx;                                          // the candidate list is still empty
                erh_bitonic_desc<ST>(cs, kBmThreads);                  // begins and ends with a barrier
                const ST p = cs[k - 1];
                __syncthreads();
                if (tid == 0 && p > (ST)0) { hdr->tau_s = (double)p; hdr->tau_idx = 0x7fffffff; }   // ties at p pass whatever their index
                __syncthreads();
            }
            // ---- sweep: move what beats the running k-th best into the list, clear the rest -----------------------
            for (;;) {
                const ST tau_s = (ST)hdr->tau_s;
                const int tau_idx = hdr->tau_idx;
                bm_sweep<ST>(hdr, acc, cs, ci, 0, TILE, kBmThreads, tid, base_doc, N, fd, dir_id, tau_s, tau_idx,
                             &hdr->total, nullptr, 0);
                ERH_SEC(3);
                __syncthreads();
                const int full = hdr->total;                          // uniform; nobody writes it again before the next sweep,
                ERH_SEC(4);                                           // which lies behind further barriers
                if (!full) break;
                __syncthreads();                                      // everyone has read it before it is cleared
                if (tid == 0) hdr->total = 0;
                bm_shrink<ST>(hdr, cs, ci, k);                        // list was full: cut to k, threshold becomes exact
                ERH_SEC(2);
            }
            // keep the list short and the threshold exact once it holds clearly more than k entries
            if (hdr->ncand > k + kBmThreads / 2) bm_shrink<ST>(hdr, cs, ci, k);   // uniform (read after a barrier)
            ERH_SEC(5);
        }
    }
    // ---- emit this segment's list, sorted --------------------------------------------------------
    bm_shrink<ST>(hdr, cs, ci, k);
    const int n = hdr->ncand < k ? hdr->ncand : k;
    for (int i = tid; i < k; i += kBmThreads) {
        if (i < n) { part_scores[out_base + i] = (double)cs[i]; part_ids[out_base + i] = ci[i]; }
        else { part_scores[out_base + i] = 0.0; part_ids[out_base + i] = -1; }
    }
    if (tid == 0) part_len[(int64_t)q * segs + seg] = n;
    ERH_SEC(6);
    if (dbg && tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&dbg[i], (unsigned long long)t_sec[i]);
    }
#undef ERH_SEC
}

// ---- wave-owned scan (default for queries of up to kWsMaxTok tokens) ------------------------------------------------
// Same tiles, accumulators, candidate list and results as bm25_scan_kernel; what changes is who applies a posting.
// The tile is cut into 16 sub-ranges of TILE/16 documents and wave w owns sub-range w: it applies EVERY query token,
// in query order, to its own documents only.  A document lives in exactly one sub-range and the LDS operations of one
// wave execute in program order, so "token j before token j+1" holds per document without any workgroup barrier --
// the token loop of the block scan (one exposed load latency plus one 16-wave barrier per (token, tile) step for
// ~490 postings) becomes 16 independent streams whose latencies overlap each other.  A wave then sweeps its own
// sub-range (nobody else wrote it) and meets the other waves once per tile, at the candidate-list bookkeeping.
//   - posting ranges per (term, sub-range) come from a fine skip table fine_off[term][sub] (one int per term and
//     sub-range, built once per index like tile_off); lane j holds the range of query token j for the current tile
//     and fetches the next tile's while this one is processed;
//   - a wave-uniform cursor walks the (token, 64-posting piece) sequence of the sub-range -- empty ranges cost
//     nothing, a frequent term simply takes several pieces (postings of one term hit distinct documents, so their
//     order is free) -- and fills 16 (fp32) / 8 (fp64) register slots per step: all requests of a step are issued before the
//     first posting is applied, and the first step of the NEXT tile is requested before this tile is swept, so its
//     latency hides behind the sweep and the bookkeeping barrier;
//   - list-full / list-long decisions are taken once per sweep pass from flag words that rotate through three
//     slots: pass p raises slot p % 3, every thread reads it after the barrier that ends pass p, thread 0 clears slot
//     (p + 1) % 3 during pass p.  Waves are never more than one pass apart, so nobody reads a word while it changes.
template <typename ST> constexpr int ws_slots() { return sizeof(ST) == 4 ? 16 : 8; }   // 64-posting pieces in flight per step (128-VGPR budget)
constexpr int kWsMaxTok = 64;       // lane j owns token j
constexpr int kWsWaves = kBmThreads / 64;

template <typename ST>
struct WsSet {
    static constexpr int SLOTS = ws_slots<ST>();
    int32_t d[SLOTS];               // document index or -1
    ST v[SLOTS];
    int used;                       // wave-uniform: slots filled
};

struct WsCursor {                   // wave-uniform
    int j, off, nj;                 // token, postings of it already taken, its postings in this sub-range
    int64_t loj;                    // its first posting
};

// Fill the slots from the cursor.  lo_lane / n_lane: lane j holds the posting range of token j (this sub-range, one tile).
template <typename ST>
__device__ __forceinline__ void ws_fill(WsSet<ST> &S, WsCursor &c, const int32_t *__restrict__ doc_ids,
                                        const ST *__restrict__ payload, int64_t lo_lane, int n_lane, int nq, int lane) {
    const int lo_l = (int)(uint32_t)lo_lane, lo_h = (int)(uint32_t)((uint64_t)lo_lane >> 32);
    int used = 0;
#pragma unroll
    for (int u = 0; u < WsSet<ST>::SLOTS; ++u) {
        while (c.j < nq && c.off >= c.nj) {                               // next token that has postings left
            ++c.j;
            c.off = 0;
            c.nj = 0;
            if (c.j < nq) {
                c.nj = __builtin_amdgcn_readlane(n_lane, c.j);
                c.loj = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(lo_h, c.j) << 32) |
                                  (uint64_t)(uint32_t)__builtin_amdgcn_readlane(lo_l, c.j));
            }
        }
        S.d[u] = -1;
        S.v[u] = (ST)0;
        if (c.j < nq) {                                                   // wave-uniform
            const int at = c.off + lane;
            if (at < c.nj) {
                S.d[u] = doc_ids[c.loj + at];
                S.v[u] = payload[c.loj + at];
            }
            c.off += 64;
            used = u + 1;
        }
    }
    S.used = used;
}

template <typename ST>
__device__ __forceinline__ void ws_apply(const WsSet<ST> &S, ST *acc, int64_t base_doc) {
#pragma unroll
    for (int u = 0; u < WsSet<ST>::SLOTS; ++u) {
        if (u < S.used && S.d[u] >= 0) {
            const int slot = (int)((int64_t)S.d[u] - base_doc);
            acc[slot] = acc[slot] + S.v[u];
        }
    }
}

// Threshold crossings.  With strictly positive payloads a document's sum only grows while the tokens are applied, so
// "the final sum reaches the running threshold th" happens exactly once per document: at the posting whose add takes
// the sum from below th to >= th.  The wave notes the accumulator slot of every crossing in its own small LDS list
// (ballot + mbcnt, no atomics; the common case costs two compares and one wave-uniform test per 64 postings).  After
// the token loop the list IS the tile's survivor set of this sub-range: the wave reads the final sums of the noted
// slots, moves the ones that pass the exact (score, index) test to the candidate list, and clears its sub-range with
// plain 16-byte stores -- no sweep over 32768 accumulators of which a few dozen matter.  A wave whose list overflows
// (kWsXCap crossings in one sub-range) sweeps its sub-range the old way, as does every wave after a full candidate
// list; indices with non-positive payloads (possible with rank-bm25's epsilon floor on a degenerate corpus) and tiles
// without a positive threshold never take this path.
constexpr int kWsXCap = 64;         // crossings a wave notes per tile
constexpr int kWsXBytes = (kBmThreads / 64) * kWsXCap * 4;   // the waves' crossing lists, behind the candidate list
// (LDS floating-point atomics with return -- ds_add_rtn_f32 / _f64, all slots of a step in flight together -- give the
// same sums bit for bit but run the scan 1.7x SLOWER than this read / add / write-back chain: profiles/r02c.)
template <typename ST>
__device__ __forceinline__ void ws_apply_x(const WsSet<ST> &S, ST *acc, int64_t base_doc, ST th, int32_t *xw, int &nx,
                                           int lane) {
#pragma unroll
    for (int u = 0; u < WsSet<ST>::SLOTS; ++u) {
        bool cross = false;
        int slot = 0;
        if (u < S.used && S.d[u] >= 0) {
            slot = (int)((int64_t)S.d[u] - base_doc);
            const ST old = acc[slot];
            const ST nw = old + S.v[u];
            acc[slot] = nw;
            cross = (old < th) && !(nw < th);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cross);
        if (m) {                                                          // wave-uniform, rare
            const int pos = nx + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (cross && pos < kWsXCap) xw[pos] = slot;
            nx += __builtin_popcountll(m);
        }
    }
}

// grid = (segs, B), block = 1024; every query of the batch has at most kWsMaxTok tokens (the host checks).
template <typename ST, bool CROSSING /* every payload > 0 and normal: threshold crossings may replace the sweep */>
__global__ __launch_bounds__(kBmThreads) void bm25_wscan_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
    const int32_t *__restrict__ fine_off, int n_fine, int n_tiles, int64_t N,
    const int32_t *__restrict__ q_indptr, const int32_t *__restrict__ q_tok,
    const int32_t *__restrict__ q_order /* workgroup y -> query: heaviest queries first, or null */, int k, int segs,
    const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    double *__restrict__ part_scores, int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len,
    unsigned long long *__restrict__ dbg /* measurement only: section clock sums of thread 0, or null */) {
    using L = BmLds<ST>;
    constexpr int TILE = L::TILE, SUB = TILE / kWsWaves;
    static_assert(SUB % (64 * 16 / (int)sizeof(ST)) == 0, "a wave sweeps its sub-range in whole 16-byte rounds");
    static_assert(L::OFF_LO + kWsXBytes <= 160 * 1024, "crossing lists must fit behind the candidate list");
    long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_mark = dbg ? clock64() : 0;
#define ERH_SEC(I) do { if (dbg) { const long long n_ = clock64(); t_sec[I] += n_ - t_mark; t_mark = n_; } } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BmHdr *hdr = reinterpret_cast<BmHdr *>(smem);
    ST *acc = reinterpret_cast<ST *>(smem + L::OFF_ACC);
    ST *cs = reinterpret_cast<ST *>(smem + L::OFF_CS);
    int32_t *ci = reinterpret_cast<int32_t *>(smem + L::OFF_CI);

    const int seg = blockIdx.x, q = q_order ? q_order[blockIdx.y] : (int)blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t *xw = reinterpret_cast<int32_t *>(smem + L::OFF_LO) + wave * kWsXCap;   // this wave's crossing list
    const int qs = q_indptr[q];
    int nq = q_indptr[q + 1] - qs;
    if (nq > kWsMaxTok) nq = kWsMaxTok;                                   // (never: the host routes longer queries to the block scan)
    const int fd = filter_dir ? (int)filter_dir[q] : -1;
    const int t_begin = (int)((int64_t)n_tiles * seg / segs);
    const int t_end = (int)((int64_t)n_tiles * (seg + 1) / segs);
    const int64_t out_base = ((int64_t)q * segs + seg) * k;

    if (tid == 0) {
        hdr->ncand = 0; hdr->total = 0; hdr->tau_idx = -1; hdr->tau_s = 0.0;
        hdr->full[0] = hdr->full[1] = hdr->full[2] = 0;
        hdr->want[0] = hdr->want[1] = hdr->want[2] = 0;
    }
    for (int i = tid; i < TILE; i += kBmThreads) acc[i] = (ST)0;
    __syncthreads();

    if (nq > 0 && t_begin < t_end) {
        // lane j: query token j -- posting base and its row of the fine skip table
        const bool has = lane < nq;
        int64_t ip = 0;
        const int32_t *fo = fine_off;
        if (has) {
            const int64_t tok = q_tok[qs + lane];
            ip = indptr[tok];
            fo = fine_off + tok * (int64_t)(n_fine + 1);
        }
        auto bounds = [&](int tile, int64_t &lo, int &n) {
            int s0 = tile * kWsWaves + wave;
            s0 = s0 < n_fine ? s0 : n_fine;
            const int s1 = s0 < n_fine ? s0 + 1 : n_fine;
            const int a = has ? fo[s0] : 0, b = has ? fo[s1] : 0;
            lo = ip + a;
            n = b - a;
        };
        int64_t lo_cur, lo_nxt = 0;
        int n_cur, n_nxt = 0;
        bounds(t_begin, lo_cur, n_cur);
        WsSet<ST> S;
        WsCursor cur;
        cur.j = -1; cur.off = 0; cur.nj = 0; cur.loj = 0;
        ws_fill<ST>(S, cur, doc_ids, payload, lo_cur, n_cur, nq, lane);
        int ph = 0;                                                       // sweep pass counter (workgroup-uniform)
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int64_t base_doc = (int64_t)tile * TILE;
            const bool more = tile + 1 < t_end;
            if (more) bounds(tile + 1, lo_nxt, n_nxt);                    // lands while this tile is processed
            ERH_SEC(0);
            // ---- all tokens, in order, onto this wave's documents ------------------------------------------------
            const ST th = (ST)hdr->tau_s;                                 // fixed for the tile (it only moves in bm_shrink)
            const int th_idx = hdr->tau_idx;
            const bool use_x = CROSSING && tile != t_begin && th > (ST)0; // workgroup-uniform
            int nx = 0;                                                   // wave-uniform: crossings noted in this tile
            if (CROSSING) {
                const ST thx = use_x ? th : (ST)-1;                       // (-1: nothing crosses, the adds are the same)
                for (;;) {
                    ws_apply_x<ST>(S, acc, base_doc, thx, xw, nx, lane);
                    if (cur.j >= nq) break;
                    ws_fill<ST>(S, cur, doc_ids, payload, lo_cur, n_cur, nq, lane);
                }
            } else {
                for (;;) {
                    ws_apply<ST>(S, acc, base_doc);
                    if (cur.j >= nq) break;                               // the sub-range's postings are exhausted
                    ws_fill<ST>(S, cur, doc_ids, payload, lo_cur, n_cur, nq, lane);
                }
            }
            if (more) {                                                   // first step of the next tile: flies during the sweep
                cur.j = -1; cur.off = 0; cur.nj = 0;
                ws_fill<ST>(S, cur, doc_ids, payload, lo_nxt, n_nxt, nq, lane);
            }
            ERH_SEC(1);
            // ---- first tile: seed the threshold from the per-thread maxima (see bm25_scan_kernel) --------------------
            if (tile == t_begin && k <= kBmThreads && fd < 0) {
                constexpr int VEC = 16 / (int)sizeof(ST);
                typedef ST VT __attribute__((ext_vector_type(VEC)));
                ST mx = (ST)0;
                for (int i = wave * SUB + lane * VEC; i < (wave + 1) * SUB; i += 64 * VEC) {
                    const VT v = *reinterpret_cast<const VT *>(acc + i);   // this wave's own sums: complete
#pragma unroll
                    for (int e = 0; e < VEC; ++e) mx = v[e] > mx ? v[e] : mx;
                }
                cs[tid] = mx;
                erh_bitonic_desc<ST>(cs, kBmThreads);
                const ST p = cs[k - 1];
                __syncthreads();
                if (tid == 0 && p > (ST)0) { hdr->tau_s = (double)p; hdr->tau_idx = 0x7fffffff; }
                __syncthreads();
            }
            // ---- survivors of the own sub-range -> candidate list; once per pass the workgroup decides about the list ---
            bool swept = false;                                           // wave-uniform: the sub-range has been swept (= cleared)
            bool from_list = use_x && nx <= kWsXCap;                      // wave-uniform: the crossing list is complete
            if (dbg && tid == 0) { t_sec[6] += from_list ? 1 : 0; t_sec[7] += nx; }   // (measurement: tiles by list, crossings)
            for (;;) {
                const int slot = ph % 3;
                if (tid == 0) { hdr->full[(ph + 1) % 3] = 0; hdr->want[(ph + 1) % 3] = 0; }
                if (from_list) {
                    // the noted slots hold final sums now; th / th_idx are still the workgroup's threshold (no shrink since)
                    for (int i = lane; i < nx; i += 64) {
                        const int sl = xw[i];
                        const ST sv = acc[sl];
                        const int64_t doc = base_doc + sl;
                        bool pass = bm_pass<ST>(sv, doc, th, th_idx);
                        if (pass && (doc >= N || (fd >= 0 && (int)dir_id[doc] != fd))) pass = false;
                        if (pass) {
                            const int pos = atomicAdd(&hdr->ncand, 1);
                            if (pos < kBmCap) {
                                cs[pos] = sv;
                                ci[pos] = (int32_t)doc;
                                acc[sl] = (ST)0;                          // moved: a later sweep must not see it again
                                if (pos >= k + kBmThreads / 2) hdr->want[slot] = 1;
                            } else {
                                hdr->full[slot] = 1;                      // list full: it stays in its accumulator for the sweep below
                            }
                        }
                    }
                    from_list = false;                                    // a further pass (full list) sweeps
                } else {
                    const ST tau_s = (ST)hdr->tau_s;
                    const int tau_idx = hdr->tau_idx;
                    bm_sweep<ST>(hdr, acc, cs, ci, wave * SUB, (wave + 1) * SUB, 64, lane, base_doc, N, fd, dir_id, tau_s,
                                 tau_idx, &hdr->full[slot], &hdr->want[slot], k + kBmThreads / 2);
                    swept = true;
                }
                ERH_SEC(2);
                __syncthreads();                                          // every sub-range done
                ERH_SEC(3);
                const int full = hdr->full[slot], want = hdr->want[slot];
                ++ph;
                if (full || want) bm_shrink<ST>(hdr, cs, ci, k);          // uniform; cut to k, threshold becomes exact
                ERH_SEC(4);
                if (!full) break;                                         // (full: survivors were left behind -- sweep again)
            }
            if (!swept) {                                                 // nothing left here that matters: clear the sub-range
                constexpr int VEC = 16 / (int)sizeof(ST);
                typedef ST VT __attribute__((ext_vector_type(VEC)));
                VT z;
#pragma unroll
                for (int e = 0; e < VEC; ++e) z[e] = (ST)0;
                for (int i = wave * SUB + lane * VEC; i < (wave + 1) * SUB; i += 64 * VEC) *reinterpret_cast<VT *>(acc + i) = z;
            }
            lo_cur = lo_nxt;
            n_cur = n_nxt;
        }
    }
    bm_shrink<ST>(hdr, cs, ci, k);
    const int n = hdr->ncand < k ? hdr->ncand : k;
    for (int i = tid; i < k; i += kBmThreads) {
        if (i < n) { part_scores[out_base + i] = (double)cs[i]; part_ids[out_base + i] = ci[i]; }
        else { part_scores[out_base + i] = 0.0; part_ids[out_base + i] = -1; }
    }
    if (tid == 0) part_len[(int64_t)q * segs + seg] = n;
    ERH_SEC(5);
    if (dbg && tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&dbg[i], (unsigned long long)t_sec[i]);
    }
#undef ERH_SEC
}

// ---- fixed-point scan + exact re-score (default when every payload of the index is a positive normal number) --------
// The block scan and the wave-owned scan above keep the library's summation order DURING the scan, which makes every
// document a serial read-add-write chain.  This kernel separates the two jobs:
//   scan     the postings of a tile are scattered with LDS INTEGER atomics (ds_add_rtn_u32: 7-8 cycles per 64 postings
//            on a busy CU; ds_add_f32 serialises at 192 -- scripts/ubench/lds_scatter.hip).  The index keeps an
//            interleaved copy of the postings for it, {document, q} with the payload in fixed point,
//            q = trunc(p32 * 2^S) + 1, 2^S chosen per index so that 64 payloads cannot overflow 32 bits (8 bytes per
//            posting for both variants, one 8-byte load per lane).  Integer sums do not depend on the order, so every
//            lane of every wave carries a posting: the (token, 64-posting piece) sequence of a tile is dealt
//            round-robin to the 16 waves, and the postings of tile t+1 are already in registers (requested one tile
//            ahead with a static number of loads: lanes and register slots past the end read a sentinel posting) while
//            tile t is applied and cleared.  Wave 0 prepares the posting ranges two tiles ahead (DPP prefix sum over
//            the tokens) and publishes them in LDS.  The kernel is bound by instruction issue (one wave-instruction
//            per cycle and CU), so the loop is written for few instructions per posting, not for few bytes;
//   select   sums only grow, so "the final sum reaches the threshold" happens at exactly one add per document: the
//            returned old value tells (thq - 1 - old < q, unsigned: exact) and the accumulator slot is noted in the
//            list of the wave that OWNS that part of the accumulators.  After the tile's barrier every wave moves its
//            own noted slots to the candidate list and clears its own 2048 accumulators -- program order inside the
//            wave, no barrier in between, no sweep over 32768 sums.  Tiles without a threshold yet (the first one,
//            seeded by a radix select over the per-thread maxima) and overflowing lists use the sweep of the block scan
//            on the integer sums;
//   bound    for a document with n <= nq matched tokens, real-arithmetic score S, library value f (fp32 / fp64 sum in
//            query-token order) and fixed-point sum a = aq / 2^S:  S~ < a <= S~ + n 2^-S  (S~: sum of the fp32 payloads,
//            within S (1 +- 2^-24)), f within S (1 +- n 2^-24).  With eps = 1.01 (nq + 2) 2^-24:
//            f < a (1 + eps) and f >= (a - n 2^-S)(1 - eps).  If k listed documents have aq >= thetaq, a document with
//            aq < thq = floor(thetaq (1 - 3 eps)) - nq - 1 has f strictly below all k of them: it can never be in the
//            top k.  Everything else stays on the list (k entries plus the near ties of the k-th); thetaq is the exact
//            k-th largest listed sum (LDS radix select, no sort);
//            Packed shape (two documents per word, 16-bit sums, q16 = (q >> sh) + 1): q16 2^sh > q and <= q + 2^sh, so the
//            sum exceeds the real one by < 2 n units of its own grid instead of n: thq = floor(thetaq (1 - 3 eps)) - 2 nq - 1,
//            everything else as above;
//   re-score the final list is scored EXACTLY: one binary search per (document, query token) inside the tile range of the
//            skip table (four searches in flight per thread), the payloads summed in query-token order in the library's
//            type (adding 0.0 for an absent token changes nothing), then ranked by (score desc, index asc) by counting.
//            ids and scores are the library's, bit for bit.
// A list that does not shrink below its capacity (thousands of documents within the margin of the k-th score -- e.g. a
// corpus of near-identical short documents) or a query whose sums could overflow sets redo[query, segment]: the host
// launches the exact block scan for those workgroups only.  Indices with non-positive payloads never come here
// (api.hip checks when an index is set).
constexpr int kAsU = 3;                                  // 128-posting pieces per wave and tile held in registers (a 16384-document tile leaves
                                                         // a wave 2.6 on average; six slots measured 4 % slower there: every empty one is two
                                                         // dummy adds).  The packed shape (32768 documents per tile and 8 waves) holds six.
constexpr int kAsTokChunk = 64;                          // lane j <-> query token j
constexpr int kAsRegion = 2048;                          // accumulators a wave owns (notes, clears)
constexpr int kAsXW = 88;                                // threshold crossings noted per region and tile
constexpr size_t kAsOffHdr2 = 64;
constexpr size_t kAsOffXcnt = 128;                       // int xz[2][NW + 1]: crossings per region, then the largest position handed out
constexpr size_t kAsOffTok = 320;                        // int32 tok[64]   (re-score stage)
constexpr size_t kAsOffIp = 576;                         // uint32 ip[64]
constexpr size_t kAsOffPx = 832;                         // int px[3][16]: the first 16 tokens' piece offsets of each range table, packed
constexpr size_t kAsOffRng = 1024;                       // int4 rng[3][64]
constexpr size_t kAsOffAcc = 4096;
static_assert(kAsOffXcnt + 2 * 17 * 4 <= kAsOffTok && kAsOffIp + 64 * 4 <= kAsOffPx && kAsOffPx + 3 * 16 * 4 <= kAsOffRng, "header tables overlap");
static_assert(kAsOffRng + 3 * 64 * 16 <= kAsOffAcc, "range tables overlap the accumulators");

// Two shapes of the same kernel.  AsBig: one 1024-thread workgroup per CU over 32768-document tiles (list capacity 2048:
// any k the API allows).  AsSmall: 512 threads over 16384-document tiles in 80 KiB of LDS, i.e. TWO workgroups (two
// queries) per CU whose phases -- adds, barrier, list + clear, barrier -- fall at different times, so the LDS pipe, the
// address path and the SIMDs of the CU are busy while one of them waits (list capacity 1024: k <= 384).
//   AsPack: the 512-thread shape over 32768-document tiles: TWO documents per accumulator word (document slot sl of the tile
// lives in half sl >> 14 of word sl & 16383), sums of 16 bits.  The stored fixed-point payloads are shifted right per query
// (as_pack_shift) until nq of them fit 16 bits, so a low half never carries into the high one.  The work per tile that
// does not depend on the postings (ranges, descriptors, barriers, list, clear) is paid once per 32768 documents.
//   AsPack16: the packed shape reading a 4-BYTE posting -- {document & 32767, ((q >> G) + 1) as 16 bits}: a piece's tile is known, so
// fifteen bits of the document suffice, and sixteen bits of payload are more than a query's sums can use anyway -- four
// postings per lane and 16-byte load, pieces of 256: half the load instructions, half the bytes in flight per CU.
template <int NT_, int TILE_, bool PACK_ = false, bool P16_ = false>
struct AsCfg {
    static constexpr int NT = NT_, TILE = TILE_, NW = NT_ / 64, CAP = 2 * NT_, RESERVE = NT_ / 4;
    static constexpr bool PACK = PACK_, P16 = P16_;
    static constexpr int PIECE = P16_ ? 256 : 128;                   // postings per piece: one 16-byte load per lane
    static_assert(!P16_ || (PACK_ && TILE_ == 32768), "the 4-byte postings carry 15 document bits and feed the packed sums");
    static constexpr int WORDS = PACK_ ? TILE_ / 2 : TILE_;          // 32-bit accumulator words of a tile
    static constexpr int U = (PACK_ && !P16_) ? 2 * kAsU : kAsU;     // posting pieces per wave and tile held in registers (768 postings when packed)
    static_assert(WORDS / (NT_ / 64) == kAsRegion, "a wave owns 2048 accumulator words");
    static_assert(!PACK_ || WORDS == 16384, "the packed slot -> (word, half) split is written for 16384 words");
    static constexpr size_t OFF_CA = kAsOffAcc + (size_t)WORDS * 4;
    static constexpr size_t OFF_CI = OFF_CA + (size_t)CAP * 4;
    static constexpr size_t OFF_XL = OFF_CI + (size_t)CAP * 4;
    static constexpr size_t OFF_HIST = OFF_XL + (size_t)NW * kAsXW * 4;
    static constexpr size_t OFF_DUMMY = OFF_HIST + 256 * 4;           // 64 words that lanes without a posting add 0 to
    static constexpr size_t BYTES = OFF_DUMMY + 64 * 4;
    static constexpr int TAB_SHIFT = TILE_ == 32768 ? 15 : 14;        // log2(TILE)
};
using AsBig = AsCfg<1024, 32768>;
using AsSmall = AsCfg<512, 16384>;
using AsPack = AsCfg<512, 32768, true>;
using AsPack16 = AsCfg<512, 32768, true, true>;
static_assert(AsBig::BYTES <= 160 * 1024 && 2 * AsSmall::BYTES <= 160 * 1024 && 2 * AsPack::BYTES <= 160 * 1024, "fixed-point scan LDS layouts");
constexpr size_t kAsMixedBytes = AsPack16::BYTES > AsSmall::BYTES ? AsPack16::BYTES : AsSmall::BYTES;   // bm25_ascan_mixed_kernel: either body
static_assert(2 * kAsMixedBytes <= 160 * 1024, "two workgroups of the mixed launch per CU");

struct AsHdr {
    uint32_t thetaq;                // k-th best fixed-point sum seen so far (0: fewer than k candidates yet)
    uint32_t thq;                   // drop threshold (>= 1): a document below it can never reach the top k
    int redo;                       // near-tie flood / possible overflow: the query goes to the exact block scan
    int sel_bin, sel_need;          // radix select scratch
    int keep;                       // compaction counter
};
static_assert(sizeof(AsHdr) <= 64, "AsHdr must fit its 64-byte slot");

typedef int as_int4 __attribute__((ext_vector_type(4)));
typedef uint32_t as_uint2 __attribute__((ext_vector_type(2)));
typedef uint32_t as_uint4 __attribute__((ext_vector_type(4)));
constexpr int kAsPiece = 128;                            // postings per piece of the 8-byte format: two consecutive ones per lane, one 16-byte load
template <int U>
struct AsSet { as_uint4 p[U]; };     // two postings per lane: .x/.z document, .y/.w fixed-point payload

// packed sums: smallest right shift of the stored payloads with nq ((qmax >> sh) + 1) <= 65535; 32: there is none
__device__ __forceinline__ int as_pack_shift(int nq, double qmax) {
    const uint64_t qm = (uint64_t)qmax;
    int sh = 0;
    while (sh < 32 && (uint64_t)nq * ((qm >> sh) + 1ull) > 65535ull) ++sh;
    return sh;
}

// inclusive prefix sum over the 64 lanes (DPP: four shifts inside the rows of 16, two row broadcasts)
__device__ __forceinline__ int as_wave_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);       // row_shr:1, lanes shifted in read 0
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return x;
}

// lane j: raw skip-table entries a, b of token j (0, 0 for lanes without a token) -> ranges of one tile:
// {pieces of the tokens before this one (0x7fffffff: no such token), postings, index of the first one, pieces of the tile}
template <int PIECE = kAsPiece>
__device__ __forceinline__ as_int4 as_make_ranges(uint32_t ip, int a, int b, int nq, int lane) {
    const int n = b - a;
    const int c = (n + PIECE - 1) / PIECE;
    const int incl = as_wave_scan(c);
    as_int4 r;
    r[0] = lane < nq ? incl - c : 0x7fffffff;
    r[1] = n;
    r[2] = (int)(ip + (uint32_t)a);
    r[3] = __builtin_amdgcn_readlane(incl, 63);
    return r;
}

// Piece descriptors of one tile, one per lane: lane u of wave w describes piece p = w + NW u of the tile whose ranges
// lie in LDS at `rt` (64 x the vector above): first posting index and how many postings the piece holds (<= 0: none --
// past the token's range or past the tile's last piece; > 128: the token goes on in its next piece).  Every lane scans
// the tokens' piece offsets with broadcast reads; no wave-uniform control flow, no scalar work.
template <int NW, int PIECE = kAsPiece>
__device__ __forceinline__ void as_describe(const as_int4 *rt, const as_int4 *px16, int nq, int lane, int wave, uint32_t &dstart,
                                            int &dcnt) {
    const int p = wave + lane * NW;                                      // (NW = the number of waves the pieces are dealt to)
    int j = -1;
    if (nq <= 16 && px16) {                                               // four 16-byte broadcast reads (lanes past nq hold 0x7fffffff)
        as_int4 px[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) px[i] = px16[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) j += (px[i >> 2][i & 3] <= p) ? 1 : 0;
    } else {
        for (int i = 0; i < nq; ++i) j += (rt[i][0] <= p) ? 1 : 0;        // non-decreasing: last token with offset <= p
    }
    const as_int4 r = rt[j < 0 ? 0 : j];
    const int o = (p - r[0]) * PIECE;
    dstart = (uint32_t)r[2] + (uint32_t)o;
    dcnt = r[1] - o;
}

// pieces [r0, r0 + kAsU) of this wave (np of them exist) -> registers; lanes without a posting read the sentinel pair
template <int U, bool P16 = false>
__device__ __forceinline__ void as_fill(AsSet<U> &S, uint32_t dstart, int dcnt, int r0, int np, const void *__restrict__ post,
                                        int lane, uint32_t sentinel) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (r0 + u < np) {                                                // wave-uniform
            const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)dstart, r0 + u);
            const int cn = __builtin_amdgcn_readlane(dcnt, r0 + u);
            if constexpr (P16) {                                          // four 4-byte postings per lane
                const uint32_t idx = 4 * lane < cn ? st + 4u * (uint32_t)lane : sentinel;
                S.p[u] = *reinterpret_cast<const as_uint4 *>(reinterpret_cast<const uint32_t *>(post) + idx);
            } else {
                const uint32_t idx = 2 * lane < cn ? st + 2u * (uint32_t)lane : sentinel;
                S.p[u] = *reinterpret_cast<const as_uint4 *>(reinterpret_cast<const as_uint2 *>(post) + idx);
            }
        }
    }
}

// rare: slot `sl` crossed the threshold -> list of the wave that owns the slot; xz[NW] keeps the largest position
template <int NW>
__device__ __forceinline__ void as_note(bool cross, int sl, int32_t *xl, int *xz) {
    if (cross) {
        const int r = (sl & (NW * kAsRegion - 1)) / kAsRegion;           // (packed: the word's region; plain: sl < NW * kAsRegion)
        const int pos = atomicAdd(&xz[r], 1);
        if (pos < kAsXW) xl[r * kAsXW + pos] = sl;
        atomicMax(&xz[NW], pos + 1);
    }
}

// the same pieces onto the integer sums.  Nothing conditional around the adds: a lane without a posting (past the end of
// its piece, or a piece the wave does not have) adds 0 to a word of its own in a dummy area, so the 2 kAsU atomics of a
// round are issued back to back and their returns are collected afterwards -- behind branches or exec masks the
// compiler waits for each return before it issues the next add.
template <int NW, int U, bool PACK>
__device__ __forceinline__ void as_apply(const AsSet<U> &S, int dcnt, int r0, int np, int lane, uint32_t *accu, uint32_t *dummy,
                                         int base_doc, uint32_t thx, int32_t *xl, int *xz, int sh) {
    uint32_t o0[U], o1[U], q0[U], q1[U];
    uint32_t *mine = dummy + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int cn = __builtin_amdgcn_readlane(dcnt, r0 + u < 64 ? r0 + u : 63);
        cn = r0 + u < np ? cn : 0;                                        // scalar select
        const bool v0 = 2 * lane < cn, v1 = 2 * lane + 1 < cn;
        if constexpr (PACK) {
            const uint32_t s0 = (uint32_t)((int)S.p[u].x - base_doc), s1 = (uint32_t)((int)S.p[u].z - base_doc);
            const uint32_t h0 = ((s0 >> 14) & 1u) << 4, h1 = ((s1 >> 14) & 1u) << 4;   // 0 / 16: the half's bit offset
            q0[u] = v0 ? (S.p[u].y >> sh) + 1u : 0u;
            q1[u] = v1 ? (S.p[u].w >> sh) + 1u : 0u;
            o0[u] = atomicAdd(v0 ? &accu[s0 & 16383u] : mine, v0 ? q0[u] << h0 : 0u);
            o1[u] = atomicAdd(v1 ? &accu[s1 & 16383u] : mine, v1 ? q1[u] << h1 : 0u);
            o0[u] = (o0[u] >> h0) & 0xffffu;                              // the half's own old sum
            o1[u] = (o1[u] >> h1) & 0xffffu;
        } else {
            q0[u] = v0 ? S.p[u].y : 0u;
            q1[u] = v1 ? S.p[u].w : 0u;
            o0[u] = atomicAdd(v0 ? &accu[(int)S.p[u].x - base_doc] : mine, q0[u]);
            o1[u] = atomicAdd(v1 ? &accu[(int)S.p[u].z - base_doc] : mine, q1[u]);
        }
    }
    if (thx) {                                                            // (no threshold yet: nothing to note, the tile is swept)
        bool any = false;
#pragma unroll
        for (int u = 0; u < U; ++u) any |= (thx - 1u - o0[u] < q0[u]) | (thx - 1u - o1[u] < q1[u]);   // old < thx <= old + q
        if (__builtin_amdgcn_ballot_w64(any)) {                           // wave-uniform, rare once the threshold has settled
#pragma unroll
            for (int u = 0; u < U; ++u) {
                as_note<NW>(thx - 1u - o0[u] < q0[u], (int)S.p[u].x - base_doc, xl, xz);
                as_note<NW>(thx - 1u - o1[u] < q1[u], (int)S.p[u].z - base_doc, xl, xz);
            }
        }
    }
}

// the 4-byte postings onto the packed sums: word w of a piece = {slot of the tile in bits 0-14, 16-bit payload in bits 16-31};
// four per lane.  Same shape as as_apply: all adds of the round issued back to back, returns collected afterwards.
template <int NW, int U>
__device__ __forceinline__ void as_apply16(const AsSet<U> &S, int dcnt, int r0, int np, int lane, uint32_t *accu, uint32_t *dummy,
                                           uint32_t thx, int32_t *xl, int *xz, int sh) {
    uint32_t o[U][4], q[U][4];
    uint32_t *mine = dummy + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int cn = __builtin_amdgcn_readlane(dcnt, r0 + u < 64 ? r0 + u : 63);
        cn = r0 + u < np ? cn : 0;                                        // scalar select
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t w = S.p[u][e];
            const bool v = 4 * lane + e < cn;
            const uint32_t hb = (w >> 10) & 16u;                          // slot bit 14 -> the half's bit offset
            q[u][e] = v ? ((w >> 16) >> sh) + 1u : 0u;
            o[u][e] = atomicAdd(v ? &accu[w & 16383u] : mine, q[u][e] << hb);
            o[u][e] = (o[u][e] >> hb) & 0xffffu;
        }
    }
    if (thx) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) any |= thx - 1u - o[u][e] < q[u][e];   // old < thx <= old + q
        if (__builtin_amdgcn_ballot_w64(any)) {                           // wave-uniform, rare once the threshold has settled
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) as_note<NW>(thx - 1u - o[u][e] < q[u][e], (int)(S.p[u][e] & 32767u), xl, xz);
        }
    }
}

// k-th largest of v[0..n) (1 <= k <= n): radix select, four passes of eight bits over an LDS histogram.  Uniform call.
__device__ __forceinline__ uint32_t as_kth_largest(const uint32_t *v, int n, int k, uint32_t *hist, AsHdr *h2) {
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t prefix = 0u, mask = 0u;
    int need = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += (int)blockDim.x) {
            const uint32_t x = v[i];
            if ((x & mask) == prefix) atomicAdd(&hist[(x >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                                   // wave 0: bins 4 lane .. 4 lane + 3, from the top
            const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2_ = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int s = (int)(h0 + h1 + h2_ + h3);
            const int incl = as_wave_scan(s);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int above = total - incl;                               // members of the bins of higher lanes
            if (above < need && need <= above + s) {                      // exactly one lane
                int a = above, bin = 4 * lane;
                const int hh[4] = {(int)h3, (int)h2_, (int)h1, (int)h0};
                bool done = false;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (!done) {
                        if (need <= a + hh[b]) { bin = 4 * lane + 3 - b; done = true; }
                        else a += hh[b];
                    }
                }
                h2->sel_bin = bin;
                h2->sel_need = need - a;
            }
        }
        __syncthreads();
        prefix |= (uint32_t)h2->sel_bin << shift;
        mask |= 255u << shift;
        need = h2->sel_need;
    }
    __syncthreads();                                                      // (everyone has read sel_* before a later call writes them)
    return prefix;
}

// n_err: how many units a sum can exceed the real one by -- nq (one per posting: the + 1 behind the truncation) or, for
// the packed sums, 2 nq (the stored payload is truncated a second time by the shift, + 1 again)
__device__ __forceinline__ uint32_t as_drop_threshold(uint32_t thetaq, double keep_frac, int n_err) {
    const long long t = (long long)((double)thetaq * keep_frac) - n_err - 1;
    return t > 1 ? (uint32_t)t : 1u;
}

// Refresh thetaq / the drop threshold from the list and keep what is not provably out (list order is arbitrary).
template <int CAP>
__device__ __forceinline__ void as_shrink(BmHdr *hdr, AsHdr *h2, uint32_t *ca, int32_t *ci, uint32_t *hist, int k,
                                          double keep_frac, int nq, int reserve) {
    __syncthreads();
    const int n = hdr->ncand < CAP ? hdr->ncand : CAP;
    if (n >= k) {                                                         // uniform
        const uint32_t th = as_kth_largest(ca, n, k, hist, h2);
        if (threadIdx.x == 0 && th > h2->thetaq) { h2->thetaq = th; h2->thq = as_drop_threshold(th, keep_frac, nq); }
    }
    if (threadIdx.x == 0) h2->keep = 0;
    __syncthreads();
    const uint32_t thq = h2->thq;
    // compaction in place: every thread holds its (at most two) entries in registers across the barrier
    const int i0 = threadIdx.x, i1 = threadIdx.x + CAP / 2;       // (CAP = 2 x threads)
    uint32_t v0 = 0u, v1 = 0u;
    int32_t d0 = 0, d1 = 0;
    if (i0 < n) { v0 = ca[i0]; d0 = ci[i0]; }
    if (i1 < n) { v1 = ca[i1]; d1 = ci[i1]; }
    __syncthreads();
    const bool k0 = i0 < n && v0 >= thq, k1 = i1 < n && v1 >= thq;
    if (k0) { const int pos = atomicAdd(&h2->keep, 1); ca[pos] = v0; ci[pos] = d0; }
    if (k1) { const int pos = atomicAdd(&h2->keep, 1); ca[pos] = v1; ci[pos] = d1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        hdr->ncand = h2->keep;
        if (h2->keep > CAP - reserve) h2->redo = 1;                    // near-tie flood: no room for the next tile
    }
    __syncthreads();
}

// bm_sweep for the packed sums: a 16-byte vector holds eight documents (word w: slots w and w + 16384 of the tile).
template <class C>
__device__ __forceinline__ void as_sweep_packed(BmHdr *hdr, uint32_t *accu, uint32_t *ca, int32_t *ci, int tid, int64_t base_doc,
                                                int64_t N, int fd, const int16_t *__restrict__ dir_id, uint32_t thq,
                                                int *full_flag, int *want_flag, int want_at) {
    typedef uint32_t UT __attribute__((ext_vector_type(4)));
    constexpr int UNR = 4, NT = C::NT, W = C::WORDS, CAP = C::CAP;
    for (int i0 = tid * 4; i0 < W; i0 += NT * 4 * UNR) {
        UT v[UNR];
        bool cand[UNR], touched[UNR];
        bool any = false;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {                                   // (all reads before the first store: see bm_sweep)
            const int i = i0 + u * NT * 4;
            if (i < W) v[u] = *reinterpret_cast<UT *>(accu + i);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * NT * 4;
            cand[u] = touched[u] = false;
            if (i < W) {
                touched[u] = (v[u][0] | v[u][1] | v[u][2] | v[u][3]) != 0u;
                uint32_t m = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = v[u][e] & 0xffffu, hi = v[u][e] >> 16;
                    m = lo > m ? lo : m;
                    m = hi > m ? hi : m;
                }
                cand[u] = touched[u] && m >= thq;
                any |= cand[u];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (touched[u] && !cand[u]) {
                const UT z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<UT *>(accu + i0 + u * NT * 4) = z;
            }
        }
        if (__builtin_amdgcn_ballot_w64(any) == 0) continue;                // wave-uniform
        // rare: a vector with a possible survivor is taken apart word by word, from LDS again (a rolled loop: few registers)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (!cand[u]) continue;
            const int i = i0 + u * NT * 4;
#pragma unroll 1
            for (int e = 0; e < 4; ++e) {
                uint32_t w = accu[i + e];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const uint32_t sv = (w >> (16 * hf)) & 0xffffu;
                    if (sv == 0u) continue;
                    const int64_t doc = base_doc + i + e + hf * W;
                    bool pass = sv >= thq;
                    if (pass && (doc >= N || (fd >= 0 && (int)dir_id[doc] != fd))) pass = false;
                    bool keep = false;
                    if (pass) {
                        const int pos = atomicAdd(&hdr->ncand, 1);
                        if (pos < CAP) {
                            ca[pos] = sv;
                            ci[pos] = (int32_t)doc;
                            if (want_flag && pos >= want_at) *want_flag = 1;
                        } else {
                            *full_flag = 1;                                 // list full: the sum stays for the next sweep
                            keep = true;
                        }
                    }
                    if (!keep) w &= hf ? 0x0000ffffu : 0xffff0000u;
                }
                accu[i + e] = w;
            }
        }
    }
}

// The tile's sums are complete and there is no threshold yet (or too many crossings were noted): sweep the whole tile
// with the workgroup -- survivors (>= the drop threshold, passing the filter) to the list, everything else cleared; the
// list is cut back (as_shrink) whenever it fills.  On the first tile of the segment the threshold is seeded from the
// k-th largest of the per-thread maxima.  Ends behind a barrier.  true: give up (redo).
template <class C>
__device__ __forceinline__ bool as_sweep_tile(BmHdr *hdr, AsHdr *h2, uint32_t *accu, uint32_t *ca, int32_t *ci, uint32_t *hist,
                                              bool first_tile, int base_doc, int64_t N, int fd,
                                              const int16_t *__restrict__ dir_id, int k, double keep_frac, int nq, int &ph) {
    constexpr int NT = C::NT, WORDS = C::WORDS, CAP = C::CAP;
    const int tid = threadIdx.x;
    if (first_tile && k <= NT && fd < 0) {
        // k-th largest of the per-thread maxima: k distinct documents reach it, so it is a valid first thetaq
        typedef uint32_t UT __attribute__((ext_vector_type(4)));
        uint32_t mx = 0u;
        for (int i = tid * 4; i < WORDS; i += NT * 4) {
            const UT v = *reinterpret_cast<const UT *>(accu + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (C::PACK) {
                    const uint32_t lo = v[e] & 0xffffu, hi = v[e] >> 16;
                    mx = lo > mx ? lo : mx;
                    mx = hi > mx ? hi : mx;
                } else {
                    mx = v[e] > mx ? v[e] : mx;
                }
            }
        }
        ca[tid] = mx;                                                     // the list is still empty
        const uint32_t p = as_kth_largest(ca, NT, k, hist, h2);           // begins and ends with barriers
        if (tid == 0 && p > 0u) { h2->thetaq = p; h2->thq = as_drop_threshold(p, keep_frac, nq); }
        __syncthreads();
    }
    for (;;) {
        const int slot = ph % 3;
        if (tid == 0) { hdr->full[(ph + 1) % 3] = 0; hdr->want[(ph + 1) % 3] = 0; }
        if constexpr (C::PACK)
            as_sweep_packed<C>(hdr, accu, ca, ci, tid, (int64_t)base_doc, N, fd, dir_id, h2->thq, &hdr->full[slot], &hdr->want[slot],
                               k + NT / 2);
        else
            bm_sweep<uint32_t, CAP>(hdr, accu, ca, ci, 0, WORDS, NT, tid, (int64_t)base_doc, N, fd, dir_id, h2->thq,
                                    0x7fffffff, &hdr->full[slot], &hdr->want[slot], k + NT / 2);
        __syncthreads();
        const int full = hdr->full[slot], want = hdr->want[slot];
        ++ph;
        if (full || want) {
            as_shrink<CAP>(hdr, h2, ca, ci, hist, k, keep_frac, nq, C::RESERVE);
            if (h2->redo) return true;                                    // (written before as_shrink's last barrier)
        }
        if (!full) return false;                                          // (full: survivors were left behind -- sweep again)
    }
}

// The scan is over: hdr->ncand list entries (document ci[], approximate sum dead) hold the top k and the near ties of
// the k-th.  Exact re-score in query-token order in the library's type (one binary search per (document, token) inside
// the skip-table range of 2^tab_shift documents that holds the document), rank by counting, write the segment's list.
template <typename ST, class C>
__device__ __forceinline__ void as_finish_query(char *smem, BmHdr *hdr, uint32_t *ca, int32_t *ci,
                                                const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids,
                                                const ST *__restrict__ payload, const int32_t *__restrict__ tile_off, int n_tab,
                                                int tab_shift, const int32_t *__restrict__ q_tok, int qs, int nq, int k,
                                                int64_t out_base, int64_t out_slot, double *__restrict__ part_scores,
                                                int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len,
                                                unsigned long long *__restrict__ dbg = nullptr /* measurement builds: stage clocks -> dbg[10..15] */) {
    constexpr int NT = C::NT, TILE = C::TILE, CAP = C::CAP;
    const int tid = threadIdx.x;
#ifdef ERH_MEASURE
    long long fq_mark = dbg ? clock64() : 0;
#define ERH_FQ(I) do { if (dbg && tid == 0) { const long long n_ = clock64(); atomicAdd(&dbg[I], (unsigned long long)(n_ - fq_mark)); fq_mark = n_; } } while (0)
#else
#define ERH_FQ(I) do { } while (0)
#endif
    int32_t *s_tok = reinterpret_cast<int32_t *>(smem + kAsOffTok);
    uint32_t *s_ip = reinterpret_cast<uint32_t *>(smem + kAsOffIp);
    // ---- exact re-score of the list, in query-token order, in the library's type ------------------------------------------
    const int n_keep = hdr->ncand;
    ST *fs = reinterpret_cast<ST *>(smem + kAsOffAcc);                    // the accumulators are dead: exact sums ...
    ST *M = fs + CAP;                                                  // ... and the (entry, token) payload matrix
    constexpr int kMCap = (int)(((size_t)C::WORDS * 4 - (size_t)CAP * sizeof(ST)) / sizeof(ST));
    for (int i = tid; i < CAP; i += NT) fs[i] = (ST)0;
    for (int c0 = 0; c0 < nq; c0 += kAsTokChunk) {
        const int nqc = nq - c0 < kAsTokChunk ? nq - c0 : kAsTokChunk;
        const int ld = nqc | 1;                                           // odd row length: conflict-free column walk
        __syncthreads();
        if (tid < nqc) {
            const int32_t tok = q_tok[qs + c0 + tid];
            s_tok[tid] = tok;
            s_ip[tid] = (uint32_t)indptr[tok];
        }
        __syncthreads();
        ERH_FQ(10);                                                       // token table
        const int ec_max = kMCap / ld;
        for (int e0 = 0; e0 < n_keep; e0 += ec_max) {
            const int ec = n_keep - e0 < ec_max ? n_keep - e0 : ec_max;
            const int items = ec * nqc;
            constexpr int R = 6;                                          // searches in flight per thread (one round for <= 3072 items)
            for (int w0 = tid; w0 < items; w0 += R * NT) {
                uint32_t lo[R], hi[R], hi0[R];
                int32_t doc[R];
                int mpos[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int w = w0 + r * NT;
                    lo[r] = hi[r] = hi0[r] = 0u;
                    doc[r] = 0;
                    mpos[r] = -1;
                    if (w < items) {
                        const int e = w / nqc, j = w - e * nqc;
                        doc[r] = ci[e0 + e];
                        const int32_t *to = tile_off + (int64_t)s_tok[j] * (n_tab + 1) + (doc[r] >> tab_shift);
                        lo[r] = s_ip[j] + (uint32_t)to[0];
                        hi[r] = hi0[r] = s_ip[j] + (uint32_t)to[1];
                        mpos[r] = e * ld + j;
                    }
                }
                ERH_FQ(11);                                               // search set-up (list entry, two skip-table entries)
                // first posting with document >= doc, R at a time.  (Binary on purpose: what this stage costs is the number of
                // cache lines the probes touch -- about one per cycle and CU -- not the round trips; a 4-ary search has half
                // the steps and 1.5 x the probes, and measured the same.)
                for (;;) {
                    bool act[R];
                    uint32_t mid[R];
                    int32_t dv[R];
                    bool any = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        act[r] = lo[r] < hi[r];
                        mid[r] = (lo[r] + hi[r]) >> 1;
                        dv[r] = act[r] ? doc_ids[mid[r]] : 0;
                        any |= act[r];
                    }
                    if (!any) break;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (act[r]) { if (dv[r] < doc[r]) lo[r] = mid[r] + 1u; else hi[r] = mid[r]; }
                    }
                }
                ERH_FQ(12);                                               // searches
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (mpos[r] >= 0) {
                        ST val = (ST)0;
                        if (lo[r] < hi0[r] && doc_ids[lo[r]] == doc[r]) val = payload[lo[r]];
                        M[mpos[r]] = val;
                    }
                }
            }
            ERH_FQ(13);                                                   // payloads
            __syncthreads();
            for (int e = tid; e < ec; e += NT) {
                ST s = fs[e0 + e];
                for (int j = 0; j < nqc; ++j) s = s + M[e * ld + j];      // token order; + 0.0 for an absent token
                fs[e0 + e] = s;
            }
            __syncthreads();
        }
    }
    ERH_FQ(14);                                                           // sums
    // ---- rank by counting: entry e goes to position #{entries that beat it} (keys (score, index) are distinct) -----------
    int *rank = reinterpret_cast<int *>(ca);                              // the approximate sums are dead
    __syncthreads();
    for (int i = tid; i < n_keep; i += NT) rank[i] = 0;
    __syncthreads();
    if (n_keep > 0) {
        typedef ST SV __attribute__((ext_vector_type(4)));
        typedef int32_t IV __attribute__((ext_vector_type(4)));
        const int parts = n_keep < NT ? NT / n_keep : 1;
        const int chunk = (((n_keep + parts - 1) / parts) + 3) & ~3;     // a multiple of four: the walk reads 16-byte groups
        for (int w = tid; w < n_keep * parts; w += NT) {
            const int part = w / n_keep, e = w - part * n_keep;           // consecutive lanes: consecutive entries, same part
            const ST se = fs[e];
            const int32_t ie = ci[e];
            const int j1 = (part + 1) * chunk < n_keep ? (part + 1) * chunk : n_keep;
            int cnt = 0;
            for (int j = part * chunk; j < j1; j += 4) {                  // (entries past n_keep: stale, masked)
                const SV sj = *reinterpret_cast<const SV *>(fs + j);
                const IV ij = *reinterpret_cast<const IV *>(ci + j);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    cnt += (j + u < j1 && (sj[u] > se || (sj[u] == se && ij[u] < ie))) ? 1 : 0;
            }
            if (cnt) atomicAdd(&rank[e], cnt);
        }
    }
    __syncthreads();
    const int n = n_keep < k ? n_keep : k;
    for (int e = tid; e < n_keep; e += NT) {
        const int r = rank[e];
        if (r < k) { part_scores[out_base + r] = (double)fs[e]; part_ids[out_base + r] = ci[e]; }
    }
    for (int i = n + tid; i < k; i += NT) { part_scores[out_base + i] = 0.0; part_ids[out_base + i] = -1; }
    if (tid == 0) part_len[out_slot] = n;
    ERH_FQ(15);                                                           // rank + output
#undef ERH_FQ
}

// grid = (segs, B), block = C::NT.  tile_off has n_tab + 1 entries per term at a granularity of C::TILE >> tshift documents;
// post = the interleaved fixed-point postings with two sentinels {document -1, q 0} at index nnz.
// The body is a device function: bm25_ascan_kernel runs it for every workgroup of a launch, bm25_ascan_mixed_kernel picks the shape per query.
template <typename ST, class C>
__device__ __forceinline__ void as_scan_query(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
    const void *__restrict__ post /* as_uint2 {document, q}, or for C::P16 the 4-byte postings */, uint32_t nnz,
    double qmax /* largest fixed-point payload of the index */, int g16 /* C::P16: the 4-byte postings hold (q >> g16) + 1 */,
    const int32_t *__restrict__ tile_off, int n_tab, int tshift, int n_tiles, int64_t N,
    const int32_t *__restrict__ q_indptr, const int32_t *__restrict__ q_tok, const int32_t *__restrict__ q_order, int k,
    int segs, int cut_mul /* segment boundaries fall on multiples of cut_mul tiles (the exact scan's tile may be larger) */,
    const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    double *__restrict__ part_scores, int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len,
    uint32_t *__restrict__ redo, unsigned long long *__restrict__ stats /* null, or erh_get_stat's device counters */,
    const int32_t *__restrict__ dir_rng /* null, or {first document, last + 1} per dir class: a filtered query walks only the tiles of its class */,
    int dir_rng_n,
    int abl /* measurement builds: 1 no adds, 2 no posting loads, 4 no clear, 8 one descriptor set */,
    unsigned long long *__restrict__ dbg,
    int32_t *__restrict__ fin_ids /* null, or [B * segs][CAP]: the workgroup hands its final list to bm25_finish_kernel instead of re-scoring it */,
    int32_t *__restrict__ fin_cnt /* ... [B * segs]: entries of that list, -1: nothing to finish (the segment goes to the exact scan) */,
    const int q, const int seg /* the workgroup's query and document-range segment (of `segs`): its lists go to slot q * segs + seg */) {
    constexpr int NT = C::NT, TILE = C::TILE, NW = C::NW, CAP = C::CAP, WORDS = C::WORDS, U = C::U;
#ifdef ERH_MEASURE
#define ERH_ABL(B) (abl & (B))
    long long t_sec[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [8] the final shrink, [9] exact re-score + rank + output
    long long t_mark = dbg ? clock64() : 0;
#define ERH_SEC(I) do { if (dbg) { const long long n_ = clock64(); t_sec[I] += n_ - t_mark; t_mark = n_; } } while (0)
#else
#define ERH_ABL(B) 0
#define ERH_SEC(I) do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BmHdr *hdr = reinterpret_cast<BmHdr *>(smem);
    AsHdr *h2 = reinterpret_cast<AsHdr *>(smem + kAsOffHdr2);
    int *xzb = reinterpret_cast<int *>(smem + kAsOffXcnt);
    as_int4 *rng = reinterpret_cast<as_int4 *>(smem + kAsOffRng);
    int *pxs = reinterpret_cast<int *>(smem + kAsOffPx);
    uint32_t *accu = reinterpret_cast<uint32_t *>(smem + kAsOffAcc);
    uint32_t *ca = reinterpret_cast<uint32_t *>(smem + C::OFF_CA);
    int32_t *ci = reinterpret_cast<int32_t *>(smem + C::OFF_CI);
    int32_t *xl = reinterpret_cast<int32_t *>(smem + C::OFF_XL);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + C::OFF_HIST);
    uint32_t *dummy = reinterpret_cast<uint32_t *>(smem + C::OFF_DUMMY);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qs = q_indptr[q];
    const int nq = q_indptr[q + 1] - qs;
    const int fd = filter_dir ? (int)filter_dir[q] : -1;
    const int n_cut = (n_tiles + cut_mul - 1) / cut_mul;
    int t_begin = (int)((int64_t)n_cut * seg / segs) * cut_mul;
    int t_end = (int)((int64_t)n_cut * (seg + 1) / segs) * cut_mul;
    t_end = t_end < n_tiles ? t_end : n_tiles;
    // Filter push-down as a tile range (round 5): the reference's `dir` is the first path component of a file and its loader walks the
    // directories one after the other (ref:src/easyrag/custom/transformation.py:70, ingestion.py:79-87), so the documents of a dir
    // are one contiguous block; a query filtered on it walks the tiles of that block only.  The segment cuts stay where they are
    // (the exact scan of a redone segment covers the same documents); interleaved classes simply span every tile.
    if (fd >= 0 && dir_rng) {
        int lo = 0, hi = 0;
        if (fd < dir_rng_n) { lo = dir_rng[2 * fd]; hi = dir_rng[2 * fd + 1]; }
        if (hi <= lo) {
            t_end = t_begin;                                              // no document carries this class
        } else {
            const int b0 = lo / TILE, b1 = (hi + TILE - 1) / TILE;
            t_begin = t_begin > b0 ? t_begin : b0;
            t_end = t_end < b1 ? t_end : b1;
            t_end = t_end > t_begin ? t_end : t_begin;
        }
    }
    const int64_t out_slot = (int64_t)q * segs + seg, out_base = out_slot * k;
    const double keep_frac = 1.0 - 3.0 * 1.01 * (double)(nq + 2) * 5.9604644775390625e-08;   // 1 - 3 eps
    // units a sum can exceed the real one by (as_drop_threshold): one per truncation + 1 the payload has gone through
    const int n_err = C::P16 ? 3 * nq : C::PACK ? 2 * nq : nq;
    const double qm_eff = C::P16 ? (double)((uint64_t)qmax >> g16) + 1.0 : qmax;
    const int sh = C::PACK ? as_pack_shift(nq, qm_eff) : 0;               // packed sums: right shift of the stored payloads

    if (tid == 0) {
        hdr->ncand = 0; hdr->total = 0; hdr->tau_idx = 0x7fffffff; hdr->tau_s = 0.0;
        hdr->full[0] = hdr->full[1] = hdr->full[2] = 0;
        hdr->want[0] = hdr->want[1] = hdr->want[2] = 0;
        h2->thetaq = 0u; h2->thq = 1u; h2->keep = 0;
        h2->redo = (C::PACK ? sh >= 32 : (double)nq * qmax >= 4294967296.0) ? 1 : 0;   // the sums could overflow: leave it to the exact scan
    }
    if (tid < 2 * (NW + 1)) xzb[tid] = 0;
    if (tid < 64) dummy[tid] = 0u;
    for (int i = tid; i < WORDS; i += NT) accu[i] = 0u;
    __syncthreads();

    int ph = 0;                                                           // sweep pass counter (workgroup-uniform)
    // The tile's sums are complete (barrier behind the adds): survivors -> list, accumulators cleared.
    // nc0 = hdr->ncand as it was before the tile (nobody changes it during the adds).  true: give up (redo).
    auto finish_tile = [&](int tile, int base_doc, uint32_t thq, int nc0) __attribute__((always_inline)) -> bool {
        if (h2->redo) return true;                                        // (the publishing wave saw a tile it cannot describe)
        const int par = tile & 1;
        const int *xz = xzb + par * (NW + 1);
        const int xmax = xz[NW];                                          // most crossings noted for one region
        const int cnt = xz[wave];                                         // ... for this wave's region
        if (tid < NW + 1) xzb[(par ^ 1) * (NW + 1) + tid] = 0;                      // the next tile's counters (idle until the next barrier)
        bool by_list = thq > 1u && xmax <= kAsXW;                         // workgroup-uniform
        if (by_list && nc0 + NW * xmax > CAP) {                  // make room first (rare)
            as_shrink<CAP>(hdr, h2, ca, ci, hist, k, keep_frac, n_err, C::RESERVE);
            if (h2->redo) return true;
            by_list = hdr->ncand + NW * xmax <= CAP;             // (stable: read behind as_shrink's last barrier)
        }
        if (by_list) {
            // this wave's own region: the noted slots hold final sums (>= the thq of the adds; the threshold may have moved
            // up in a shrink just now), then the clear -- program order inside the wave, no barrier in between
            if (cnt > 0) {
                const uint32_t thn = h2->thq;
                for (int i = lane; i < cnt; i += 64) {
                    const int sl = xl[wave * kAsXW + i];
                    const uint32_t av = C::PACK ? (accu[sl & (WORDS - 1)] >> ((sl >> 14) << 4)) & 0xffffu : accu[sl];
                    const int64_t doc = (int64_t)base_doc + sl;
                    bool pass = av >= thn;
                    if (pass && (doc >= N || (fd >= 0 && (int)dir_id[doc] != fd))) pass = false;
                    if (pass) {
                        const int pos = atomicAdd(&hdr->ncand, 1);        // fits: ncand + 16 xmax <= CAP
                        ca[pos] = av;
                        ci[pos] = (int32_t)doc;
                    }
                }
            }
            typedef uint32_t UT __attribute__((ext_vector_type(4)));
            const UT z = {0u, 0u, 0u, 0u};
            uint32_t *mine = accu + wave * kAsRegion + lane * 4;
            if (!ERH_ABL(4)) {
#pragma unroll
                for (int i = 0; i < kAsRegion / 256; ++i) *reinterpret_cast<UT *>(mine + i * 256) = z;
            }
            ERH_SEC(3);
            __syncthreads();
            ERH_SEC(4);
            if (hdr->ncand > k + NT / 2) {                        // uniform: keep the list short, the threshold current
                as_shrink<CAP>(hdr, h2, ca, ci, hist, k, keep_frac, n_err, C::RESERVE);
                ERH_SEC(5);
                if (h2->redo) return true;
            }
            return false;
        }
        return as_sweep_tile<C>(hdr, h2, accu, ca, ci, hist, tile == t_begin, base_doc, N, fd, dir_id, k, keep_frac, n_err, ph);
    };

    bool stop = h2->redo != 0;
    if (nq > 0 && t_begin < t_end && !stop) {
        AsSet<U> S;
        auto apply = [&](int dc, int r0, int np, int base_doc, uint32_t thx, int *xz) __attribute__((always_inline)) {
            if constexpr (C::P16) as_apply16<NW, U>(S, dc, r0, np, lane, accu, dummy, thx, xl, xz, sh);
            else as_apply<NW, U, C::PACK>(S, dc, r0, np, lane, accu, dummy, base_doc, thx, xl, xz, sh);
        };
        auto pieces_of = [&](int pt, int nd) __attribute__((always_inline)) -> int {   // this wave's share of a tile's pt pieces, dealt to waves 0 .. nd - 1
            return (wave < nd && pt > wave) ? (pt - wave + nd - 1) / nd : 0;
        };
        if (nq <= kAsTokChunk) {
            // the publishing wave, lane j: token j's posting base and its row of the skip table; ranges are published two tiles ahead
            uint32_t ip = 0u;
            const int32_t *fo = tile_off;
            constexpr int PW = NW - 1;                                    // the publishing wave: pieces are dealt round-robin from wave 0, the last wave has the fewest
            constexpr int ND = NW;                                        // waves the tile's pieces are dealt to (a publishing wave WITHOUT pieces measured +2 %)
            const bool tokl = wave == PW && lane < nq;
            if (tokl) {
                const int64_t tok = q_tok[qs + lane];
                ip = (uint32_t)indptr[tok];
                fo = tile_off + tok * (int64_t)(n_tab + 1);
            }
            auto raw = [&](int t, int &a, int &b) __attribute__((always_inline)) {   // publishing wave only; t is clamped
                t = t < t_end ? t : t_end - 1;
                int i0 = t << tshift, i1 = (t + 1) << tshift;
                i0 = i0 < n_tab ? i0 : n_tab;
                i1 = i1 < n_tab ? i1 : n_tab;
                a = tokl ? fo[i0] : 0;
                b = tokl ? fo[i1] : 0;
            };
            auto publish = [&](int slot, int a, int b) __attribute__((always_inline)) {   // publishing wave only
                const as_int4 r = as_make_ranges<C::PIECE>(ip, a, b, nq, lane);
                rng[slot * 64 + lane] = r;
                if (lane < 16) pxs[slot * 16 + lane] = r[0];
                if (lane == 0 && r[3] > 64 * ND) h2->redo = 1;      // more pieces than the waves' lanes can describe
            };
            int ra = 0, rb = 0;
            if (wave == PW) {
                raw(t_begin, ra, rb);
                publish(0, ra, rb);
                raw(t_begin + 1, ra, rb);
                publish(1, ra, rb);
                raw(t_begin + 2, ra, rb);                                 // consumed in the first tile body
            }
            __syncthreads();
            uint32_t ds_c, ds_n = 0u;                                     // this tile's / the next tile's piece descriptors
            int dc_c, dc_n = 0;
            as_describe<ND, C::PIECE>(rng, reinterpret_cast<const as_int4 *>(pxs), nq, lane, wave, ds_c, dc_c);
            int np_c = pieces_of(__builtin_amdgcn_readfirstlane(rng[0][3]), ND), np_n = 0;
            as_fill<U, C::P16>(S, ds_c, dc_c, 0, np_c, post, lane, nnz);
            int r3 = 0;                                                   // (tile - t_begin) % 3: slot of the current tile's ranges
            for (int tile = t_begin; tile < t_end && !stop; ++tile) {
                const int base_doc = tile * TILE;
                const uint32_t thq = h2->thq;                             // fixed for the tile (it only moves in as_shrink)
                const int nc0 = hdr->ncand;
                int *xz = xzb + (tile & 1) * (NW + 1);
                const int r_n = r3 == 2 ? 0 : r3 + 1, r_nn = r_n == 2 ? 0 : r_n + 1;
                if (!ERH_ABL(8)) {                                        // next tile (clamped past the end)
                    as_describe<ND, C::PIECE>(rng + r_n * 64, reinterpret_cast<const as_int4 *>(pxs + r_n * 16), nq, lane, wave, ds_n, dc_n);
                    np_n = pieces_of(__builtin_amdgcn_readfirstlane(rng[r_n * 64][3]), ND);
                }
                ERH_SEC(0);
#ifdef ERH_MEASURE
                if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ERH_SEC(6); }   // (measurement: load wait on its own, booked under [6])
#endif
                const uint32_t thx = thq > 1u ? thq : 0u;
                if (!ERH_ABL(1)) apply(dc_c, 0, np_c, base_doc, thx, xz);   // (requested a tile ago)
#ifdef ERH_MEASURE
                if (dbg) ERH_SEC(7);                                      // (measurement: the adds on their own, booked under [7])
#endif
                for (int r0 = U; r0 < np_c; r0 += U) {              // more pieces than the register slots hold (long posting lists)
                    as_fill<U, C::P16>(S, ds_c, dc_c, r0, np_c, post, lane, nnz);
                    apply(dc_c, r0, np_c, base_doc, thx, xz);
                }
                if (wave == PW) {                                          // ranges of tile + 2 -> LDS, skip-table entries of tile + 3
                    publish(r_nn, ra, rb);
                    raw(tile + 3, ra, rb);
                }
                // (the next tile's postings are requested LAST: the publishing wave's wait for its two table entries would otherwise
                // -- the counter is in order -- also wait for them)
                if (!ERH_ABL(2)) as_fill<U, C::P16>(S, ds_n, dc_n, 0, np_n, post, lane, nnz);   // lands during the bookkeeping below
                ERH_SEC(1);
                __syncthreads();                                          // every posting of the tile is in its sum
                ERH_SEC(2);
                stop = finish_tile(tile, base_doc, thq, nc0);
                ds_c = ds_n; dc_c = dc_n; np_c = np_n;
                r3 = r_n;
            }
        } else {
            // long queries: chunks of 64 tokens; wave 0 publishes the chunk's ranges, everybody fetches and applies on the spot
            for (int tile = t_begin; tile < t_end && !stop; ++tile) {
                const int base_doc = tile * TILE;
                const uint32_t thq = h2->thq;
                const int nc0 = hdr->ncand;
                const uint32_t thx = thq > 1u ? thq : 0u;
                int *xz = xzb + (tile & 1) * (NW + 1);
                for (int c0 = 0; c0 < nq; c0 += kAsTokChunk) {
                    const int nqc = nq - c0 < kAsTokChunk ? nq - c0 : kAsTokChunk;
                    if (wave == 0) {
                        int a = 0, b = 0;
                        uint32_t ipc = 0u;
                        if (lane < nqc) {
                            const int64_t tok = q_tok[qs + c0 + lane];
                            ipc = (uint32_t)indptr[tok];
                            const int32_t *foc = tile_off + tok * (int64_t)(n_tab + 1);
                            int i0 = tile << tshift, i1 = (tile + 1) << tshift;
                            i0 = i0 < n_tab ? i0 : n_tab;
                            i1 = i1 < n_tab ? i1 : n_tab;
                            a = foc[i0];
                            b = foc[i1];
                        }
                        const as_int4 r = as_make_ranges<C::PIECE>(ipc, a, b, nqc, lane);
                        rng[lane] = r;
                        if (lane == 0 && r[3] > 64 * NW) h2->redo = 1;
                    }
                    __syncthreads();
                    uint32_t ds;
                    int dc;
                    as_describe<NW, C::PIECE>(rng, nullptr, nqc, lane, wave, ds, dc);
                    const int np = pieces_of(__builtin_amdgcn_readfirstlane(rng[0][3]), NW);
                    for (int r0 = 0; r0 < np; r0 += U) {
                        as_fill<U, C::P16>(S, ds, dc, r0, np, post, lane, nnz);
                        apply(dc, r0, np, base_doc, thx, xz);
                    }
                    __syncthreads();                                      // (the ranges are overwritten by the next chunk)
                }
                ERH_SEC(2);
                stop = finish_tile(tile, base_doc, thq, nc0);
            }
        }
    }
    if (!stop) as_shrink<CAP>(hdr, h2, ca, ci, hist, k, keep_frac, n_err, 0);     // final list: k entries + the near ties of the k-th
    if (h2->redo) {                                                       // workgroup-uniform
        if (tid == 0) {
            redo[out_slot] = 1u;
            part_len[out_slot] = 0;
            if (fin_cnt) fin_cnt[out_slot] = -1;
            if (stats) atomicAdd(&stats[1], 1ull);
        }
        return;
    }
    if (ERH_ABL(0xff)) { if (tid == 0) { part_len[out_slot] = 0; if (fin_cnt) fin_cnt[out_slot] = -1; } return; }   // (ablations: the list is garbage)
    ERH_SEC(8);
    if (fin_ids) {
        // split finish (round 6): the exact re-score + rank of all lists of the batch run in ONE kernel behind the scan (2 M independent
        // binary searches spread over the chip at 16 waves per CU), and this workgroup's CU slot goes to the next query now
        const int64_t slot = out_slot;
        const int n_keep = hdr->ncand;
        for (int i = tid; i < n_keep; i += NT) fin_ids[slot * CAP + i] = ci[i];
        if (tid == 0) fin_cnt[slot] = n_keep;
        return;
    }
    as_finish_query<ST, C>(smem, hdr, ca, ci, indptr, doc_ids, payload, tile_off, n_tab, C::TAB_SHIFT - tshift, q_tok, qs, nq, k,
                           out_base, out_slot, part_scores, part_ids, part_len, dbg);
    ERH_SEC(9);
#ifdef ERH_MEASURE
    if (dbg && tid == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) atomicAdd(&dbg[i], (unsigned long long)t_sec[i]);
    }
#endif
#undef ERH_SEC
#undef ERH_ABL
}

template <typename ST, class C>
__global__ __launch_bounds__(C::NT, 4 /* waves per SIMD: two 512-thread workgroups per CU */) void bm25_ascan_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
    const void *__restrict__ post, uint32_t nnz, double qmax, int g16,
    const int32_t *__restrict__ tile_off, int n_tab, int tshift, int n_tiles, int64_t N,
    const int32_t *__restrict__ q_indptr, const int32_t *__restrict__ q_tok, const int32_t *__restrict__ q_order, int k,
    int segs, int cut_mul, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    double *__restrict__ part_scores, int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len,
    uint32_t *__restrict__ redo, unsigned long long *__restrict__ stats, const int32_t *__restrict__ dir_rng, int dir_rng_n,
    int abl, unsigned long long *__restrict__ dbg, int32_t *__restrict__ fin_ids, int32_t *__restrict__ fin_cnt) {
    as_scan_query<ST, C>(indptr, doc_ids, payload, post, nnz, qmax, g16, tile_off, n_tab, tshift, n_tiles, N, q_indptr, q_tok, q_order, k, segs, cut_mul,
                         filter_dir, dir_id, part_scores, part_ids, part_len, redo, stats, dir_rng, dir_rng_n, abl, dbg, fin_ids, fin_cnt,
                         q_order ? q_order[blockIdx.y] : (int)blockIdx.y, (int)blockIdx.x);
}

// One launch, two shapes (round 6): a query of more than `long_tokens` tokens is too coarse on 16-bit sums (as_pack_shift leaves it 65535 / nq
// payload levels and an error of 3 nq of them: its lists never shrink), but sending the WHOLE batch to the 32-bit shape for the sake of a few long
// questions costs every short one 20 % (1024 queries with the reference's question lengths, 32 of them longer than 28 tokens: 0.66 ms on the
// 32-bit shape, 0.54 when the long ones are left out and the rest scans packed; profiles/r06m_bm25_segs_probe.log).  Here the workgroup of a long
// query runs the 32-bit body over 16384-document tiles (AsSmall), every other one the packed body over the 4-byte postings (AsPack16) --
// both 512 threads, 80 KiB of LDS, 128 VGPRs, the same segment cuts (cut_mul), the same outputs.  *_s: the 32-bit shape's posting copy and skip table.
template <typename ST>
__global__ __launch_bounds__(AsPack16::NT, 4) void bm25_ascan_mixed_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
    const void *__restrict__ post, uint32_t nnz, double qmax, int g16,
    const int32_t *__restrict__ tile_off, int n_tab, int tshift, int n_tiles, int64_t N,
    const int32_t *__restrict__ q_indptr, const int32_t *__restrict__ q_tok, const int32_t *__restrict__ q_order, int k,
    int segs, int cut_mul, const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    double *__restrict__ part_scores, int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len,
    uint32_t *__restrict__ redo, unsigned long long *__restrict__ stats, const int32_t *__restrict__ dir_rng, int dir_rng_n,
    unsigned long long *__restrict__ dbg,
    int long_tokens, const void *__restrict__ post_s, const int32_t *__restrict__ tile_off_s, int n_tab_s, int tshift_s, int n_tiles_s, int cut_mul_s,
    const int32_t *__restrict__ q_items /* null: grid = (segs, B) as every other scan; else grid = (1, items), item = query | segment << 24 -- a long
                                           query comes as segs_l items (its documents cut into segs_l ranges: one 45-token question in ONE workgroup
                                           is the launch's tail otherwise), every other query as one */,
    int segs_l, double *__restrict__ l_scores, int32_t *__restrict__ l_ids, int32_t *__restrict__ l_len /* [B][segs_l] partial lists of the long queries */,
    uint32_t *__restrict__ l_redo) {
    static_assert(AsPack16::NT == AsSmall::NT, "one block size for both bodies");
    int q, seg;
    if (q_items) {
        const int item = q_items[blockIdx.y];
        q = item & 0xffffff;
        seg = (int)((uint32_t)item >> 24);
    } else {
        q = q_order ? q_order[blockIdx.y] : (int)blockIdx.y;
        seg = (int)blockIdx.x;
    }
    const int nq = q_indptr[q + 1] - q_indptr[q];
    if (nq > long_tokens) {                                                // workgroup-uniform
        if (q_items)
            as_scan_query<ST, AsSmall>(indptr, doc_ids, payload, post_s, nnz, qmax, g16, tile_off_s, n_tab_s, tshift_s, n_tiles_s, N, q_indptr, q_tok, q_order, k,
                                       segs_l, cut_mul_s, filter_dir, dir_id, l_scores, l_ids, l_len, l_redo, stats, dir_rng, dir_rng_n, 0, dbg, nullptr, nullptr, q, seg);
        else
            as_scan_query<ST, AsSmall>(indptr, doc_ids, payload, post_s, nnz, qmax, g16, tile_off_s, n_tab_s, tshift_s, n_tiles_s, N, q_indptr, q_tok, q_order, k,
                                       segs, cut_mul_s, filter_dir, dir_id, part_scores, part_ids, part_len, redo, stats, dir_rng, dir_rng_n, 0, dbg, nullptr, nullptr, q, seg);
    } else {
        as_scan_query<ST, AsPack16>(indptr, doc_ids, payload, post, nnz, qmax, g16, tile_off, n_tab, tshift, n_tiles, N, q_indptr, q_tok, q_order, k,
                                    segs, cut_mul, filter_dir, dir_id, part_scores, part_ids, part_len, redo, stats, dir_rng, dir_rng_n, 0, dbg, nullptr, nullptr, q, seg);
    }
}

// The exact re-score + rank of the fixed-point scan's final lists as a kernel of its own (option bm25_split_finish): grid = (segs, B),
// 256 threads, ~37 KiB of LDS = four workgroups per CU.  Same arithmetic, same order, same output as the scan kernel's tail
// (as_finish_query is shared): one binary search per (listed document, query token) inside the skip table's tile range, payloads
// summed in token order in the library's type, rank by counting.
template <int CAP_>
struct AsFinCfg {
    static constexpr int NT = 256, TILE = 0, CAP = CAP_;
    static constexpr int WORDS = 6144;                                  // 24 KiB: exact sums [CAP] + the (entry, token) payload matrix
    static constexpr size_t OFF_CA = kAsOffAcc + (size_t)WORDS * 4;
    static constexpr size_t OFF_CI = OFF_CA + (size_t)CAP * 4;
    static constexpr size_t BYTES = OFF_CI + (size_t)CAP * 4;
    static_assert((size_t)WORDS * 4 >= (size_t)CAP * 8 + 2048, "room for the payload matrix behind the exact sums");
};
template <typename ST, class F>
__global__ __launch_bounds__(F::NT) void bm25_finish_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
    const int32_t *__restrict__ tile_off, int n_tab, int tab_shift, const int32_t *__restrict__ q_indptr,
    const int32_t *__restrict__ q_tok, const int32_t *__restrict__ q_order, int k, int segs, const int32_t *__restrict__ fin_ids,
    const int32_t *__restrict__ fin_cnt, double *__restrict__ part_scores, int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BmHdr *hdr = reinterpret_cast<BmHdr *>(smem);
    uint32_t *ca = reinterpret_cast<uint32_t *>(smem + F::OFF_CA);
    int32_t *ci = reinterpret_cast<int32_t *>(smem + F::OFF_CI);
    const int seg = blockIdx.x, q = q_order ? q_order[blockIdx.y] : (int)blockIdx.y, tid = threadIdx.x;   // (the scan launch's queries)
    const int64_t slot = (int64_t)q * segs + seg;
    const int n_keep = fin_cnt[slot];
    if (n_keep < 0) return;                                              // (redo: the exact block scan answers this segment)
    if (tid == 0) hdr->ncand = n_keep;
    for (int i = tid; i < n_keep; i += F::NT) ci[i] = fin_ids[slot * F::CAP + i];
    __syncthreads();
    const int qs = q_indptr[q];
    as_finish_query<ST, F>(smem, hdr, ca, ci, indptr, doc_ids, payload, tile_off, n_tab, tab_shift, q_tok, qs, q_indptr[q + 1] - qs, k,
                           slot * k, slot, part_scores, part_ids, part_len);
}

// interleaved fixed-point postings of the scan above: post[i] = {document, trunc(p32 * scale) + 1}, post[nnz] = post[nnz + 1] = {-1, 0}
__global__ void bm25_post_kernel(const int32_t *__restrict__ doc_ids, const float *__restrict__ pay32, int64_t nnz, float scale,
                                 as_uint2 *__restrict__ post) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nnz + 1; i += (int64_t)gridDim.x * blockDim.x) {
        as_uint2 v;
        if (i < nnz) { v.x = (uint32_t)doc_ids[i]; v.y = (uint32_t)(pay32[i] * scale) + 1u; }
        else { v.x = 0xffffffffu; v.y = 0u; }
        post[i] = v;
    }
}

// 4-byte postings of the packed scan: post16[i] = (document & 32767) | (((q >> g) + 1) << 16), zeros from index nnz on
__global__ void bm25_post16_kernel(const as_uint2 *__restrict__ post, int64_t nnz, int g, uint32_t *__restrict__ post16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz + 8; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0u;
        if (i < nnz) { const as_uint2 p = post[i]; v = (p.x & 32767u) | (((p.y >> g) + 1u) << 16); }
        post16[i] = v;
    }
}

// bits[0] = max over the fp32 payloads (positive floats order like their bit patterns)
__global__ void bm25_payload_max_kernel(const float *__restrict__ p, int64_t n, uint32_t *__restrict__ bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = p[i] > m ? p[i] : m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(bits, __float_as_uint(m));
}

__global__ void narrow_f64_kernel(const double *__restrict__ in, int64_t n, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}

// Merge `segs` sorted partial lists of one query: grid = B, block = 1024, LDS = P*(8+4) (+64), P = pow2 >= segs*k.
__global__ __launch_bounds__(kBmThreads) void bm25_merge_kernel(
    int k, int segs, int P, const double *__restrict__ part_scores, const int32_t *__restrict__ part_ids,
    const int32_t *__restrict__ part_len, int32_t *__restrict__ out_ids, double *__restrict__ out_scores,
    int32_t *__restrict__ out_len, const int32_t *__restrict__ q_list /* null, or workgroup -> query (the long queries of a mixed launch) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int &s_n = *reinterpret_cast<int *>(smem);
    double *cs = reinterpret_cast<double *>(smem + 64);
    int32_t *ci = reinterpret_cast<int32_t *>(smem + 64 + (size_t)P * 8);
    const int q = q_list ? q_list[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    for (int i = tid; i < P; i += kBmThreads) { cs[i] = -1.0; ci[i] = 0x7fffffff; }
    __syncthreads();
    for (int i = tid; i < segs * k; i += kBmThreads) {
        const int seg = i / k, r = i % k;
        if (r < part_len[(int64_t)q * segs + seg]) {
            const int64_t src = ((int64_t)q * segs + seg) * k + r;
            cs[i] = part_scores[src];
            ci[i] = part_ids[src];
            atomicAdd(&s_n, 1);
        }
    }
    erh_bitonic_rec_desc<double>(cs, ci, P);
    const int n = s_n < k ? s_n : k;
    for (int i = tid; i < k; i += kBmThreads) {
        const int64_t dst = (int64_t)q * k + i;
        if (i < n) { out_ids[dst] = ci[i]; out_scores[dst] = cs[i]; }
        else { out_ids[dst] = -1; out_scores[dst] = 0.0; }
    }
    if (tid == 0) out_len[q] = n;
}

// get_scores parity path: one launch per query token, in order; inside a term each doc occurs once.
template <typename ST>
__global__ void bm25_add_term_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids,
                                     const ST *__restrict__ payload, int32_t term, ST *__restrict__ scores) {
    const int64_t s = indptr[term], e = indptr[term + 1];
    for (int64_t p = s + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t doc = doc_ids[p];
        scores[doc] = scores[doc] + payload[p];
    }
}

// flag[0] = 1 if any payload is <= 0, subnormal or NaN: the crossing path of the wave-owned scan needs strictly growing
// sums of normal numbers
template <typename ST>
__global__ void bm25_payload_sign_kernel(const ST *__restrict__ payload, int64_t nnz, uint32_t *__restrict__ flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
        bad |= !(payload[i] >= (sizeof(ST) == 4 ? (ST)1.17549435e-38f : (ST)2.2250738585072014e-308));   // positive and normal
    if (__builtin_amdgcn_ballot_w64(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

__global__ void widen_f32_kernel(const float *__restrict__ in, int64_t n, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (double)in[i];
}

}  // namespace

namespace erh {

static int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }

hipError_t bm25_init() {
    hipError_t e;
    e = hipFuncSetAttribute((const void *)bm25_scan_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)BmLds<float>::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_scan_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)BmLds<double>::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_wscan_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)BmLds<float>::OFF_LO + kWsXBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_wscan_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)BmLds<float>::OFF_LO + kWsXBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_wscan_kernel<double, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)BmLds<double>::OFF_LO + kWsXBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_wscan_kernel<double, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)BmLds<double>::OFF_LO + kWsXBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<float, AsBig>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsBig::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<double, AsBig>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsBig::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<float, AsSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsSmall::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<double, AsSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsSmall::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<float, AsPack>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsPack::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<double, AsPack>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsPack::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<float, AsPack16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsPack16::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_kernel<double, AsPack16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsPack16::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_mixed_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAsMixedBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_ascan_mixed_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAsMixedBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_finish_kernel<float, AsFinCfg<AsPack16::CAP>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsFinCfg<AsPack16::CAP>::BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bm25_finish_kernel<double, AsFinCfg<AsPack16::CAP>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AsFinCfg<AsPack16::CAP>::BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)bm25_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               8192 * 12 + 64);
}

hipError_t launch_bm25_tile_off(const int64_t *indptr, const int32_t *doc_ids, int64_t V, int tile_docs,
                                int n_tiles, int32_t *tile_off, hipStream_t st) {
    const int64_t total = V * (n_tiles + 1);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(bm25_tile_off_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       indptr, doc_ids, V, tile_docs, n_tiles, tile_off);
    return hipGetLastError();
}

hipError_t launch_bm25_payload(int variant, int64_t V, int64_t nnz, const int64_t *indptr, const int32_t *doc_ids,
                               const int32_t *tf, const int32_t *doc_len, const void *idf, double avgdl,
                               double k1, double b, void *payload, hipStream_t st) {
    if (nnz <= 0) return hipSuccess;
    dim3 grid((unsigned)((nnz + 255) / 256)), block(256);
    if (variant == 0)
        hipLaunchKernelGGL(bm25_payload_kernel<true>, grid, block, 0, st, V, nnz, indptr, doc_ids, tf, doc_len, idf,
                           avgdl, k1, b, payload);
    else
        hipLaunchKernelGGL(bm25_payload_kernel<false>, grid, block, 0, st, V, nnz, indptr, doc_ids, tf, doc_len, idf,
                           avgdl, k1, b, payload);
    return hipGetLastError();
}

hipError_t launch_bm25_scan(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                            const int32_t *tile_off, int n_tiles, int64_t N,
                            const int32_t *q_indptr, const int32_t *q_tok, const int32_t *q_order, int B, int k, int segs,
                            const int16_t *filter_dir, const int16_t *dir_id,
                            double *part_scores, int32_t *part_ids, int32_t *part_len, const uint32_t *only, int cut_tiles,
                            int cut_shift, int ablate, unsigned long long *dbg, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    if (cut_tiles <= 0) { cut_tiles = n_tiles; cut_shift = 0; }
    dim3 grid(segs, B), block(kBmThreads);
    if (variant == 0)
        hipLaunchKernelGGL(bm25_scan_kernel<double>, grid, block, BmLds<double>::BYTES, st, indptr, doc_ids,
                           (const double *)payload, tile_off, n_tiles, N, q_indptr, q_tok, q_order, k, segs, filter_dir, dir_id,
                           part_scores, part_ids, part_len, only, cut_tiles, cut_shift, ablate, dbg);
    else
        hipLaunchKernelGGL(bm25_scan_kernel<float>, grid, block, BmLds<float>::BYTES, st, indptr, doc_ids,
                           (const float *)payload, tile_off, n_tiles, N, q_indptr, q_tok, q_order, k, segs, filter_dir, dir_id,
                           part_scores, part_ids, part_len, only, cut_tiles, cut_shift, ablate, dbg);
    return hipGetLastError();
}

int bm25_ascan_tile_docs(int shape) { return shape == 1 ? AsSmall::TILE : AsBig::TILE; }   // (shape 2, packed: AsBig's tiles)
int bm25_ascan_small_max_k() { return AsSmall::CAP - AsSmall::NT / 2 - 3 * AsSmall::RESERVE; }   // 384: the shrink trigger (k + 256) leaves room for the notes of a tile

hipError_t launch_bm25_ascan(int variant, int small, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                             const void *post, const void *post16, int g16, uint32_t nnz, double qmax, const int32_t *tile_off,
                             int n_tab, int tshift, int64_t N, const int32_t *q_indptr, const int32_t *q_tok,
                             const int32_t *q_order, int B, int k, int segs, int cut_mul, const int16_t *filter_dir,
                             const int16_t *dir_id, double *part_scores, int32_t *part_ids, int32_t *part_len, uint32_t *redo,
                             unsigned long long *stats, const int32_t *dir_rng, int dir_rng_n, int ablate, unsigned long long *dbg,
                             hipStream_t st, int32_t *fin_ids, int32_t *fin_cnt) {
    if (B <= 0) return hipSuccess;
    if (cut_mul < 1) cut_mul = 1;
    const int tile = bm25_ascan_tile_docs(small);
    const int n_tiles = (int)((N + tile - 1) / tile);
    dim3 grid(segs, B);
    if (!small) { fin_ids = nullptr; fin_cnt = nullptr; }              // (the split finish is built for the 512-thread shapes' list capacity)
#define ERH_AS_LAUNCH(ST, CFG, POST)                                                                                 \
    hipLaunchKernelGGL((bm25_ascan_kernel<ST, CFG>), grid, dim3(CFG::NT), CFG::BYTES, st, indptr, doc_ids,           \
                       (const ST *)payload, POST, nnz, qmax, g16, tile_off, n_tab, tshift, n_tiles, N, q_indptr,      \
                       q_tok, q_order, k, segs, cut_mul, filter_dir, dir_id, part_scores, part_ids, part_len, redo, stats, dir_rng, dir_rng_n, ablate, dbg, \
                       fin_ids, fin_cnt)
#define ERH_AS_SHAPES(ST)                                                                                            \
    do {                                                                                                             \
        if (small == 2 && post16) ERH_AS_LAUNCH(ST, AsPack16, post16);                                               \
        else if (small == 2) ERH_AS_LAUNCH(ST, AsPack, post);                                                        \
        else if (small) ERH_AS_LAUNCH(ST, AsSmall, post);                                                            \
        else ERH_AS_LAUNCH(ST, AsBig, post);                                                                         \
    } while (0)
    if (variant == 0) ERH_AS_SHAPES(double); else ERH_AS_SHAPES(float);
#undef ERH_AS_SHAPES
#undef ERH_AS_LAUNCH
    if (fin_ids && fin_cnt) {
        using F = AsFinCfg<AsPack16::CAP>;
        static_assert(AsPack16::CAP == AsPack::CAP && AsPack::CAP == AsSmall::CAP, "one list capacity for the 512-thread shapes");
        const int tab_shift = (small == 1 ? AsSmall::TAB_SHIFT : AsPack::TAB_SHIFT) - tshift;
        if (variant == 0)
            hipLaunchKernelGGL((bm25_finish_kernel<double, F>), grid, dim3(F::NT), F::BYTES, st, indptr, doc_ids, (const double *)payload, tile_off,
                               n_tab, tab_shift, q_indptr, q_tok, q_order, k, segs, fin_ids, fin_cnt, part_scores, part_ids, part_len);
        else
            hipLaunchKernelGGL((bm25_finish_kernel<float, F>), grid, dim3(F::NT), F::BYTES, st, indptr, doc_ids, (const float *)payload, tile_off,
                               n_tab, tab_shift, q_indptr, q_tok, q_order, k, segs, fin_ids, fin_cnt, part_scores, part_ids, part_len);
    }
    return hipGetLastError();
}

// The packed shape for every query of at most `long_tokens` tokens, the 32-bit 16384-document shape for the longer ones, in ONE launch.
// post16 / tile_off / n_tab / tshift: what the packed body walks (32768-document tiles); *_s: the same for the 32-bit body.  cut_mul / cut_mul_s:
// both bodies cut their segments at the same documents.
hipError_t launch_bm25_ascan_mixed(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                                   const void *post16, int g16, uint32_t nnz, double qmax, const int32_t *tile_off, int n_tab, int tshift, int cut_mul,
                                   const void *post_s, const int32_t *tile_off_s, int n_tab_s, int tshift_s, int cut_mul_s, int long_tokens,
                                   int64_t N, const int32_t *q_indptr, const int32_t *q_tok, const int32_t *q_order, int B, int k, int segs,
                                   const int16_t *filter_dir, const int16_t *dir_id, double *part_scores, int32_t *part_ids, int32_t *part_len,
                                   uint32_t *redo, unsigned long long *stats, const int32_t *dir_rng, int dir_rng_n, unsigned long long *dbg,
                                   hipStream_t st, const int32_t *q_items, int n_items, int segs_l, double *l_scores, int32_t *l_ids, int32_t *l_len,
                                   uint32_t *l_redo) {
    if (B <= 0) return hipSuccess;
    const int n_tiles = (int)((N + AsPack16::TILE - 1) / AsPack16::TILE), n_tiles_s = (int)((N + AsSmall::TILE - 1) / AsSmall::TILE);
    if (q_items && (segs != 1 || n_items < B || !l_scores || !l_ids || !l_len || !l_redo)) return hipErrorInvalidValue;
    const dim3 grid = q_items ? dim3(1, n_items) : dim3(segs, B);
    if (variant == 0)
        hipLaunchKernelGGL((bm25_ascan_mixed_kernel<double>), grid, dim3(AsPack16::NT), kAsMixedBytes, st, indptr, doc_ids, (const double *)payload, post16,
                           nnz, qmax, g16, tile_off, n_tab, tshift, n_tiles, N, q_indptr, q_tok, q_order, k, segs, cut_mul, filter_dir, dir_id,
                           part_scores, part_ids, part_len, redo, stats, dir_rng, dir_rng_n, dbg,
                           long_tokens, post_s, tile_off_s, n_tab_s, tshift_s, n_tiles_s, cut_mul_s, q_items, segs_l, l_scores, l_ids, l_len, l_redo);
    else
        hipLaunchKernelGGL((bm25_ascan_mixed_kernel<float>), grid, dim3(AsPack16::NT), kAsMixedBytes, st, indptr, doc_ids, (const float *)payload, post16,
                           nnz, qmax, g16, tile_off, n_tab, tshift, n_tiles, N, q_indptr, q_tok, q_order, k, segs, cut_mul, filter_dir, dir_id,
                           part_scores, part_ids, part_len, redo, stats, dir_rng, dir_rng_n, dbg,
                           long_tokens, post_s, tile_off_s, n_tab_s, tshift_s, n_tiles_s, cut_mul_s, q_items, segs_l, l_scores, l_ids, l_len, l_redo);
    return hipGetLastError();
}

int bm25_ascan_fin_cap() { return AsPack16::CAP; }

// 4-byte postings: the smallest shift g with (qmax >> g) + 1 <= 65535
int bm25_post16_shift(double qmax) {
    const uint64_t qm = (uint64_t)qmax;
    int g = 0;
    while (((qm >> g) + 1ull) > 65535ull) ++g;
    return g;
}

hipError_t launch_bm25_post16(const void *post, int64_t nnz, int g, void *post16, hipStream_t st) {
    const unsigned grid = (unsigned)std::min<int64_t>((nnz + 8 + 255) / 256, 8192);
    hipLaunchKernelGGL(bm25_post16_kernel, dim3(grid), dim3(256), 0, st, (const as_uint2 *)post, nnz, g, (uint32_t *)post16);
    return hipGetLastError();
}

// scale of the fixed-point copy: the largest power of two with 64 * (pmax * scale + 1) < 2^32 (at most 2^30)
float bm25_post_scale(float pmax) {
    const float bound = 64.0f * pmax * 1.001f + 66.0f;
    int S = 0;
    if (bound < 2147483648.0f) { uint32_t b = (uint32_t)bound; S = 0; while (S < 30 && ((uint64_t)(b + 1) << (S + 1)) <= (1ull << 32)) ++S; }
    return ldexpf(1.0f, S);
}

hipError_t launch_bm25_post(const int32_t *doc_ids, const float *pay32, int64_t nnz, float scale, void *post, hipStream_t st) {
    const unsigned g = (unsigned)std::min<int64_t>((nnz + 2 + 255) / 256, 8192);
    hipLaunchKernelGGL(bm25_post_kernel, dim3(g), dim3(256), 0, st, doc_ids, pay32, nnz, scale, (as_uint2 *)post);
    return hipGetLastError();
}


hipError_t launch_bm25_payload_max(const float *pay32, int64_t nnz, uint32_t *bits, hipStream_t st) {
    if (nnz <= 0) return hipSuccess;
    const unsigned g = (unsigned)std::min<int64_t>((nnz + 255) / 256, 4096);
    hipLaunchKernelGGL(bm25_payload_max_kernel, dim3(g), dim3(256), 0, st, pay32, nnz, bits);
    return hipGetLastError();
}

hipError_t launch_narrow_f64(const double *in, int64_t n, float *out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(narrow_f64_kernel, dim3(2048), dim3(256), 0, st, in, n, out);
    return hipGetLastError();
}

int bm25_wscan_max_tokens() { return kWsMaxTok; }
int bm25_wscan_sub_docs(int variant) { return (variant == 0 ? kBm25TileF64 : kBm25TileF32) / kWsWaves; }

hipError_t launch_bm25_wscan(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                             const int32_t *fine_off, int n_fine, int n_tiles, int64_t N,
                             const int32_t *q_indptr, const int32_t *q_tok, const int32_t *q_order, int B, int k, int segs,
                             const int16_t *filter_dir, const int16_t *dir_id,
                             double *part_scores, int32_t *part_ids, int32_t *part_len, int crossing,
                             unsigned long long *dbg, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    dim3 grid(segs, B), block(kBmThreads);
#define ERH_WS_LAUNCH(ST, X)                                                                               \
    hipLaunchKernelGGL((bm25_wscan_kernel<ST, X>), grid, block, BmLds<ST>::OFF_LO + kWsXBytes, st, indptr, doc_ids, \
                       (const ST *)payload, fine_off, n_fine, n_tiles, N, q_indptr, q_tok, q_order, k, segs, filter_dir, dir_id, \
                       part_scores, part_ids, part_len, dbg)
    if (variant == 0) {
        if (crossing) ERH_WS_LAUNCH(double, true); else ERH_WS_LAUNCH(double, false);
    } else {
        if (crossing) ERH_WS_LAUNCH(float, true); else ERH_WS_LAUNCH(float, false);
    }
#undef ERH_WS_LAUNCH
    return hipGetLastError();
}

hipError_t launch_bm25_payload_sign(int variant, const void *payload, int64_t nnz, uint32_t *flag, hipStream_t st) {
    if (nnz <= 0) return hipSuccess;
    const unsigned g = (unsigned)std::min<int64_t>((nnz + 255) / 256, 8192);
    if (variant == 0)
        hipLaunchKernelGGL(bm25_payload_sign_kernel<double>, dim3(g), dim3(256), 0, st, (const double *)payload, nnz, flag);
    else
        hipLaunchKernelGGL(bm25_payload_sign_kernel<float>, dim3(g), dim3(256), 0, st, (const float *)payload, nnz, flag);
    return hipGetLastError();
}

hipError_t launch_bm25_merge(int B, int k, int segs, const double *part_scores, const int32_t *part_ids,
                             const int32_t *part_len, int32_t *out_ids, double *out_scores, int32_t *out_len,
                             hipStream_t st, const int32_t *q_list) {
    if (B <= 0) return hipSuccess;
    const int P = pow2_ge(segs * k < 2 ? 2 : segs * k);
    hipLaunchKernelGGL(bm25_merge_kernel, dim3(B), dim3(kBmThreads), (size_t)P * 12 + 64, st,
                       k, segs, P, part_scores, part_ids, part_len, out_ids, out_scores, out_len, q_list);
    return hipGetLastError();
}

hipError_t launch_bm25_add_term(int variant, const int64_t *indptr, const int32_t *doc_ids, const void *payload,
                                int32_t term, void *scores, hipStream_t st) {
    if (variant == 0)
        hipLaunchKernelGGL(bm25_add_term_kernel<double>, dim3(256), dim3(256), 0, st, indptr, doc_ids,
                           (const double *)payload, term, (double *)scores);
    else
        hipLaunchKernelGGL(bm25_add_term_kernel<float>, dim3(256), dim3(256), 0, st, indptr, doc_ids,
                           (const float *)payload, term, (float *)scores);
    return hipGetLastError();
}

hipError_t launch_widen_f32(const float *in, int64_t n, double *out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(widen_f32_kernel, dim3(1024), dim3(256), 0, st, in, n, out);
    return hipGetLastError();
}

}  // namespace erh
