// Sparse route, kernels K2/K3: BM25 scored over CSR inverted postings with a deterministic, atomics-free
// scatter-add into LDS accumulators and a running top-k.  Replaces BM25Retriever.get_scores + filter
// (/root/reference/src/easyrag/custom/retrievers.py:128-151, 191-210), whose arithmetic is
// rank_bm25.BM25Okapi.get_scores (float64) or bm25s.BM25.get_scores (float32) -- SURVEY.md A.1/A.2.
//
// Bit parity rule: for every document the per-term contributions must be added in query-token order,
// repeats included, in the library's accumulation type.  So:
//   - postings carry the precomputed per-(term, doc) contribution ("eager" payload, as bm25s stores it;
//     for Okapi the same thing in float64), built on the host or by bm25_payload_kernel below;
//   - a workgroup owns one query and walks document tiles; a tile's accumulators live in LDS
//     (32768 fp32 / 16384 fp64 sums); query tokens are applied one after the other with a barrier in
//     between; inside one token every document occurs at most once, so plain LDS read-add-write by the
//     thread that holds the posting is race free and needs no atomics;
//   - per-term tile boundaries come from a skip table tile_off[term][tile] built once per index, so a
//     tile touches exactly its postings (algorithmic bytes = 8 or 12 per posting touched);
//   - the tile is then swept once: entries that beat the running k-th best (score desc, index asc;
//     score > 0 only, retrievers.py:195-196; optional dir filter, retrievers.py:198-202) are compacted
//     into an LDS candidate list that is re-sorted and cut to k whenever it fills.
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
template <typename ST>
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
                        if (pos < kBmCap) {
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
    int ablate /* measurement only: 1 no add, 2 no sweep, 4 no token barrier, 8 no posting loads */,
    unsigned long long *__restrict__ dbg /* measurement only: per-section shader-clock sums of thread 0, or null */) {
    using L = BmLds<ST>;
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
    const int t_begin = (int)((int64_t)n_tiles * seg / segs);
    const int t_end = (int)((int64_t)n_tiles * (seg + 1) / segs);
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

// Merge `segs` sorted partial lists of one query: grid = B, block = 1024, LDS = P*(8+4) (+64), P = pow2 >= segs*k.
__global__ __launch_bounds__(kBmThreads) void bm25_merge_kernel(
    int k, int segs, int P, const double *__restrict__ part_scores, const int32_t *__restrict__ part_ids,
    const int32_t *__restrict__ part_len, int32_t *__restrict__ out_ids, double *__restrict__ out_scores,
    int32_t *__restrict__ out_len) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int &s_n = *reinterpret_cast<int *>(smem);
    double *cs = reinterpret_cast<double *>(smem + 64);
    int32_t *ci = reinterpret_cast<int32_t *>(smem + 64 + (size_t)P * 8);
    const int q = blockIdx.x, tid = threadIdx.x;
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
                            double *part_scores, int32_t *part_ids, int32_t *part_len, int ablate,
                            unsigned long long *dbg, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    dim3 grid(segs, B), block(kBmThreads);
    if (variant == 0)
        hipLaunchKernelGGL(bm25_scan_kernel<double>, grid, block, BmLds<double>::BYTES, st, indptr, doc_ids,
                           (const double *)payload, tile_off, n_tiles, N, q_indptr, q_tok, q_order, k, segs, filter_dir, dir_id,
                           part_scores, part_ids, part_len, ablate, dbg);
    else
        hipLaunchKernelGGL(bm25_scan_kernel<float>, grid, block, BmLds<float>::BYTES, st, indptr, doc_ids,
                           (const float *)payload, tile_off, n_tiles, N, q_indptr, q_tok, q_order, k, segs, filter_dir, dir_id,
                           part_scores, part_ids, part_len, ablate, dbg);
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
                             hipStream_t st) {
    if (B <= 0) return hipSuccess;
    const int P = pow2_ge(segs * k < 2 ? 2 : segs * k);
    hipLaunchKernelGGL(bm25_merge_kernel, dim3(B), dim3(kBmThreads), (size_t)P * 12 + 64, st,
                       k, segs, P, part_scores, part_ids, part_len, out_ids, out_scores, out_len);
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
