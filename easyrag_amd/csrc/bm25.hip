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
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBmThreads = 1024;
constexpr int kBmCap = 2048;       // LDS candidate slots (k <= 1024 so that k + one sweep chunk always fits)
constexpr int kBmTokChunk = 256;   // query tokens whose tile ranges are staged at once

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
template <typename ST>
struct BmLds {
    static constexpr int TILE = (sizeof(ST) == 4) ? erh::kBm25TileF32 : erh::kBm25TileF64;
    // layout (bytes): [0,64) header | acc TILE*ST | cand_s CAP*ST | cand_i CAP*4 | lo CHUNK*8 | hi CHUNK*8
    static constexpr size_t OFF_ACC = 64;
    static constexpr size_t OFF_CS = OFF_ACC + (size_t)TILE * sizeof(ST);
    static constexpr size_t OFF_CI = OFF_CS + (size_t)kBmCap * sizeof(ST);
    static constexpr size_t OFF_LO = OFF_CI + (size_t)kBmCap * 4;
    static constexpr size_t OFF_HI = OFF_LO + (size_t)kBmTokChunk * 8;
    static constexpr size_t BYTES = OFF_HI + (size_t)kBmTokChunk * 8;
};

struct BmHdr {
    int ncand;      // live entries in the candidate list
    int total;      // scratch for block-wide counts
    int tau_idx;    // running k-th best: index part (INT_MAX when fewer than k so far)
    int pad;
    double tau_s;   // running k-th best: score part (0 => "score > 0" is the only condition)
};

// Sort the candidate list (score desc, idx asc), cut to k, refresh the running threshold.  Uniform call.
template <typename ST>
__device__ __forceinline__ void bm_shrink(BmHdr *hdr, ST *cs, int32_t *ci, int k) {
    __syncthreads();
    const int n = hdr->ncand;
    for (int i = n + (int)threadIdx.x; i < kBmCap; i += kBmThreads) { cs[i] = (ST)-1; ci[i] = 0x7fffffff; }
    erh_bitonic_rec_desc<ST>(cs, ci, kBmCap);
    if (threadIdx.x == 0 && n >= k) {
        hdr->ncand = k;
        hdr->tau_s = (double)cs[k - 1];
        hdr->tau_idx = ci[k - 1];
    }
    __syncthreads();
}

// grid = (segs, B), block = 1024.  Segment `seg` of query q walks tiles [n_tiles*seg/segs, n_tiles*(seg+1)/segs).
template <typename ST>
__global__ __launch_bounds__(kBmThreads) void bm25_scan_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ doc_ids, const ST *__restrict__ payload,
    const int32_t *__restrict__ tile_off, int n_tiles, int64_t N,
    const int32_t *__restrict__ q_indptr, const int32_t *__restrict__ q_tok, int k, int segs,
    const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    double *__restrict__ part_scores, int32_t *__restrict__ part_ids, int32_t *__restrict__ part_len) {
    using L = BmLds<ST>;
    constexpr int TILE = L::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BmHdr *hdr = reinterpret_cast<BmHdr *>(smem);
    ST *acc = reinterpret_cast<ST *>(smem + L::OFF_ACC);
    ST *cs = reinterpret_cast<ST *>(smem + L::OFF_CS);
    int32_t *ci = reinterpret_cast<int32_t *>(smem + L::OFF_CI);
    int64_t *s_lo = reinterpret_cast<int64_t *>(smem + L::OFF_LO);
    int64_t *s_hi = reinterpret_cast<int64_t *>(smem + L::OFF_HI);

    const int seg = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
    const int qs = q_indptr[q], nq = q_indptr[q + 1] - qs;
    const int fd = filter_dir ? (int)filter_dir[q] : -1;
    const int t_begin = (int)((int64_t)n_tiles * seg / segs);
    const int t_end = (int)((int64_t)n_tiles * (seg + 1) / segs);
    const int64_t out_base = ((int64_t)q * segs + seg) * k;

    if (tid == 0) { hdr->ncand = 0; hdr->total = 0; hdr->tau_idx = -1; hdr->tau_s = 0.0; }
    for (int i = tid; i < TILE; i += kBmThreads) acc[i] = (ST)0;
    __syncthreads();

    if (nq > 0) {
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int64_t base_doc = (int64_t)tile * TILE;
            // ---- scatter-add, one query token after the other --------------------------------
            for (int c0 = 0; c0 < nq; c0 += kBmTokChunk) {
                const int nqc = (nq - c0 < kBmTokChunk) ? (nq - c0) : kBmTokChunk;
                for (int j = tid; j < nqc; j += kBmThreads) {
                    const int64_t tok = q_tok[qs + c0 + j];
                    const int64_t ip = indptr[tok];
                    const int32_t *to = tile_off + tok * (n_tiles + 1) + tile;
                    s_lo[j] = ip + to[0];
                    s_hi[j] = ip + to[1];
                }
                __syncthreads();
                for (int j = 0; j < nqc; ++j) {
                    const int64_t lo = s_lo[j], hi = s_hi[j];      // block-uniform
                    if (lo < hi) {
                        for (int64_t p = lo + tid; p < hi; p += kBmThreads) {
                            const int slot = (int)((int64_t)doc_ids[p] - base_doc);
                            acc[slot] = acc[slot] + payload[p];
                        }
                        __syncthreads();                          // token j complete before token j+1
                    }
                }
                __syncthreads();                                  // s_lo/s_hi free for the next chunk
            }
            // ---- sweep: keep what beats the running k-th best, clear the rest -------------------
            double tau_s = hdr->tau_s;
            int tau_idx = hdr->tau_idx;
            int mine = 0;
            for (int i = tid; i < TILE; i += kBmThreads) {
                const ST s = acc[i];
                if (s != (ST)0) {
                    const int64_t doc = base_doc + i;
                    bool pass = ((double)s > tau_s) || ((double)s == tau_s && doc < (int64_t)tau_idx);
                    if (pass && (doc >= N || (fd >= 0 && (int)dir_id[doc] != fd))) pass = false;
                    if (pass) ++mine; else acc[i] = (ST)0;
                }
            }
            for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o);
            if ((tid & 63) == 0 && mine) atomicAdd(&hdr->total, mine);
            __syncthreads();
            const int total = hdr->total;
            const int have = hdr->ncand;
            __syncthreads();
            if (tid == 0) hdr->total = 0;
            if (total == 0) continue;                             // uniform
            if (have + total <= kBmCap) {
                for (int i = tid; i < TILE; i += kBmThreads) {
                    const ST s = acc[i];
                    if (s != (ST)0) {
                        const int pos = atomicAdd(&hdr->ncand, 1);
                        cs[pos] = s;
                        ci[pos] = (int32_t)(base_doc + i);
                        acc[i] = (ST)0;
                    }
                }
                __syncthreads();
            } else {
                // warm-up / adversarial path: go chunk by chunk, shrinking whenever the list may overflow
                for (int cb = 0; cb < TILE; cb += kBmThreads) {
                    __syncthreads();
                    const int have_now = hdr->ncand;              // read between two barriers: uniform
                    __syncthreads();
                    if (have_now + kBmThreads > kBmCap) {
                        bm_shrink<ST>(hdr, cs, ci, k);
                        tau_s = hdr->tau_s;
                        tau_idx = hdr->tau_idx;
                    }
                    const int i = cb + tid;
                    const ST s = acc[i];
                    if (s != (ST)0) {
                        const int64_t doc = base_doc + i;
                        const bool pass = ((double)s > tau_s) || ((double)s == tau_s && doc < (int64_t)tau_idx);
                        if (pass) {
                            const int pos = atomicAdd(&hdr->ncand, 1);
                            cs[pos] = s;
                            ci[pos] = (int32_t)doc;
                        }
                        acc[i] = (ST)0;
                    }
                }
                __syncthreads();
            }
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
                            const int32_t *q_indptr, const int32_t *q_tok, int B, int k, int segs,
                            const int16_t *filter_dir, const int16_t *dir_id,
                            double *part_scores, int32_t *part_ids, int32_t *part_len, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    dim3 grid(segs, B), block(kBmThreads);
    if (variant == 0)
        hipLaunchKernelGGL(bm25_scan_kernel<double>, grid, block, BmLds<double>::BYTES, st, indptr, doc_ids,
                           (const double *)payload, tile_off, n_tiles, N, q_indptr, q_tok, k, segs, filter_dir, dir_id,
                           part_scores, part_ids, part_len);
    else
        hipLaunchKernelGGL(bm25_scan_kernel<float>, grid, block, BmLds<float>::BYTES, st, indptr, doc_ids,
                           (const float *)payload, tile_off, n_tiles, N, q_indptr, q_tok, k, segs, filter_dir, dir_id,
                           part_scores, part_ids, part_len);
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
