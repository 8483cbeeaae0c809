// Fusion of the two rank lists, kernel K4.  Replaces HybridRetriever.reciprocal_rank_fusion and
// HybridRetriever.fusion (/root/reference/src/easyrag/custom/retrievers.py:256-274 and 239-253).
//
// Items are the concatenation [list a | list b] (the reference passes [sparse, dense], retrievers.py:290).
// The key is the node TEXT (get_content(), retrievers.py:245,263): content_id[doc] here.
//   RRF     score(c) = sum over occurrences, in item order, of 1/(rank + K) in float64, rank 1-based inside
//           its list; Python dict order = first-seen order, and the stable sort keeps it among equal
//           scores; the node returned is the LAST one seen for the content (text_to_node[content] = item).
//   fusion  first occurrence wins, ordered by its raw route score (stable).
// Lists are tiny (<= 192 + 288 in the reference config): one workgroup per query, two small LDS sorts
// (common.h: wave-local bitonic steps), no host round trip between the routes and the fusion.
//   1. sort items by (content, position): equal contents become adjacent, in item order;
//   2. the first item of each run is the content's leader (first seen); its score is the sum of the run's
//      weights in item order (RRF) or its own route score (fusion); the node is the run's last (RRF) / first item;
//   3. sort the leaders by (score desc, first-seen position asc) = Python's stable sort over dict order; emit topk.
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int kFuseThreads = 512;

// LDS layout for P = pow2 >= depth_a + depth_b slots:
//   key u64[P] | w f64[P] | score f64[P] | doc i32[P] | lidx i32[P] | node i32[P]     (36 bytes per slot + 16 header)
__host__ __device__ inline size_t fuse_lds_bytes(int P) { return 16 + (size_t)P * (8 + 8 + 8 + 4 + 4 + 4); }

template <bool RRF>
__global__ __launch_bounds__(kFuseThreads) void fuse_kernel(
    const int32_t *__restrict__ ids_a, const double *__restrict__ sc_a, const int32_t *__restrict__ len_a, int depth_a,
    const int32_t *__restrict__ ids_b, const double *__restrict__ sc_b, const int32_t *__restrict__ len_b, int depth_b,
    const int32_t *__restrict__ content_id, int K, int topk, int P,
    int32_t *__restrict__ out_ids, double *__restrict__ out_scores, int32_t *__restrict__ out_len) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int &n_leaders = *reinterpret_cast<int *>(smem);
    uint64_t *key = reinterpret_cast<uint64_t *>(smem + 16);
    double *w = reinterpret_cast<double *>(key + P);       // by item: RRF 1/(rank + K); fusion: raw route score
    double *score = w + P;                                 // by sorted slot
    int32_t *doc = reinterpret_cast<int32_t *>(score + P); // by item
    int32_t *lidx = doc + P;                               // by sorted slot: leader's item position (INT_MAX: none)
    int32_t *node = lidx + P;                              // by item (leaders only): the node to return
    const int q = blockIdx.x, tid = threadIdx.x;
    int la = len_a ? len_a[q] : depth_a;
    int lb = len_b ? len_b[q] : depth_b;
    if (la > depth_a) la = depth_a;
    if (lb > depth_b) lb = depth_b;
    if (la < 0) la = 0;
    if (lb < 0) lb = 0;
    const int n = la + lb;
    if (tid == 0) n_leaders = 0;
    for (int i = tid; i < P; i += kFuseThreads) {
        uint64_t kv = 0;                                   // empty slots sort to the end
        if (i < n) {
            const bool in_a = i < la;
            const int32_t id = in_a ? ids_a[(int64_t)q * depth_a + i] : ids_b[(int64_t)q * depth_b + (i - la)];
            doc[i] = id;
            const uint32_t c = (uint32_t)(content_id ? content_id[id] : id);
            kv = ~(((uint64_t)c << 32) | (uint32_t)i);     // descending sort of ~v = ascending (content, position)
            if (RRF) {
                const int rank = in_a ? (i + 1) : (i - la + 1);
                w[i] = 1.0 / (double)(rank + K);
            } else {
                w[i] = in_a ? sc_a[(int64_t)q * depth_a + i] : sc_b[(int64_t)q * depth_b + (i - la)];
            }
        }
        key[i] = kv;
    }
    erh_bitonic_desc<uint64_t>(key, P);                    // begins and ends with a barrier
    for (int p = tid; p < P; p += kFuseThreads) {
        double s = -INFINITY;
        int32_t li = 0x7fffffff;
        if (p < n) {
            const uint64_t v = ~key[p];
            const uint32_t c = (uint32_t)(v >> 32);
            const bool start = (p == 0) || ((uint32_t)((~key[p - 1]) >> 32) != c);
            if (start) {
                const int i0 = (int)(uint32_t)v;
                int32_t last = doc[i0];
                if (RRF) {
                    s = 0.0;                               // occurrences are added in item order
                    for (int pp = p; pp < n; ++pp) {
                        const uint64_t vv = ~key[pp];
                        if ((uint32_t)(vv >> 32) != c) break;
                        const int it = (int)(uint32_t)vv;
                        s = s + w[it];
                        last = doc[it];                    // text_to_node[content] = the LAST node seen
                    }
                } else {
                    s = w[i0];                             // first occurrence wins
                }
                li = i0;
                node[i0] = last;
                atomicAdd(&n_leaders, 1);
            }
        }
        score[p] = s;
        lidx[p] = li;
    }
    erh_bitonic_rec_desc<double>(score, lidx, P);          // (score desc, first-seen position asc)
    const int nl = n_leaders;
    const int kk = topk < nl ? topk : nl;
    for (int r = tid; r < topk; r += kFuseThreads) {
        if (r < kk) {
            out_ids[(int64_t)q * topk + r] = node[lidx[r]];
            out_scores[(int64_t)q * topk + r] = score[r];
        } else {
            out_ids[(int64_t)q * topk + r] = -1;
            out_scores[(int64_t)q * topk + r] = 0.0;
        }
    }
    if (tid == 0) out_len[q] = kk;
}


// ---- multi-GPU exchange: one packed row per query -----------------------------------------------------------------
// The query batch is sharded contiguously over the ranks (corpus replicated); the only
// exchange is the all-gather of the fused top-k.  Three arrays (scores f64[k], ids i32[k], len i32) travel as ONE
// buffer: row = [k x f64 | k x i32 | i32 len | pad to 8 bytes], every rank contributing `m` = ceil(n / world) rows
// (padding rows: len 0).  pack / unpack are the only per-step work besides the collective, and allocate nothing.
__global__ void pack_topk_kernel(const int32_t *__restrict__ ids, const double *__restrict__ sc,
                                 const int32_t *__restrict__ len, int b_local, int k, int m, int row_bytes,
                                 char *__restrict__ out) {
    const int r = blockIdx.x;
    char *row = out + (size_t)r * row_bytes;
    double *o_sc = reinterpret_cast<double *>(row);
    int32_t *o_id = reinterpret_cast<int32_t *>(row + (size_t)k * 8);
    const bool live = r < b_local;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        o_sc[i] = live ? sc[(size_t)r * k + i] : 0.0;
        o_id[i] = live ? ids[(size_t)r * k + i] : -1;
    }
    if (threadIdx.x == 0) o_id[k] = live ? len[r] : 0;
}

// Global query g lives in rank r's block at row g - lo(r); shards differ by at most one query (dist.py: shard_bounds).
__global__ void unpack_topk_kernel(const char *__restrict__ gathered, int n_queries, int world, int k, int m,
                                   int row_bytes, int32_t *__restrict__ ids, double *__restrict__ sc,
                                   int32_t *__restrict__ len) {
    const int g = blockIdx.x;
    const int base = n_queries / world, rem = n_queries % world;
    int r, lo;
    if (g < (base + 1) * rem) { r = g / (base + 1); lo = r * (base + 1); }
    else { r = rem + (base ? (g - (base + 1) * rem) / base : 0); lo = r * base + rem; }
    const char *row = gathered + ((size_t)r * m + (g - lo)) * row_bytes;
    const double *i_sc = reinterpret_cast<const double *>(row);
    const int32_t *i_id = reinterpret_cast<const int32_t *>(row + (size_t)k * 8);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        sc[(size_t)g * k + i] = i_sc[i];
        ids[(size_t)g * k + i] = i_id[i];
    }
    if (threadIdx.x == 0) len[g] = i_id[k];
}

}  // namespace

namespace erh {

hipError_t fuse_init() {
    hipError_t e = hipFuncSetAttribute((const void *)fuse_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)fuse_lds_bytes(kFuseMaxItems));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)fuse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)fuse_lds_bytes(kFuseMaxItems));
}

hipError_t launch_rrf(const int32_t *ids_a, const int32_t *len_a, int depth_a,
                      const int32_t *ids_b, const int32_t *len_b, int depth_b,
                      const int32_t *content_id, int B, int K, int topk,
                      int32_t *out_ids, double *out_scores, int32_t *out_len, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    int P = 2;
    while (P < depth_a + depth_b) P <<= 1;
    hipLaunchKernelGGL(fuse_kernel<true>, dim3(B), dim3(kFuseThreads), fuse_lds_bytes(P), st,
                       ids_a, (const double *)nullptr, len_a, depth_a, ids_b, (const double *)nullptr, len_b, depth_b,
                       content_id, K, topk, P, out_ids, out_scores, out_len);
    return hipGetLastError();
}

hipError_t launch_fusion(const int32_t *ids_a, const double *sc_a, const int32_t *len_a, int depth_a,
                         const int32_t *ids_b, const double *sc_b, const int32_t *len_b, int depth_b,
                         const int32_t *content_id, int B, int topk,
                         int32_t *out_ids, double *out_scores, int32_t *out_len, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    int P = 2;
    while (P < depth_a + depth_b) P <<= 1;
    hipLaunchKernelGGL(fuse_kernel<false>, dim3(B), dim3(kFuseThreads), fuse_lds_bytes(P), st,
                       ids_a, sc_a, len_a, depth_a, ids_b, sc_b, len_b, depth_b,
                       content_id, 0, topk, P, out_ids, out_scores, out_len);
    return hipGetLastError();
}

int topk_row_bytes(int k) { return (k * 12 + 4 + 7) / 8 * 8; }

hipError_t launch_pack_topk(const int32_t *ids, const double *sc, const int32_t *len, int b_local, int k, int m,
                            void *out, hipStream_t st) {
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_topk_kernel, dim3(m), dim3(64), 0, st, ids, sc, len, b_local, k, m, topk_row_bytes(k),
                       (char *)out);
    return hipGetLastError();
}

hipError_t launch_unpack_topk(const void *gathered, int n_queries, int world, int k, int m, int32_t *ids, double *sc,
                              int32_t *len, hipStream_t st) {
    if (n_queries <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_topk_kernel, dim3(n_queries), dim3(64), 0, st, (const char *)gathered, n_queries, world, k,
                       m, topk_row_bytes(k), ids, sc, len);
    return hipGetLastError();
}

}  // namespace erh
