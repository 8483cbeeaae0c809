// Fusion of the two rank lists, kernel K4.  Replaces HybridRetriever.reciprocal_rank_fusion and
// HybridRetriever.fusion (/root/reference/src/easyrag/custom/retrievers.py:256-274 and 239-253).
//
// Items are the concatenation [list a | list b] (the reference passes [sparse, dense], retrievers.py:290).
// The key is the node TEXT (get_content(), retrievers.py:245,263): content_id[doc] here.
//   RRF     score(c) = sum over occurrences, in item order, of 1/(rank + K) in float64, rank 1-based inside
//           its list; Python dict order = first-seen order, and the stable sort keeps it among equal
//           scores; the node returned is the LAST one seen for the content (text_to_node[content] = item).
//   fusion  first occurrence wins, ordered by its raw route score (stable).
// Lists are tiny (<= 192 + 288 in the reference config), so one workgroup per query does the
// quadratic leader / rank counting out of LDS; no host round trip between the routes and the fusion.
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int kFuseThreads = 512;

// LDS layout, sized at launch for cap = round_up(depth_a + depth_b, 64) items (480 in the reference config:
// 16 KiB, four workgroups per CU instead of the two a kFuseMaxItems-sized block allows):
//   [0,16) header | w f64[cap] | score f64[cap] | content i32[cap] | doc i32[cap] | out_doc i32[cap] | leader i32[cap]
struct FuseLds {
    int *n_leaders_p;
    double *w;                           // RRF: 1/(rank + K) of the item; fusion: its raw route score
    double *score;
    int32_t *content, *doc, *out_doc, *leader;
    int &n_leaders;
    __device__ FuseLds(char *smem, int cap)
        : n_leaders_p(reinterpret_cast<int *>(smem)), w(reinterpret_cast<double *>(smem + 16)), score(w + cap),
          content(reinterpret_cast<int32_t *>(score + cap)), doc(content + cap), out_doc(doc + cap),
          leader(out_doc + cap), n_leaders(*n_leaders_p) {}
};
__host__ __device__ inline size_t fuse_lds_bytes(int cap) { return 16 + (size_t)cap * (8 + 8 + 4 * 4); }

// One item per thread; the inner scans are branch-free counting loops over LDS (broadcast reads, unrolled so
// that several reads are in flight) instead of early-exit walks.
template <bool RRF>
__global__ __launch_bounds__(kFuseThreads) void fuse_kernel(
    const int32_t *__restrict__ ids_a, const double *__restrict__ sc_a, const int32_t *__restrict__ len_a, int depth_a,
    const int32_t *__restrict__ ids_b, const double *__restrict__ sc_b, const int32_t *__restrict__ len_b, int depth_b,
    const int32_t *__restrict__ content_id, int K, int topk,
    int32_t *__restrict__ out_ids, double *__restrict__ out_scores, int32_t *__restrict__ out_len) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FuseLds L(smem, (depth_a + depth_b + 63) / 64 * 64);
    const int q = blockIdx.x, tid = threadIdx.x;
    int la = len_a ? len_a[q] : depth_a;
    int lb = len_b ? len_b[q] : depth_b;
    if (la > depth_a) la = depth_a;
    if (lb > depth_b) lb = depth_b;
    if (la < 0) la = 0;
    if (lb < 0) lb = 0;
    const int n = la + lb;
    if (tid == 0) L.n_leaders = 0;
    for (int i = tid; i < n; i += kFuseThreads) {
        const bool in_a = i < la;
        const int32_t id = in_a ? ids_a[(int64_t)q * depth_a + i] : ids_b[(int64_t)q * depth_b + (i - la)];
        L.doc[i] = id;
        L.content[i] = content_id ? content_id[id] : id;
        if (RRF) {
            const int rank = in_a ? (i + 1) : (i - la + 1);
            L.w[i] = 1.0 / (double)(rank + K);
        } else {
            L.w[i] = in_a ? sc_a[(int64_t)q * depth_a + i] : sc_b[(int64_t)q * depth_b + (i - la)];
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += kFuseThreads) {
        const int32_t c = L.content[i];
        int earlier = 0;
#pragma unroll 8
        for (int j = 0; j < i; ++j) earlier += (L.content[j] == c) ? 1 : 0;
        const bool lead = earlier == 0;
        L.leader[i] = lead ? 1 : 0;
        if (lead) {
            atomicAdd(&L.n_leaders, 1);
            if (RRF) {
                // occurrences are added in item order; a non-occurrence adds +0.0, which never changes the bits
                double s = 0.0;
                int32_t last = L.doc[i];
#pragma unroll 8
                for (int j = i; j < n; ++j) {
                    const bool m = L.content[j] == c;
                    s = s + (m ? L.w[j] : 0.0);
                    last = m ? L.doc[j] : last;
                }
                L.score[i] = s;
                L.out_doc[i] = last;
            } else {
                L.score[i] = L.w[i];
                L.out_doc[i] = L.doc[i];
            }
        }
    }
    __syncthreads();
    const int nl = L.n_leaders;
    const int kk = topk < nl ? topk : nl;
    for (int i = tid; i < n; i += kFuseThreads) {
        if (!L.leader[i]) continue;
        const double s = L.score[i];
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
            const double sj = L.score[j];
            const bool better = (sj > s) || (sj == s && j < i);
            rank += (L.leader[j] && better) ? 1 : 0;
        }
        if (rank < kk) {
            out_ids[(int64_t)q * topk + rank] = L.out_doc[i];
            out_scores[(int64_t)q * topk + rank] = s;
        }
    }
    for (int i = kk + tid; i < topk; i += kFuseThreads) {
        out_ids[(int64_t)q * topk + i] = -1;
        out_scores[(int64_t)q * topk + i] = 0.0;
    }
    if (tid == 0) out_len[q] = kk;
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
    hipLaunchKernelGGL(fuse_kernel<true>, dim3(B), dim3(kFuseThreads), fuse_lds_bytes((depth_a + depth_b + 63) / 64 * 64), st,
                       ids_a, (const double *)nullptr, len_a, depth_a, ids_b, (const double *)nullptr, len_b, depth_b,
                       content_id, K, topk, out_ids, out_scores, out_len);
    return hipGetLastError();
}

hipError_t launch_fusion(const int32_t *ids_a, const double *sc_a, const int32_t *len_a, int depth_a,
                         const int32_t *ids_b, const double *sc_b, const int32_t *len_b, int depth_b,
                         const int32_t *content_id, int B, int topk,
                         int32_t *out_ids, double *out_scores, int32_t *out_len, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(fuse_kernel<false>, dim3(B), dim3(kFuseThreads), fuse_lds_bytes((depth_a + depth_b + 63) / 64 * 64), st,
                       ids_a, sc_a, len_a, depth_a, ids_b, sc_b, len_b, depth_b,
                       content_id, 0, topk, out_ids, out_scores, out_len);
    return hipGetLastError();
}

}  // namespace erh
