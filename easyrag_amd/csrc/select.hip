// Dense route, selection side: query preparation, threshold seeding, candidate refinement and the
// final ranking with the order-pinned fp64 re-score.  Together with dense_scan.hip this replaces the
// "argsort(scores)[::-1], walk, take `limit`" of Qdrant's exact COSINE search behind
// QdrantRetriever (/root/reference/src/easyrag/custom/retrievers.py:37-52).
//
// Pruning is exact, not approximate: a chunk may be dropped only if its fp32 MFMA score is below
// (a valid lower bound of the k-th best fp32 score) - margin, margin = 2*delta(q),
// delta(q) = d * 2^-23 * ||q|| * max_i ||x_i|| >= |fp32 score - exact score| for any summation order.
// Everything that survives is ranked by (fp64 score desc, index asc) in EXACT mode.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int kSelThreads = 1024;

__device__ __forceinline__ float block_sum_256(float v, float *red) {
    // blockDim.x == 256
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float margin_of(float qn, float xn, int d) {
    // 2 * delta, delta = d * 2^-23 * |q| * max|x|  (2^-23: twice the unit round-off of fp32, to stay
    // a bound whatever rounding the matrix core applies to its internal partial sums)
    return 2.0f * ((float)d * 1.1920929e-7f * qn * xn) + 1e-30f;
}

// ---- query block preparation -----------------------------------------------------------------
template <typename TIN>
__global__ __launch_bounds__(256) void prep_queries_kernel(const TIN *__restrict__ q, int normalize, int B, int d,
                                                          _Float16 *__restrict__ Q16, float *__restrict__ qnorm,
                                                          uint32_t *__restrict__ zero_bad /* null, or B words to clear */,
                                                          uint32_t *__restrict__ zero_flags /* null, or the call's 16 flag words */,
                                                          const int32_t *__restrict__ q_src /* null, or row -> caller's row (-1: padding) */,
                                                          uint32_t *__restrict__ zero_extra, int n_extra) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    if (zero_extra && row == 0 && (int)threadIdx.x >= 16 && (int)threadIdx.x < 16 + n_extra) zero_extra[threadIdx.x - 16] = 0u;
    const int src = q_src ? q_src[row] : (row < B ? row : -1);
    // the call's per-query "uncertified" marks and its flag words start at zero: cleared here instead of by two memset launches
    if (zero_bad && row < B && threadIdx.x == 0) zero_bad[row] = 0u;
    if (zero_flags && row == 0 && threadIdx.x < 16) zero_flags[threadIdx.x] = 0u;
    _Float16 *out = Q16 + (int64_t)row * d;
    if (src < 0) {
        for (int i = threadIdx.x; i < d; i += 256) out[i] = (_Float16)0.f;
        if (threadIdx.x == 0) qnorm[row] = 0.f;
        return;
    }
    const TIN *in = q + (int64_t)src * d;
    float inv = 1.f;
    if (normalize) {
        float ss = 0.f;
        for (int i = threadIdx.x; i < d; i += 256) { const float v = (float)in[i]; ss += v * v; }
        ss = block_sum_256(ss, red);
        inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
    }
    float s2 = 0.f;
    for (int i = threadIdx.x; i < d; i += 256) {
        const _Float16 hv = (_Float16)((float)in[i] * inv);
        out[i] = hv;
        const float f = (float)hv;
        s2 += f * f;
    }
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) qnorm[row] = sqrtf(s2) * 1.0001f;
}

// fp32 rows [r0, r0 + n) of the caller's matrix -> fp16 rows at their stored positions ((r0 + row) * mul) mod N.
__global__ __launch_bounds__(256) void convert_rows_kernel(const float *__restrict__ x, int64_t n, int d, int normalize,
                                                          _Float16 *__restrict__ out_base, int64_t r0, int64_t mul,
                                                          int64_t N) {
    __shared__ float red[4];
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        const float *in = x + row * d;
        _Float16 *out = out_base + erh_mulmod(r0 + row, mul, N) * d;
        float inv = 1.f;
        if (normalize) {
            float ss = 0.f;
            for (int i = threadIdx.x; i < d; i += 256) { const float v = in[i]; ss += v * v; }
            ss = block_sum_256(ss, red);
            inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
        }
        for (int i = threadIdx.x; i < d; i += 256) out[i] = (_Float16)(in[i] * inv);
    }
}

// fp16 rows [r0, r0 + n) -> stored positions (same placement as convert_rows_kernel); 16 bytes per thread and step.
__global__ __launch_bounds__(256) void permute_rows_kernel(const _Float16 *__restrict__ x, int64_t n, int d,
                                                          _Float16 *__restrict__ out_base, int64_t r0, int64_t mul,
                                                          int64_t N) {
    const int vec = d / 8;
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        const half8 *in = reinterpret_cast<const half8 *>(x + row * d);
        half8 *out = reinterpret_cast<half8 *>(out_base + erh_mulmod(r0 + row, mul, N) * d);
        for (int i = threadIdx.x; i < vec; i += 256) out[i] = in[i];
    }
}

// Original rows [row0, row0 + n) -- or, with a list, the original rows rows[row0 .. row0 + n) -- gathered back into a contiguous block
// (debug scores path; the per-dir block copies).
__global__ __launch_bounds__(256) void gather_rows_kernel(const _Float16 *__restrict__ X, const int32_t *__restrict__ rows, int64_t row0,
                                                         int64_t n, int d, int64_t mul, int64_t N, _Float16 *__restrict__ out) {
    const int vec = d / 8;
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        const int64_t orig = rows ? (int64_t)rows[row0 + row] : row0 + row;
        const half8 *in = reinterpret_cast<const half8 *>(X + erh_mulmod(orig, mul, N) * d);
        half8 *o = reinterpret_cast<half8 *>(out + row * d);
        for (int i = threadIdx.x; i < vec; i += 256) o[i] = in[i];
    }
}

// dir id by stored position: out[s] = dir_id[(s * inv) mod N]
__global__ __launch_bounds__(256) void permute_dir_kernel(const int16_t *__restrict__ dir_id, int64_t N, int64_t inv,
                                                         int16_t *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < N) out[s] = dir_id[erh_mulmod(s, inv, N)];
}

__global__ __launch_bounds__(256) void row_norm_max_kernel(const _Float16 *__restrict__ x, int64_t n, int d,
                                                          float *__restrict__ out) {
    __shared__ float red[4];
    float best = 0.f;
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        const _Float16 *in = x + row * d;
        float ss = 0.f;
        for (int i = threadIdx.x; i < d; i += 256) { const float v = (float)in[i]; ss += v * v; }
        ss = block_sum_256(ss, red);
        best = fmaxf(best, ss);
    }
    if (threadIdx.x == 0) atomicMax((unsigned int *)out, __float_as_uint(sqrtf(best) * 1.0001f));
}

// ---- seed stage: k-th best of the stored prefix scores ------------------------------------------
// grid = B, block = 1024.  Exact k-th largest without sorting the row: every thread keeps the maximum of its
// (strided) keys; the k-th largest of those 1024 maxima (one small bitonic sort) is a lower bound p of the true
// k-th value, and only ~k*(1+k/2048) keys are >= p.  Those are compacted and sorted.
// Fast kernel (`seed_select_kernel`): the row is read twice from memory -- thread maxima, then the gather above the pivot
// (it was written by the store kernel a moment ago and sits in L2 / Infinity Cache); 36 KiB of LDS, two workgroups per CU.
// Inputs that defeat the pivot (more than kSeedBuf keys >= p, or fewer than k threads holding a valid key) take the kernel's own
// fall-back: a radix select of the rank-th largest key over the row itself (round 6; a second, full-sort kernel before).
constexpr int kSeedBuf = 4096;

__device__ __forceinline__ uint32_t seed_key(const float *__restrict__ row, int i, int n0, int fd,
                                             const int16_t *__restrict__ dir_id, int64_t c0) {
    if (i >= n0) return 0u;
    const float s = row[i];
    bool ok = s > -INFINITY;                                          // chunks past N were stored as -inf
    if (ok && fd >= 0) ok = ((int)dir_id[c0 + i] == fd);
    return ok ? erh_f2ord(s) : 0u;
}

// Visit every prefix score of the row once: f(ordered key or 0, score, index).  Thread t takes the 16-byte groups
// t, t + 1024, ... (the row is 16-byte aligned: ld_s0 is a multiple of 256), the first threads the ragged tail.
// (Keeping the thread's 32 scores in registers across the passes instead of re-reading the row measured
// SLOWER -- 0.146 vs 0.106 ms per 1024 queries: the row comes from L2 / Infinity Cache, the unrolled register walk costs
// more than the reads.)
template <class F>
__device__ __forceinline__ void seed_visit(const float *__restrict__ row, int n0, int fd,
                                           const int16_t *__restrict__ dir_id, int64_t c0, F f) {
    const int tid = threadIdx.x;
    const int n4 = n0 >> 2;
    const float4 *row4 = reinterpret_cast<const float4 *>(row);
#pragma unroll 4
    for (int g = tid; g < n4; g += kSelThreads) {
        const float4 v = row4[g];
        const float sv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            bool ok = sv[e] > -INFINITY;                                // chunks past N were stored as -inf
            if (ok && fd >= 0) ok = ((int)dir_id[c0 + i] == fd);
            f(ok ? erh_f2ord(sv[e]) : 0u, sv[e], i);
        }
    }
    const int i = 4 * n4 + tid;
    if (i < n0) {
        const float sc = row[i];
        bool ok = sc > -INFINITY;
        if (ok && fd >= 0) ok = ((int)dir_id[c0 + i] == fd);
        f(ok ? erh_f2ord(sc) : 0u, sc, i);
    }
}

// every stored prefix score >= prune becomes a candidate; tau[q] = prune
__device__ __forceinline__ void seed_emit(const float *__restrict__ row, int n0, int64_t c0, int fd,
                                          const int16_t *__restrict__ dir_id, float prune, int q,
                                          float *__restrict__ tau, ErhCand *__restrict__ cand,
                                          uint32_t *__restrict__ cand_cnt, int cap, uint32_t *__restrict__ bad,
                                          int *s_cnt) {
    const int tid = threadIdx.x;
    if (tid == 0) tau[q] = prune;
    seed_visit(row, n0, fd, dir_id, c0, [&](uint32_t key, float s, int i) {
        if (key != 0u && s >= prune) {
            const int pos = atomicAdd(s_cnt, 1);
            if (pos < cap) {
                ErhCand c;
                c.s = s;
                c.idx = (int32_t)(c0 + i);
                cand[(int64_t)q * cap + pos] = c;
            }
        }
    });
    __syncthreads();
    if (tid == 0) {
        const int c = *s_cnt;
        cand_cnt[q] = (uint32_t)(c < cap ? c : cap);
        if (c > cap) bad[q] = 1u;                        // list too short for this query: the exhaustive path answers it
    }
}

// `rank` <= k is the position in the prefix whose score seeds the threshold.  rank == k gives a guaranteed bound (the
// k-th best of a subset never exceeds the k-th best of the whole).  rank < k is a SPECULATIVE bound: an estimate of
// where the k-th best of the whole corpus lies, extrapolated from the prefix being an even sample of it (pipeline_dense.hip picks
// the rank so that fewer than `rank` of the true top k land in the prefix except with probability < 1e-7 per query).
// dense_finalize_kernel verifies it -- at least k candidates must score >= threshold + margin -- and hands the
// query to the exhaustive path otherwise, so the result is exact either way; what the speculation buys is a
// threshold ~3x tighter than any guaranteed one, from the first scanned tile on.
__global__ __launch_bounds__(kSelThreads) void seed_select_kernel(
    const float *__restrict__ S0, int ld_s0, int n0, int np2, int64_t c0, int k, int rank,
    const float *__restrict__ qnorm, float xnorm_max, int d,
    const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id,
    float *__restrict__ tau, ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap,
    uint32_t *__restrict__ bad, uint32_t *__restrict__ need_full,
    const erh::ErhDenseView *__restrict__ views /* null, or the grouped call's table: prefix length and rank per query tile */) {
    __shared__ int s_nvalid, s_cnt, s_cnt2, s_keep, s_bin, s_need;
    __shared__ uint32_t tmax[kSelThreads];
    __shared__ uint64_t buf[kSeedBuf];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (views) {
        const erh::ErhDenseView &v = views[q >> 8];
        if ((q & 255) >= v.nq) {                       // a padding row of the tile: nothing can pass its threshold
            if (tid == 0) { tau[q] = INFINITY; cand_cnt[q] = 0u; need_full[q] = 0u; }
            return;
        }
        n0 = v.n0; rank = v.rank;
    }
    const float *row = S0 + (int64_t)q * ld_s0;
    const int fd = filter_dir ? (int)filter_dir[q] : -1;
    if (tid == 0) { s_nvalid = 0; s_cnt = 0; s_cnt2 = 0; s_keep = 0; need_full[q] = 0u; }
    __syncthreads();
    int myvalid = 0;
    uint32_t mx = 0;
    seed_visit(row, n0, fd, dir_id, c0, [&](uint32_t key, float, int) {
        myvalid += key ? 1 : 0;
        mx = key > mx ? key : mx;
    });
    tmax[tid] = mx;
    for (int o = 32; o >= 1; o >>= 1) myvalid += __shfl_xor(myvalid, o);
    if ((tid & 63) == 0 && myvalid) atomicAdd(&s_nvalid, myvalid);
    erh_bitonic_desc<uint32_t>(tmax, kSelThreads);   // begins and ends with a barrier
    const int nv = s_nvalid;
    if (nv >= k) {                                   // uniform
        const uint32_t p = (rank <= kSelThreads) ? tmax[rank - 1] : 0u;
        bool full_sort = (p == 0u);
        if (!full_sort) {
            // p bounds the rank-th score from below, so the final threshold (rank-th score - margin) is >= p - margin:
            // one pass gathers everything that can become a candidate, with its index, and the candidates are then
            // emitted from LDS -- the row (128 KiB per query out of L2 / Infinity Cache) is read twice, not three times
            const float margin = margin_of(qnorm[q], xnorm_max, d);
            const float g_thr = erh_ord2f(p) - margin;
            seed_visit(row, n0, fd, dir_id, c0, [&](uint32_t key, float sc, int i) {
                if (key != 0u && sc >= g_thr) {
                    const int pos = atomicAdd(&s_cnt2, 1);
                    if (pos < kSeedBuf) buf[pos] = erh_key32(sc, (int32_t)(c0 + i));
                }
            });
            __syncthreads();
            const int c2 = s_cnt2;
            if (c2 > kSeedBuf) {
                full_sort = true;
            } else {
                const int ns = erh_next_pow2(c2 < 2 ? 2 : c2);
                for (int i = c2 + tid; i < ns; i += kSelThreads) buf[i] = 0ull;
                erh_bitonic_desc<uint64_t>(buf, ns);
                const float prune = erh_key32_score(buf[rank - 1]) - margin;
                for (int i = tid; i < c2; i += kSelThreads) {             // survivors are a prefix of the sorted keys
                    const bool keep_i = erh_key32_score(buf[i]) >= prune;
                    const bool keep_n = (i + 1 < c2) ? (erh_key32_score(buf[i + 1]) >= prune) : false;
                    if (keep_i && !keep_n) s_keep = i + 1;
                }
                __syncthreads();
                const int m = s_keep;
                for (int i = tid; i < m && i < cap; i += kSelThreads) {
                    ErhCand c;
                    c.s = erh_key32_score(buf[i]);
                    c.idx = erh_key32_idx(buf[i]);
                    cand[(int64_t)q * cap + i] = c;
                }
                if (tid == 0) {
                    tau[q] = prune;
                    cand_cnt[q] = (uint32_t)(m < cap ? m : cap);
                    if (m > cap) bad[q] = 1u;
                }
                return;
            }
        }
        if (full_sort) {
            // uniform, rare (a filter that leaves few strides a valid score, a tie cluster above the pivot): the rank-th largest key of the
            // WHOLE prefix by a radix select over the row itself -- four passes of eight bits, the row read from L2 each time, a 256-bin
            // histogram in the (now dead) gather buffer -- then every score >= that - margin is emitted.  Same threshold, same candidates
            // as sorting the row would give.  (Until round 6 a second kernel with the whole row in 128 KiB of LDS did this; it was launched
            // behind every seed select -- 5 us + a launch boundary per dense call of < 512 queries -- and returned at once almost always.)
            uint32_t *const hist = reinterpret_cast<uint32_t *>(buf);
            uint32_t prefix = 0u, mask = 0u;
            int need = rank;
            for (int shift = 24; shift >= 0; shift -= 8) {
                __syncthreads();                                          // (everyone has read s_bin / s_need of the pass before)
                if (tid < 256) hist[tid] = 0u;
                __syncthreads();
                seed_visit(row, n0, fd, dir_id, c0, [&](uint32_t key, float, int) {
                    if (key != 0u && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                });
                __syncthreads();
                if (tid == 0) {                                           // from the top bin down: the bin that holds the need-th key
                    int above = 0, b = 255;
                    for (; b > 0; --b) {
                        if (above + (int)hist[b] >= need) break;
                        above += (int)hist[b];
                    }
                    s_bin = b;
                    s_need = need - above;
                }
                __syncthreads();
                prefix |= (uint32_t)s_bin << shift;
                mask |= 255u << shift;
                need = s_need;
            }
            const float prune = erh_ord2f(prefix) - margin_of(qnorm[q], xnorm_max, d);
            __syncthreads();                                              // (s_cnt is still 0: seed_emit counts from there)
            seed_emit(row, n0, c0, fd, dir_id, prune, q, tau, cand, cand_cnt, cap, bad, &s_cnt);
            return;
        }
    }
    seed_emit(row, n0, c0, fd, dir_id, -INFINITY, q, tau, cand, cand_cnt, cap, bad, &s_cnt);   // fewer than k valid rows: all of them
}

// Threshold from the sample pass of the ping-pong scan (round 4; kernels.h: ErhSeedIo): per query n_vals scores -- the two
// best of every cell (64 sampled chunk rows) -- of which the rank-th largest, minus the margin, is the speculative pruning
// threshold.  The rank-th largest of the cells' two best never exceeds the rank-th largest of the whole sample (a cell with
// more than two of the sample's best hides the others), so the threshold errs on the safe side; dense_finalize_kernel verifies
// it as it verifies seed_select_kernel's.  The candidate lists start empty: the main launch scans the sampled rows again.
// grid = B, block = 256, dynamic LDS = np2 * 4 bytes.
__global__ __launch_bounds__(256) void seed_cells_select_kernel(
    const float *__restrict__ seed_top, int n_vals, int np2, int rank, const float *__restrict__ qnorm, float xnorm_max, int d,
    float *__restrict__ tau, uint32_t *__restrict__ cand_cnt, const erh::ErhDenseView *__restrict__ views) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem);
    const int q = blockIdx.x, tid = threadIdx.x;
    const float *row = seed_top + (int64_t)q * n_vals;
    if (views) {                                                     // grouped call: this tile's cells and rank (n_vals: the row stride)
        const erh::ErhDenseView &v = views[q >> 8];
        if ((q & 255) >= v.nq) {                                     // a padding row
            if (tid == 0) { tau[q] = INFINITY; cand_cnt[q] = 0u; }
            return;
        }
        n_vals = v.n_cells * 2; rank = v.rank;
        np2 = erh_next_pow2(n_vals < 2 ? 2 : n_vals);
    }
    for (int i = tid; i < np2; i += 256) {
        uint32_t key = 0u;
        if (i < n_vals) {
            const float v = row[i];
            if (v > -INFINITY) key = erh_f2ord(v);
        }
        keys[i] = key;
    }
    erh_bitonic_desc<uint32_t>(keys, np2);                           // begins and ends with a barrier
    if (tid == 0) {
        const uint32_t v = (rank >= 1 && rank <= n_vals) ? keys[rank - 1] : 0u;
        tau[q] = v ? erh_ord2f(v) - margin_of(qnorm[q], xnorm_max, d) : -INFINITY;
        cand_cnt[q] = 0u;
    }
}

// ---- refine: tighten tau from the candidates gathered so far --------------------------------------
// grid = B, block = 1024, dynamic LDS = cp2 * 8 bytes.
// Launched twice: with LDS for kRefineLight entries (two workgroups per CU; the usual few thousand candidates)
// handling the queries whose list fits, and with LDS for the full capacity handling only the others.
constexpr int kRefineLight = 4096;

__global__ __launch_bounds__(kSelThreads) void cand_refine_kernel(
    int k, int cp2, const float *__restrict__ qnorm, float xnorm_max, int d,
    float *__restrict__ tau, ErhCand *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int cap, int lo_excl,
    int hi_incl /* this launch handles lists with lo_excl < count <= hi_incl */, uint32_t *__restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int &s_keep = *reinterpret_cast<int *>(smem);
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem + 64);
    const int q = blockIdx.x, tid = threadIdx.x;
    int c = (int)cand_cnt[q];
    if (c > cap) {                                         // appends were dropped in the stage before this boundary
        if (threadIdx.x == 0 && hi_incl >= cap) bad[q] = 1u;
        c = cap;
    }
    if (c <= lo_excl || c > hi_incl) return;               // the other launch's query (uniform)
    if (c < k) return;                                     // nothing to learn yet (uniform)
    ErhCand *mine = cand + (int64_t)q * cap;
    const int ns = erh_next_pow2(c < 2 ? 2 : c);           // sort only what is there (ns <= cp2)
    for (int i = tid; i < ns; i += kSelThreads) {
        uint64_t key = 0;
        if (i < c) { const ErhCand e = mine[i]; key = erh_key32(e.s, e.idx); }
        keys[i] = key;
    }
    if (tid == 0) s_keep = c;
    erh_bitonic_desc<uint64_t>(keys, ns);
    const float kth = erh_key32_score(keys[k - 1]);
    float t_new = kth - margin_of(qnorm[q], xnorm_max, d);
    const float t_old = tau[q];
    if (t_old > t_new) t_new = t_old;
    // survivors are a prefix of the sorted keys
    for (int i = tid; i < c; i += kSelThreads) {
        const bool keep_i = erh_key32_score(keys[i]) >= t_new;
        const bool keep_n = (i + 1 < c) ? (erh_key32_score(keys[i + 1]) >= t_new) : false;
        if (keep_i && !keep_n) s_keep = i + 1;
    }
    __syncthreads();
    const int m = s_keep;
    for (int i = tid; i < m; i += kSelThreads) {
        ErhCand e;
        e.s = erh_key32_score(keys[i]);
        e.idx = erh_key32_idx(keys[i]);
        mine[i] = e;
    }
    if (tid == 0) { cand_cnt[q] = (uint32_t)m; tau[q] = t_new; }
}

// ---- dense calls routed by dir block (pipeline_dense.hip: dense_topk_routed): the queries of one group gathered into a contiguous block, and
// the group's results scattered back to the caller's rows with the block's first document added to the ids --------------------
__global__ __launch_bounds__(256) void gather_query_rows_kernel(const int4 *__restrict__ q, const int32_t *__restrict__ idx, int n,
                                                               int row_vec /* 16-byte pieces per row */, int4 *__restrict__ out) {
    const int64_t total = (int64_t)n * row_vec;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int r = (int)(o / row_vec), c = (int)(o - (int64_t)r * row_vec);
        out[o] = q[(int64_t)idx[r] * row_vec + c];
    }
}

__global__ __launch_bounds__(256) void scatter_topk_rows_kernel(const int32_t *__restrict__ ids, const double *__restrict__ sc,
                                                               const int32_t *__restrict__ len, const int32_t *__restrict__ idx, int n,
                                                               int k, int32_t id_offset, const int32_t *__restrict__ id_map,
                                                               int32_t *__restrict__ out_ids, double *__restrict__ out_sc,
                                                               int32_t *__restrict__ out_len) {
    const int64_t total = (int64_t)n * k;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int r = (int)(o / k), c = (int)(o - (int64_t)r * k);
        const int64_t dst = (int64_t)idx[r] * k + c;
        const int32_t id = ids[o];
        out_ids[dst] = id < 0 ? id : (id_map ? id_map[id_offset + id] : id + id_offset);   // (a block's rows -> the caller's document ids)
        out_sc[dst] = sc[o];
        if (c == 0) out_len[idx[r]] = len[r];
    }
}

// ---- final: select, pinned fp64 re-score of the margin set, rank, emit ------------------------------------------
// grid = B, block = 1024, static LDS ~ 40 KiB (two workgroups per CU).
// The candidate list (a few thousand entries, a few hundred of which matter) is never fully sorted: every thread
// takes the maximum of its strided share, the k-th largest of those 1024 maxima is a lower bound p of the k-th best
// score, and only entries >= p - margin (about k of them) are gathered into LDS and sorted.
constexpr int kFinBuf = 2048;

// Product of two fp16 values as fp32, straight from the packed halves: v_fma_mix_f32 widens both factors inside the instruction
// (op_sel picks the half, op_sel_hi marks the source as fp16) and adds +0 -- one VALU instruction where v_cvt_f32_f16 + v_mul_f32 take
// two.  The product has 22 significant bits, so it is exact either way; the +0 can only turn a -0 product into +0, which adds
// the same nothing to the fp64 sum (the sum starts at +0 and +0 + -0 = +0: it is never -0).  HI: the dword's upper half.
template <bool HI>
__device__ __forceinline__ float fin_mul_f16(uint32_t x2, uint32_t q2) {
    float p;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(p) : "v"(x2), "v"(q2));
    else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(p) : "v"(x2), "v"(q2));
    return p;
}
typedef uint32_t fin_u4 __attribute__((ext_vector_type(4)));

// 512 threads (a 256-VGPR budget): two workgroups per CU, and room to fetch the next pair of rows while the current
// pair is accumulated -- the row gathers (2 KiB each, anywhere in HBM) are what this kernel waits for.
constexpr int kFinThreads = 512;
constexpr int kFinSlots = 1024;                      // pivot granularity: strided maxima of 1024 sub-sequences

// QR = rounds of 512 row elements held in registers (d <= 512 * QR runs entirely from them); with QR <= 2 the kernel
// fits 128 VGPRs, i.e. TWO workgroups per CU (at 144 registers only one fits, and the row gathers of one workgroup do
// not keep a CU's memory queue busy).  Round 6: with the products from v_fma_mix_f32 (fin_mul_f16: no fp32 copy of the query
// fragments), two rows per wave and iteration instead of three (as many again in flight) and the pivot's maxima in r_idx's
// storage the QR <= 2 kernels fit 64 VGPRs and 36 KiB of LDS = FOUR workgroups per CU (launch bound 8 waves per SIMD; 24 bytes of
// scratch outside the row loop) -- at 1024 queries every CU's four queries are resident at once instead of two and two (or three and
// one: the last round gathered with a quarter of the chip's waves).  Select class per 1024 queries, interleaved
// (profiles/r06zd_ab_fin_wgs4_*.log; option dense_fin_wgs = 2 / 3 keeps the others out with unused dynamic LDS): 0.200 / 0.194 / 0.177 ms
// at 2 / 3 / 4 workgroups (-11.5 %; 0.198 with the 112-VGPR kernel of the rounds before), the filtered 1024-query step 1.289 -> 1.263;
// 256 and 512 queries (one / two queries per CU) +-0.
template <int QR>
__global__ __launch_bounds__(kFinThreads, (QR <= 2) ? 8 : 2) void dense_finalize_kernel(
    int k, int mode, const float *__restrict__ qnorm, float xnorm_max, int d,
    const _Float16 *__restrict__ X, const _Float16 *__restrict__ Q16,
    const ErhCand *__restrict__ cand, const uint32_t *__restrict__ cand_cnt, int cap,
    int32_t *__restrict__ out_ids, double *__restrict__ out_scores, int32_t *__restrict__ out_len,
    float *__restrict__ diag_maxerr, uint32_t *__restrict__ diag_uncert, uint32_t *__restrict__ bad,
    int64_t N, int64_t pos_mul, int64_t pos_inv /* candidates carry stored positions: orig = pos * pos_inv mod N */,
    const float *__restrict__ tau /* nullptr: thresholds were guaranteed bounds, nothing to verify */,
    int G /* workgroups per query (round 5; small batches, EXACT mode): each re-scores every G-th row of the margin set, the last one
             to finish ranks and emits.  1: one workgroup per query does everything */,
    double *__restrict__ ws_s64 /* G > 1: [B][kDenseRescoreMax] exact scores handed to the last workgroup */,
    uint32_t *__restrict__ ws_sync /* G > 1: [B][2] {arrival counter, max error bits}, zero between calls */,
    const erh::ErhGroupIo gio /* views != null: the grouped call (G == 1) -- matrix and placement of the query's tile, results to the
                                 caller's row, block rows mapped to document ids */) {
    __shared__ __attribute__((aligned(16))) uint64_t buf[kFinBuf];
    __shared__ double r_s64[erh::kDenseRescoreMax];
    __shared__ int32_t r_idx[erh::kDenseRescoreMax];
    static_assert(erh::kDenseRescoreMax >= kFinSlots, "the pivot's maxima live in r_idx's storage");
    uint32_t *const tmax = reinterpret_cast<uint32_t *>(r_idx);         // (dead before r_idx is written: 36 KiB of LDS = four workgroups per CU)
    __shared__ float r_s32[erh::kDenseRescoreMax];
    __shared__ int32_t r_pos[erh::kDenseRescoreMax];                    // stored position of the candidate (row of X)
    __shared__ int s_cnt, s_m, s_lvl, s_last;
    __shared__ unsigned int s_maxerr;
    // Small batches are latency-bound here: ONE workgroup gathering a query's ~300 rows of 2 KiB takes 51 us at one query per call (a
    // dozen dependent gather rounds).  With G workgroups per query every one repeats the cheap part (pivot, gather of the candidates
    // above it, sort: the same result in each, the keys are distinct), re-scores its share of the rows, and the last to arrive ranks.
    const int q = G > 1 ? (int)blockIdx.x / G : (int)blockIdx.x, part = G > 1 ? (int)blockIdx.x % G : 0, tid = threadIdx.x;
    const bool lead = part == 0;                                        // side effects that must happen once per query
    int out_row = q, id_lo = 0;
    if (gio.views) {                                                    // uniform
        const erh::ErhDenseView &v = gio.views[q >> 8];
        if ((q & 255) >= v.nq) return;                                  // a padding row
        X = v.X; N = v.N; pos_mul = v.mul; pos_inv = v.inv; id_lo = v.id_lo;
        out_row = gio.q_src[q];
    } else {                                                            // one view for the launch: rows / ids of a routed group
        if (gio.row_map) out_row = gio.row_map[q];
        id_lo = gio.single_lo;
    }
    const int32_t *const id_map = gio.id_map;
    int c = (int)cand_cnt[q];
    if (c > cap) {                                                      // appends were dropped: not answerable from the list
        if (tid == 0 && lead) bad[q] = 1u;
        c = cap;
    }
    const ErhCand *mine = cand + (int64_t)q * cap;
    const int kk = k < c ? k : c;
    int32_t *o_ids = out_ids + (int64_t)out_row * k;
    double *o_sc = out_scores + (int64_t)out_row * k;
    if (tid == 0) { if (lead) out_len[out_row] = kk; s_cnt = 0; s_m = 0; s_maxerr = 0; s_lvl = 0; s_last = 1; }
    if (lead) for (int i = kk + tid; i < k; i += kFinThreads) { o_ids[i] = -1; o_sc[i] = 0.0; }
    if (kk == 0) return;                                                // uniform

    const float delta = 0.5f * margin_of(qnorm[q], xnorm_max, d);
    // A speculative threshold tau = T - 2*delta is valid iff k chunks reach T in fp32: then the k-th best exact score is
    // >= T - delta and every member of the exact top k has an fp32 score >= T - 2*delta, i.e. is in the list.  (T is
    // rebuilt from tau and nudged up a few ulps, which can only make the test stricter.)
    float lvl = INFINITY;
    if (tau) {
        const float t = tau[q];
        lvl = (t > -INFINITY) ? (t + 2.0f * delta) : -INFINITY;
        if (lvl > -INFINITY) lvl += fabsf(lvl) * 1e-6f;
    }
    int n_lvl = 0;
    // pivot: k-th largest of the 1024 strided maxima (all of them candidates, so it bounds the k-th best from below)
#pragma unroll
    for (int h = 0; h < kFinSlots / kFinThreads; ++h) {
        uint32_t mx = 0;
        for (int i = tid + h * kFinThreads; i < c; i += kFinSlots) {
            const float sc = mine[i].s;
            const uint32_t o = erh_f2ord(sc);
            mx = o > mx ? o : mx;
            n_lvl += (sc >= lvl) ? 1 : 0;
        }
        tmax[tid + h * kFinThreads] = mx;
    }
    if (tau) {
        for (int o = 32; o >= 1; o >>= 1) n_lvl += __shfl_xor(n_lvl, o);
        if ((tid & 63) == 0 && n_lvl) atomicAdd(&s_lvl, n_lvl);
    }
    erh_bitonic_desc<uint32_t>(tmax, kFinSlots);                        // begins and ends with a barrier
    if (tau && tid == 0 && lead && lvl > -INFINITY && s_lvl < k) bad[q] = 1u;   // the speculation failed: exhaustive path
    float gather_thr = -INFINITY;
    if (kk <= kFinSlots && tmax[kk - 1] != 0u) gather_thr = erh_ord2f(tmax[kk - 1]) - ((mode == 1) ? 0.f : 2.0f * delta);
    for (int i = tid; i < c; i += kFinThreads) {
        const ErhCand e = mine[i];
        if (e.s >= gather_thr) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < kFinBuf) buf[pos] = erh_key32(e.s, (int32_t)erh_mulmod(e.idx, pos_inv, N));   // ties rank by ORIGINAL index
        }
    }
    __syncthreads();
    int g = s_cnt;
    if (g > kFinBuf) {                                                  // uniform; pathological tie clusters only
        if (tid == 0 && lead) { bad[q] = 1u; atomicAdd(diag_uncert, 1u); }
        g = kFinBuf;
    }
    const int ns = erh_next_pow2(g < 2 ? 2 : g);
    for (int i = g + tid; i < ns; i += kFinThreads) buf[i] = 0ull;
    erh_bitonic_desc<uint64_t>(buf, ns);

    if (mode == 1 /* ERH_DENSE_FAST */) {
        for (int i = tid; i < kk; i += kFinThreads) {
            const int32_t o = erh_key32_idx(buf[i]);
            o_ids[i] = id_map ? id_map[id_lo + o] : o;
            o_sc[i] = (double)erh_key32_score(buf[i]);
        }
        return;
    }

    // EXACT: everything whose fp32 score is within the margin of the kk-th best can still be in the top kk
    const float thr = erh_key32_score(buf[kk - 1]) - 2.0f * delta;
    for (int i = tid; i < g; i += kFinThreads) {
        const bool in_i = erh_key32_score(buf[i]) >= thr;
        const bool in_n = (i + 1 < g) ? (erh_key32_score(buf[i + 1]) >= thr) : false;
        if (in_i && !in_n) s_m = i + 1;
    }
    __syncthreads();
    int m = s_m;
    bool uncertified = false;
    if (m > erh::kDenseRescoreMax) { m = erh::kDenseRescoreMax; uncertified = true; }
    for (int i = tid; i < m; i += kFinThreads) {
        const int32_t o = erh_key32_idx(buf[i]);
        r_idx[i] = o;
        r_s32[i] = erh_key32_score(buf[i]);
        r_pos[i] = (int32_t)erh_mulmod(o, pos_mul, N);                  // once per candidate, not once per lane and row
    }
    __syncthreads();
    // one wave per pair of candidates: lane j accumulates elements 512*t + 8*j + e (e = 0..7) sequentially in fp64,
    // then an xor butterfly 32,16,..,1.  Products of two fp16 values are exact in fp64, so the result depends only
    // on this order -- which oracle/dense.py: dense_exact_scores reproduces.  The query's fragments stay in
    // registers (rounds of 512 elements, up to 4 = d <= 2048; longer rows reload them).
    const int lane = tid & 63, wave = tid >> 6;
    const _Float16 *qrow = Q16 + (int64_t)q * d;
    constexpr int RW = (QR == 1) ? 3 : 2;               // rows per wave iteration (as many again are in flight)
    constexpr int kWaves = kFinThreads / 64;
    half8 qreg[QR];
#pragma unroll
    for (int t = 0; t < QR; ++t) {
        const int off = 512 * t + 8 * lane;
#pragma unroll
        for (int u = 0; u < 8; ++u) qreg[t][u] = (_Float16)0.f;
        if (off < d) qreg[t] = *reinterpret_cast<const half8 *>(qrow + off);
    }
    half8 xa[RW][QR], xb[RW][QR];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int t = 0; t < QR; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) { xa[r][t][u] = (_Float16)0.f; xb[r][t][u] = (_Float16)0.f; }
#define ERH_FIN_LOAD(DST, E0)                                                                         \
    do {                                                                                              \
        _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                              \
            const int e_ = ((E0) + r < m) ? (E0) + r : (E0);                                          \
            const _Float16 *xr_ = X + (int64_t)r_pos[e_] * d;                                         \
            _Pragma("unroll") for (int t = 0; t < QR; ++t) {                                          \
                const int off_ = 512 * t + 8 * lane;                                                  \
                if (off_ < d) DST[r][t] = *reinterpret_cast<const half8 *>(xr_ + off_);               \
            }                                                                                         \
        }                                                                                             \
    } while (0)
    const int e_step = G * kWaves * RW;                                 // (G > 1: workgroup `part` takes every G-th group of rows)
    int e0 = (part * kWaves + wave) * RW;
    if (e0 < m) ERH_FIN_LOAD(xa, e0);
    for (; e0 < m; e0 += e_step) {
        const int e1 = e0 + e_step;
        if (e1 < m) ERH_FIN_LOAD(xb, e1);                               // next pair in flight during the sums below
        double acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r] = 0.0;
#pragma unroll
        for (int t = 0; t < QR; ++t) {
            if (512 * t + 8 * lane < d) {
                const fin_u4 q2 = __builtin_bit_cast(fin_u4, qreg[t]);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        // the product of two fp16 values has 22 significant bits: exact in fp32 (and far from its
                        // subnormal range), so widening the PRODUCT gives the same double as multiplying widened factors
                        const fin_u4 x2 = __builtin_bit_cast(fin_u4, xa[r][t]);
                        const float p32 = (u & 1) ? fin_mul_f16<true>(x2[u >> 1], q2[u >> 1]) : fin_mul_f16<false>(x2[u >> 1], q2[u >> 1]);
                        acc[r] = acc[r] + (double)p32;
                    }
            }
        }
        if (d > 512 * QR) {                                             // d > 2048: the tail of the rows straight from memory
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int e = (e0 + r < m) ? e0 + r : e0;
                const _Float16 *xr = X + (int64_t)r_pos[e] * d;
                for (int off = 512 * QR + 8 * lane; off < d; off += 512) {
                    const half8 qv = *reinterpret_cast<const half8 *>(qrow + off);
                    const half8 xv = *reinterpret_cast<const half8 *>(xr + off);
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[r] = acc[r] + (double)((float)xv[u] * (float)qv[u]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r)
            for (int o = 32; o >= 1; o >>= 1) acc[r] = acc[r] + __shfl_xor(acc[r], o);
        if (lane == 0) {
            float err = 0.f;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (e0 + r < m) {
                    r_s64[e0 + r] = acc[r];
                    if (G > 1) ws_s64[(int64_t)q * erh::kDenseRescoreMax + e0 + r] = acc[r];
                    err = fmaxf(err, fabsf((float)(acc[r] - (double)r_s32[e0 + r])));
                }
            }
            atomicMax(&s_maxerr, __float_as_uint(err));
        }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int t = 0; t < QR; ++t) xa[r][t] = xb[r][t];
    }
#undef ERH_FIN_LOAD
    if (G > 1) {
        // hand-over: scores and error to memory, fence (every writer its own stores), take a ticket; all but the last workgroup of
        // the query are done
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            atomicMax(&ws_sync[2 * q + 1], s_maxerr);
            __threadfence();
            const unsigned int t = atomicAdd(&ws_sync[2 * q], 1u);
            s_last = (t == (unsigned int)G - 1u) ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;                                            // uniform
        __threadfence();
        for (int i = tid; i < m; i += kFinThreads)
            r_s64[i] = __builtin_nontemporal_load(ws_s64 + (int64_t)q * erh::kDenseRescoreMax + i);   // (every row was written by exactly one workgroup)
        if (tid == 0) {
            s_maxerr = atomicExch(&ws_sync[2 * q + 1], 0u);             // ... and the words are zero again for the next call
            ws_sync[2 * q] = 0u;
        }
        __syncthreads();
    }
    // order by (fp64 desc, index asc): one small record sort (m <= 1024), then the first kk entries are the answer
    const int mp = erh_next_pow2(m < 2 ? 2 : m);
    for (int t = m + tid; t < mp; t += kFinThreads) { r_s64[t] = -INFINITY; r_idx[t] = 0x7fffffff; }
    erh_bitonic_rec_desc<double>(r_s64, r_idx, mp);                     // begins and ends with a barrier
    for (int t = tid; t < kk; t += kFinThreads) { o_ids[t] = id_map ? id_map[id_lo + r_idx[t]] : r_idx[t]; o_sc[t] = r_s64[t]; }
    if (tid == 0) {
        const float me = __uint_as_float(s_maxerr);
        atomicMax((unsigned int *)diag_maxerr, __float_as_uint(me));
        if (uncertified || me > delta) { atomicAdd(diag_uncert, 1u); bad[q] = 1u; }   // re-score set cut, or the error bound failed
    }
}

// ---- exhaustive path: queries the pruned pipeline could not certify ----------------------------------------------
// The reference always answers (Qdrant's exact scan has no candidate budget).  The pruned pipeline has three: a
// 16384-entry candidate list per query, a 2048-entry gather and a 1024-row fp64 re-score set -- a corpus with tens of
// thousands of (near-)duplicates of what a query asks for exhausts them.  Every such query is flagged in bad[q]
// and answered here instead, with the same contract and no budget at all:
//   1. dense_exact_all_kernel   the pinned-order fp64 score of EVERY chunk (one more pass over the matrix per group
//                               of kExGroup flagged queries; products of fp16 values are exact in fp64, the sum order
//                               is dense_finalize_kernel's) -> S64[slot][position];
//   2. dense_exact_select_kernel one workgroup per flagged query streams its S64 row through a 2048-entry list kept
//                               under an exact (score, index) threshold -- ties resolve by original index as always --
//                               and writes the query's final top-k over whatever the pruned pipeline wrote.
// Round 5: a call itself enqueues only the collect kernel (one workgroup: how many queries are flagged -> the call's flag words and
// the device counter of erh_get_stat); the two exact kernels are enqueued by the call's synchronisation point (pipeline_dense.hip:
// dense_check_flags -- inside every host-output call, erh_dense_check for device outputs, which the ABI has always required
// before results are read) and only when the flag word says that a query needs them, kExMax flagged queries per round.  The
// common case -- nothing flagged -- used to pay two empty launches per call for keeping up to kExMax answers free of a host
// round trip that every caller makes anyway.
constexpr int kExGroup = 4;
constexpr int kExMax = 16;
constexpr int kExCap = 2048;

// list[0 .. kExMax) = flagged queries of rank skip .. skip + kExMax - 1 (ascending), count[0] = how many, count[1] = all flagged
__global__ __launch_bounds__(1024) void dense_bad_collect_kernel(const uint32_t *__restrict__ bad, int B, int skip,
                                                                int32_t *__restrict__ list, int32_t *__restrict__ count,
                                                                uint32_t *__restrict__ flags,
                                                                unsigned long long *__restrict__ stats /* erh_get_stat: [0] += flagged queries (the call's own collect only) */,
                                                                int answer /* 1: the exact kernels of this round follow; 0: count only (what a call enqueues) */) {
    __shared__ int s_base, s_taken;
    const int tid = threadIdx.x;
    if (tid == 0) { s_base = 0; s_taken = 0; }
    __syncthreads();
    for (int q0 = 0; q0 < B; q0 += 1024) {                               // ascending order by construction
        const int q = q0 + tid;
        const bool f = q < B && bad[q] != 0u;
        // rank of q among the flagged: block-wide exclusive scan by wave ballots
        __shared__ int wsum[16];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
        const int lane = tid & 63, w = tid >> 6;
        const int within = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __builtin_popcountll(m);
        __syncthreads();
        int before = s_base;
        for (int i = 0; i < w; ++i) before += wsum[i];
        if (f) {
            const int r = before + within - skip;
            if (r >= 0 && r < kExMax) { list[r] = q; atomicAdd(&s_taken, 1); }
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int i = 0; i < 16; ++i) t += wsum[i]; s_base += t; }
        __syncthreads();
    }
    if (tid == 0) {
        count[0] = s_taken;
        count[1] = s_base;
        flags[0] = (answer ? s_base > skip + kExMax : s_base > 0) ? 1u : 0u;   // flagged queries still unanswered after this launch
        flags[3] = (uint32_t)s_base;
        if (answer && s_base <= skip + kExMax) flags[2] = 0u;            // every flagged query gets its exact answer
        if (stats && !answer && s_base) atomicAdd(&stats[0], (unsigned long long)s_base);
    }
}

// grid = any, block = 256 (4 waves), dynamic LDS = kExGroup * d * 2 bytes.  One wave per row and iteration.
__global__ __launch_bounds__(256) void dense_exact_all_kernel(const _Float16 *__restrict__ X, int64_t N, int d,
                                                            const _Float16 *__restrict__ Q16,
                                                            const int32_t *__restrict__ list,
                                                            const int32_t *__restrict__ count, double *__restrict__ S64) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *qs = reinterpret_cast<_Float16 *>(smem);
    const int n_bad = count[0];
    if (n_bad <= 0) return;                                              // the normal case: nothing to do
    const int lane = threadIdx.x & 63;
    const int64_t wave_g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int g0 = 0; g0 < n_bad; g0 += kExGroup) {
        const int ng = (n_bad - g0 < kExGroup) ? n_bad - g0 : kExGroup;
        __syncthreads();
        for (int i = threadIdx.x; i < ng * (d / 8); i += 256) {
            const int g = i / (d / 8), o = i % (d / 8);
            reinterpret_cast<half8 *>(qs)[g * (d / 8) + o] =
                reinterpret_cast<const half8 *>(Q16 + (int64_t)list[g0 + g] * d)[o];
        }
        __syncthreads();
        for (int64_t pos = wave_g; pos < N; pos += n_waves) {
            const _Float16 *xr = X + pos * d;
            double acc[kExGroup];
#pragma unroll
            for (int g = 0; g < kExGroup; ++g) acc[g] = 0.0;
            for (int off = 8 * lane; off < d; off += 512) {               // lane j: elements 512*t + 8*j + e, sequentially
                const half8 xv = *reinterpret_cast<const half8 *>(xr + off);
#pragma unroll
                for (int g = 0; g < kExGroup; ++g) {
                    if (g < ng) {
                        const half8 qv = *reinterpret_cast<const half8 *>(qs + g * d + off);
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[g] = acc[g] + (double)((float)xv[u] * (float)qv[u]);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < kExGroup; ++g)
                for (int o = 32; o >= 1; o >>= 1) acc[g] = acc[g] + __shfl_xor(acc[g], o);
            if (lane == 0) {
#pragma unroll
                for (int g = 0; g < kExGroup; ++g)
                    if (g < ng) S64[(int64_t)(g0 + g) * N + pos] = acc[g];
            }
        }
    }
}

// grid = kExMax, block = 1024.
__global__ __launch_bounds__(1024) void dense_exact_select_kernel(
    const double *__restrict__ S64, int64_t N, int k, const int32_t *__restrict__ list, const int32_t *__restrict__ count,
    const int16_t *__restrict__ filter_dir, const int16_t *__restrict__ dir_id /* by ORIGINAL index */, int64_t pos_inv,
    int32_t *__restrict__ out_ids, double *__restrict__ out_scores, int32_t *__restrict__ out_len) {
    __shared__ double cs[kExCap];
    __shared__ int32_t ci[kExCap];
    __shared__ int s_n;
    __shared__ double s_tau;
    __shared__ int s_tau_idx, s_have_tau;
    const int slot = blockIdx.x, tid = threadIdx.x;
    if (slot >= count[0]) return;                                        // uniform
    const int q = list[slot];
    const int fd = filter_dir ? (int)filter_dir[q] : -1;
    const double *row = S64 + (int64_t)slot * N;
    if (tid == 0) { s_n = 0; s_have_tau = 0; s_tau = 0.0; s_tau_idx = 0; }
    __syncthreads();
    auto shrink = [&]() {                                                // uniform call: sort, keep k, refresh the threshold
        __syncthreads();
        const int n = s_n;
        const int np2 = erh_next_pow2(n < 2 ? 2 : n);
        for (int i = n + tid; i < np2; i += 1024) { cs[i] = -INFINITY; ci[i] = 0x7fffffff; }
        erh_bitonic_rec_desc<double>(cs, ci, np2);
        if (tid == 0) {
            s_n = n < k ? n : k;
            if (n >= k) { s_have_tau = 1; s_tau = cs[k - 1]; s_tau_idx = ci[k - 1]; }
        }
        __syncthreads();
    };
    for (int64_t p0 = 0; p0 < N; p0 += 1024) {
        if (s_n > kExCap - 1024) shrink();                               // uniform (read after a barrier): appends below always fit
        const int64_t pos = p0 + tid;
        if (pos < N) {
            const double s = row[pos];
            const int32_t orig = (int32_t)erh_mulmod(pos, pos_inv, N);
            bool pass = !s_have_tau || s > s_tau || (s == s_tau && orig < s_tau_idx);
            if (pass && fd >= 0 && (int)dir_id[orig] != fd) pass = false;
            if (pass) {
                const int at = atomicAdd(&s_n, 1);
                cs[at] = s;
                ci[at] = orig;
            }
        }
        __syncthreads();
    }
    shrink();
    const int n = s_n < k ? s_n : k;
    for (int i = tid; i < k; i += 1024) {
        out_ids[(int64_t)q * k + i] = i < n ? ci[i] : -1;
        out_scores[(int64_t)q * k + i] = i < n ? cs[i] : 0.0;
    }
    if (tid == 0) out_len[q] = n;
}

}  // namespace

namespace erh {

static int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }

hipError_t select_init() {
    hipError_t e;
    e = hipFuncSetAttribute((const void *)cand_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDenseCapMax * 8 + 64);
    if (e != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_prep_queries(const void *q, int q_dtype, int normalize, int B, int Bpad, int d,
                               _Float16 *Q16, float *qnorm, uint32_t *zero_bad, uint32_t *zero_flags, hipStream_t st,
                               const int32_t *q_src, uint32_t *zero_extra, int n_extra) {
    if (n_extra > 16) return hipErrorInvalidValue;
    if (q_dtype == 0)
        hipLaunchKernelGGL(prep_queries_kernel<_Float16>, dim3(Bpad), dim3(256), 0, st,
                           (const _Float16 *)q, normalize, B, d, Q16, qnorm, zero_bad, zero_flags, q_src, zero_extra, n_extra);
    else
        hipLaunchKernelGGL(prep_queries_kernel<float>, dim3(Bpad), dim3(256), 0, st,
                           (const float *)q, normalize, B, d, Q16, qnorm, zero_bad, zero_flags, q_src, zero_extra, n_extra);
    return hipGetLastError();
}

hipError_t launch_convert_rows(const float *x, int64_t n, int d, int normalize, _Float16 *out_base, int64_t r0,
                               int64_t mul, int64_t N, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n < 65536 ? n : 65536);
    hipLaunchKernelGGL(convert_rows_kernel, dim3(grid), dim3(256), 0, st, x, n, d, normalize, out_base, r0, mul, N);
    return hipGetLastError();
}

hipError_t launch_permute_rows(const _Float16 *x, int64_t n, int d, _Float16 *out_base, int64_t r0, int64_t mul,
                               int64_t N, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n < 65536 ? n : 65536);
    hipLaunchKernelGGL(permute_rows_kernel, dim3(grid), dim3(256), 0, st, x, n, d, out_base, r0, mul, N);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const _Float16 *X, const int32_t *rows, int64_t row0, int64_t n, int d, int64_t mul, int64_t N, _Float16 *out,
                              hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n < 65536 ? n : 65536);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, st, X, rows, row0, n, d, mul, N, out);
    return hipGetLastError();
}

hipError_t launch_permute_dir(const int16_t *dir_id, int64_t N, int64_t inv, int16_t *out, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(permute_dir_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dir_id, N, inv, out);
    return hipGetLastError();
}

hipError_t launch_row_norm_max(const _Float16 *x, int64_t n, int d, float *out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    if (e != hipSuccess || n <= 0) return e;
    const unsigned grid = (unsigned)(n < 8192 ? n : 8192);
    hipLaunchKernelGGL(row_norm_max_kernel, dim3(grid), dim3(256), 0, st, x, n, d, out);
    return hipGetLastError();
}

hipError_t launch_seed_select(const float *S0, int ld_s0, int n0, int64_t c0, int B, int k, int rank,
                              const float *qnorm, float xnorm_max, int d,
                              const int16_t *filter_dir, const int16_t *dir_id,
                              float *tau, ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *bad,
                              uint32_t *need_full, hipStream_t st, const ErhDenseView *views) {
    const int np2 = pow2_ge(n0 < 2 ? 2 : n0);         // (grouped call: n0 = the longest prefix of the table; every tile uses its own)
    if (rank < 1 || rank > k) rank = k;
    hipLaunchKernelGGL(seed_select_kernel, dim3(B), dim3(kSelThreads), 0, st,
                       S0, ld_s0, n0, np2, c0, k, rank, qnorm, xnorm_max, d, filter_dir, dir_id, tau, cand, cand_cnt, cap,
                       bad, need_full, views);
    return hipGetLastError();
}

bool seed_cells_select_fits(int n_vals) { return (size_t)pow2_ge(n_vals < 2 ? 2 : n_vals) * 4 <= 48 * 1024; }

hipError_t launch_seed_cells_select(const float *seed_top, int n_vals, int B, int rank, const float *qnorm, float xnorm_max, int d,
                                    float *tau, uint32_t *cand_cnt, hipStream_t st, const ErhDenseView *views) {
    if (B <= 0) return hipSuccess;
    const int np2 = pow2_ge(n_vals < 2 ? 2 : n_vals);
    if (!seed_cells_select_fits(n_vals)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(seed_cells_select_kernel, dim3(B), dim3(256), (size_t)np2 * 4, st, seed_top, n_vals, np2, rank, qnorm,
                       xnorm_max, d, tau, cand_cnt, views);
    return hipGetLastError();
}

hipError_t launch_cand_refine(int B, int k, const float *qnorm, float xnorm_max, int d,
                              float *tau, ErhCand *cand, uint32_t *cand_cnt, int cap, uint32_t *bad, hipStream_t st) {
    const int cp2 = pow2_ge(cap);
    const int light = cp2 < kRefineLight ? cp2 : kRefineLight;
    hipLaunchKernelGGL(cand_refine_kernel, dim3(B), dim3(kSelThreads), (size_t)light * 8 + 64, st,
                       k, light, qnorm, xnorm_max, d, tau, cand, cand_cnt, cap, -1, light, bad);
    if (cp2 > light)
        hipLaunchKernelGGL(cand_refine_kernel, dim3(B), dim3(kSelThreads), (size_t)cp2 * 8 + 64, st,
                           k, cp2, qnorm, xnorm_max, d, tau, cand, cand_cnt, cap, light, cp2, bad);
    return hipGetLastError();
}

// Batches up to this size run 16 workgroups per query in the final kernel: one query per call -2.4 % wall (0.502 -> 0.490 ms), two -1.9 %,
// four -0.5 %, eight +2.7 %, sixteen +11 % (profiles/r05m_ab_fin_split.log): the redundant pivot / sort stage of 16 x B workgroups costs more
// than the shared row gathers save as soon as the queries alone spread over the chip.
int dense_finalize_split_max() { return 2; }
// option dense_fin_wgs: workgroups of the final kernel per CU -- 4 (what 64 VGPRs and 36 KiB of LDS allow) or 3 / 2 (A/B arms: unused
// dynamic LDS per workgroup keeps the others out)
static int g_fin_wgs = 4;
void dense_finalize_set_wgs(int v) { g_fin_wgs = v <= 2 ? 2 : v == 3 ? 3 : 4; }

hipError_t launch_dense_finalize(int B, int k, int mode, const float *qnorm, float xnorm_max, int d,
                                 const _Float16 *X, const _Float16 *Q16,
                                 const ErhCand *cand, const uint32_t *cand_cnt, int cap,
                                 int32_t *out_ids, double *out_scores, int32_t *out_len,
                                 float *diag_maxerr, uint32_t *diag_uncert, uint32_t *bad, int64_t N,
                                 int64_t pos_mul, int64_t pos_inv, const float *tau_verify, int n_cus, double *ws_s64,
                                 uint32_t *ws_sync, hipStream_t st, const ErhGroupIo *gio) {
    // workgroups per query: small batches spread a query's row gathers over the chip (EXACT mode only; the work space is optional)
    int G = 1;
    if (mode == 0 && ws_s64 && ws_sync && B <= dense_finalize_split_max() && !(gio && gio->views))
        G = 16;
    (void)n_cus;
    ErhGroupIo gv{};
    if (gio) gv = *gio;
#define ERH_FIN_LAUNCH(QR)                                                                                  \
    hipLaunchKernelGGL(dense_finalize_kernel<QR>, dim3(B * G), dim3(kFinThreads), g_fin_wgs == 2 ? 20480 : g_fin_wgs == 3 ? 4608 : 0, st, k, mode, qnorm, xnorm_max, d, X, Q16, \
                       cand, cand_cnt, cap, out_ids, out_scores, out_len, diag_maxerr, diag_uncert, bad, N, pos_mul, \
                       pos_inv, tau_verify, G, ws_s64, ws_sync, gv)
    if (d <= 512) ERH_FIN_LAUNCH(1);
    else if (d <= 1024) ERH_FIN_LAUNCH(2);
    else ERH_FIN_LAUNCH(4);
#undef ERH_FIN_LAUNCH
    return hipGetLastError();
}

hipError_t launch_gather_query_rows(const void *q, const int32_t *idx, int n, int row_bytes, void *out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (row_bytes % 16 != 0) return hipErrorInvalidValue;
    const int64_t total = (int64_t)n * (row_bytes / 16);
    hipLaunchKernelGGL(gather_query_rows_kernel, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, st,
                       (const int4 *)q, idx, n, row_bytes / 16, (int4 *)out);
    return hipGetLastError();
}

hipError_t launch_scatter_topk_rows(const int32_t *ids, const double *sc, const int32_t *len, const int32_t *idx, int n, int k,
                                    int32_t id_offset, const int32_t *id_map, int32_t *out_ids, double *out_sc, int32_t *out_len,
                                    hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t total = (int64_t)n * k;
    hipLaunchKernelGGL(scatter_topk_rows_kernel, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, st, ids,
                       sc, len, idx, n, k, id_offset, id_map, out_ids, out_sc, out_len);
    return hipGetLastError();
}

int dense_exhaustive_max() { return kExMax; }
size_t dense_exhaustive_bytes(int64_t N) { return (size_t)kExMax * (size_t)N * 8 + 64 * 4; }

// ws = [S64 kExMax x N doubles | list int32[kExMax] | count int32[2]]; flags = the call's flag words.
hipError_t launch_dense_exhaustive(const uint32_t *bad, int B, int skip, int k, const _Float16 *X, int64_t N, int d,
                                   const _Float16 *Q16, const int16_t *filter_dir, const int16_t *dir_id,
                                   int64_t pos_inv, void *ws, uint32_t *flags, int n_cus,
                                   int32_t *out_ids, double *out_scores, int32_t *out_len, unsigned long long *stats,
                                   int collect_only, hipStream_t st) {
    double *S64 = reinterpret_cast<double *>(ws);
    int32_t *list = reinterpret_cast<int32_t *>(S64 + (size_t)kExMax * (size_t)N);
    int32_t *count = list + kExMax;
    hipLaunchKernelGGL(dense_bad_collect_kernel, dim3(1), dim3(1024), 0, st, bad, B, skip, list, count, flags, stats,
                       collect_only ? 0 : 1);
    if (collect_only) return hipGetLastError();
    const size_t lds = (size_t)kExGroup * d * 2;
    hipError_t e = hipFuncSetAttribute((const void *)dense_exact_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dense_exact_all_kernel, dim3((unsigned)(n_cus * 8)), dim3(256), lds, st, X, N, d, Q16, list, count,
                       S64);
    hipLaunchKernelGGL(dense_exact_select_kernel, dim3(kExMax), dim3(1024), 0, st, S64, N, k, list, count, filter_dir,
                       dir_id, pos_inv, out_ids, out_scores, out_len);
    return hipGetLastError();
}

}  // namespace erh
