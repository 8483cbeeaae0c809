// BM25 index build on the device: token-id stream -> CSR inverted postings (term-major, documents ascending inside a
// term) with term frequencies.  Replaces the O(corpus) Python loops of BM25Retriever.__init__
// (/root/reference/src/easyrag/custom/retrievers.py:94-118 -> rank_bm25.BM25Okapi.__init__ / bm25s.BM25.index) on this
// side of the boundary; easyrag_amd/index.py (numpy) stays as the checker.  SURVEY.md section 8 row f3, first half
// (the tokeniser itself stays host-side: jieba is not available offline).
//
//   1. key[p] = term[p] << 32 | doc(p)           doc(p) by binary search in the document offsets (one thread per token)
//      first[term] = min p                        (rank_bm25 sums idf in first-appearance order of the terms)
//   2. radix sort of the 64-bit keys              (hipCUB DeviceRadixSort over the bits actually used)
//   3. run-length encode                          unique (term, doc) pairs + tf = run length  -> nnz
//   4. split keys into doc_ids / tf, df[term] += 1 per pair; indptr = exclusive scan of df (host, V entries)
// idf, the epsilon floor and avgdl are V-sized / scalar work on the host in api.hip (libm's log is what Python's
// math.log calls, so the values are the library's bit for bit); the per-posting payload is bm25_payload_kernel.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "kernels.h"

namespace {

__global__ void csr_keys_kernel(const int32_t *__restrict__ tok, const int64_t *__restrict__ doc_off, int64_t T, int64_t N,
                                uint64_t *__restrict__ keys, unsigned long long *__restrict__ first_pos,
                                uint32_t *__restrict__ bad_token, int64_t V) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= T) return;
    // document of token p: last d with doc_off[d] <= p
    int64_t lo = 0, hi = N;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (doc_off[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int32_t t = tok[p];
    if (t < 0 || (int64_t)t >= V) { atomicOr(bad_token, 1u); keys[p] = ~0ull; return; }
    keys[p] = ((uint64_t)(uint32_t)t << 32) | (uint64_t)(uint32_t)lo;
    atomicMin(&first_pos[t], (unsigned long long)p);
}

__global__ void csr_split_kernel(const uint64_t *__restrict__ uniq, const int32_t *__restrict__ counts, int64_t nnz,
                                 int32_t *__restrict__ doc_ids, int32_t *__restrict__ tf,
                                 unsigned long long *__restrict__ df) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const uint64_t k = uniq[i];
    doc_ids[i] = (int32_t)(uint32_t)k;
    tf[i] = counts[i];
    atomicAdd(&df[k >> 32], 1ull);
}

__global__ void fill_u64_kernel(unsigned long long *p, int64_t n, unsigned long long v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

namespace erh {

hipError_t launch_csr_keys(const int32_t *tok, const int64_t *doc_off, int64_t T, int64_t N, int64_t V, uint64_t *keys,
                           unsigned long long *first_pos, uint32_t *bad_token, hipStream_t st) {
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, first_pos, V, ~0ull);
    if (T > 0)
        hipLaunchKernelGGL(csr_keys_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, tok, doc_off, T, N, keys,
                           first_pos, bad_token, V);
    return hipGetLastError();
}

// temp == nullptr: only *temp_bytes is set (the larger of the two primitives' needs).
hipError_t csr_sort_rle(const uint64_t *keys_in, uint64_t *keys_sorted, int64_t T, int key_bits, uint64_t *uniq,
                        int32_t *counts, int32_t *num_runs, void *temp, size_t *temp_bytes, hipStream_t st) {
    size_t a = 0, b = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, a, keys_in, keys_sorted, (int)T, 0, key_bits, st);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceRunLengthEncode::Encode(nullptr, b, keys_sorted, uniq, counts, num_runs, (int)T, st);
    if (e != hipSuccess) return e;
    const size_t need = a > b ? a : b;
    if (!temp) { *temp_bytes = need; return hipSuccess; }
    if (*temp_bytes < need) return hipErrorInvalidValue;
    size_t n = *temp_bytes;
    e = hipcub::DeviceRadixSort::SortKeys(temp, n, keys_in, keys_sorted, (int)T, 0, key_bits, st);
    if (e != hipSuccess) return e;
    n = *temp_bytes;
    return hipcub::DeviceRunLengthEncode::Encode(temp, n, keys_sorted, uniq, counts, num_runs, (int)T, st);
}

hipError_t launch_csr_split(const uint64_t *uniq, const int32_t *counts, int64_t nnz, int64_t V, int32_t *doc_ids,
                            int32_t *tf, unsigned long long *df, hipStream_t st) {
    hipError_t e = hipMemsetAsync(df, 0, (size_t)V * 8, st);
    if (e != hipSuccess || nnz <= 0) return e;
    hipLaunchKernelGGL(csr_split_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, uniq, counts, nnz, doc_ids,
                       tf, df);
    return hipGetLastError();
}

}  // namespace erh
