// Micro-benchmark (measurement only, not part of the library): what does FETCH_SIZE count on gfx950 for the access shapes the library's
// kernels use?  MI355X_MICROARCH.md says to double it ("128-byte requests are tallied at 64 bytes"); that was calibrated on wide coalesced
// streams.  The BM25 scan mixes 16-byte-per-lane posting loads with the 4-byte gathers of its binary searches, so its
// traffic / algorithmic ratio was "no evidence of anything" (VERDICT r5, 8).  Four kernels of KNOWN volume over a 4 GiB buffer (16 x the
// 256 MiB Infinity Cache, so every line comes from HBM), each launched once per rocprofv3 --pmc FETCH_SIZE pass (scripts/gpu_fetch_calib.sh):
//   stream16   every lane loads 16 consecutive bytes, a wave 1 KiB              -> 4 GiB, every 128-byte line fetched whole
//   stream4    every lane loads 4 consecutive bytes, a wave 256 B               -> 4 GiB
//   stride128  lane i of the grid loads the dword at byte 128 * i               -> one dword per 128-byte line: 32 Mi lines touched once
//   gather4    every lane loads a dword at a pseudo-random address              -> 32 Mi loads; almost all hit distinct lines
// The table printed by the script gives FETCH_SIZE [KiB] x 1024 / known bytes per kernel: 0.5 means "double it".
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch_calib.hip -o scripts/ubench/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_stream16(const u4 *__restrict__ p, size_t n16, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u4 v = __builtin_nontemporal_load(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_stream4(const uint32_t *__restrict__ p, size_t n4, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x12345u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_stride128(const uint32_t *__restrict__ p, size_t n_lines, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * 256) acc ^= p[i * 32];
    if (acc == 0x12345u) *sink = acc;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(256) void calib_gather4(const uint32_t *__restrict__ p, size_t n4, size_t loads, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < loads; i += (size_t)gridDim.x * 256) acc ^= p[mix64(i + 1) % n4];
    if (acc == 0x12345u) *sink = acc;
}

int main() {
    const size_t bytes = 4ull << 30;
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&sink), 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    const size_t n_lines = bytes / 128;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto timed = [&](const char *name, double known_bytes, auto launch) {
        (void)hipEventRecord(a);
        launch();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("%-10s known_bytes %.0f  %.3f ms  %.0f GB/s of known bytes\n", name, known_bytes, ms, known_bytes / ms / 1e6);
    };
    timed("stream16", (double)bytes, [&] { hipLaunchKernelGGL(calib_stream16, dim3(8192), dim3(256), 0, 0, (const u4 *)buf, bytes / 16, sink); });
    timed("stream4", (double)bytes, [&] { hipLaunchKernelGGL(calib_stream4, dim3(8192), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink); });
    timed("stride128", (double)n_lines * 128.0, [&] { hipLaunchKernelGGL(calib_stride128, dim3(8192), dim3(256), 0, 0, (const uint32_t *)buf, n_lines, sink); });
    timed("gather4", (double)n_lines * 128.0, [&] { hipLaunchKernelGGL(calib_gather4, dim3(8192), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, n_lines, sink); });
    printf("lines (128 B) in the buffer: %zu; stride128 and gather4 issue one dword load per line count (known_bytes = lines x 128; a 64-byte fetch granule would move half)\n", n_lines);
    return 0;
}
