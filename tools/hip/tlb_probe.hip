// Does it matter for scattered reads whether 16 GiB are ONE allocation or many?  (TLB reach: the driver maps a large contiguous
// allocation with large page-table fragments; many separate hipMallocs of a few MiB each can at best get 2 MiB ones.)
//   every lane reads 8 B from a random 128-byte line of a random chunk; chunks: 1 x 16 GiB | 256 x 64 MiB | 4096 x 4 MiB | 16384 x 1 MiB
//   (the small ones allocated in an interleaved order with dummy allocations between them, as a batch of handles would)
// build: hipcc --offload-arch=gfx950 -O2 -o tlb_probe tlb_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t h) { h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33; return h; }
__global__ __launch_bounds__(256) void k(const double* const* chunks, uint64_t nchunks, uint64_t lines_per_chunk, int reps, double* out) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t h = mix(gid * 0x9E3779B97F4A7C15ull + (uint64_t)(r * 8 + u));
            const double* base = chunks[h % nchunks];
            v[u] = base[((h >> 20) % lines_per_chunk) * 16 + ((h >> 50) & 15)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 1.2345e-300) out[0] = acc;
}
int main() {
    const uint64_t total = 16ull << 30;
    double* out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t counts[4] = {1, 256, 4096, 16384};
    for (int c = 0; c < 4; ++c) {
        const uint64_t n = counts[c], sz = total / n;
        std::vector<double*> ptrs(n), dummies;
        for (uint64_t i = 0; i < n; ++i) {
            CK(hipMalloc(&ptrs[i], sz));
            if (n > 1 && (i % 3) == 0) { double* d; CK(hipMalloc(&d, 300 * 1024)); dummies.push_back(d); }  // (other small buffers in between)
        }
        for (uint64_t i = 0; i < n; ++i) CK(hipMemsetAsync(ptrs[i], 0, sz, 0));
        const double** dtab; CK(hipMalloc(&dtab, n * 8)); CK(hipMemcpy(dtab, ptrs.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        const int grid = 256 * 32, reps = 32;
        float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0));
            k<<<grid, 256>>>((const double* const*)dtab, n, sz / 128, reps, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double loads = (double)grid * 256 * reps * 8;
        printf("%6llu chunk(s) of %8.1f MiB: %.2f ms  %.1f G random lines/s = %.2f TB/s at 128 B per line\n", (unsigned long long)n, sz / 1048576.0, ms, loads / ms / 1e6, loads * 128 / ms / 1e9);
        for (auto p : ptrs) CK(hipFree(p));
        for (auto p : dummies) CK(hipFree(p));
        CK(hipFree(dtab));
    }
    return 0;
}
