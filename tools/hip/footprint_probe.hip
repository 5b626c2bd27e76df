// Does the FOOTPRINT of the randomly accessed data cost latency (address translation), beyond the caches?  BASELINE config 5 keeps 96 dense
// sequences x ~1.3 GB of sparsely touched map table and voxel blocks: ~130 GB of working set touched one 128-byte line at a time.
// Every thread walks a chain of DEPENDENT random line reads (the next address is formed from the value just loaded - the buffers are zero, the
// compiler cannot know) over a working set of W GiB made of 1-GiB allocations; 256 workgroups x 512 threads = the free-running kernel's
// residency (2 wavefronts per SIMD), U independent chains per thread.  Prints ns per dependent step and lines/s for W = 1 .. 224 GiB.
// build: hipcc --offload-arch=gfx950 -O2 -o footprint_probe footprint_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t h) { h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33; return h; }
struct Bufs { const double* p[256]; };
template <int U>
__global__ __launch_bounds__(512) void k(Bufs b, int nbuf, uint64_t lines_per_buf, int steps, double* out) {
    const uint64_t gid = (uint64_t)blockIdx.x * 512 + threadIdx.x;
    uint64_t h[U];
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) h[u] = mix(gid * 0x9E3779B97F4A7C15ull + u);
    for (int s = 0; s < steps; ++s) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t q = h[u] % ((uint64_t)nbuf * lines_per_buf);
            v[u] = b.p[q / lines_per_buf][(q % lines_per_buf) * 16 + ((h[u] >> 50) & 15)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += v[u]; h[u] = mix(h[u] + (uint64_t)__double_as_longlong(v[u]) + 1u); }
    }
    if (acc == 1.2345e-300) out[0] = acc;
}
int main(int argc, char** argv) {
    const int max_gib = argc > 1 ? atoi(argv[1]) : 224;
    std::vector<double*> bufs;
    double* out;
    CK(hipMalloc(&out, 8));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("device memory: %.1f GiB free of %.1f\n", free_b / 1073741824.0, total_b / 1073741824.0);
    const uint64_t lines_per_buf = (1ull << 30) / 128;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int sizes[] = {1, 4, 16, 64, 128, 192, 224};
    for (int W : sizes) {
        if (W > max_gib || W > 250) break;
        while ((int)bufs.size() < W) {
            double* p;
            if (hipMalloc(&p, 1ull << 30) != hipSuccess) { printf("allocation stopped at %zu GiB\n", bufs.size()); W = -1; break; }
            CK(hipMemsetAsync(p, 0, 1ull << 30));
            bufs.push_back(p);
        }
        if (W < 0) break;
        CK(hipDeviceSynchronize());
        Bufs b;
        for (int i = 0; i < 256; ++i) b.p[i] = bufs[i < W ? i : 0];
        for (int U : {1, 4}) {
            const int steps = 400;
            float best = 1e30f;
            for (int pass = 0; pass < 3; ++pass) {
                CK(hipEventRecord(e0));
                if (U == 1) k<1><<<256, 512>>>(b, W, lines_per_buf, steps, out); else k<4><<<256, 512>>>(b, W, lines_per_buf, steps, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (pass && ms < best) best = ms;
            }
            const double lines = 256.0 * 512 * steps * U;
            printf("W = %3d GiB  U = %d chains per thread: %.0f ns per dependent step, %.2f G lines/s = %.2f TB/s of 128-B lines\n", W, U, best * 1e6 / steps, lines / best / 1e6, lines * 128 / best / 1e9);
        }
    }
    return 0;
}
