// What does FETCH_SIZE count for scattered 8-byte reads, and how fast can the part serve them?  (MI355X_MICROARCH.md calibrates the
// counter for wide streaming reads only: it reports half their bytes.)
//   mode 0: streaming, 16 B per lane, coalesced                        (the calibrated case)
//   mode 1: every lane reads 8 B from its own random 128-B line        (one line per lane)
//   mode 2: every lane reads 8 B, 16 consecutive lanes share a random 128-B line (one full line per quarter wavefront)
//   mode 3: every lane reads 8 B from its own random 64-B half line, the other half never touched
// buffer: 16 GiB (far beyond the 256 MiB Infinity Cache).  Prints bytes requested, time, lines/s; run under
// rocprofv3 --pmc FETCH_SIZE for the counter's view of the same launches.
// build: hipcc --offload-arch=gfx950 -O2 -o gather_probe gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t h) { h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33; return h; }
template <int MODE>
__global__ __launch_bounds__(256) void k(const double* buf, uint64_t nlines, int reps, double* out) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (uint64_t)gridDim.x * 256;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t q = (uint64_t)(r * 8 + u);
            if (MODE == 0) {
                const double2 w = ((const double2*)buf)[(q * nthreads + gid) % (nlines * 8)];
                v[u] = w.x + w.y;
            } else if (MODE == 1) {
                const uint64_t h = mix(gid * 0x9E3779B97F4A7C15ull + q);
                v[u] = buf[(h % nlines) * 16 + ((h >> 40) & 15)];
            } else if (MODE == 2) {
                const uint64_t h = mix((gid >> 4) * 0x9E3779B97F4A7C15ull + q);
                v[u] = buf[(h % nlines) * 16 + (gid & 15)];
            } else {
                const uint64_t h = mix(gid * 0x9E3779B97F4A7C15ull + q);
                v[u] = buf[(h % nlines) * 16 + ((h >> 40) & 7)];  // first half of the line only
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 1.2345e-300) out[0] = acc;
}
int main() {
    const uint64_t bytes = 16ull << 30, nlines = bytes / 128;
    double *buf, *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 32, reps = 64;
    const char* names[4] = {"streaming 16 B / lane", "8 B from a random line per lane", "8 B per lane, 16 lanes share a random line", "8 B from a random half line per lane"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0));
            if (mode == 0) k<0><<<grid, 256>>>(buf, nlines, reps, out);
            else if (mode == 1) k<1><<<grid, 256>>>(buf, nlines, reps, out);
            else if (mode == 2) k<2><<<grid, 256>>>(buf, nlines, reps, out);
            else k<3><<<grid, 256>>>(buf, nlines, reps, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double loads = (double)grid * 256 * reps * 8;
            const double lines = mode == 0 ? loads / 8 : mode == 2 ? loads / 16 : loads;
            if (pass) printf("mode %d %-46s %.2f ms  %.1f G loads/s  %.2f G distinct-line touches/s  = %.2f TB/s at 128 B per line, %.2f TB/s at 64 B;  requested %.1f GB\n", mode, names[mode], ms,
                             loads / ms / 1e6, lines / ms / 1e6, lines * 128 / ms / 1e9, lines * 64 / ms / 1e9, loads * (mode == 0 ? 16 : 8) / 1e9);
        }
    }
    return 0;
}
