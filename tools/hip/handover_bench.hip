// Cross-stream hand-over latency on one GPU: kernel A (stream 1) ends -> kernel B (stream 2) starts, with
//   (1) hipEventRecord + hipStreamWaitEvent,  (2) a flag A writes + hipStreamWaitValue32 on stream 2,
//   (3) B already resident and spinning on the flag.   Times from wall_clock64() (100 MHz) inside the kernels.
// build: hipcc --offload-arch=gfx950 -O2 -o handover_bench handover_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void kA(unsigned* flag, unsigned v, long long* t_end, int spin_us) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 100LL * spin_us) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *t_end = wall_clock64();
        __threadfence_system();
        if (flag) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void kB(const unsigned* flag, unsigned v, long long* t_start, int spin) {
    if (spin) while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) *t_start = wall_clock64();
}

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned* flag; long long *ta, *tb;
    const int N = 200;
    CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    CK(hipMalloc(&ta, 8 * N)); CK(hipMalloc(&tb, 8 * N));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<long long> a(N), b(N);
    for (int mode = 1; mode <= 3; ++mode) {
        CK(hipMemset(flag, 0, 4)); CK(hipDeviceSynchronize());
        for (int i = 0; i < N; ++i) {
            if (mode == 1) {
                kA<<<1, 64, 0, s1>>>(nullptr, 0, ta + i, 50);
                CK(hipEventRecord(ev, s1));
                CK(hipStreamWaitEvent(s2, ev, 0));
                kB<<<1, 64, 0, s2>>>(flag, 0, tb + i, 0);
            } else if (mode == 2) {
                kA<<<1, 64, 0, s1>>>(flag, (unsigned)(i + 1), ta + i, 50);
                hipError_t e = hipStreamWaitValue32(s2, flag, (unsigned)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (e != hipSuccess) { printf("hipStreamWaitValue32 -> %s\n", hipGetErrorString(e)); break; }
                kB<<<1, 64, 0, s2>>>(flag, 0, tb + i, 0);
            } else {
                kB<<<1, 64, 0, s2>>>(flag, (unsigned)(i + 1), tb + i, 1);  // resident, spinning
                kA<<<1, 64, 0, s1>>>(flag, (unsigned)(i + 1), ta + i, 50);
            }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(a.data(), ta, 8 * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), tb, 8 * N, hipMemcpyDeviceToHost));
        double s = 0; int n = 0; double mn = 1e9, mx = 0;
        for (int i = 20; i < N; ++i) { const double d = (b[i] - a[i]) / 100.0; s += d; ++n; if (d < mn) mn = d; if (d > mx) mx = d; }
        printf("mode %d (%s): A end -> B start mean %.1f us  min %.1f  max %.1f\n", mode,
               mode == 1 ? "event record + stream wait" : mode == 2 ? "flag + hipStreamWaitValue32" : "B resident, spinning", s / n, mn, mx);
    }
    return 0;
}
