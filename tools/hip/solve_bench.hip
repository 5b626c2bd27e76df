// micro-benchmark of the per-iteration serial tail of gn8_body (moments -> 27 sums, 6x6 solve, Exp, flag): clock ticks per
// piece, measured the way the kernel runs it - wavefront 0 of a 512-thread workgroup does the work, the other seven wait at
// the barrier - and with the kernel's own functions.
// build: hipcc --offload-arch=gfx950 -O2 -o solve_bench solve_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../ptudes-lab_amd/csrc/devmath.h"
#include "../../ptudes-lab_amd/csrc/icp_kernels.h"
#define CLK() ((long long)__builtin_readcyclecounter())
__global__ __launch_bounds__(512) void k(const double* in, double* out, long long* clk, int reps) {
    __shared__ double mom[GN8_ROW_ENTRIES];
    __shared__ double tot[32];
    __shared__ double Esh2[2][12];
    __shared__ int flag_done2[2];
    const int tid = threadIdx.x;
    if (tid < GN8_ROW_ENTRIES) mom[tid] = in[tid];
    __syncthreads();
    long long t[8] = {0};
    double sink = 0;
    for (int it = 0; it < reps; ++it) {
        __syncthreads();
        if (tid < 64) {
            const long long c0 = CLK();
            const long long c0b = CLK();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                double mm[GN8_ROW_ENTRIES];
#pragma unroll
                for (int e = 0; e < GN8_ROW_ENTRIES; ++e) mm[e] = mom[e];
                if (tid < 27) tot[tid] = sums_from_moments(tid, mm);
                else if (tid == 27) tot[27] = mm[16];
                else if (tid == 28) tot[28] = mm[17];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const long long c1 = CLK();
            double dx[6];
            solve6_ldlt_wave(tot, tid, dx);
            const long long c2 = CLK();
            if (tid == 0) {
                const Rt e = se3_exp_gn(dx);
                for (int q = 0; q < 9; ++q) Esh2[it & 1][q] = e.R[q];
                for (int q = 0; q < 3; ++q) Esh2[it & 1][9 + q] = e.t[q];
                double nn = 0.0;
                for (int q = 0; q < 6; ++q) nn += dx[q] * dx[q];
                flag_done2[it & 1] = (nn < 1e-8) ? 1 : 0;
            }
            const long long c3 = CLK();
            t[0] += c0b - c0; t[1] += c1 - c0b; t[2] += c2 - c1; t[3] += c3 - c2;
            sink += dx[0];
            if (tid == 0 && it == reps - 1) for (int q = 0; q < 6; ++q) out[q] = dx[q];
        }
        __syncthreads();
        if (tid == 0) mom[10 + (it % 6)] += 1e-9 * (sink + Esh2[it & 1][3]) + flag_done2[it & 1];
    }
    if (tid == 0) { for (int i = 0; i < 4; ++i) clk[i] = t[i] / reps; out[12] = sink; }
}
int main() {
    // moments of a plausible scan: W, W s, second moments, sum w r, sum w s x r, pairs, candidates
    double h[GN8_ROW_ENTRIES] = {5200.0, 310.0, -120.0, 45.0, 9.1e4, 1.2e3, -3.0e2, 8.7e4, 2.2e2, 1.9e3, 1.3, -0.7, 0.2, 3.1, -2.2, 0.9, 5600.0, 1.2e5};
    double *din, *dout; long long* dclk;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, 13 * 8); hipMalloc(&dclk, 8 * 8);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        k<<<1, 512>>>(din, dout, dclk, 256);
        long long c[4]; double r[13];
        hipMemcpy(c, dclk, sizeof c, hipMemcpyDeviceToHost); hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
        printf("ticks per iteration: clock read %lld | moments -> sums %lld | solve (wave) %lld | Exp + flag %lld   dx0 %.17g\n", c[0], c[1], c[2], c[3], r[0]);
    }
    return 0;
}
