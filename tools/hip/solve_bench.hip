// micro-benchmark of the per-iteration serial tail of k_gn_loop (solve, exp, compose): clock ticks per piece
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../ptudes-lab_amd/csrc/devmath.h"
#include "../ptudes-lab_amd/csrc/icp_kernels.h"
__global__ void k(const double* in, double* out, long long* clk) {
    __shared__ double tot[32];
    __shared__ double Tsh[12];
    const int tid = threadIdx.x;
    if (tid < 27) tot[tid] = in[tid];
    if (tid < 12) Tsh[tid] = (tid < 9) ? ((tid % 4 == 0) ? 1.0 : 0.0) : 0.0;
    __syncthreads();
    long long t[8] = {0};
    double sink = 0;
    for (int rep = 0; rep < 64; ++rep) {
        __syncthreads();
        long long c0 = __builtin_readcyclecounter();
        double dx[6];
        if (tid == 0) { solve6_ldlt(tot, dx); sink += dx[0]; }
        __syncthreads();
        long long c1 = __builtin_readcyclecounter();
        double dy[6];
        if (tid < 64) { solve6_ldlt_wave(tot, tid, dy); sink += dy[1]; }
        __syncthreads();
        long long c2 = __builtin_readcyclecounter();
        Rt e;
        if (tid == 0) { e = se3_exp(dy); sink += e.R[3]; }
        __syncthreads();
        long long c3 = __builtin_readcyclecounter();
        if (tid == 0) {
            Rt T;
            for (int k = 0; k < 9; ++k) T.R[k] = Tsh[k];
            for (int k = 0; k < 3; ++k) T.t[k] = Tsh[9 + k];
            T = rt_mul(e, T);
            for (int k = 0; k < 9; ++k) Tsh[k] = T.R[k];
            for (int k = 0; k < 3; ++k) Tsh[9 + k] = T.t[k];
            double nn = 0.0;
            for (int k = 0; k < 6; ++k) nn += dy[k] * dy[k];
            sink += sqrt(nn);
        }
        __syncthreads();
        long long c4 = __builtin_readcyclecounter();
        __syncthreads();
        long long c5 = __builtin_readcyclecounter();
        t[0] += c1 - c0; t[1] += c2 - c1; t[2] += c3 - c2; t[3] += c4 - c3; t[4] += c5 - c4;
        if (tid == 0) tot[21 + (rep % 6)] += 1e-9 * sink;
        for (int k = 0; k < 6; ++k) if (tid == 0 && rep == 63) { out[k] = dx[k]; out[6 + k] = dy[k]; }
    }
    if (tid == 0) { for (int i = 0; i < 5; ++i) clk[i] = t[i] / 64; out[12] = sink; }
}
int main() {
    double h[27]; int o = 0;
    for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) h[o++] = (a == b) ? 50.0 + a : 0.3 * (a + 1) - 0.1 * b;
    for (int a = 0; a < 6; ++a) h[o++] = 0.01 * (a + 1);
    double *din, *dout; long long* dclk;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, 13 * 8); hipMalloc(&dclk, 8 * 8);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 1024>>>(din, dout, dclk);
    long long c[5]; double r[13];
    hipMemcpy(c, dclk, sizeof c, hipMemcpyDeviceToHost); hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    printf("ticks: solve_scalar %lld solve_wave %lld se3_exp %lld compose+norm %lld empty_sync %lld\n", c[0], c[1], c[2], c[3], c[4]);
    int same = 1; for (int i = 0; i < 6; ++i) same &= (r[i] == r[6 + i]);
    printf("dx identical: %d  dx0 %.17g\n", same, r[0]);
    return 0;
}
