// micro-benchmark of k_ekf_step's parts: update only, predicts only (1 / 10 samples), both; clock ticks per launch shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../ptudes-lab_amd/csrc/devmath.h"
#include "../ptudes-lab_amd/csrc/ekf_kernels.h"
__global__ __launch_bounds__(EKF_THREADS) void k(EkfState* e, const double* imu, const double* pose, long long* clk) {
    long long t[4];
    for (int rep = 0; rep < 3; ++rep) {
        __syncthreads();
        long long c0 = __builtin_readcyclecounter();
        d_ekf_step(e, nullptr, 0, 0, pose, nullptr, nullptr, nullptr, nullptr, 1);   // update only
        __syncthreads();
        long long c1 = __builtin_readcyclecounter();
        d_ekf_step(e, imu, 1 + 20 * rep, 2 + 20 * rep, nullptr, nullptr, nullptr, nullptr, nullptr, 0);   // one predict
        __syncthreads();
        long long c2 = __builtin_readcyclecounter();
        d_ekf_step(e, imu, 2 + 20 * rep, 12 + 20 * rep, nullptr, nullptr, nullptr, nullptr, nullptr, 0);  // ten predicts
        __syncthreads();
        long long c3 = __builtin_readcyclecounter();
        d_ekf_step(e, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);  // nothing: head + tail
        __syncthreads();
        long long c4 = __builtin_readcyclecounter();
        t[0] = c1 - c0; t[1] = c2 - c1; t[2] = c3 - c2; t[3] = c4 - c3;
    }
    if (threadIdx.x == 0) for (int i = 0; i < 4; ++i) clk[i] = t[i];
}
int main() {
    EkfState* e; double *imu, *pose; long long* clk;
    hipMalloc(&e, sizeof(EkfState)); hipMalloc(&imu, 7 * 100 * 8); hipMalloc(&pose, 128); hipMalloc(&clk, 64);
    double g[3] = {0, 0, -9.78}, z[3] = {0, 0, 0};
    double *dg, *dz; hipMalloc(&dg, 24); hipMalloc(&dz, 24);
    hipMemcpy(dg, g, 24, hipMemcpyHostToDevice); hipMemcpy(dz, z, 24, hipMemcpyHostToDevice);
    k_ekf_init<<<1, 64>>>(e, dg, dz, dz);
    double h[700]; for (int i = 0; i < 100; ++i) { h[7*i] = 100.0 + 0.01 * i; h[7*i+1] = 0.1; h[7*i+2] = 0.0; h[7*i+3] = 9.8; h[7*i+4] = 0.01; h[7*i+5] = 0.0; h[7*i+6] = -0.01; }
    hipMemcpy(imu, h, sizeof h, hipMemcpyHostToDevice);
    double P[16] = {1,0,0,0.1, 0,1,0,0.0, 0,0,1,0.0, 0,0,0,1};
    hipMemcpy(pose, P, 128, hipMemcpyHostToDevice);
    k_ekf_step<<<1, EKF_THREADS>>>(e, imu, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0);  // latch the first sample
    k<<<1, EKF_THREADS>>>(e, imu, pose, clk);
    long long c[4]; hipMemcpy(c, clk, 32, hipMemcpyDeviceToHost);
    printf("ticks: update %lld  one predict %lld  ten predicts %lld  empty (load + store) %lld\n", c[0], c[1], c[2], c[3]);
    return 0;
}
