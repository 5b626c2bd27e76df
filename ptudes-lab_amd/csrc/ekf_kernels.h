// ekf_kernels.h -- the ptudes 18-state error-state EKF on device (gfx950), fp64.
//
// Replaces reference src/ptudes/ins/es_ekf.py: ESEKF.processImu (:191-237, with _insMech :239-257) and
// ESEKF.processPose (:259-329).  One 384-thread workgroup (6 wavefronts) owns the filter: thread (i,j)
// of the first 324 owns covariance entry P[i][j]; the 18x18 products go through LDS.  A launch consumes
// a whole batch of IMU samples and, optionally, one pose update, so the per-scan EKF work of the
// reference's driver loop (cli/ekf_bench.py:493-563) is a single launch on the handle's stream.
//
// State order (es_ekf.py:65-71): POS 0, VEL 3, PHI 6, BG 9, BA 12, G 15.  Attitude is kept as an xyzw
// quaternion and every use converts through it, as the reference's NavState does (ins/data.py:76-82).
#pragma once
#include "devmath.h"

#define EKF_N 18
#define EKF_POS 0
#define EKF_VEL 3
#define EKF_PHI 6
#define EKF_BG 9
#define EKF_BA 12
#define EKF_G 15
#define EKF_CHUNK 64

struct EkfNav {
    double pos[3], q[4], vel[3], bg[3], ba[3], grav[3];
    double cur_lacc[3], cur_avel[3], cur_ts, cur_dt;
    int initialized, n_updates;
};
struct EkfState {
    EkfNav nav;
    double P[EKF_N * EKF_N];
    double Fx[EKF_N * EKF_N];  // persistent: identity + rewritten blocks (es_ekf.py:142, :216-223)
    double W[EKF_N * EKF_N];   // persistent: zero + rewritten diagonal blocks (es_ekf.py:145, :226-233)
    double pose[16];  // NavState.pose_mat() after the last step (ins/data.py:70-74)
};

__device__ __forceinline__ void ekf_write_pose(EkfState* e) {
    double R[9];
    quat_to_R(e->nav.q, R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) e->pose[4 * i + j] = R[3 * i + j];
        e->pose[4 * i + 3] = e->nav.pos[i];
    }
    e->pose[12] = e->pose[13] = e->pose[14] = 0.0;
    e->pose[15] = 1.0;
}

// ESEKF.__init__ (es_ekf.py:73-179)
__global__ void k_ekf_init(EkfState* e, const double* grav, const double* bacc, const double* bgyr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < 3; ++i) { e->nav.pos[i] = 0.0; e->nav.vel[i] = 0.0; e->nav.grav[i] = grav[i]; e->nav.ba[i] = bacc[i]; e->nav.bg[i] = bgyr[i]; }
    e->nav.q[0] = e->nav.q[1] = e->nav.q[2] = 0.0; e->nav.q[3] = 1.0;
    for (int i = 0; i < EKF_N * EKF_N; ++i) { e->P[i] = 0.0; e->Fx[i] = 0.0; e->W[i] = 0.0; }
    for (int i = 0; i < EKF_N; ++i) e->Fx[i * EKF_N + i] = 1.0;
    // initial attitude std = rotation vector of intrinsic-XYZ euler (10, 10, 10) deg (es_ekf.py:104-107)
    const double a = 10.0 * 3.14159265358979323846 / 180.0, cs = cos(a), sn = sin(a);
    const double Rx[9] = {1, 0, 0, 0, cs, -sn, 0, sn, cs}, Ry[9] = {cs, 0, sn, 0, 1, 0, -sn, 0, cs}, Rz[9] = {cs, -sn, 0, sn, cs, 0, 0, 0, 1};
    double R[9], rv[3];
    mat3_mul(Rx, Ry, R);
    mat3_mul(R, Rz, R);
    R_to_rotvec(R, rv);
    for (int i = 0; i < 3; ++i) {
        e->P[(EKF_POS + i) * EKF_N + EKF_POS + i] = 100.0;
        e->P[(EKF_VEL + i) * EKF_N + EKF_VEL + i] = 25.0;
        e->P[(EKF_PHI + i) * EKF_N + EKF_PHI + i] = rv[i] * rv[i];
        e->P[(EKF_BG + i) * EKF_N + EKF_BG + i] = 2.25;
        e->P[(EKF_BA + i) * EKF_N + EKF_BA + i] = 0.25;
        e->P[(EKF_G + i) * EKF_N + EKF_G + i] = 6.25;
    }
    e->nav.cur_ts = 0.0; e->nav.cur_dt = 0.0; e->nav.initialized = 0; e->nav.n_updates = 0;
    for (int i = 0; i < 3; ++i) { e->nav.cur_lacc[i] = 0.0; e->nav.cur_avel[i] = 0.0; }
    ekf_write_pose(e);
}

__device__ __forceinline__ void set_blk3(double* M, int r, int c, const double* B) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(r + i) * EKF_N + c + j] = B[3 * i + j];
}
__device__ __forceinline__ void set_diag3(double* M, int r, int c, double v) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(r + i) * EKF_N + c + j] = (i == j) ? v : 0.0;
}

// One filter step batch.  Order inside a launch: [pose update, if `pose` and update_first] -> IMU samples
// [i0, i1) in order -> [pose update, if `pose` and !update_first].  The sequence runner uses
// update_first = 1 so that one launch per scan does "update with scan k's pose, then predict through the
// IMU samples that precede scan k+1" (reference loop order, cli/ekf_bench.py:493-563).
// imu: rows of 7 doubles (ts, lacc[3], avel[3]).  pose (nullable): 4x4 measurement; meas_cov (nullable): 6x6.
// out_* (nullable): res_poses / res_t / NC-GT row written right after the update (ekf_bench.py:560-563).
// P, Fx and W live in LDS for the whole launch (Fx / W are persistent members in the reference: identity /
// zero plus the blocks each predict rewrites, es_ekf.py:142-145, :216-233).
__device__ __forceinline__ void lds_blk3(double* M, int r, int c, const double* B) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(r + i) * EKF_N + c + j] = B[3 * i + j];
}
__device__ __forceinline__ void lds_diag3(double* M, int r, int c, double v) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(r + i) * EKF_N + c + j] = (i == j) ? v : 0.0;
}

// columns of Fx row i that can be non-zero: the diagonal plus the blocks es_ekf.py:216-223 writes
// ((POS,VEL) diag; (VEL,PHI), (VEL,BA), (PHI,PHI) dense 3x3; (PHI,BG) diag)
__device__ __forceinline__ unsigned ekf_row_mask(int i) {
    if (i < 3) return (1u << i) | (1u << (EKF_VEL + i));
    if (i < 6) return (1u << i) | (7u << EKF_PHI) | (7u << EKF_BA);
    if (i < 9) return (7u << EKF_PHI) | (1u << (EKF_BG + i - EKF_PHI));
    return 1u << i;
}

__device__ __forceinline__ void d_ekf_step(EkfState* e, const double* imu, int i0, int i1, const double* pose,
                                           const double* meas_cov, double* out_pose, double* out_t,
                                           double* out_row8, int update_first) {
    // Fx / W are double-buffered: the mechanisation thread writes the blocks of sample j into buffer j & 1 while the
    // covariance products of sample j - 1 still read buffer (j - 1) & 1 (every active sample rewrites all the blocks,
    // the other entries never change, so both buffers always agree outside the blocks in flight)
    __shared__ double sP[EKF_N * EKF_N], sF2[2][EKF_N * EKF_N], sW2[2][EKF_N * EKF_N], sT[EKF_N * EKF_N];
    __shared__ int sAct[2], s_lastbuf;
    __shared__ double sK[EKF_N * 6], sSi[36], sr[6], sdx[EKF_N];
    __shared__ double sM[6][12];
    __shared__ int sPiv;
    __shared__ double sRd[EKF_CHUNK][10];  // per-sample Exp(dtheta) (9) and dt
    __shared__ double sBias[8];
    const int tid = threadIdx.x;
    const int ti = tid / EKF_N, tj = tid % EKF_N;
    const bool cell = tid < EKF_N * EKF_N;
    if (cell) { sP[tid] = e->P[tid]; sF2[0][tid] = sF2[1][tid] = e->Fx[tid]; sW2[0][tid] = sW2[1][tid] = e->W[tid]; }
    if (tid == 0) s_lastbuf = 0;
    // the mechanisation (one lane) runs in a wavefront of its own when the launch has one to spare, so that it overlaps
    // the covariance products instead of preceding them
    const int TS = (blockDim.x > EKF_N * EKF_N + 60) ? 384 : 0;
    // the nav state lives in LDS for the whole launch (only thread 0 touches it; written back once at the end)
    __shared__ EkfNav nv;
    if (tid < (int)(sizeof(EkfNav) / 8)) ((double*)&nv)[tid] = ((const double*)&e->nav)[tid];
    __syncthreads();
    const int sel[6] = {EKF_POS, EKF_POS + 1, EKF_POS + 2, EKF_PHI, EKF_PHI + 1, EKF_PHI + 2};
    for (int phase = 0; phase < 3; ++phase) {
        const bool do_update = pose && ((phase == 0 && update_first) || (phase == 2 && !update_first));
        if (phase == 1) {
            // The gyro / accelerometer biases are constant between pose updates, so the per-sample rotation
            // increments Exp((w - b_g) dt) - the only transcendental work of a predict - do not depend on the
            // running state: they are computed for a chunk of samples in parallel, one thread per sample.
            for (int c0 = i0; c0 < i1; c0 += EKF_CHUNK) {
                const int nc = (i1 - c0) < EKF_CHUNK ? (i1 - c0) : EKF_CHUNK;
                if (tid == 0) { for (int k = 0; k < 3; ++k) { sBias[k] = nv.bg[k]; sBias[3 + k] = nv.ba[k]; } sBias[6] = nv.cur_ts; }
                __syncthreads();
                if (tid < nc) {
                    const double* row = imu + 7 * (size_t)(c0 + tid);
                    const double tprev = (tid == 0) ? sBias[6] : imu[7 * (size_t)(c0 + tid - 1)];
                    const double dt = row[0] - tprev;  // es_ekf.py:196
                    double dth[3], Rd[9];
                    for (int k = 0; k < 3; ++k) dth[k] = (row[4 + k] - sBias[k]) * dt;
                    rotvec_to_R(dth, Rd);
                    for (int k = 0; k < 9; ++k) sRd[tid][k] = Rd[k];
                    sRd[tid][9] = dt;
                }
                __syncthreads();
                // Software pipeline over the samples, two barriers per sample: while the mechanisation lane works on sample
                // j (stage 1: _insMech; stage 2: the Fx / W blocks), the 324 cell threads form P = Fx P Fx^T + W of
                // sample j - 1 (stage 1: Fx P; stage 2: (Fx P) Fx^T + W).
                double Rp[9], a[3], Rd[9], dtj = 0.0;
                int act_j = 0;
                for (int j = 0; j <= nc; ++j) {
                    const int bj = j & 1, bp = (j - 1) & 1;
                    const bool prev_active = j >= 1 && sAct[bp] != 0;
                    // ---- stage 1
                    if (tid == TS && j < nc) {  // processImu of sample j: latch + mechanisation (:191-214, _insMech :239-257)
                        const double* row = imu + 7 * (size_t)(c0 + j);
                        nv.cur_dt = sRd[j][9];
                        nv.cur_ts = row[0];
                        for (int k = 0; k < 3; ++k) { nv.cur_lacc[k] = row[1 + k]; nv.cur_avel[k] = row[4 + k]; }
                        if (!nv.initialized) {  // :201-203 the first sample only latches
                            nv.initialized = 1;
                            act_j = 0;
                        } else {
                            act_j = 1;
                            dtj = nv.cur_dt;
                            double Rn[9];
                            quat_to_R(nv.q, Rp);  // nav_prev.att_h
                            for (int k = 0; k < 3; ++k) a[k] = row[1 + k] - nv.ba[k];
                            for (int k = 0; k < 9; ++k) Rd[k] = sRd[j][k];
                            for (int k = 0; k < 3; ++k) {
                                const double ag = (Rp[3 * k] * a[0] + Rp[3 * k + 1] * a[1] + Rp[3 * k + 2] * a[2]) + nv.grav[k];
                                nv.pos[k] = nv.pos[k] + nv.vel[k] * dtj + 0.5 * ag * dtj * dtj;
                                nv.vel[k] = nv.vel[k] + ag * dtj;
                            }
                            mat3_mul(Rp, Rd, Rn);
                            R_to_quat(Rn, nv.q);
                        }
                        sAct[bj] = act_j;
                    }
                    if (prev_active && cell) {
                        // P = Fx P Fx^T + W (:235), dense like the reference.  Only the structurally non-zero columns of
                        // each Fx row are visited, in ascending order: skipping exact-zero products leaves every
                        // partial sum bit-identical to the dense loop.
                        const double* sF = sF2[bp];
                        double acc = 0.0;
                        for (unsigned m = ekf_row_mask(ti); m; m &= m - 1) {
                            const int k = __ffs(m) - 1;
                            acc += sF[ti * EKF_N + k] * sP[k * EKF_N + tj];
                        }
                        sT[tid] = acc;
                    }
                    __syncthreads();
                    // ---- stage 2
                    if (tid == TS && j < nc && act_j) {  // Fx blocks (:216-223) and W blocks (:226-233) of sample j
                        double* sF = sF2[bj];
                        double* sW = sW2[bj];
                        double K[9], B[9];
                        lds_diag3(sF, EKF_POS, EKF_VEL, dtj);
                        skew(a, K);
                        mat3_mul(Rp, K, B);
                        for (int k = 0; k < 9; ++k) B[k] = -dtj * B[k];
                        lds_blk3(sF, EKF_VEL, EKF_PHI, B);
                        for (int k = 0; k < 9; ++k) B[k] = -dtj * Rp[k];
                        lds_blk3(sF, EKF_VEL, EKF_BA, B);
                        for (int r = 0; r < 3; ++r)
                            for (int cc = 0; cc < 3; ++cc) B[3 * r + cc] = Rd[3 * cc + r];
                        lds_blk3(sF, EKF_PHI, EKF_PHI, B);
                        lds_diag3(sF, EKF_PHI, EKF_BG, -dtj);
                        // the reference's names of the noise terms do not match their use, this copies the use
                        lds_diag3(sW, EKF_VEL, EKF_VEL, dtj * dtj * (0.049 * 0.049));
                        lds_diag3(sW, EKF_PHI, EKF_PHI, dtj * dtj * (0.38 * 0.38));
                        lds_diag3(sW, EKF_BA, EKF_BA, dtj * (0.0043 * 0.0043));
                        lds_diag3(sW, EKF_BG, EKF_BG, dtj * (0.000466 * 0.000466));
                        s_lastbuf = bj;
                    }
                    if (prev_active && cell) {
                        const double* sF = sF2[bp];
                        double acc = 0.0;
                        for (unsigned m = ekf_row_mask(tj); m; m &= m - 1) {
                            const int k = __ffs(m) - 1;
                            acc += sT[ti * EKF_N + k] * sF[tj * EKF_N + k];
                        }
                        sP[tid] = acc + sW2[bp][tid];
                    }
                    __syncthreads();
                }
            }
            continue;
        }
        if (!do_update) continue;
        // ---- processPose (:259-329); the error state is zero on entry (reset at :327)
        if (tid == 0) {
            double Rk[9], RkT[9], Rm[9], D[9];
            quat_to_R(nv.q, Rk);
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) { RkT[3 * r + cc] = Rk[3 * cc + r]; Rm[3 * r + cc] = pose[4 * r + cc]; }
            for (int k = 0; k < 3; ++k) sr[k] = pose[4 * k + 3] - nv.pos[k];  // :294
            mat3_mul(RkT, Rm, D);
            R_to_rotvec(D, sr + 3);  // :297
        }
        // S = Jp P Jp^T + R (:299) and its inverse by Gauss-Jordan with partial pivoting (np.linalg.inv, :300), held
        // in LDS as the 6 x 12 augmented matrix [S | I]; thread (a, b) owns one entry, every entry sees exactly
        // the operations of the sequential algorithm.
        if (tid < 72) {
            const int a = tid / 12, bb = tid % 12;
            double v;
            if (bb < 6) {
                double rm;
                if (meas_cov) rm = meas_cov[6 * a + bb];
                else rm = (a == bb) ? ((a < 3) ? 0.02 * 0.02 : 0.01 * 0.01) : 0.0;  // :289-292
                v = sP[sel[a] * EKF_N + sel[bb]] + rm;
            } else {
                v = (a == bb - 6) ? 1.0 : 0.0;
            }
            sM[a][bb] = v;
        }
        __syncthreads();
        for (int cI = 0; cI < 6; ++cI) {
            if (tid == 0) {
                int p = cI;
                for (int r = cI + 1; r < 6; ++r)
                    if (fabs(sM[r][cI]) > fabs(sM[p][cI])) p = r;
                sPiv = p;
            }
            __syncthreads();
            const int p = sPiv;
            double mine = 0.0, other = 0.0;
            if (tid < 12 && p != cI) { mine = sM[cI][tid]; other = sM[p][tid]; }
            __syncthreads();
            if (tid < 12 && p != cI) { sM[cI][tid] = other; sM[p][tid] = mine; }
            __syncthreads();
            const double d = sM[cI][cI];
            double pv = 0.0;
            if (tid < 12) pv = sM[cI][tid] / d;
            __syncthreads();
            if (tid < 12) sM[cI][tid] = pv;
            __syncthreads();
            double f = 0.0, prow = 0.0, cur = 0.0;
            const int a = tid / 12, bb = tid % 12;
            if (tid < 72 && a != cI) { f = sM[a][cI]; prow = sM[cI][bb]; cur = sM[a][bb]; }
            __syncthreads();
            if (tid < 72 && a != cI && f != 0.0) sM[a][bb] = cur - f * prow;
            __syncthreads();
        }
        if (tid < 36) sSi[tid] = sM[tid / 6][6 + tid % 6];
        __syncthreads();
        if (tid < EKF_N * 6) {  // K = P Jp^T S^-1 (:300)
            const int i = tid / 6, j = tid % 6;
            double acc = 0.0;
            for (int k = 0; k < 6; ++k) acc += sP[i * EKF_N + sel[k]] * sSi[6 * k + j];
            sK[tid] = acc;
        }
        __syncthreads();
        if (tid < EKF_N) {  // dx = K r (:301)
            double acc = 0.0;
            for (int k = 0; k < 6; ++k) acc += sK[tid * 6 + k] * sr[k];
            sdx[tid] = acc;
        }
        // P = (I - K Jp) P (:303): build (I - K Jp) in sT, then multiply
        if (cell) {
            double v = (ti == tj) ? 1.0 : 0.0;
            for (int k = 0; k < 6; ++k)
                if (sel[k] == tj) v -= sK[ti * 6 + k];
            sT[tid] = v;
        }
        __syncthreads();
        double pn = 0.0;
        if (cell)
            for (int k = 0; k < EKF_N; ++k) pn += sT[ti * EKF_N + k] * sP[k * EKF_N + tj];
        __syncthreads();
        if (cell) sP[tid] = pn;
        __syncthreads();
        if (tid == 0) {
            // inject (:314-319)
            double dth[3], R[9], Rd[9], Rn[9];
            for (int k = 0; k < 3; ++k) {
                nv.pos[k] += sdx[EKF_POS + k];
                nv.vel[k] += sdx[EKF_VEL + k];
                dth[k] = sdx[EKF_PHI + k];
                nv.bg[k] += sdx[EKF_BG + k];
                nv.ba[k] += sdx[EKF_BA + k];
                nv.grav[k] += sdx[EKF_G + k];
            }
            quat_to_R(nv.q, R);
            rotvec_to_R(dth, Rd);
            mat3_mul(R, Rd, Rn);
            R_to_quat(Rn, nv.q);
            // covariance projection of the PHI diagonal block only (:322-324): G = I - hat(dth / 2)
            const double h[3] = {0.5 * dth[0], 0.5 * dth[1], 0.5 * dth[2]};
            double Gm[9], GT[9], B[9], C[9];
            skew(h, Gm);
            for (int k = 0; k < 9; ++k) Gm[k] = -Gm[k];
            Gm[0] += 1.0; Gm[4] += 1.0; Gm[8] += 1.0;
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) { GT[3 * r + cc] = Gm[3 * cc + r]; B[3 * r + cc] = sP[(EKF_PHI + r) * EKF_N + EKF_PHI + cc]; }
            mat3_mul(Gm, B, C);
            mat3_mul(C, GT, C);
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) sP[(EKF_PHI + r) * EKF_N + EKF_PHI + cc] = C[3 * r + cc];
            nv.n_updates += 1;
            // outputs of this update (ekf_bench.py:560-563): nav pose and the last IMU's timestamp
            e->nav = nv;
            ekf_write_pose(e);
            if (out_pose) for (int k = 0; k < 16; ++k) out_pose[k] = e->pose[k];
            if (out_t) *out_t = nv.cur_ts;
            if (out_row8) {  // [t, x, y, z, qx, qy, qz, qw] for the trajectory gather
                out_row8[0] = nv.cur_ts;
                for (int k = 0; k < 3; ++k) out_row8[1 + k] = nv.pos[k];
                for (int k = 0; k < 4; ++k) out_row8[4 + k] = nv.q[k];
            }
        }
        __syncthreads();
    }
    if (cell) { e->P[tid] = sP[tid]; e->Fx[tid] = sF2[s_lastbuf][tid]; e->W[tid] = sW2[s_lastbuf][tid]; }
    if (tid == 0) { e->nav = nv; ekf_write_pose(e); }
}

#define EKF_THREADS 448  /* 324 covariance cells (six wavefronts) + a seventh wavefront for the mechanisation lane */
__global__ __launch_bounds__(EKF_THREADS) void k_ekf_step(EkfState* e, const double* imu, int i0, int i1, const double* pose,
                                                  const double* meas_cov, double* out_pose, double* out_t,
                                                  double* out_row8, int update_first) {
    d_ekf_step(e, imu, i0, i1, pose, meas_cov, out_pose, out_t, out_row8, update_first);
}
// One IMU sample (processImu) / one pose measurement (processPose) with the payload passed by value in the kernel
// arguments: no staging copy, so the per-call API neither copies nor synchronises - the call returns once the
// launch is queued, and the state is read back (and the stream drained) only when the caller asks for it.
struct EkfRow7 { double v[7]; };
struct EkfPoseArg { double pose[16]; double cov[36]; int has_cov; };
__global__ __launch_bounds__(384) void k_ekf_imu_value(EkfState* e, EkfRow7 row) {
    __shared__ double srow[7];
    if (threadIdx.x < 7) srow[threadIdx.x] = row.v[threadIdx.x];
    __syncthreads();
    d_ekf_step(e, srow, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
}
__global__ __launch_bounds__(384) void k_ekf_pose_value(EkfState* e, EkfPoseArg a) {
    __shared__ double sp[16 + 36];
    if (threadIdx.x < 16) sp[threadIdx.x] = a.pose[threadIdx.x];
    if (threadIdx.x < 36) sp[16 + threadIdx.x] = a.cov[threadIdx.x];
    __syncthreads();
    d_ekf_step(e, nullptr, 0, 0, sp, a.has_cov ? sp + 16 : nullptr, nullptr, nullptr, nullptr, 0);
}
// S filters in one launch (blockIdx.x = sequence)
#define EKF_MAX_SEQ 64
struct EkfBatchArgs {
    EkfState* e[EKF_MAX_SEQ];
    const double* imu[EKF_MAX_SEQ];
    int i0[EKF_MAX_SEQ], i1[EKF_MAX_SEQ];
    const double* pose[EKF_MAX_SEQ];
    double* out_pose[EKF_MAX_SEQ];
    double* out_t[EKF_MAX_SEQ];
    double* out_row8[EKF_MAX_SEQ];
    int update_first;
};
__global__ __launch_bounds__(384) void kb_ekf_step(EkfBatchArgs a) {
    const int s = blockIdx.x;
    d_ekf_step(a.e[s], a.imu[s], a.i0[s], a.i1[s], a.pose[s], nullptr, a.out_pose[s], a.out_t[s], a.out_row8[s], a.update_first);
}
