/*
 * oracle_ekf.c -- CPU restatement of the reference's 18-state error-state EKF and its metric.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Follows /root/reference/src/ptudes/ins/es_ekf.py and
 * ins/data.py line by line in meaning, not in code: plain C, dense loops, fp64.
 * PINNED against tests/golden/ekf_steps_*.npz and ekf_sim.npz (outputs of the reference itself).
 */
#include "oracle.h"
#include "oracle_math.h"

#include <stdlib.h>

#define NS 18
enum { POS = 0, VEL = 3, PHI = 6, BG = 9, BA = 12, GRV = 15 }; /* es_ekf.py:65-71 */

static const double GRAV = 9.782940329221166; /* ins/data.py:10 */

struct orc_ekf {
    /* NavState (ins/data.py:34-43): attitude is STORED as an xyzw quaternion, every matrix access
     * converts through it (ins/data.py:76-82) */
    double pos[3], q[4], vel[3], bg[3], ba[3], grav[3];
    double P[NS * NS];  /* _cov */
    double Fx[NS * NS]; /* persistent, identity + rewritten blocks (es_ekf.py:142, 216-223) */
    double W[NS * NS];  /* persistent, zero + rewritten blocks (es_ekf.py:145, 226-233) */
    double cur_lacc[3], cur_avel[3], cur_ts, cur_dt; /* _imu_curr */
    int initialized;                                 /* _imu_initialized */
};

static void set_blk3(double *M, int r, int c, const double B[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(r + i) * NS + c + j] = B[3 * i + j];
}
static void set_diag3(double *M, int r, int c, double v) {
    double B[9] = {v, 0, 0, 0, v, 0, 0, 0, v};
    set_blk3(M, r, c, B);
}

orc_ekf *orc_ekf_create(const double *init_grav, const double *init_bacc, const double *init_bgyr) {
    orc_ekf *e = (orc_ekf *)calloc(1, sizeof *e);
    if (!e) return NULL;
    e->q[3] = 1.0; /* att_v = 0 (es_ekf.py:153) */
    for (int i = 0; i < 3; ++i) {
        e->grav[i] = init_grav ? init_grav[i] : (i == 2 ? -GRAV : 0.0); /* GRAV * DOWN, es_ekf.py:75 */
        e->ba[i] = init_bacc ? init_bacc[i] : 0.0;
        e->bg[i] = init_bgyr ? init_bgyr[i] : 0.0;
    }
    /* initial covariance, es_ekf.py:98-137 */
    set_diag3(e->P, POS, POS, 10.0 * 10.0);
    set_diag3(e->P, VEL, VEL, 5.0 * 5.0);
    {
        /* rotvec of intrinsic-XYZ euler (10,10,10) deg => R = Rx Ry Rz (es_ekf.py:104-107) */
        double a = 10.0 * M_PI / 180.0;
        double c = cos(a), s = sin(a);
        double Rx[9] = {1, 0, 0, 0, c, -s, 0, s, c};
        double Ry[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
        double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
        double R[9], rv[3];
        m3_mul(Rx, Ry, R);
        m3_mul(R, Rz, R);
        so3_log(R, rv);
        for (int i = 0; i < 3; ++i) e->P[(PHI + i) * NS + PHI + i] = rv[i] * rv[i];
    }
    set_diag3(e->P, BG, BG, 1.5 * 1.5);
    set_diag3(e->P, BA, BA, 0.5 * 0.5);
    set_diag3(e->P, GRV, GRV, 2.5 * 2.5);
    for (int i = 0; i < NS; ++i) e->Fx[i * NS + i] = 1.0;
    return e;
}

void orc_ekf_destroy(orc_ekf *e) { free(e); }

/* es_ekf.py:239-257 */
static void ins_mech(orc_ekf *e) {
    double a[3], w[3], R[9], aw[3], dth[3], Rd[9], Rn[9];
    double dt = e->cur_dt;
    for (int i = 0; i < 3; ++i) {
        a[i] = e->cur_lacc[i] - e->ba[i];
        w[i] = e->cur_avel[i] - e->bg[i];
        dth[i] = w[i] * dt;
    }
    quat_to_mat(e->q, R);
    m3_vec(R, a, aw);
    so3_exp(dth, Rd);
    for (int i = 0; i < 3; ++i) {
        double ag = aw[i] + e->grav[i];
        e->pos[i] = e->pos[i] + e->vel[i] * dt + 0.5 * ag * dt * dt;
        e->vel[i] = e->vel[i] + ag * dt;
    }
    m3_mul(R, Rd, Rn);
    mat_to_quat(Rn, e->q); /* att_h setter renormalises through a quaternion, ins/data.py:80-82 */
}

/* es_ekf.py:191-237 */
void orc_ekf_process_imu(orc_ekf *e, const double lacc[3], const double avel[3], double ts) {
    double prev_ts = e->cur_ts;
    e->cur_dt = ts - prev_ts; /* :196, overrides any packet dt */
    e->cur_ts = ts;
    memcpy(e->cur_lacc, lacc, sizeof e->cur_lacc);
    memcpy(e->cur_avel, avel, sizeof e->cur_avel);
    if (!e->initialized) { /* :201-203, first sample only latches */
        e->initialized = 1;
        return;
    }
    double Rprev[9];
    quat_to_mat(e->q, Rprev); /* nav_prev.att_h */
    ins_mech(e);

    double dt = e->cur_dt, ab[3], dth[3], Rd[9], RdT[9], H[9], B[9];
    for (int i = 0; i < 3; ++i) {
        ab[i] = e->cur_lacc[i] - e->ba[i];
        dth[i] = (e->cur_avel[i] - e->bg[i]) * dt;
    }
    so3_exp(dth, Rd);
    m3_T(Rd, RdT);
    set_diag3(e->Fx, POS, VEL, dt);       /* :216 */
    hat3(ab, H);
    m3_mul(Rprev, H, B);
    for (int i = 0; i < 9; ++i) B[i] = -dt * B[i];
    set_blk3(e->Fx, VEL, PHI, B);          /* :217 */
    for (int i = 0; i < 9; ++i) B[i] = -dt * Rprev[i];
    set_blk3(e->Fx, VEL, BA, B);           /* :218 */
    set_blk3(e->Fx, PHI, PHI, RdT);        /* :222 */
    set_diag3(e->Fx, PHI, BG, -dt);        /* :223 */
    /* noise: the names in the reference do not match their use; this copies the USE (:226-233) */
    set_diag3(e->W, VEL, VEL, dt * dt * (0.049 * 0.049));
    set_diag3(e->W, PHI, PHI, dt * dt * (0.38 * 0.38));
    set_diag3(e->W, BA, BA, dt * (0.0043 * 0.0043));
    set_diag3(e->W, BG, BG, dt * (0.000466 * 0.000466));

    /* P = Fx P Fx^T + W  (:235) */
    double FP[NS * NS];
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
            for (int k = 0; k < NS; ++k) s += e->Fx[i * NS + k] * e->P[k * NS + j];
            FP[i * NS + j] = s;
        }
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
            for (int k = 0; k < NS; ++k) s += FP[i * NS + k] * e->Fx[j * NS + k];
            e->P[i * NS + j] = s + e->W[i * NS + j];
        }
}

/* general n x n inverse by Gauss-Jordan with partial pivoting (np.linalg.inv stand-in, es_ekf.py:300) */
static int inv_n(const double *A, double *Ai, int n) {
    double M[6 * 12];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            M[i * 2 * n + j] = A[i * n + j];
            M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r)
            if (fabs(M[r * 2 * n + c]) > fabs(M[p * 2 * n + c])) p = r;
        if (M[p * 2 * n + c] == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < 2 * n; ++j) {
                double t = M[c * 2 * n + j];
                M[c * 2 * n + j] = M[p * 2 * n + j];
                M[p * 2 * n + j] = t;
            }
        double d = M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) M[c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Ai[i * n + j] = M[i * 2 * n + n + j];
    return 0;
}

/* es_ekf.py:259-329.  The error state is zero on entry (it is reset at :327 after every update and
 * nothing else writes it), so dR = I and the dpos term vanishes (:276, :294, :297). */
void orc_ekf_process_pose(orc_ekf *e, const double pose[16], const double *meas_cov36) {
    double Rm[36];
    if (meas_cov36) {
        memcpy(Rm, meas_cov36, sizeof Rm);
    } else { /* :289-292 */
        memset(Rm, 0, sizeof Rm);
        for (int i = 0; i < 3; ++i) {
            Rm[i * 6 + i] = 0.02 * 0.02;
            Rm[(3 + i) * 6 + 3 + i] = 0.01 * 0.01;
        }
    }
    /* Jp rows: 0..2 select POS, 3..5 select PHI (:284-286) */
    const int sel[6] = {POS, POS + 1, POS + 2, PHI, PHI + 1, PHI + 2};
    double r[6], Rk[9], RkT[9], Rmeas[9], dRm[9];
    for (int i = 0; i < 3; ++i) r[i] = pose[4 * i + 3] - e->pos[i]; /* :294 */
    quat_to_mat(e->q, Rk);
    m3_T(Rk, RkT);
    T_get_R(pose, Rmeas);
    m3_mul(RkT, Rmeas, dRm);
    so3_log(dRm, r + 3); /* :297 */

    double S[36], Si[36], K[NS * 6], dx[NS];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) S[i * 6 + j] = e->P[sel[i] * NS + sel[j]] + Rm[i * 6 + j]; /* :299 */
    if (inv_n(S, Si, 6) != 0) return;
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += e->P[i * NS + sel[k]] * Si[k * 6 + j];
            K[i * 6 + j] = s; /* :300 */
        }
    for (int i = 0; i < NS; ++i) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += K[i * 6 + k] * r[k];
        dx[i] = s; /* :301 */
    }
    /* P = (I - K Jp) P  (:303, non-Joseph form) */
    double IKH[NS * NS], Pn[NS * NS];
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) IKH[i * NS + j] = (i == j) ? 1.0 : 0.0;
    for (int i = 0; i < NS; ++i)
        for (int k = 0; k < 6; ++k) IKH[i * NS + sel[k]] -= K[i * 6 + k];
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
            for (int k = 0; k < NS; ++k) s += IKH[i * NS + k] * e->P[k * NS + j];
            Pn[i * NS + j] = s;
        }
    memcpy(e->P, Pn, sizeof Pn);

    /* inject (:314-319) */
    double dth[3], Rd[9], R[9], Rn[9];
    for (int i = 0; i < 3; ++i) {
        e->pos[i] += dx[POS + i];
        e->vel[i] += dx[VEL + i];
        dth[i] = dx[PHI + i];
        e->bg[i] += dx[BG + i];
        e->ba[i] += dx[BA + i];
        e->grav[i] += dx[GRV + i];
    }
    quat_to_mat(e->q, R);
    so3_exp(dth, Rd);
    m3_mul(R, Rd, Rn);
    mat_to_quat(Rn, e->q);

    /* covariance projection of the PHI diagonal block only (:322-324) */
    double half[3] = {0.5 * dth[0], 0.5 * dth[1], 0.5 * dth[2]}, G[9], GT[9], B[9], C[9];
    hat3(half, G);
    for (int i = 0; i < 9; ++i) G[i] = -G[i];
    G[0] += 1.0; G[4] += 1.0; G[8] += 1.0;
    m3_T(G, GT);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) B[3 * i + j] = e->P[(PHI + i) * NS + PHI + j];
    m3_mul(G, B, C);
    m3_mul(C, GT, C);
    set_blk3(e->P, PHI, PHI, C);
}

void orc_ekf_get_nav(const orc_ekf *e, double nav[19]) {
    memcpy(nav + 0, e->pos, 24);
    memcpy(nav + 3, e->q, 32);
    memcpy(nav + 7, e->vel, 24);
    memcpy(nav + 10, e->bg, 24);
    memcpy(nav + 13, e->ba, 24);
    memcpy(nav + 16, e->grav, 24);
}
void orc_ekf_get_cov(const orc_ekf *e, double cov[324]) { memcpy(cov, e->P, sizeof e->P); }
void orc_ekf_pose_mat(const orc_ekf *e, double T[16]) {
    double R[9];
    quat_to_mat(e->q, R);
    T_set(T, R, e->pos);
}
double orc_ekf_ts(const orc_ekf *e) { return e->cur_ts; }

/* ins/data.py:124-153.  "ATE" here is a MEAN OF SQUARES (m^2; rad^2 * 180/pi), as the reference prints. */
int orc_calc_ate(const double *nav, const double *gt, int64_t n, double ate[2]) {
    if (n <= 0) return -1;
    double g0i[16], A[16];
    orc_se3_inv(gt, g0i);
    orc_se3_mul(nav, g0i, A); /* pose0_inv = nav[0] @ inv(gt[0]) */
    double st = 0.0, sr = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double G[16], Rn[9], RnT[9], Rg[9], D[9], rv[3], d[3];
        orc_se3_mul(A, gt + 16 * i, G);
        for (int k = 0; k < 3; ++k) d[k] = G[4 * k + 3] - nav[16 * i + 4 * k + 3];
        double td = norm3(d);
        T_get_R(nav + 16 * i, Rn);
        m3_T(Rn, RnT);
        T_get_R(G, Rg);
        m3_mul(RnT, Rg, D);
        so3_log(D, rv);
        double rd = norm3(rv);
        st += td * td;
        sr += rd * rd;
    }
    ate[1] = st / (double)n;
    ate[0] = (sr / (double)n) * 180.0 / M_PI;
    return 0;
}
