/*
 * oracle_math.h -- small fixed-size linear algebra for the CPU oracle (test infrastructure only).
 * Plain C, fp64 throughout (the reference path is fp64 end to end).
 */
#ifndef PTL_ORACLE_MATH_H
#define PTL_ORACLE_MATH_H

#include <math.h>
#include <string.h>

static inline void m3_mul(const double A[9], const double B[9], double C[9]) {
    double r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            r[3 * i + j] = s;
        }
    memcpy(C, r, sizeof r);
}
static inline void m3_T(const double A[9], double C[9]) {
    double r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[3 * i + j] = A[3 * j + i];
    memcpy(C, r, sizeof r);
}
static inline void m3_vec(const double A[9], const double v[3], double o[3]) {
    double r[3];
    for (int i = 0; i < 3; ++i) r[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
/* hat(v): the reference's `vee` (utils.py:28-36) builds exactly this skew matrix */
static inline void hat3(const double v[3], double H[9]) {
    H[0] = 0.0;   H[1] = -v[2]; H[2] = v[1];
    H[3] = v[2];  H[4] = 0.0;   H[5] = -v[0];
    H[6] = -v[1]; H[7] = v[0];  H[8] = 0.0;
}
static inline double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

static inline void T_get_R(const double T[16], double R[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = T[4 * i + j];
}
static inline void T_set(double T[16], const double R[9], const double t[3]) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = t[i];
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}
static inline void T_identity(double T[16]) {
    memset(T, 0, 16 * sizeof(double));
    T[0] = T[5] = T[10] = T[15] = 1.0;
}
static inline void T_apply(const double T[16], const double p[3], double o[3]) {
    double x = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
    double y = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
    double z = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
    o[0] = x; o[1] = y; o[2] = z;
}

/* ---- SO(3): rotation vector <-> matrix through a unit quaternion, the way scipy's Rotation does it
 * (the reference's EKF goes through scipy for every attitude access, ins/data.py:76-90). ---- */

/* Rotation.from_rotvec: quaternion xyzw */
static inline void rotvec_to_quat(const double v[3], double q[4]) {
    double angle = norm3(v), scale;
    if (angle <= 1e-3) {
        double a2 = angle * angle;
        scale = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
    } else {
        scale = sin(angle / 2.0) / angle;
    }
    q[0] = scale * v[0]; q[1] = scale * v[1]; q[2] = scale * v[2];
    q[3] = cos(angle / 2.0);
}
/* Rotation.as_matrix for a unit quaternion */
static inline void quat_to_mat(const double q[4], double R[9]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    R[0] = x2 - y2 - z2 + w2; R[1] = 2.0 * (xy - zw);    R[2] = 2.0 * (xz + yw);
    R[3] = 2.0 * (xy + zw);   R[4] = -x2 + y2 - z2 + w2; R[5] = 2.0 * (yz - xw);
    R[6] = 2.0 * (xz - yw);   R[7] = 2.0 * (yz + xw);    R[8] = -x2 - y2 + z2 + w2;
}
static inline void quat_normalize(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
/* Rotation.from_matrix -> as_quat: largest of {m00, m11, m22, trace} picks the branch, then normalise */
static inline void mat_to_quat(const double R[9], double q[4]) {
    double d[4] = {R[0], R[4], R[8], R[0] + R[4] + R[8]};
    int c = 0;
    for (int i = 1; i < 4; ++i)
        if (d[i] > d[c]) c = i;
    if (c != 3) {
        int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1.0 - d[3] + 2.0 * R[3 * i + i];
        q[j] = R[3 * j + i] + R[3 * i + j];
        q[k] = R[3 * k + i] + R[3 * i + k];
        q[3] = R[3 * k + j] - R[3 * j + k];
    } else {
        q[0] = R[7] - R[5];
        q[1] = R[2] - R[6];
        q[2] = R[3] - R[1];
        q[3] = 1.0 + d[3];
    }
    quat_normalize(q);
}
/* Rotation.as_rotvec */
static inline void quat_to_rotvec(const double qin[4], double v[3]) {
    double q[4] = {qin[0], qin[1], qin[2], qin[3]};
    if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    double angle = 2.0 * atan2(norm3(q), q[3]), scale;
    if (angle <= 1e-3) {
        double a2 = angle * angle;
        scale = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
    } else {
        scale = angle / sin(angle / 2.0);
    }
    v[0] = scale * q[0]; v[1] = scale * q[1]; v[2] = scale * q[2];
}
static inline void so3_exp(const double v[3], double R[9]) {
    double q[4];
    rotvec_to_quat(v, q);
    quat_to_mat(q, R);
}
static inline void so3_log(const double R[9], double v[3]) {
    double q[4];
    mat_to_quat(R, q);
    quat_to_rotvec(q, v);
}

/* np.linalg.inv of a general 4x4 and a plain 4x4 product: the reference forms the innovation as
 * np.linalg.inv(initial_guess) @ new_pose on the caller's RAW guess (kiss.py:116, :128), whatever its
 * rotation block and last row look like.  Gauss-Jordan with partial pivoting (first largest pivot);
 * returns 0 for a singular matrix (the identity is written then). */
static inline int m4_inv(const double A[16], double out[16]) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = A[4 * i + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) {
            for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.0 : 0.0;
            return 0;
        }
        if (piv != col)
            for (int j = 0; j < 8; ++j) { double t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
        const double p = a[col][col];
        for (int j = 0; j < 8; ++j) a[col][j] = a[col][j] / p;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[col][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = a[i][4 + j];
    return 1;
}
/* nearest rotation to a 3x3 with positive determinant - the orthogonal polar factor U V^T, which is what scipy's
 * Rotation.from_matrix makes of a matrix that is not orthogonal (SVD) before kiss.py:119 takes its rotation vector.  Newton's
 * iteration X <- (X + X^-T) / 2 (Higham), X^-T = cof(X) / det(X); eight steps (quadratic convergence: exact to rounding from
 * anything within a few per cent of a rotation; an exact rotation is a fixed point). */
static inline void m3_polar(const double A[9], double Q[9]) {
    double X[9];
    memcpy(X, A, sizeof X);
    for (int it = 0; it < 8; ++it) {
        double C[9];
        C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
        C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
        C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
        if (!(det > 0.0)) break;
        for (int i = 0; i < 9; ++i) X[i] = 0.5 * (X[i] + C[i] / det);
    }
    memcpy(Q, X, sizeof X);
}
static inline void m4_mul(const double A[16], const double B[16], double C[16]) {
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(C, r, sizeof r);
}

#endif
