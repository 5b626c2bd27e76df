// devmath.h -- small fixed-size fp64 algebra used by the HIP kernels (SE(3)/SO(3), 6x6 solve).
// Product code: written independently of the CPU oracle (oracle/ is test infrastructure and is never
// included from here).  The reference path is fp64 end to end, so is this.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define PTL_HD __host__ __device__ __forceinline__

struct V3 {
    double x, y, z;
};
PTL_HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
PTL_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PTL_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PTL_HD V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
PTL_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PTL_HD double norm(V3 a) { return sqrt(dot(a, a)); }

// rigid transform: R row-major 3x3, t
struct Rt {
    double R[9];
    double t[3];
};
PTL_HD Rt rt_identity() {
    Rt a;
    for (int i = 0; i < 9; ++i) a.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    a.t[0] = a.t[1] = a.t[2] = 0.0;
    return a;
}
PTL_HD Rt rt_from16(const double* T) {
    Rt a;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) a.R[3 * i + j] = T[4 * i + j];
        a.t[i] = T[4 * i + 3];
    }
    return a;
}
PTL_HD void rt_to16(const Rt& a, double* T) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = a.R[3 * i + j];
        T[4 * i + 3] = a.t[i];
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}
PTL_HD V3 rt_apply(const Rt& a, V3 p) {
    return V3{a.R[0] * p.x + a.R[1] * p.y + a.R[2] * p.z + a.t[0], a.R[3] * p.x + a.R[4] * p.y + a.R[5] * p.z + a.t[1],
              a.R[6] * p.x + a.R[7] * p.y + a.R[8] * p.z + a.t[2]};
}
PTL_HD Rt rt_mul(const Rt& a, const Rt& b) {
    Rt c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += a.R[3 * i + k] * b.R[3 * k + j];
            c.R[3 * i + j] = s;
        }
        c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    return c;
}
PTL_HD Rt rt_inv(const Rt& a) {
    Rt c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * j + i];
    for (int i = 0; i < 3; ++i) c.t[i] = -(c.R[3 * i] * a.t[0] + c.R[3 * i + 1] * a.t[1] + c.R[3 * i + 2] * a.t[2]);
    return c;
}

// np.linalg.inv of a general 4x4 (row-major) and a plain 4x4 product: the reference forms the innovation of a registration as
// np.linalg.inv(initial_guess) @ new_pose on the caller's RAW guess (kiss.py:116, :128).  Gauss-Jordan with partial pivoting
// (the first largest pivot); every index is a compile-time constant after unrolling, rows are exchanged by selects - nothing
// here may end up in scratch memory (it is inlined into the tail of the Gauss-Newton kernels).  A singular matrix gives the
// identity and false.
PTL_HD bool mat4_inv(const double* A, double* out) {
    double a[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[i][j] = A[4 * i + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    bool ok = true;
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double pv = fabs(a[col][col]);
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {
            const double v = fabs(a[r][col]);
            if (v > pv) { pv = v; piv = r; }
        }
        if (pv == 0.0) ok = false;
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {
            const bool sw = piv == r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double x = a[col][j], y = a[r][j];
                a[col][j] = sw ? y : x;
                a[r][j] = sw ? x : y;
            }
        }
        const double p = a[col][col];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[col][j] = a[col][j] / p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[col][j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * i + j] = ok ? a[i][4 + j] : ((i == j) ? 1.0 : 0.0);
    return ok;
}
// nearest rotation to a 3x3 with positive determinant: the orthogonal polar factor U V^T - what scipy's Rotation.from_matrix makes of a
// matrix that is not orthogonal before kiss.py:119 takes its rotation vector.  Newton's iteration X <- (X + X^-T) / 2, eight steps.
PTL_HD void mat3_polar(const double* A, double* Q) {
    double X[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) X[i] = A[i];
    for (int it = 0; it < 8; ++it) {
        double C[9];
        C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
        C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
        C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
        if (!(det > 0.0)) break;
#pragma unroll
        for (int i = 0; i < 9; ++i) X[i] = 0.5 * (X[i] + C[i] / det);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Q[i] = X[i];
}
PTL_HD void mat4_mul(const double* A, const double* B, double* C) {
    double r[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            r[4 * i + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) C[i] = r[i];
}

PTL_HD void skew(const double w[3], double K[9]) {
    K[0] = 0.0;   K[1] = -w[2]; K[2] = w[1];
    K[3] = w[2];  K[4] = 0.0;   K[5] = -w[0];
    K[6] = -w[1]; K[7] = w[0];  K[8] = 0.0;
}
PTL_HD void mat3_mul(const double* A, const double* B, double* C) {
    double r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    for (int i = 0; i < 9; ++i) C[i] = r[i];
}

// ---- SO(3) through a unit quaternion (xyzw), the path scipy's Rotation takes for the reference EKF
PTL_HD void quat_to_R(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double xx = x * x, yy = y * y, zz = z * z, ww = w * w;
    R[0] = xx - yy - zz + ww; R[1] = 2.0 * (x * y - z * w); R[2] = 2.0 * (x * z + y * w);
    R[3] = 2.0 * (x * y + z * w); R[4] = -xx + yy - zz + ww; R[5] = 2.0 * (y * z - x * w);
    R[6] = 2.0 * (x * z - y * w); R[7] = 2.0 * (y * z + x * w); R[8] = -xx - yy + zz + ww;
}
// scipy's Rotation.from_matrix(...).as_quat(): the largest of the diagonal entries and the trace picks the formula.
// Written out per case with constant indices (the compact form with i, j, k = c, c+1, c+2 mod 3 indexes R and q by
// run-time values, which puts both on scratch memory - in the EKF's per-sample mechanisation); same sums, same order.
PTL_HD void R_to_quat(const double R[9], double q[4]) {
    const double tr = R[0] + R[4] + R[8];
    int c = 0;
    double best = R[0];
    if (R[4] > best) { c = 1; best = R[4]; }
    if (R[8] > best) { c = 2; best = R[8]; }
    if (tr > best) c = 3;
    if (c == 3) {
        q[0] = R[7] - R[5]; q[1] = R[2] - R[6]; q[2] = R[3] - R[1]; q[3] = 1.0 + tr;
    } else if (c == 0) {
        q[0] = 1.0 - tr + 2.0 * R[0]; q[1] = R[3] + R[1]; q[2] = R[6] + R[2]; q[3] = R[7] - R[5];
    } else if (c == 1) {
        q[1] = 1.0 - tr + 2.0 * R[4]; q[2] = R[7] + R[5]; q[0] = R[1] + R[3]; q[3] = R[2] - R[6];
    } else {
        q[2] = 1.0 - tr + 2.0 * R[8]; q[0] = R[2] + R[6]; q[1] = R[5] + R[7]; q[3] = R[3] - R[1];
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
// The poses of kiss-icp are Sophus::SE3d (unit quaternion + translation): the guess becomes one when it enters the
// registration, the result is the matrix of one - every registration re-orthonormalises the rotation.  On bare matrices
// that has to be done explicitly (see oracle/oracle_icp.c so3_project16): without it the constant-velocity recursion
// amplifies rounding-level non-orthogonality by ~2.4x per sweep and the track is lost around sweep 35-40.
PTL_HD Rt rt_project(const Rt& a) {
    double q[4];
    Rt r = a;
    R_to_quat(a.R, q);
    quat_to_R(q, r.R);
    return r;
}
PTL_HD void rotvec_to_R(const double v[3], double R[9]) {
    const double a = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double s;
    if (a <= 1e-3) {
        const double a2 = a * a;
        s = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
    } else {
        s = sin(0.5 * a) / a;
    }
    const double q[4] = {s * v[0], s * v[1], s * v[2], cos(0.5 * a)};
    quat_to_R(q, R);
}
PTL_HD void R_to_rotvec(const double R[9], double v[3]) {
    double q[4];
    R_to_quat(R, q);
    if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double a = 2.0 * atan2(sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), q[3]);
    double s;
    if (a <= 1e-3) {
        const double a2 = a * a;
        s = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
    } else {
        s = a / sin(0.5 * a);
    }
    v[0] = s * q[0]; v[1] = s * q[1]; v[2] = s * q[2];
}

// rotation angle in [0, pi] of a rotation matrix (atan2 form: accurate near zero)
PTL_HD double rot_angle(const double R[9]) {
    const double ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1];
    return atan2(0.5 * sqrt(ax * ax + ay * ay + az * az), 0.5 * (R[0] + R[4] + R[8] - 1.0));
}

// SE(3) exp of xi = (upsilon, omega) [Sophus order]: R = I + a K + b K^2, t = (I + b K + c K^2) upsilon
PTL_HD Rt se3_exp(const double xi[6]) {
    const double* w = xi + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double a, b, c;
    if (th < 1e-6) {
        a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0;
    } else {
        const double sn = sin(th), cs = cos(th);
        a = sn / th; b = (1.0 - cs) / th2; c = (th - sn) / (th2 * th);
    }
    double K[9], K2[9];
    skew(w, K);
    mat3_mul(K, K, K2);
    Rt o;
    double V[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        o.R[i] = I + a * K[i] + b * K2[i];
        V[i] = I + b * K[i] + c * K2[i];
    }
    for (int i = 0; i < 3; ++i) o.t[i] = V[3 * i] * xi[0] + V[3 * i + 1] * xi[1] + V[3 * i + 2] * xi[2];
    return o;
}
// se3_exp for the Gauss-Newton increments.  Below 0.05 rad the coefficients come from their Maclaurin series
// (six terms: truncation < 1e-20, and none of the cancellation the closed forms (1 - cos) / th^2, (th - sin) / th^3
// suffer at small angles) - no sin / cos / three divisions on the serial tail of every iteration.  The result
// differs from se3_exp() by rounding only (<= 1e-16 in R and t for the increments a registration produces).
PTL_HD Rt se3_exp_gn(const double xi[6]) {
    const double* w = xi + 3;
    const double t = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (!(t < 2.5e-3)) return se3_exp(xi);
    const double a = 1.0 + t * (-1.0 / 6.0 + t * (1.0 / 120.0 + t * (-1.0 / 5040.0 + t * (1.0 / 362880.0 + t * (-1.0 / 39916800.0)))));
    const double b = 0.5 + t * (-1.0 / 24.0 + t * (1.0 / 720.0 + t * (-1.0 / 40320.0 + t * (1.0 / 3628800.0 + t * (-1.0 / 479001600.0)))));
    const double c = 1.0 / 6.0 + t * (-1.0 / 120.0 + t * (1.0 / 5040.0 + t * (-1.0 / 362880.0 + t * (1.0 / 39916800.0 + t * (-1.0 / 6227020800.0)))));
    double K[9], K2[9];
    skew(w, K);
    mat3_mul(K, K, K2);
    Rt o;
    double V[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        o.R[i] = I + a * K[i] + b * K2[i];
        V[i] = I + b * K[i] + c * K2[i];
    }
    for (int i = 0; i < 3; ++i) o.t[i] = V[3 * i] * xi[0] + V[3 * i + 1] * xi[1] + V[3 * i + 2] * xi[2];
    return o;
}
// SE(3) log
PTL_HD void se3_log(const Rt& T, double xi[6]) {
    double w[3];
    R_to_rotvec(T.R, w);
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double k;
    if (th < 1e-6) {
        k = 1.0 / 12.0 + th2 / 720.0;
    } else {
        const double h = 0.5 * th;
        k = (1.0 - th * cos(h) / (2.0 * sin(h))) / th2;
    }
    double K[9], K2[9];
    skew(w, K);
    mat3_mul(K, K, K2);
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0;
            s += (I - 0.5 * K[3 * i + j] + k * K2[3 * i + j]) * T.t[j];
        }
        xi[i] = s;
    }
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

// Solve JTJ dx = -JTr from the 27 packed sums (21 upper-triangle JTJ row-major, then 6 JTr) by an
// unpivoted LDL^T; a zero pivot yields a zero component (all-zero system of a scan with no pairs).
PTL_HD void solve6_ldlt(const double* s, double dx[6]) {
    // fully unrolled so that every array stays in registers on the device (no scratch traffic)
    double A[36], L[36], D[6], y[6];
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) {
            A[6 * i + j] = s[o];
            A[6 * j + i] = s[o];
            ++o;
        }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < j) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        D[j] = d;
        L[6 * j + j] = 1.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i > j) {
                double v = A[6 * i + j];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k < j) v -= L[6 * i + k] * L[6 * j + k] * D[k];
                L[6 * i + j] = (d != 0.0) ? v / d : 0.0;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = -s[21 + i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < i) v -= L[6 * i + k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > i) v -= L[6 * k + i] * dx[k];
        dx[i] = v;
    }
}
