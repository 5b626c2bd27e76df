/*
 * oracle_icp.c -- CPU restatement of the scan-to-local-map registration the reference drives at
 * /root/reference/src/ptudes/kiss.py:83-131.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The arithmetic lives in the third-party package kiss-icp (reference setup.py:23; effective pin
 * v0.2.10, see oracle.h).  It is absent from /root/reference and from this image, so the published
 * v0.2.10 algorithm is restated here; each function names the upstream unit it follows.
 * PARITY UNPINNED: no golden vector exists for this part; known-answer tests in tests/ pin it
 * algebraically (exact-GT scenes, identity when scan == map, first scan returns the guess).
 *
 * Order-dependent semantics, made deterministic (upstream's own orders depend on tsl::robin_map
 * iteration order and TBB scheduling, i.e. are unspecified):
 *   - voxel downsample keeps the FIRST point per voxel in input order; output is in input order;
 *   - a map voxel keeps its first max_points_per_voxel points in insertion (= input) order;
 *   - nearest neighbour: candidates visited in voxel order (i, j, k ascending) then insertion order,
 *     a candidate replaces the best only when STRICTLY closer;
 *   - voxel index = (int)(coordinate / voxel_size), C truncation toward zero (upstream casts an
 *     Eigen::Vector3d to int the same way), so voxels touching a zero coordinate are double width;
 *   - sums are accumulated in source-point order.
 */
#include "oracle.h"
#include "oracle_math.h"

#include <stdint.h>
#include <stdlib.h>

/* ================================================================= SE(3) (Sophus::SE3d semantics) */

void orc_se3_inv(const double T[16], double Ti[16]) {
    double R[9], Rt[9], t[3] = {T[3], T[7], T[11]}, ti[3];
    T_get_R(T, R);
    m3_T(R, Rt);
    m3_vec(Rt, t, ti);
    ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
    T_set(Ti, Rt, ti);
}
void orc_se3_mul(const double A[16], const double B[16], double C[16]) {
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(C, r, sizeof r);
}
/* rotation angle in [0, pi]: atan2(|axis part|, cos) is accurate for small angles */
double orc_rot_angle(const double T[16]) {
    double ax = T[9] - T[6], ay = T[2] - T[8], az = T[4] - T[1];
    double s = 0.5 * sqrt(ax * ax + ay * ay + az * az);
    double c = 0.5 * (T[0] + T[5] + T[10] - 1.0);
    return atan2(s, c);
}
/* SE3::exp: xi = (upsilon, omega); R = Exp(omega), t = V(omega) upsilon */
void orc_se3_exp(const double xi[6], double T[16]) {
    const double *u = xi, *w = xi + 3;
    double th = norm3(w), R[9], Om[9], Om2[9], V[9], t[3];
    hat3(w, Om);
    m3_mul(Om, Om, Om2);
    double a, b, c; /* R = I + a Om + b Om2 ; V = I + b Om + c Om2 */
    if (th < 1e-6) {
        double t2 = th * th;
        a = 1.0 - t2 / 6.0;
        b = 0.5 - t2 / 24.0;
        c = 1.0 / 6.0 - t2 / 120.0;
    } else {
        double t2 = th * th;
        a = sin(th) / th;
        b = (1.0 - cos(th)) / t2;
        c = (th - sin(th)) / (t2 * th);
    }
    for (int i = 0; i < 9; ++i) {
        double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * Om[i] + b * Om2[i];
        V[i] = I + b * Om[i] + c * Om2[i];
    }
    m3_vec(V, u, t);
    T_set(T, R, t);
}
/* SE3::log */
void orc_se3_log(const double T[16], double xi[6]) {
    double R[9], w[3], Om[9], Om2[9], Vi[9], t[3] = {T[3], T[7], T[11]};
    T_get_R(T, R);
    so3_log(R, w);
    double th = norm3(w);
    hat3(w, Om);
    m3_mul(Om, Om, Om2);
    double k;
    if (th < 1e-6) {
        k = 1.0 / 12.0 + th * th / 720.0;
    } else {
        double h = 0.5 * th;
        k = (1.0 - th * cos(h) / (2.0 * sin(h))) / (th * th);
    }
    for (int i = 0; i < 9; ++i) {
        double I = (i % 4 == 0) ? 1.0 : 0.0;
        Vi[i] = I - 0.5 * Om[i] + k * Om2[i];
    }
    m3_vec(Vi, t, xi);
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

/* ================================================================= voxel hash containers */

typedef struct { int32_t x, y, z; } vox_t;

static inline vox_t vox_of(const double p[3], double size) {
    vox_t v = {(int32_t)(p[0] / size), (int32_t)(p[1] / size), (int32_t)(p[2] / size)};
    return v;
}
static inline uint64_t vox_hash(vox_t v) {
    uint64_t h = (uint64_t)(uint32_t)v.x * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)(uint32_t)v.y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= (uint64_t)(uint32_t)v.z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}
static inline int vox_eq(vox_t a, vox_t b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

typedef struct {
    vox_t *keys;
    int32_t *vals; /* -1 = empty */
    int64_t cap, used;
} vtab;

static void vtab_init(vtab *t, int64_t expect) {
    int64_t cap = 64;
    while (cap < 2 * expect + 2) cap <<= 1;
    t->cap = cap;
    t->used = 0;
    t->keys = (vox_t *)malloc((size_t)cap * sizeof(vox_t));
    t->vals = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    for (int64_t i = 0; i < cap; ++i) t->vals[i] = -1;
}
static void vtab_free(vtab *t) {
    free(t->keys);
    free(t->vals);
    t->keys = NULL;
    t->vals = NULL;
}
static inline int32_t vtab_find(const vtab *t, vox_t k) {
    int64_t m = t->cap - 1, i = (int64_t)(vox_hash(k) & (uint64_t)m);
    while (t->vals[i] >= 0) {
        if (vox_eq(t->keys[i], k)) return t->vals[i];
        i = (i + 1) & m;
    }
    return -1;
}
static void vtab_put_nogrow(vtab *t, vox_t k, int32_t v) {
    int64_t m = t->cap - 1, i = (int64_t)(vox_hash(k) & (uint64_t)m);
    while (t->vals[i] >= 0) i = (i + 1) & m;
    t->keys[i] = k;
    t->vals[i] = v;
    t->used++;
}
static void vtab_put(vtab *t, vox_t k, int32_t v) {
    if (2 * (t->used + 1) > t->cap) {
        vtab n;
        vtab_init(&n, t->cap);
        for (int64_t i = 0; i < t->cap; ++i)
            if (t->vals[i] >= 0) vtab_put_nogrow(&n, t->keys[i], t->vals[i]);
        vtab_free(t);
        *t = n;
    }
    vtab_put_nogrow(t, k, v);
}

/* Thread count of the multi-core timing mode (bench.py's cpu_baseline leg).  1 (the default, what every parity
 * test uses) = one sequential pass in point order.  > 1 = the two loops kiss-icp itself runs under TBB
 * (Registration.cpp: correspondences + linear system; Deskew.cpp / the per-point transforms) are split over
 * OpenMP threads in fixed chunks of ORC_CHUNK points whose partial sums are added in chunk order, so the result
 * does not depend on the thread count (it differs from the sequential pass by summation order only). */
#define ORC_CHUNK 128
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

/* ================================================================= Deskew / Preprocess / VoxelDownsample */

/* Deskew.cpp DeSkewScan (reference call site kiss.py:90): mid-pose timestamp 0.5 */
void orc_deskew(const double *xyz, const double *t01, int64_t n, const double start[16],
                const double finish[16], double *out) {
    double si[16], d[16], xi[6];
    orc_se3_inv(start, si);
    orc_se3_mul(si, finish, d);
    orc_se3_log(d, xi);
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads)
    for (int64_t i = 0; i < n; ++i) {
        double s = t01[i] - 0.5, x[6], M[16];
        for (int k = 0; k < 6; ++k) x[k] = s * xi[k];
        orc_se3_exp(x, M);
        T_apply(M, xyz + 3 * i, out + 3 * i);
    }
}

/* Preprocessing.cpp Preprocess (kiss.py:93): strict inequalities on the Euclidean norm */
int64_t orc_preprocess(const double *xyz, int64_t n, double max_range, double min_range, double *out) {
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        double r = norm3(xyz + 3 * i);
        if (r < max_range && r > min_range) {
            out[3 * m] = xyz[3 * i];
            out[3 * m + 1] = xyz[3 * i + 1];
            out[3 * m + 2] = xyz[3 * i + 2];
            ++m;
        }
    }
    return m;
}

/* Preprocessing.cpp VoxelDownsample (kiss.py:96 via KissICP.voxelize) */
int64_t orc_voxel_downsample(const double *xyz, int64_t n, double voxel, double *out, int64_t *src_index) {
    vtab t;
    vtab_init(&t, n);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        vox_t v = vox_of(xyz + 3 * i, voxel);
        if (vtab_find(&t, v) >= 0) continue;
        vtab_put(&t, v, 0);
        out[3 * m] = xyz[3 * i];
        out[3 * m + 1] = xyz[3 * i + 1];
        out[3 * m + 2] = xyz[3 * i + 2];
        if (src_index) src_index[m] = i;
        ++m;
    }
    vtab_free(&t);
    return m;
}

/* ================================================================= VoxelHashMap */

struct orc_map {
    double voxel_size, max_distance;
    int32_t max_pts;
    vtab tab;       /* voxel -> block id */
    vox_t *bkey;    /* per block */
    int32_t *bcnt;
    double *bpts;   /* [block][max_pts][3] */
    int64_t nblk, cap_blk;
};

orc_map *orc_map_create(double voxel_size, double max_distance, int32_t max_points_per_voxel) {
    orc_map *m = (orc_map *)calloc(1, sizeof *m);
    m->voxel_size = voxel_size;
    m->max_distance = max_distance;
    m->max_pts = max_points_per_voxel;
    vtab_init(&m->tab, 1024);
    m->cap_blk = 1024;
    m->bkey = (vox_t *)malloc((size_t)m->cap_blk * sizeof(vox_t));
    m->bcnt = (int32_t *)malloc((size_t)m->cap_blk * sizeof(int32_t));
    m->bpts = (double *)malloc((size_t)m->cap_blk * (size_t)m->max_pts * 3 * sizeof(double));
    return m;
}
void orc_map_destroy(orc_map *m) {
    if (!m) return;
    vtab_free(&m->tab);
    free(m->bkey);
    free(m->bcnt);
    free(m->bpts);
    free(m);
}
int64_t orc_map_num_voxels(const orc_map *m) { return m->nblk; }
int64_t orc_map_num_points(const orc_map *m) {
    int64_t s = 0;
    for (int64_t b = 0; b < m->nblk; ++b) s += m->bcnt[b];
    return s;
}
int64_t orc_map_export(const orc_map *m, double *out) {
    int64_t k = 0;
    for (int64_t b = 0; b < m->nblk; ++b)
        for (int32_t j = 0; j < m->bcnt[b]; ++j) {
            const double *p = m->bpts + ((size_t)b * m->max_pts + j) * 3;
            out[3 * k] = p[0]; out[3 * k + 1] = p[1]; out[3 * k + 2] = p[2];
            ++k;
        }
    return k;
}

/* VoxelHashMap::AddPoints: append to an existing voxel while it holds < max_points, else create it */
void orc_map_add_points(orc_map *m, const double *xyz, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        vox_t v = vox_of(xyz + 3 * i, m->voxel_size);
        int32_t b = vtab_find(&m->tab, v);
        if (b < 0) {
            if (m->nblk == m->cap_blk) {
                m->cap_blk *= 2;
                m->bkey = (vox_t *)realloc(m->bkey, (size_t)m->cap_blk * sizeof(vox_t));
                m->bcnt = (int32_t *)realloc(m->bcnt, (size_t)m->cap_blk * sizeof(int32_t));
                m->bpts = (double *)realloc(m->bpts, (size_t)m->cap_blk * (size_t)m->max_pts * 3 * sizeof(double));
            }
            b = (int32_t)m->nblk++;
            m->bkey[b] = v;
            m->bcnt[b] = 0;
            vtab_put(&m->tab, v, b);
        }
        if (m->bcnt[b] < m->max_pts) {
            double *p = m->bpts + ((size_t)b * m->max_pts + m->bcnt[b]) * 3;
            p[0] = xyz[3 * i]; p[1] = xyz[3 * i + 1]; p[2] = xyz[3 * i + 2];
            m->bcnt[b]++;
        }
    }
}

/* VoxelHashMap::RemovePointsFarFromLocation: a voxel goes when its FIRST point is beyond max_distance */
void orc_map_prune(orc_map *m, const double origin[3]) {
    double md2 = m->max_distance * m->max_distance;
    int64_t w = 0;
    for (int64_t b = 0; b < m->nblk; ++b) {
        const double *p = m->bpts + (size_t)b * m->max_pts * 3;
        double dx = p[0] - origin[0], dy = p[1] - origin[1], dz = p[2] - origin[2];
        if (dx * dx + dy * dy + dz * dz > md2) continue;
        if (w != b) {
            m->bkey[w] = m->bkey[b];
            m->bcnt[w] = m->bcnt[b];
            memcpy(m->bpts + (size_t)w * m->max_pts * 3, p, (size_t)m->max_pts * 3 * sizeof(double));
        }
        ++w;
    }
    if (w == m->nblk) return;
    m->nblk = w;
    vtab_free(&m->tab);
    vtab_init(&m->tab, w > 512 ? w : 512);
    for (int64_t b = 0; b < w; ++b) vtab_put_nogrow(&m->tab, m->bkey[b], (int32_t)b);
}

/* VoxelHashMap::Update(points, pose): transform, AddPoints, prune around the pose's translation */
void orc_map_update(orc_map *m, const double *xyz, int64_t n, const double pose[16]) {
    double *w = (double *)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(double));
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads)
    for (int64_t i = 0; i < n; ++i) T_apply(pose, xyz + 3 * i, w + 3 * i);
    orc_map_add_points(m, w, n);
    double o[3] = {pose[3], pose[7], pose[11]};
    orc_map_prune(m, o);
    free(w);
}

/* VoxelHashMap::GetCorrespondences' per-point search: 27 voxels around the point's voxel */
static inline int nearest_in_map(const orc_map *m, const double p[3], double best[3], double *best_d2,
                                 int64_t *n_cand) {
    vox_t c = vox_of(p, m->voxel_size);
    double bd = 1.7976931348623157e308;
    int found = 0;
    for (int32_t i = c.x - 1; i <= c.x + 1; ++i)
        for (int32_t j = c.y - 1; j <= c.y + 1; ++j)
            for (int32_t k = c.z - 1; k <= c.z + 1; ++k) {
                vox_t v = {i, j, k};
                int32_t b = vtab_find(&m->tab, v);
                if (b < 0) continue;
                const double *q = m->bpts + (size_t)b * m->max_pts * 3;
                int32_t cnt = m->bcnt[b];
                *n_cand += cnt;
                for (int32_t a = 0; a < cnt; ++a, q += 3) {
                    double dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                    double d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < bd) {
                        bd = d2;
                        best[0] = q[0]; best[1] = q[1]; best[2] = q[2];
                        found = 1;
                    }
                }
            }
    *best_d2 = bd;
    return found;
}

/* Registration.cpp BuildLinearSystem, one pair: residual r = s - t, J = [I | -hat(s)], w = k^2 / (k + |r|^2)^2 */
static inline void accumulate_pair(const double s[3], const double t[3], double kernel, double k2, double JTJ[36],
                                   double JTr[6]) {
    double r[3] = {s[0] - t[0], s[1] - t[1], s[2] - t[2]};
    double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    double den = kernel + r2;
    double w = k2 / (den * den);
    double J[18], H[9]; /* J row-major 3x6 */
    hat3(s, H);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            J[6 * a + b] = (a == b) ? 1.0 : 0.0;
            J[6 * a + 3 + b] = -H[3 * a + b];
        }
    for (int a = 0; a < 6; ++a) {
        for (int b = a; b < 6; ++b) {
            double acc = 0.0;
            for (int q = 0; q < 3; ++q) acc += (J[6 * q + a] * w) * J[6 * q + b];
            JTJ[6 * a + b] += acc;
        }
        double acc = 0.0;
        for (int q = 0; q < 3; ++q) acc += (J[6 * q + a] * w) * r[q];
        JTr[a] += acc;
    }
}

/* points [i0, i1) of the correspondence search + linear system, accumulated in index order */
static void linear_system_range(const orc_map *m, const double *src, int64_t i0, int64_t i1, double max_dist,
                                double kernel, double JTJ[36], double JTr[6], int64_t *nc, int64_t *cand,
                                double *tgt) {
    double k2 = kernel * kernel;
    for (int64_t i = i0; i < i1; ++i) {
        const double *s = src + 3 * i;
        double t[3], d2;
        int ok = nearest_in_map(m, s, t, &d2, cand);
        /* upstream keeps the pair when (closest - point).norm() < max_correspondance_distance */
        if (!ok || !(sqrt(d2) < max_dist)) {
            if (tgt) tgt[3 * i] = tgt[3 * i + 1] = tgt[3 * i + 2] = NAN;
            continue;
        }
        if (tgt) { tgt[3 * i] = t[0]; tgt[3 * i + 1] = t[1]; tgt[3 * i + 2] = t[2]; }
        ++*nc;
        accumulate_pair(s, t, kernel, k2, JTJ, JTr);
    }
}

/* GetCorrespondences + BuildLinearSystem on transformed source points */
void orc_map_linear_system(const orc_map *m, const double *src, int64_t n, double max_dist, double kernel,
                           double sums[27], int64_t *n_corr, int64_t *n_cand, double *tgt) {
    double JTJ[36], JTr[6];
    memset(JTJ, 0, sizeof JTJ);
    memset(JTr, 0, sizeof JTr);
    int64_t nc = 0, cand = 0;
    if (g_threads <= 1 || n < 2 * ORC_CHUNK) {
        linear_system_range(m, src, 0, n, max_dist, kernel, JTJ, JTr, &nc, &cand, tgt);
    } else {
        const int64_t nchunk = (n + ORC_CHUNK - 1) / ORC_CHUNK;
        double *part = (double *)calloc((size_t)nchunk * 44, sizeof(double)); /* 36 JTJ + 6 JTr + nc + cand */
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
        for (int64_t ch = 0; ch < nchunk; ++ch) {
            double *q = part + ch * 44;
            int64_t c_nc = 0, c_cand = 0;
            int64_t i1 = (ch + 1) * ORC_CHUNK < n ? (ch + 1) * ORC_CHUNK : n;
            linear_system_range(m, src, ch * ORC_CHUNK, i1, max_dist, kernel, q, q + 36, &c_nc, &c_cand, tgt);
            q[42] = (double)c_nc;
            q[43] = (double)c_cand;
        }
        for (int64_t ch = 0; ch < nchunk; ++ch) {
            const double *q = part + ch * 44;
            for (int a = 0; a < 36; ++a) JTJ[a] += q[a];
            for (int a = 0; a < 6; ++a) JTr[a] += q[36 + a];
            nc += (int64_t)q[42];
            cand += (int64_t)q[43];
        }
        free(part);
    }
    int o = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) sums[o++] = JTJ[6 * a + b];
    for (int a = 0; a < 6; ++a) sums[o++] = JTr[a];
    if (n_corr) *n_corr = nc;
    if (n_cand) *n_cand = cand;
}

/* dx = JTJ.ldlt().solve(-JTr): LDL^T without pivoting; a zero pivot zeroes that component (what a
 * rank-revealing LDLT returns for the all-zero system of a scan with no correspondences). */
void orc_solve6(const double sums[27], double dx[6]) {
    double A[36], L[36], D[6], y[6], b[6];
    int o = 0;
    for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c) {
            A[6 * a + c] = sums[o];
            A[6 * c + a] = sums[o];
            ++o;
        }
    for (int a = 0; a < 6; ++a) b[a] = -sums[21 + a];
    memset(L, 0, sizeof L);
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        D[j] = d;
        L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[6 * i + j];
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = (d != 0.0) ? v / d : 0.0;
        }
    }
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k];
        y[i] = v;
    }
    for (int i = 0; i < 6; ++i) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * dx[k];
        dx[i] = v;
    }
}

/* The poses of kiss-icp are Sophus::SE3d: a UNIT QUATERNION and a translation.  The 4x4 guess the reference hands over
 * (kiss.py:108-114 -> pybind `Sophus::SE3d initial_guess(T_guess)`) becomes one on entry, and what comes back,
 * `(T_icp * initial_guess).matrix()`, is the matrix of one: every registration re-orthonormalises the rotation.  A
 * restatement on bare 3x3 matrices must do that explicitly - without it the constant-velocity recursion
 * last (prev^-1 last) amplifies rounding-level non-orthogonality by ~2.4x per sweep (1e-16 -> 1e-5 in 30 sweeps) and the
 * track is lost around sweep 35-40 on ANY input (round-2 finding).  Projection = matrix -> unit quaternion
 * (the branch on the largest of the diagonal entries and the trace, then normalise) -> matrix. */
static void so3_project16(double T[16]) {
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]}, q[4];
    mat_to_quat(R, q);
    quat_to_mat(q, R);
    T[0] = R[0]; T[1] = R[1]; T[2] = R[2]; T[4] = R[3]; T[5] = R[4]; T[6] = R[5]; T[8] = R[6]; T[9] = R[7]; T[10] = R[8];
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
}
void orc_so3_project(double T[16]) { so3_project16(T); }

/* Registration.cpp RegisterFrame (reference call site kiss.py:108-114) */
void orc_register(const orc_map *m, const double *frame, int64_t n, const double guess_in[16], double max_dist,
                  double kernel, int32_t max_iter, double conv, double out_pose[16], int32_t *iters,
                  int32_t *n_corr_last, int64_t *sum_cand) {
    if (iters) *iters = 0;
    if (n_corr_last) *n_corr_last = 0;
    double guess[16];
    memcpy(guess, guess_in, sizeof guess);
    so3_project16(guess); /* Sophus::SE3d initial_guess(T_guess) */
    if (m->nblk == 0) { /* voxel_map.Empty() => return initial_guess */
        memcpy(out_pose, guess, 16 * sizeof(double));
        return;
    }
    double *src = (double *)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(double));
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads)
    for (int64_t i = 0; i < n; ++i) T_apply(guess, frame + 3 * i, src + 3 * i);
    double Ticp[16];
    T_identity(Ticp);
    for (int32_t j = 0; j < max_iter; ++j) {
        double sums[27], dx[6], E[16];
        int64_t nc = 0, cand = 0;
        orc_map_linear_system(m, src, n, max_dist, kernel, sums, &nc, &cand, NULL);
        orc_solve6(sums, dx);
        orc_se3_exp(dx, E);
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads)
        for (int64_t i = 0; i < n; ++i) T_apply(E, src + 3 * i, src + 3 * i);
        orc_se3_mul(E, Ticp, Ticp);
        if (iters) *iters = j + 1;
        if (n_corr_last) *n_corr_last = (int32_t)nc;
        if (sum_cand) *sum_cand += cand;
        double nn = 0.0;
        for (int k = 0; k < 6; ++k) nn += dx[k] * dx[k];
        if (sqrt(nn) < conv) break;
    }
    orc_se3_mul(Ticp, guess, out_pose);
    so3_project16(out_pose); /* (T_icp * initial_guess).matrix() of an SE3d */
    free(src);
}

/* ================================================================= the per-scan pipeline */

struct orc_icp {
    orc_icp_cfg cfg;
    orc_map *map;
    double *poses; /* [n][16] */
    int64_t n_poses, cap_poses;
    /* Threshold.cpp AdaptiveThreshold state */
    double sse;
    int64_t n_samples;
    double model_deviation[16];
    /* last intermediates */
    double *frame_down, *source;
    int64_t n_down, n_src;
};

void orc_icp_default_cfg(orc_icp_cfg *c, double max_range, double min_range) {
    c->max_range = max_range;
    c->min_range = min_range;
    c->voxel_size = max_range / 100.0; /* kiss-icp config/parser.py load_config */
    c->max_points_per_voxel = 20;
    c->initial_threshold = 2.0;
    c->min_motion_th = 0.1;
    c->deskew = 1;
    c->max_iterations = 500;
    c->convergence = 1e-4;
}

orc_icp *orc_icp_create(const orc_icp_cfg *cfg) {
    orc_icp *h = (orc_icp *)calloc(1, sizeof *h);
    h->cfg = *cfg;
    h->map = orc_map_create(cfg->voxel_size, cfg->max_range, cfg->max_points_per_voxel);
    h->cap_poses = 256;
    h->poses = (double *)malloc((size_t)h->cap_poses * 16 * sizeof(double));
    T_identity(h->model_deviation);
    return h;
}
void orc_icp_destroy(orc_icp *h) {
    if (!h) return;
    orc_map_destroy(h->map);
    free(h->poses);
    free(h->frame_down);
    free(h->source);
    free(h);
}
int64_t orc_icp_num_poses(const orc_icp *h) { return h->n_poses; }
void orc_icp_get_pose(const orc_icp *h, int64_t idx, double out[16]) {
    if (idx < 0) idx += h->n_poses;
    if (idx < 0 || idx >= h->n_poses) { T_identity(out); return; }
    memcpy(out, h->poses + 16 * idx, 16 * sizeof(double));
}
const orc_map *orc_icp_map(const orc_icp *h) { return h->map; }

/* KissICP.get_prediction_model */
void orc_icp_prediction(const orc_icp *h, double out[16]) {
    if (h->n_poses < 2) { T_identity(out); return; }
    double inv[16];
    orc_se3_inv(h->poses + 16 * (h->n_poses - 2), inv);
    orc_se3_mul(inv, h->poses + 16 * (h->n_poses - 1), out);
}
/* KissICP.has_moved: |(P0^-1 Plast).t| > 5 * min_motion_th */
int orc_icp_has_moved(const orc_icp *h) {
    if (h->n_poses < 1) return 0;
    double inv[16], d[16];
    orc_se3_inv(h->poses, inv);
    orc_se3_mul(inv, h->poses + 16 * (h->n_poses - 1), d);
    double t[3] = {d[3], d[7], d[11]};
    return norm3(t) > 5.0 * h->cfg.min_motion_th;
}
/* KissICP.get_adaptive_threshold -> Threshold.cpp ComputeThreshold (stateful: accumulates) */
double orc_icp_sigma(orc_icp *h) {
    if (!orc_icp_has_moved(h)) return h->cfg.initial_threshold;
    double theta = orc_rot_angle(h->model_deviation);
    double t[3] = {h->model_deviation[3], h->model_deviation[7], h->model_deviation[11]};
    double err = norm3(t) + 2.0 * h->cfg.max_range * sin(theta / 2.0);
    if (err > h->cfg.min_motion_th) {
        h->sse += err * err;
        h->n_samples++;
    }
    if (h->n_samples < 1) return h->cfg.initial_threshold;
    return sqrt(h->sse / (double)h->n_samples);
}

int64_t orc_icp_last_frame_down(const orc_icp *h, double *out) {
    if (out && h->n_down) memcpy(out, h->frame_down, (size_t)h->n_down * 3 * sizeof(double));
    return h->n_down;
}
int64_t orc_icp_last_source(const orc_icp *h, double *out) {
    if (out && h->n_src) memcpy(out, h->source, (size_t)h->n_src * 3 * sizeof(double));
    return h->n_src;
}

/* reference kiss.py:83-131, statement by statement */
int orc_icp_register_frame(orc_icp *h, const double *xyz, const double *t01, int64_t n, const double *guess_in,
                           double out_pose[16], orc_icp_stats *st) {
    const orc_icp_cfg *c = &h->cfg;
    size_t bytes = (size_t)(n > 0 ? n : 1) * 3 * sizeof(double);
    double *a = (double *)malloc(bytes), *b = (double *)malloc(bytes);
    if (!a || !b) { free(a); free(b); return -1; }

    /* :90 deskew with KISS's own last two poses (never the guess / EKF) */
    if (c->deskew && h->n_poses >= 2 && t01)
        orc_deskew(xyz, t01, n, h->poses + 16 * (h->n_poses - 2), h->poses + 16 * (h->n_poses - 1), a);
    else
        memcpy(a, xyz, (size_t)n * 3 * sizeof(double));
    /* :93 preprocess */
    int64_t nv = orc_preprocess(a, n, c->max_range, c->min_range, b);
    /* :96 voxelize: frame_downsample at 0.5*vs, source at 1.5*vs of frame_downsample */
    free(h->frame_down);
    free(h->source);
    h->frame_down = (double *)malloc((size_t)(nv > 0 ? nv : 1) * 3 * sizeof(double));
    h->n_down = orc_voxel_downsample(b, nv, c->voxel_size * 0.5, h->frame_down, NULL);
    h->source = (double *)malloc((size_t)(h->n_down > 0 ? h->n_down : 1) * 3 * sizeof(double));
    h->n_src = orc_voxel_downsample(h->frame_down, h->n_down, c->voxel_size * 1.5, h->source, NULL);
    free(a);
    free(b);
    /* :99 */
    double sigma = orc_icp_sigma(h);
    /* :102-105 */
    double guess[16];
    if (guess_in) {
        memcpy(guess, guess_in, sizeof guess);
    } else {
        double pred[16], last[16];
        orc_icp_prediction(h, pred);
        if (h->n_poses) memcpy(last, h->poses + 16 * (h->n_poses - 1), sizeof last);
        else T_identity(last);
        orc_se3_mul(last, pred, guess);
    }
    double guess_raw[16]; /* :116, :128 invert the guess as the caller handed it over */
    memcpy(guess_raw, guess, sizeof guess);
    so3_project16(guess); /* the registration works on the SE3d of it (orc_register does the same to its copy) */
    /* :108-114 */
    double new_pose[16];
    int32_t iters = 0, nc = 0;
    int64_t cand = 0;
    orc_register(h->map, h->source, h->n_src, guess, 3.0 * sigma, sigma / 3.0, c->max_iterations,
                 c->convergence, new_pose, &iters, &nc, &cand);
    /* :116-124 innovation; :128 model deviation */
    /* pose_gain = np.linalg.inv(initial_guess) @ new_pose with the RAW guess (a general 4x4 inverse and product).  dt is the
     * norm of its translation column; drot = |Rotation.from_matrix(gain[:3,:3]).as_rotvec()| and the model deviation -
     * a Sophus::SE3d on the C++ side - see its rotation block through a unit quaternion. */
    double gi[16], gain[16];
    m4_inv(guess_raw, gi);
    m4_mul(gi, new_pose, gain);
    double t[3] = {gain[3], gain[7], gain[11]};
    double drot;
    {   /* :119-120 scipy: Rotation.from_matrix orthogonalises (SVD = the polar factor), then the rotation vector's norm */
        double Rg[9], Qg[9], Tq[16];
        T_get_R(gain, Rg);
        m3_polar(Rg, Qg);
        T_set(Tq, Qg, t);
        drot = orc_rot_angle(Tq);
    }
    so3_project16(gain); /* :128 the model deviation as the Sophus::SE3d the C++ side makes of it (Eigen: matrix -> quaternion, normalised) */
    gain[3] = t[0]; gain[7] = t[1]; gain[11] = t[2];
    memcpy(h->model_deviation, gain, sizeof gain);
    /* :129 map update with frame_downsample */
    orc_map_update(h->map, h->frame_down, h->n_down, new_pose);
    /* :130 */
    if (h->n_poses == h->cap_poses) {
        h->cap_poses *= 2;
        h->poses = (double *)realloc(h->poses, (size_t)h->cap_poses * 16 * sizeof(double));
    }
    memcpy(h->poses + 16 * h->n_poses, new_pose, sizeof new_pose);
    h->n_poses++;
    if (out_pose) memcpy(out_pose, new_pose, sizeof new_pose);
    if (st) {
        st->sigma = sigma;
        st->err_dt = norm3(t);
        st->err_drot = drot;
        st->iterations = iters;
        st->n_corr_last = nc;
        st->n_in = n;
        st->n_valid = nv;
        st->n_down = h->n_down;
        st->n_src = h->n_src;
        st->sum_cand = cand;
        st->map_voxels = orc_map_num_voxels(h->map);
        st->map_points = orc_map_num_points(h->map);
    }
    return 0;
}
