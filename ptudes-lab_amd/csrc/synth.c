/*
 * synth.c -- analytic ray caster for synthetic spinning-lidar sweeps (SURVEY.md 8(d) "Synthetic inputs").
 *
 * Host-side data tooling for tests and bench.py: the reference ships no sample data and the real
 * datasets (OS-0-128 pcap, Newer College bags) are not available offline, so 128x1024 / 64x2048
 * sweeps are rendered from a closed room with boxes and cylinders.  Not part of the measured path.
 *
 * Layout of a sweep: H x W points, row-major beam-outer (index = row * W + col), float32 xyz in the
 * sensor(=IMU) frame of the instant each COLUMN was fired (so motion distortion is present and the
 * deskew stage has work to do).  A missing / out-of-range return is (0,0,0), which is what an Ouster
 * XYZLut yields for RANGE == 0 (reference kiss.py:59-60 masks those out).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    int32_t H, W;
    double el_top_deg, el_bot_deg; /* beam elevations, row 0 = top */
    double min_range, max_range;   /* returns outside are dropped */
    double noise_std;              /* range noise (m) */
    double dropout;                /* probability of a missing return */
    double rough_amp;              /* surface roughness: world-anchored displacement amplitude (m) */
    double rough_len;              /* ... and its longest wavelength (m) */
    double ray_jitter_deg;         /* per-ray, per-sweep angular jitter of the firing direction (deg, uniform +-): the direction that
                                    * is cast is jittered, the point is reported along the NOMINAL direction's jittered twin - i.e. the
                                    * sensor knows where it fired (experiments on sampling-lattice effects; 0 = a perfect grid) */
} ptl_synth_sensor;

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

/* world-anchored value noise in [-1,1]: lattice hashes, trilinear blend.  Gives every surface a fixed
 * bumpy relief (grass / rubble / facade detail) so that scans are not made of ideal planes only. */
static inline double lat(int64_t x, int64_t y, int64_t z) {
    uint64_t h = (uint64_t)x * 0x9E3779B97F4A7C15ull ^ (uint64_t)y * 0xC2B2AE3D27D4EB4Full ^
                 (uint64_t)z * 0x165667B19E3779F9ull;
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
    return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
static double vnoise(double x, double y, double z) {
    double fx = floor(x), fy = floor(y), fz = floor(z);
    int64_t ix = (int64_t)fx, iy = (int64_t)fy, iz = (int64_t)fz;
    double u = x - fx, v = y - fy, w = z - fz;
    u = u * u * (3 - 2 * u); v = v * v * (3 - 2 * v); w = w * w * (3 - 2 * w);
    double c = 0.0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 2; ++k)
                c += (a ? u : 1 - u) * (b ? v : 1 - v) * (k ? w : 1 - w) * lat(ix + a, iy + b, iz + k);
    return c;
}

/* distance along unit ray (o, d) to the first surface; the origin is inside the room.
 * cand (nullable): the objects this ray can meet at all - cand[0] boxes' indices from cand + 2, then cand[1] cylinders' */
static double cast(const double o[3], const double d[3], const double room[6], const double *boxes, int nb,
                   const double *cyls, int nc, const int16_t *cand) {
    double best = 1e300;
    /* room: exit distance of the AABB [x0,x1]x[y0,y1]x[z0,z1] */
    for (int a = 0; a < 3; ++a) {
        if (d[a] > 1e-12) {
            double t = (room[2 * a + 1] - o[a]) / d[a];
            if (t < best) best = t;
        } else if (d[a] < -1e-12) {
            double t = (room[2 * a] - o[a]) / d[a];
            if (t < best) best = t;
        }
    }
    /* boxes: (cx, cy, cz, hx, hy, hz), slab test, entry point */
    const int nbc = cand ? cand[0] : nb, ncc = cand ? cand[1] : nc;
    for (int bi = 0; bi < nbc; ++bi) {
        const int b = cand ? cand[2 + bi] : bi;
        const double *B = boxes + 6 * b;
        double tn = -1e300, tf = 1e300;
        int miss = 0;
        for (int a = 0; a < 3 && !miss; ++a) {
            double lo = B[a] - B[3 + a], hi = B[a] + B[3 + a];
            if (fabs(d[a]) < 1e-12) {
                if (o[a] < lo || o[a] > hi) miss = 1;
            } else {
                double t1 = (lo - o[a]) / d[a], t2 = (hi - o[a]) / d[a];
                if (t1 > t2) { double t = t1; t1 = t2; t2 = t; }
                if (t1 > tn) tn = t1;
                if (t2 < tf) tf = t2;
                if (tn > tf) miss = 1;
            }
        }
        if (!miss && tn > 1e-6 && tn < best) best = tn;
    }
    /* vertical cylinders: (cx, cy, r, h) standing on z = room z0 */
    for (int ci = 0; ci < ncc; ++ci) {
        const int c = cand ? cand[2 + nbc + ci] : ci;
        const double *Cy = cyls + 4 * c;
        double ox = o[0] - Cy[0], oy = o[1] - Cy[1], r = Cy[2], z0 = room[4], z1 = room[4] + Cy[3];
        double A = d[0] * d[0] + d[1] * d[1];
        if (A > 1e-14) {
            double Bq = ox * d[0] + oy * d[1], Cq = ox * ox + oy * oy - r * r;
            double disc = Bq * Bq - A * Cq;
            if (disc > 0.0) {
                double t = (-Bq - sqrt(disc)) / A;
                if (t > 1e-6 && t < best) {
                    double z = o[2] + t * d[2];
                    if (z >= z0 && z <= z1) best = t;
                }
            }
        }
        if (fabs(d[2]) > 1e-12) { /* top cap */
            double t = (z1 - o[2]) / d[2];
            if (t > 1e-6 && t < best) {
                double x = ox + t * d[0], y = oy + t * d[1];
                if (x * x + y * y <= r * r) best = t;
            }
        }
    }
    return best;
}

/* Rows are rendered by at most this many threads (every ray has its own seed, so the result does not depend on
 * it). The default OpenMP team is one thread per visible core, which is far more than a container's CPU quota
 * allows to run: 256 threads on a 16-core quota made one sweep take 124 ms instead of 4. */
static int synth_threads = 1;
void ptl_synth_set_threads(int32_t n) { synth_threads = n < 1 ? 1 : n; }

/* col_poses: W x 12 doubles, per column R (row-major 9) then t (3): world <- sensor at firing time */
void ptl_synth_render(const double room[6], const double *boxes, int32_t nb, const double *cyls, int32_t nc,
                      const ptl_synth_sensor *s, const double *col_poses, uint64_t seed, float *xyz_out) {
    const int H = s->H, W = s->W;
    const double deg = M_PI / 180.0;
    const int nthr = (H * W < 8192) ? 1 : (synth_threads < H ? synth_threads : H);
    /* All rays of a column leave one origin inside one half-plane (spanned by the column's azimuth direction and the sensor's
     * up axis), so an object whose bounding sphere does not reach that half-plane cannot be hit by any of them: per column,
     * the few objects that can (3-6 of 60) are listed once and the H rays of the column test only those.  A culled object
     * would have reported a miss, so the sweep is bit-identical to the full test.  (Not with per-ray angular jitter.) */
    const int stride = 2 + nb + nc;
    int16_t *cands = NULL;
    double *caz = NULL;
    if (s->ray_jitter_deg <= 0.0 && nb + nc > 0 && nb + nc < 32000) {
        cands = (int16_t *)malloc((size_t)W * stride * sizeof(int16_t));
        caz = (double *)malloc((size_t)W * 2 * sizeof(double));
    }
    if (cands && caz) {
        for (int col = 0; col < W; ++col) {
            const double az = 2.0 * M_PI * (double)col / (double)W;
            const double ca = cos(az), sa = sin(az);
            caz[2 * col] = ca; caz[2 * col + 1] = sa;
            const double *P = col_poses + 12 * col;
            /* world-frame forward direction and normal of the column's half-plane */
            const double f[3] = {P[0] * ca + P[1] * sa, P[3] * ca + P[4] * sa, P[6] * ca + P[7] * sa};
            const double nn[3] = {-P[0] * sa + P[1] * ca, -P[3] * sa + P[4] * ca, -P[6] * sa + P[7] * ca};
            int16_t *cd = cands + (size_t)col * stride;
            int kb = 0, kc = 0;
            for (int b = 0; b < nb; ++b) {
                const double *B = boxes + 6 * b;
                const double rad = sqrt(B[3] * B[3] + B[4] * B[4] + B[5] * B[5]) * (1.0 + 1e-9) + 1e-6;
                const double dx = B[0] - P[9], dy = B[1] - P[10], dz = B[2] - P[11];
                if (fabs(nn[0] * dx + nn[1] * dy + nn[2] * dz) <= rad && f[0] * dx + f[1] * dy + f[2] * dz >= -rad) cd[2 + kb++] = (int16_t)b;
            }
            for (int c = 0; c < nc; ++c) {
                const double *Cy = cyls + 4 * c;
                const double hh = 0.5 * Cy[3];
                const double rad = sqrt(Cy[2] * Cy[2] + hh * hh) * (1.0 + 1e-9) + 1e-6;
                const double dx = Cy[0] - P[9], dy = Cy[1] - P[10], dz = room[4] + hh - P[11];
                if (fabs(nn[0] * dx + nn[1] * dy + nn[2] * dz) <= rad && f[0] * dx + f[1] * dy + f[2] * dz >= -rad) cd[2 + kb + kc++] = (int16_t)c;
            }
            cd[0] = (int16_t)kb; cd[1] = (int16_t)kc;
        }
    } else {
        free(cands); free(caz);
        cands = NULL; caz = NULL;
    }
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int row = 0; row < H; ++row) {
        double el = (H > 1) ? s->el_top_deg + (s->el_bot_deg - s->el_top_deg) * (double)row / (double)(H - 1)
                            : s->el_top_deg;
        double ce = cos(el * deg), se = sin(el * deg);
        for (int col = 0; col < W; ++col) {
            double az = 2.0 * M_PI * (double)col / (double)W;
            double cej = ce, sej = se;
            if (s->ray_jitter_deg > 0.0) {
                uint64_t sj = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(row * W + col) * 0xD6E8FEB86659FD93ull + 0x51ull;
                double ja = (2.0 * u01(&sj) - 1.0) * s->ray_jitter_deg * deg, je = (2.0 * u01(&sj) - 1.0) * s->ray_jitter_deg * deg;
                az += ja;
                cej = cos(el * deg + je); sej = sin(el * deg + je);
            }
            const double caz_c = caz ? caz[2 * col] : cos(az), saz_c = caz ? caz[2 * col + 1] : sin(az);
            double db[3] = {cej * caz_c, cej * saz_c, sej};
            const double *P = col_poses + 12 * col;
            double dw[3] = {P[0] * db[0] + P[1] * db[1] + P[2] * db[2],
                            P[3] * db[0] + P[4] * db[1] + P[5] * db[2],
                            P[6] * db[0] + P[7] * db[1] + P[8] * db[2]};
            double r = cast(P + 9, dw, room, boxes, nb, cyls, nc, cands ? cands + (size_t)col * stride : NULL);
            if (s->rough_amp > 0.0) {
                double q = 1.0 / s->rough_len;
                double hx = (P[9] + r * dw[0]) * q, hy = (P[10] + r * dw[1]) * q, hz = (P[11] + r * dw[2]) * q;
                /* octaves: rough_len (1), /2.7 (0.5), /9 (0.3), /27 (0.2): relief from metres down to
                 * centimetres, so sliding a sampling pattern along a surface changes what it measures */
                r += s->rough_amp * (vnoise(hx, hy, hz) + 0.5 * vnoise(2.7 * hx + 11.0, 2.7 * hy - 5.0, 2.7 * hz + 3.0) +
                                     0.3 * vnoise(9.1 * hx - 7.0, 9.1 * hy + 2.0, 9.1 * hz - 13.0) +
                                     0.2 * vnoise(27.3 * hx + 5.0, 27.3 * hy + 17.0, 27.3 * hz + 1.0));
            }
            uint64_t st = seed * 0xD1342543DE82EF95ull + (uint64_t)(row * W + col) * 0x2545F4914F6CDD1Dull;
            double u1 = u01(&st), u2 = u01(&st), u3 = u01(&st);
            if (u1 < 1e-300) u1 = 1e-300;
            r += s->noise_std * sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
            float *o = xyz_out + 3 * ((size_t)row * W + col);
            if (u3 < s->dropout || !(r > s->min_range) || !(r < s->max_range)) {
                o[0] = o[1] = o[2] = 0.0f;
            } else {
                o[0] = (float)(r * db[0]);
                o[1] = (float)(r * db[1]);
                o[2] = (float)(r * db[2]);
            }
        }
    }
    free(cands);
    free(caz);
}

/* ------------------------------------------------------------------------------------------------
 * Random-walk SE(3) ground truth, integrated at a fine step (SURVEY.md 8(d) "Trajectory").
 * v_knots / w_knots: body-frame velocity / angular-rate commands at knot_dt spacing (AR(1) draws made
 * by the Python side with numpy's default_rng); linearly interpolated, then
 *   - a soft repulsion keeps the sensor >= ~2 m from walls and outside obstacles' bounding circles,
 *   - height is pulled to `height`, roll/pitch are pulled level (a hand-held / vehicle-like motion).
 * Outputs per step i (time i*dt): R (world<-body, row-major 9), p, world velocity, body angular rate.
 */
static void rodrigues(const double w[3], double R[9]) {
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double a, b;
    if (th < 1e-8) { a = 1.0; b = 0.5; }
    else { a = sin(th) / th; b = (1.0 - cos(th)) / (th * th); }
    double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
            K2[3 * i + j] = s;
        }
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

void ptl_synth_trajectory(const double room[6], const double *obst_xyr, int32_t n_obst, const double *v_knots,
                          const double *w_knots, int32_t n_knots, double knot_dt, double dt, int64_t n_steps,
                          const double p0[3], double height, double *R_out, double *p_out, double *vw_out,
                          double *wb_out) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {p0[0], p0[1], p0[2]};
    const double margin = 5.0, k_wall = 1.0, k_z = 1.0, k_att = 1.5, k_obst = 2.0;
    for (int64_t i = 0; i < n_steps; ++i) {
        double t = (double)i * dt, u = t / knot_dt;
        int32_t k = (int32_t)u;
        if (k > n_knots - 2) k = n_knots - 2;
        double f = u - (double)k;
        if (f > 1.0) f = 1.0;
        double vb[3], wb[3], vw[3];
        for (int a = 0; a < 3; ++a) {
            vb[a] = (1.0 - f) * v_knots[3 * k + a] + f * v_knots[3 * k + 3 + a];
            wb[a] = (1.0 - f) * w_knots[3 * k + a] + f * w_knots[3 * k + 3 + a];
        }
        for (int a = 0; a < 3; ++a) vw[a] = R[3 * a] * vb[0] + R[3 * a + 1] * vb[1] + R[3 * a + 2] * vb[2];
        for (int a = 0; a < 2; ++a) {
            double lo = room[2 * a] + margin, hi = room[2 * a + 1] - margin;
            if (p[a] < lo) vw[a] += k_wall * (lo - p[a]);
            if (p[a] > hi) vw[a] -= k_wall * (p[a] - hi);
        }
        for (int o = 0; o < n_obst; ++o) {
            double dx = p[0] - obst_xyr[3 * o], dy = p[1] - obst_xyr[3 * o + 1], rr = obst_xyr[3 * o + 2] + 1.5;
            double d = sqrt(dx * dx + dy * dy);
            if (d < rr && d > 1e-9) {
                vw[0] += k_obst * (rr - d) * dx / d;
                vw[1] += k_obst * (rr - d) * dy / d;
            }
        }
        vw[2] += k_z * (height - p[2]);
        /* level the body: up vector in body coords is row 2 of R (= R^T e_z) */
        double ub[3] = {R[6], R[7], R[8]};
        wb[0] += k_att * (-ub[1]); /* e_z x u_b = (-u_y, u_x, 0) */
        wb[1] += k_att * (ub[0]);
        for (int a = 0; a < 9; ++a) R_out[9 * i + a] = R[a];
        for (int a = 0; a < 3; ++a) {
            p_out[3 * i + a] = p[a];
            vw_out[3 * i + a] = vw[a];
            wb_out[3 * i + a] = wb[a];
        }
        double dth[3] = {wb[0] * dt, wb[1] * dt, wb[2] * dt}, dR[9], Rn[9];
        rodrigues(dth, dR);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int q = 0; q < 3; ++q) s += R[3 * r + q] * dR[3 * q + c];
                Rn[3 * r + c] = s;
            }
        /* re-orthonormalise (Gram-Schmidt on rows) so 1e5 steps stay a rotation */
        double n0 = sqrt(Rn[0] * Rn[0] + Rn[1] * Rn[1] + Rn[2] * Rn[2]);
        for (int c = 0; c < 3; ++c) Rn[c] /= n0;
        double d01 = Rn[0] * Rn[3] + Rn[1] * Rn[4] + Rn[2] * Rn[5];
        for (int c = 0; c < 3; ++c) Rn[3 + c] -= d01 * Rn[c];
        double n1 = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4] + Rn[5] * Rn[5]);
        for (int c = 0; c < 3; ++c) Rn[3 + c] /= n1;
        Rn[6] = Rn[1] * Rn[5] - Rn[2] * Rn[4];
        Rn[7] = Rn[2] * Rn[3] - Rn[0] * Rn[5];
        Rn[8] = Rn[0] * Rn[4] - Rn[1] * Rn[3];
        for (int a = 0; a < 9; ++a) R[a] = Rn[a];
        for (int a = 0; a < 3; ++a) p[a] += vw[a] * dt;
    }
}
