/*
 * ptudes_mi.h -- C-ABI of libptudes_mi.so: the MI355X-native lidar-odometry core.
 *
 * Drop-in boundary for ONE hot path of bexcite/ptudes-lab: the `ptudes ekf-bench ouster` loop
 * (reference src/ptudes/cli/ekf_bench.py:493-563) = KISS-ICP scan-to-local-map registration
 * (reference src/ptudes/kiss.py:54-131, arithmetic in the third-party kiss-icp 0.2.10) + the ptudes
 * error-state EKF (reference src/ptudes/ins/es_ekf.py:191-329).
 *
 * The reference has NO C/FFI boundary of its own: the path sits behind two Python classes
 * (KissICPWrapper, ESEKF) and below them kiss-icp's private pybind11 module.  The entry points here
 * are what a ctypes/pybind11 binding for those classes binds; each one names the reference
 * interface it replaces.  INTEGRATION.md shows the Python-side stubs.
 *
 * Conventions (same as the reference): poses are 4x4 row-major doubles, T_world<-body, body = IMU
 * frame; quaternions xyzw; ICP tangent order (translation, rotation); the EKF attitude error is a
 * right perturbation.  Every function returns 0 on success, < 0 on error (ptl_last_error() gives the
 * text, thread-local).  A handle is not thread-safe; distinct handles are independent.  Each handle
 * owns one HIP stream on cfg.device_id.  Caller owns all buffers passed in; the library copies.
 * All numerics run on the GPU (hand-written HIP kernels, gfx950); there is no CPU fallback.
 */
#ifndef PTUDES_MI_H
#define PTUDES_MI_H

#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTL_OK 0
#define PTL_ERR_ARG -1
#define PTL_ERR_HIP -2      /* HIP runtime error / no device */
#define PTL_ERR_CAPACITY -3 /* a device-side capacity (map pool / hash table / scan size) was exceeded */
#define PTL_ERR_STATE -4

#define PTL_F32 0
#define PTL_F64 1

/* ---- ABI guard.  Every configuration struct starts with {struct_size, abi_version}: the CALLER writes sizeof(its own struct) and the
 * PTL_ABI_VERSION it was compiled / written against (PTL_CFG_INIT does both) BEFORE handing the struct to the library, and
 * ptl_*_default_cfg / ptl_*_create refuse a struct whose size or version is not the library's with PTL_ERR_ARG and both numbers in
 * ptl_last_error() - without writing a byte.  A binding in another language (ctypes, cgo, JNI) that was written against an older header
 * therefore fails at its first call instead of being overrun by the library's memset or read past its end (a field appended to
 * ptl_icp_cfg did exactly that to round 5's documented stub).  ptl_abi_version() / ptl_sizeof_cfg() let such a binding assert its
 * layout at load time (INTEGRATION.md section 2 does).  Bump PTL_ABI_VERSION with every change of a struct or a prototype. */
#define PTL_ABI_VERSION 6
#define PTL_CFG_ICP 0
#define PTL_CFG_EKF 1
#define PTL_CFG_SEQ 2
#define PTL_CFG_ICP_STATS 3
#define PTL_CFG_INIT(cfg_ptr) do { memset((cfg_ptr), 0, sizeof *(cfg_ptr)); (cfg_ptr)->struct_size = (uint32_t)sizeof *(cfg_ptr); (cfg_ptr)->abi_version = PTL_ABI_VERSION; } while (0)
int ptl_abi_version(void);
/* the library's sizeof for PTL_CFG_ICP / _EKF / _SEQ / _ICP_STATS (the three configuration structs and the per-scan counter row);
 * -1 for an unknown id */
int64_t ptl_sizeof_cfg(int which);

const char *ptl_last_error(void);
/* 1 when a HIP device is usable (the only backend); never falls back to a CPU path */
int ptl_backend(void);
int ptl_device_count(void);
int ptl_device_sync(int device_id); /* hipDeviceSynchronize on that device */
/* Page-lock (hipHostRegister) / release a HOST buffer the caller will upload sweeps from - a recording read into memory, a capture ring:
 * the uploads (ptl_*_upload_scan / _range, synchronous copies) then go by DMA straight from it instead of through the runtime's staging
 * buffer (12.6 -> 2x GB/s on the measurement box, tools/stream_feed.py).  Optional; any buffer may be passed to the uploads unpinned. */
int ptl_host_pin(int device_id, void *host_ptr, uint64_t bytes);
int ptl_host_unpin(void *host_ptr);

/* ------------------------------------------------------------------------------------------------
 * ICP handle == reference KissICPWrapper (src/ptudes/kiss.py:18-166) + the kiss_icp.KissICP it owns
 * ---------------------------------------------------------------------------------------------- */
typedef struct ptl_icp ptl_icp;

typedef struct {
    uint32_t struct_size;         /* sizeof(ptl_icp_cfg) as the CALLER knows it - set before ptl_icp_default_cfg (PTL_CFG_INIT) */
    uint32_t abi_version;         /* PTL_ABI_VERSION as the caller knows it */
    /* algorithm parameters: reference kiss.py:40-43 + kiss-icp 0.2.10 config defaults */
    double max_range;             /* kiss.py:25, ekf_bench.py:360-363 */
    double min_range;             /* kiss.py:24,43 */
    double voxel_size;            /* max_range / 100 */
    int32_t max_points_per_voxel; /* 20 */
    double initial_threshold;     /* 2.0 */
    double min_motion_th;         /* 0.1 */
    int32_t deskew;               /* 1 (kiss.py:41) */
    int32_t max_iterations;       /* 500 */
    double convergence;           /* 1e-4 */
    /* device / capacity parameters (no reference counterpart) */
    int32_t device_id;
    int32_t scan_cols;            /* W: when t01 == NULL, point i gets t = (i % W) * (1/W) (kiss.py:34-35) */
    int64_t max_points_per_scan;  /* upper bound of n per register_frame */
    int64_t map_block_capacity;   /* voxel blocks in the local-map pool */
    int64_t map_table_capacity;   /* hash slots (power of two) */
    int32_t gn_workgroups;        /* workgroups of the persistent Gauss-Newton kernel */
    int32_t rebuild_every;        /* rebuild the map hash table every this many scans (drops tombstones) */
    int32_t gn_threads;           /* threads per workgroup of that kernel (multiple of 64, 256..1024) */
    int32_t gn_lanes_per_point;   /* 32: latency form (one point per 32 lanes, one sequence over the whole chip);
                                   * 8: throughput form (lane-per-point answer cache, 8 lanes per searched point, moment
                                   * accumulation; gn_threads <= 512) - what the batched runner uses per sequence, and the
                                   * faster form for dense scans (tens of thousands of source points) */
    int64_t map_small_blocks;     /* 0 (default): every voxel owns a full block of max_points_per_voxel points.  > 0: that many SMALL blocks
                                   * (one 128-byte line, 5 points) beside the map_block_capacity full ones - a voxel starts small and moves to a
                                   * full block when a batch takes it past 5 points.  For sparse maps (BASELINE config 5, 0.1 m voxels: 3.6
                                   * points per voxel): the pool shrinks to a third and a sparse voxel is one line.  Free-running batches only. */
} ptl_icp_cfg;

/* per-scan counters; identical meaning to oracle/oracle.h orc_icp_stats (SURVEY.md 8(d) byte model) */
typedef struct {
    double sigma;        /* kiss.py:99 */
    double err_dt;       /* kiss.py:118 (KissICPWrapper._err_dt) */
    double err_drot;     /* kiss.py:119-120 (KissICPWrapper._err_drot) */
    int32_t iterations;
    int32_t n_corr_last;
    int64_t n_in;
    int64_t n_valid;     /* N_v */
    int64_t n_down;      /* N_d */
    int64_t n_src;       /* N_s */
    int64_t sum_cand;    /* sum_i C_i */
    int64_t map_voxels;  /* M_v */
    int64_t map_points;
} ptl_icp_stats;

/* fills every field with the reference defaults for (max_range, min_range) (kiss.py:21-43); cfg->struct_size / abi_version must
 * already hold the caller's values (PTL_CFG_INIT) - a mismatch is refused with PTL_ERR_ARG and nothing is written */
int ptl_icp_default_cfg(ptl_icp_cfg *cfg, double max_range, double min_range);
/* KissICPWrapper.__init__ (kiss.py:21-52) */
int ptl_icp_create(const ptl_icp_cfg *cfg, ptl_icp **out);
int ptl_icp_destroy(ptl_icp *h);

/* KissICPWrapper.register_frame / _kiss_register_frame (kiss.py:54-131).
 * xyz: n x 3 (dtype PTL_F32 / PTL_F64), host memory; points with |p| outside (min_range, max_range)
 * are dropped, so RANGE==0 returns may be passed as (0,0,0) instead of being masked (kiss.py:59-60).
 * t01: n per-point normalised times (kiss.py:61) or NULL => column-implicit (cfg.scan_cols).
 * guess: 4x4 initial guess (ekf_bench.py:533-548) or NULL => constant-velocity prediction (kiss.py:102-105).
 * out_pose: poses[-1] after the call (kiss.py:74).  stats may be NULL. */
int ptl_icp_register_frame(ptl_icp *h, const void *xyz, int dtype, int64_t n, const double *t01,
                           double scan_ts, const double *guess, double out_pose[16], ptl_icp_stats *stats);
/* KissICPWrapper.poses / .pose (kiss.py:142-153): copies up to max poses (16 doubles each) */
/* Per-call registrations (ptl_icp_register_frame / _range) normally return after the scan's MAP UPDATE, because the statistics
 * row carries the map size.  on != 0: return as soon as the registration itself (the Gauss-Newton kernel) is done - what the
 * reference's register_frame returns is the pose (kiss.py:74) - while the map update runs beside the caller's next steps and the
 * next call's upload and K0-K4; stats->map_voxels / map_points are then -1 (ptl_icp_map_size gives them on demand) and an error
 * flag raised by that update is reported by the next call. */
int ptl_icp_set_lazy_map_stats(ptl_icp *h, int32_t on);
int ptl_icp_num_poses(ptl_icp *h, int64_t *n);
int ptl_icp_get_poses(ptl_icp *h, double *out, int64_t max_poses, int64_t *n_written);
/* kiss_icp.KissICP.get_prediction_model, used at ekf_bench.py:545 */
int ptl_icp_get_prediction(ptl_icp *h, double out[16]);
/* KissICPWrapper.local_map_points (kiss.py:159-161) */
int ptl_icp_map_size(ptl_icp *h, int64_t *voxels, int64_t *points);
int ptl_icp_map_points(ptl_icp *h, double *xyz_out, int64_t max_points, int64_t *n_written);
/* intermediates of the last register_frame, what _kiss_register_frame returns (kiss.py:131): frame_downsample, source */
int ptl_icp_last_frame_down(ptl_icp *h, double *xyz_out, int64_t max_points, int64_t *n_written);
int ptl_icp_last_source(ptl_icp *h, double *xyz_out, int64_t max_points, int64_t *n_written);

/* KissICPWrapper.deskew (kiss.py:76-78): motion-compensate `xyz` (n x 3 f64) with the last two poses; fewer than two
 * poses => returned unchanged */
int ptl_icp_deskew(ptl_icp *h, const double *xyz, const double *t01, int64_t n, double *out);

/* Stage-level entry points (teacher-forced parity checks of single kernels; kiss-icp 0.2.10 units):
 * VoxelHashMap::AddPoints + RemovePointsFarFromLocation on world-frame points */
int ptl_icp_map_add(ptl_icp *h, const double *xyz_world, int64_t n, const double origin[3], int prune);
/* GetCorrespondences + BuildLinearSystem for already-transformed source points: 21 JTJ upper + 6 JTr */
int ptl_icp_linear_system(ptl_icp *h, const double *src_world, int64_t n, double max_dist, double kernel,
                          double sums[27], int64_t *n_corr, int64_t *n_cand);
/* RegisterFrame: full Gauss-Newton loop of `frame` (sensor frame) against the current map */
int ptl_icp_align(ptl_icp *h, const double *frame, int64_t n, const double guess[16], double max_dist,
                  double kernel, double out_pose[16], int32_t *iterations);

/* ------------------------------------------------------------------------------------------------
 * Range-image input == ouster client.XYZLut as the reference uses it (kiss.py:28-29, 59-60) + reduce_active_beams
 * (utils.py:328-341).  SURVEY.md 8(f) rank 1: a sweep enters as a 512 KB u32 range image instead of 3 MB of xyz.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ptl_lut ptl_lut;
/* beam angles in degrees (H each), beam-origin offset in mm, lidar_to_sensor 4x4 with mm translation (SensorInfo
 * fields of the same names); extrinsic16_m (nullable): 4x4 with metre translation, applied on top
 * (use_extrinsics=True, ekf_bench.py:440-452) */
int ptl_lut_create(int device_id, int32_t H, int32_t W, const double *beam_altitude_deg, const double *beam_azimuth_deg,
                   double lidar_origin_to_beam_origin_mm, const double *lidar_to_sensor16_mm,
                   const double *extrinsic16_m, ptl_lut **out);
int ptl_lut_destroy(ptl_lut *l);
/* XYZLut.__call__: range image (H*W u32, mm, 0 = no return) -> H*W x 3 f64 metres */
int ptl_lut_apply(ptl_lut *l, const uint32_t *range_mm, double *xyz_out);
/* StreamStatsTracker.trackScan (reference ins/data.py:284-308): over the non-zero ranges of the rows
 * np.linspace(0, H, beams_num, endpoint=False, dtype=int) (all rows when beams_num <= 0), scaled by range_to_m
 * (0.001; 0.008 for the RNG15 profile, :242-252): out5 = {count, mean, population variance, min, max} */
int ptl_range_stats(int device_id, const uint32_t *range, int32_t H, int32_t W, int32_t beams_num, double range_to_m,
                    double out5[5]);
/* Posed scans for the map fly-by.
 * ouster.sdk.pose_util.TrajectoryEvaluator.poses_at as the reference uses it (utils.py:368 time_bounds 1.5,
 * cli/ekf_bench.py:489 time_bounds 1.0; third-party): pose at each of n timestamps on the SE(3) geodesic between the
 * bracketing knots (ts strictly increasing, poses 4x4 row-major); the first / last segment is extended by bound_before /
 * bound_after seconds; timestamps further out get the identity and are counted in n_outside (the reference skips them). */
int ptl_traj_poses_at(int device_id, const double *knot_ts, const double *knot_poses16, int64_t n_knots, double bound_before,
                      double bound_after, const double *ts, int64_t n, double *poses16_out, int64_t *n_outside);
/* ouster client.dewarp(XYZLut(scan), column_poses = scan.pose) (reference fly.py:75-86 through ScansAccumulator): world
 * xyz (H*W x 3 f64) of a range image whose W columns carry their own pose (W x 16).  As upstream, pixels without a
 * return land on their column's sensor origin; n_valid (nullable) counts the pixels with a return. */
int ptl_lut_dewarp(ptl_lut *l, const uint32_t *range_mm, const double *col_poses16, double *xyz_out, int64_t *n_valid);
/* rows kept active by reduce_active_beams(ls, beams_num); beams_num <= 0 = all rows.  Applies to range-image input. */
int ptl_icp_set_active_beams(ptl_icp *h, int32_t H, int32_t beams_num);
/* register_frame on a raw range image; per-pixel times are column-implicit (kiss.py:34-35) */
int ptl_icp_register_range(ptl_icp *h, ptl_lut *lut, const uint32_t *range_mm, double scan_ts, const double *guess,
                           double out_pose[16], ptl_icp_stats *stats);

/* profiling: HIP-event time of the dominant kernel (the persistent Gauss-Newton loop) since last reset; enable = n > 1
 * times every n-th launch only (the two event records cost ~18 us per scan) */
int ptl_icp_profile(ptl_icp *h, int enable, double *gn_ms_total, int64_t *gn_launches, int reset);
/* diagnostic: accumulated clock ticks of workgroup 0 per phase of that kernel (nn, wg-reduce+publish, barrier,
 * grid-reduce, solve) and out[5] = iterations, since the handle was created / reset */
int ptl_icp_gn_phases(ptl_icp *h, int64_t out[8]);
/* diagnostic: ticks each Gauss-Newton workgroup spent in the search phase since creation; out holds 2 * gn_workgroups
 * values (until the last / the first wavefront finished) */
int ptl_icp_gn_wg_clocks(ptl_icp *h, int64_t *out, int32_t max_wgs);
/* diagnostic: the device state's 32 debug sums (phase-clock builds park sub-step clocks there) */
int ptl_icp_debug_sums(ptl_icp *h, double out[32]);
/* test hook: workgroup `wg` of the following Gauss-Newton launches returns at once (-1 = back to normal, also clears the
 * time-out flag): the others must run into the exchange time-out, abort together and report it - not hang */
int ptl_icp_debug_stall_workgroup(ptl_icp *h, int32_t wg);
/* test hook: at most `free_blocks` free voxel blocks left in the pool (< 0: unchanged), `table_used` map-table entries declared
 * taken (< 0: unchanged): the following map updates raise the pool / table capacity flags (PTL_ERR_CAPACITY at the next wait) */
int ptl_icp_debug_limit_capacity(ptl_icp *h, int32_t free_blocks, int64_t table_used);
/* test hook: set the 22-bit launch epoch of the Gauss-Newton exchange (exercises its wrap-around) */
int ptl_icp_debug_set_epoch(ptl_icp *h, uint32_t epoch);

/* ------------------------------------------------------------------------------------------------
 * EKF handle == reference ESEKF (src/ptudes/ins/es_ekf.py:57-365)
 * ---------------------------------------------------------------------------------------------- */
typedef struct ptl_ekf ptl_ekf;

typedef struct {
    uint32_t struct_size; /* sizeof(ptl_ekf_cfg) as the caller knows it (PTL_CFG_INIT) */
    uint32_t abi_version;
    double init_grav[3]; /* es_ekf.py:75 */
    double init_bacc[3]; /* es_ekf.py:76 */
    double init_bgyr[3]; /* es_ekf.py:77 */
    int32_t device_id;
} ptl_ekf_cfg;

int ptl_ekf_default_cfg(ptl_ekf_cfg *cfg);
int ptl_ekf_create(const ptl_ekf_cfg *cfg, ptl_ekf **out);        /* ESEKF.__init__  (:73-179) */
int ptl_ekf_destroy(ptl_ekf *h);
/* processImu (:191-237) and processPose (:259-329) only queue their launch (the payload travels in the kernel
 * arguments): they return before the step has run; ptl_ekf_get_state / pose_mat / ts wait for it */
int ptl_ekf_process_imu(ptl_ekf *h, const double lacc[3], const double avel[3], double ts);
/* batch form: imu is n x 7 rows (ts, lacc[3], avel[3]); one launch for the whole batch */
int ptl_ekf_process_imu_batch(ptl_ekf *h, const double *imu, int64_t n);
int ptl_ekf_process_pose(ptl_ekf *h, const double pose[16], const double *meas_cov36);
/* nav[19] = pos(3) quat_xyzw(4) vel(3) bias_gyr(3) bias_acc(3) grav(3) (ESEKF.nav, :181-184); cov 18x18 (._cov) */
int ptl_ekf_get_state(ptl_ekf *h, double nav[19], double cov[324]);
int ptl_ekf_pose_mat(ptl_ekf *h, double T[16]); /* NavState.pose_mat (ins/data.py:70-74) */
int ptl_ekf_ts(ptl_ekf *h, double *ts);         /* ESEKF.ts (:186-189) */

/* ------------------------------------------------------------------------------------------------
 * Whole-sequence runner == the reference's driver loop (cli/ekf_bench.py:493-563) on a pre-uploaded
 * sequence: all scans + IMU samples resident in HBM, no host round trip per scan.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ptl_seq ptl_seq;

typedef struct {
    uint32_t struct_size; /* sizeof(ptl_seq_cfg) as the caller knows it (PTL_CFG_INIT); icp and ekf carry their own */
    uint32_t abi_version;
    ptl_icp_cfg icp;
    ptl_ekf_cfg ekf;
    int64_t n_scans;
    int64_t points_per_scan;    /* H * W, every scan has exactly this many (invalid returns = (0,0,0)) */
    int64_t n_imu;
    int32_t use_imu_prediction; /* ekf_bench.py:533-535 */
    int32_t with_ekf;           /* 0 => ICP only (BASELINE config 2) */
    int32_t range_input;        /* batches: 1 => every sweep arrives as a raw range image (ptl_batch_set_lut + ptl_batch_upload_range) and is KEPT as
                                 * one: 4 resident bytes per pixel instead of the 12 of a float32 xyz sweep - 240 sequences x 1 000 sweeps of 128x1024 are
                                 * 126 GB instead of 377 GB (the reference walks a recording of any length, data.py:31-77).  ptl_batch_upload_scan is then
                                 * refused.  0 (default): 12-byte slots, either form may be uploaded.  Ignored by ptl_seq_create. */
    int32_t resident_scans;     /* batches, free-running driver: 0 (default) = every sweep of the run resident (n_scans slots per sequence).  R >= 2 = a RING of
                                 * R sweep slots per sequence - a recording of any length at the full batch width (the reference walks a file scan by scan,
                                 * data.py:31-77, cli/ekf_bench.py:493-563): sweeps are uploaded in order, sweep k takes slot k % R once scan k - R is known
                                 * to be complete (ptl_batch_wait), and uploads of later sweeps may run while a launch works on earlier ones - between
                                 * ptl_batch_enqueue and ptl_batch_wait, or from another host thread (the only calls of a handle that may overlap).
                                 * ptl_batch_enqueue refuses scans whose sweeps are not there.  Ignored by ptl_seq_create. */
} ptl_seq_cfg;

int ptl_seq_create(const ptl_seq_cfg *cfg, ptl_seq **out);
int ptl_seq_destroy(ptl_seq *s);
/* host -> HBM: scan k (points_per_scan x 3 float32, row-major beam-outer) */
int ptl_seq_upload_scan(ptl_seq *s, int64_t k, const float *xyz);
/* the same sweep as a raw range image (points_per_scan u32, mm, 0 = no return): 1/3 of the bytes; the LUT (and
 * optionally reduce_active_beams) must be set with ptl_seq_set_lut before the run */
int ptl_seq_upload_range(ptl_seq *s, int64_t k, const uint32_t *range_mm);
int ptl_seq_set_lut(ptl_seq *s, ptl_lut *lut, int32_t active_beams);
/* imu: n_imu x 7 (ts, lacc, avel); imu_end[k] = number of IMU samples that precede scan k in the event stream */
int ptl_seq_upload_imu(ptl_seq *s, const double *imu, const int64_t *imu_end);
/* cold-start the filters/map and run scans [0, n) (n <= n_scans); returns after the stream has drained */
int ptl_seq_run(ptl_seq *s, int64_t n);
/* continue with the next n scans from where the last run / advance stopped (state kept) */
int ptl_seq_advance(ptl_seq *s, int64_t n);
/* the same split in two: enqueue the next n scans on the handle's stream without waiting, then wait + check
 * device error flags (lets several sequences on one GPU overlap on their own streams) */
int ptl_seq_enqueue(ptl_seq *s, int64_t n);
int ptl_seq_wait(ptl_seq *s);
/* res_poses (n x 16), res_t (n), kiss_poses (n x 16), stats (n): any may be NULL; n_out = scans processed
 * (scans with no IMU since the previous one are skipped when with_ekf, ekf_bench.py:512-518) */
int ptl_seq_results(ptl_seq *s, double *res_poses, double *res_t, double *kiss_poses, ptl_icp_stats *stats,
                    int64_t max_n, int64_t *n_out);
/* device pointer + row count of the (T x 8) NC-GT rows [t, x,y,z, qx,qy,qz,qw] of the last run, for the
 * RCCL trajectory gather (no host copy) */
int ptl_seq_traj_device(ptl_seq *s, void **dev_ptr, int64_t *rows);
/* device-to-device copy of those rows into a caller-owned DEVICE buffer (e.g. a torch tensor's data_ptr) */
int ptl_seq_copy_traj(ptl_seq *s, void *dst_device, int64_t max_rows, int64_t *rows);
int ptl_seq_icp(ptl_seq *s, ptl_icp **icp);
int ptl_seq_profile(ptl_seq *s, int enable, double *gn_ms_total, int64_t *gn_launches, int reset);

/* One scan of the reference's loop body (cli/ekf_bench.py:493-563) for the per-call handles in ONE host round trip: the n_imu IMU samples
 * that precede the scan (rows [ts, lacc(3), avel(3)]: ESEKF.processImu each, es_ekf.py:191-237), the registration (kiss.py:83-131) with
 * the filter's pose as its guess (use_imu_prediction, ekf_bench.py:533-535), else `guess` (nullable: the constant-velocity model),
 * ESEKF.processPose with the new pose (es_ekf.py:259-329).  kiss_pose = the registration's pose, ekf_pose / ekf_ts = the filter's pose
 * after the update and its timestamp (what the loop appends, ekf_bench.py:561-563); any output may be NULL.  Same kernels and order as
 * the separate calls - same bits - with the hand-overs done on the device (the host waits once instead of three times). */
int ptl_icp_ekf_step(ptl_icp *icp, ptl_ekf *ekf, const double *imu_rows, int64_t n_imu, const void *xyz, int dtype, int64_t n,
                     const double *t01, const double *guess, int32_t use_imu_prediction, double kiss_pose[16],
                     double ekf_pose[16], double *ekf_ts, ptl_icp_stats *stats);

/* ------------------------------------------------------------------------------------------------
 * Batched runner: up to 256 independent sequences on ONE GPU.  The sequences s = x (mod 8) live on XCD x (their maps,
 * probe rows and exchange stay in its L2); the workgroups with blockIdx & 7 == x form 1 / 2 / 4 teams of gn_workgroups /
 * 8, / 16, / 32 workgroups (<= 8 / <= 16 / <= 64 sequences), 8 teams for up to 128 sequences, 16 beyond
 * (ptl_batch_set_team_workgroups overrides).  Two drivers, see ptl_batch_set_driver: the free-running
 * kernel (default) and lockstep (one launch per stage for all sequences, <= 32 sequences).  Each sequence's results are
 * bit-identical to running it alone with a team's workgroups and the same gn_lanes_per_point.  cfg describes every sequence (same n_scans / points_per_scan / n_imu; gn_workgroups = 8 x
 * workgroups per XCD); with_ekf requires >= 1 IMU sample between consecutive scans.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ptl_batch ptl_batch;
int ptl_batch_create(const ptl_seq_cfg *cfg, int32_t n_sequences, ptl_batch **out);
int ptl_batch_destroy(ptl_batch *b);
int ptl_batch_upload_scan(ptl_batch *b, int32_t seq, int64_t k, const float *xyz);
int ptl_batch_set_lut(ptl_batch *b, ptl_lut *lut, int32_t active_beams);
int ptl_batch_upload_range(ptl_batch *b, int32_t seq, int64_t k, const uint32_t *range_mm);
int ptl_batch_upload_imu(ptl_batch *b, int32_t seq, const double *imu, const int64_t *imu_end);
int ptl_batch_run(ptl_batch *b, int64_t n);     /* cold start + scans [0, n), waits */
int ptl_batch_enqueue(ptl_batch *b, int64_t n); /* next n scans, does not wait */
int ptl_batch_wait(ptl_batch *b);
int ptl_batch_results(ptl_batch *b, int32_t seq, double *res_poses, double *res_t, double *kiss_poses,
                      ptl_icp_stats *stats, int64_t max_n, int64_t *n_out);
int ptl_batch_copy_traj(ptl_batch *b, int32_t seq, void *dst_device, int64_t max_rows, int64_t *rows);
int ptl_batch_gn_phases(ptl_batch *b, int64_t out[8]); /* like ptl_icp_gn_phases, for the shared launch */
int ptl_batch_icp(ptl_batch *b, int32_t seq, ptl_icp **icp); /* the ICP handle of one sequence (diagnostics, map export) */
int ptl_batch_profile(ptl_batch *b, int enable, double *gn_ms_total, int64_t *gn_launches, int reset);
/* Which driver advances the batch (choose before the first scan of a run):
 *   free_running != 0 (the default when gn_lanes_per_point = 8): one persistent launch (kx_seq_run) carries up to
 *     scans_per_launch scans (0 = keep, default 256) of EVERY sequence; the workgroups of a sequence walk its whole
 *     per-scan pipeline - reference cli/ekf_bench.py:493-563 - at their own pace, so a sequence whose Gauss-Newton loop
 *     converges early starts its next scan instead of waiting for the slowest one.  ptl_batch_profile then times these
 *     launches.  Needs gn_lanes_per_point = 8 and a grid whose workgroups can all be resident (checked here, or at the
 *     first enqueue when the driver is the default); scans_per_launch <= 4096.
 *   free_running == 0: lockstep, one launch per stage for all sequences; a step lasts as long as its slowest sequence.
 * Same results either way (bit-identical). */
int ptl_batch_set_driver(ptl_batch *b, int32_t free_running, int64_t scans_per_launch);
/* back to the cold start (empty maps, fresh filters, scan 0 next) without running anything - what ptl_batch_run does first;
 * after it the driver and the team size may be chosen again for the same handle and the same uploaded sweeps */
int ptl_batch_reset(ptl_batch *b);
/* phase clocks of sequence s in the free-running kernel, 100 MHz wall-clock ticks summed since the cold start:
 * workgroup 0's K0-K4 | its wait before the Gauss-Newton loop | the loop | its wait after it | its map update;
 * out[5] the filter workgroup's step; out[6] scans */
int ptl_batch_seq_clocks(ptl_batch *b, int32_t seq, int64_t out[8]);
/* Team size of the free-running kernel: `team_workgroups` workgroups (1 .. min(64, gn_workgroups / 8); 0 = the default for
 * the number of sequences) walk one scan of one sequence together; an XCD's gn_workgroups / 8 / team_workgroups teams serve
 * its sequences scan by scan.  Smaller teams spread the per-iteration fixed costs of the Gauss-Newton loop (workgroup
 * reduction, exchange, 6x6 solve) over more points each and want more sequences in flight; the results of a sequence
 * are those of a run alone with `team_workgroups` workgroups.  Before the first scan of a run. */
int ptl_batch_set_team_workgroups(ptl_batch *b, int32_t team_workgroups);
int ptl_batch_team_workgroups(ptl_batch *b, int32_t *team_workgroups, int32_t *teams /* nullable: teams that can get work */);
/* Executed-work counters of sequence `seq`, cumulative since the cold start - what the kernels requested from memory, as
 * opposed to the brute-force counts of ptl_icp_stats: [0] full 27-voxel searches (points the answer cache did not
 * settle), [1] probe rows rebuilt (27 hash probes each), [2] stored map points read by the searches, [3] Gauss-Newton
 * iterations, [4] / [5] voxel claims of down-sampling pass 1 / 2, [6] source point-iterations, [7] scans, [8] point-iterations settled as
 * "nothing in reach" (the point's last search found its 27 voxels empty and it has not left its voxel: neither a row evaluation nor a
 * search - exact, the map does not change inside a registration), [9..15] reserved (0). */
int ptl_batch_exec_counters(ptl_batch *b, int32_t seq, uint64_t out[16]);
/* Where the scans of sequence `seq` ran in the free-running kernel, cumulative since the cold start: out[0] scans run by a
 * team of another XCD than the sequence's home XCD (seq & 7) - a team whose own XCD had nothing for it; out[1] scans that
 * ran on another XCD than the sequence's previous scan (the cross-XCD hand-overs: the previous scan's data sits in another
 * L2); out[2] XCC id of the last scan + 1; out[3] reserved. */
int ptl_batch_sched_counters(ptl_batch *b, int32_t seq, uint64_t out[4]);
/* Sticky status word of the free-running driver (cleared by ptl_batch_run / ptl_batch_reset): why teams left a launch
 * other than "all scans done" - 1 a teammate never reached the head-of-launch barrier, 2 ... the job barrier, 4 a team
 * found no work for its whole idle budget while sequences were pending, 8 a team gave up on a sequence (see that
 * sequence's error flags), 16 a sequence did not reach the last scan of a launch (it gets the time-out flag).
 * ptl_batch_wait returns PTL_ERR_CAPACITY for a flagged sequence and PTL_ERR_STATE for a status without one (bit 4 alone -
 * a team that only idled past its budget while every sequence reached its last scan - is reported here, not as an error):
 * no exit of the persistent kernel is silent. */
int ptl_batch_status(ptl_batch *b, uint32_t *status);
/* test hook: workgroup `block` of the free-running grid returns right before the job barrier of its `round`-th job of every
 * launch from now on (0 = the first); block < 0 switches it off */
int ptl_batch_debug_stall_block(ptl_batch *b, int32_t block, int32_t round);
/* test hook: points per thread and pass of the free-running kernel's map update (reference kiss.py:129) - 0 = only ask, else the
 * build's default (8) or its second instance (4); *points_out = the value in effect.  Results must not depend on it. */
int ptl_batch_debug_set_map_points_per_thread(ptl_batch *b, int32_t points, int32_t *points_out);
/* Environment: PTL_TEAM_SYNC=agent when the batch is created keeps the agent-scope release (L2 write-back) at every team
 * barrier of the free-running kernel instead of the XCD-local shortcut (same results; a diagnostic switch).
 * PTL_SCHED_MARGIN=n (default 1): a team takes the least-advanced free sequence of ANOTHER XCD as soon as it is n scans behind its own
 * XCD's least-advanced one (all sequences of the device stay within a scan of each other); -1 = only when its own XCD has nothing left.
 *
 * Memory: one sequence of a batch holds 480 B per point of points_per_scan of work buffers (probe and answer rows, ...:
 * 63 MB at 128x1024), its two per-scan voxel tables (64 and 16 slots of 16 B per point: 134 + 34 MB - sparse on purpose,
 * a taken line costs a claim a dependent read), map_table_capacity x 16 B (256 MB at the default 2^24 slots: sparse for
 * the same reason, see ptl_icp_default_cfg), map_block_capacity x (block size + 44) B (512-B blocks of points at 20 points per
 * voxel, 40 B of block directory - header and first point - and 4 B of free stack each: 278 MB at the default 512 k blocks),
 * map_small_blocks x (128 + 48) B when the second block class is on, and its resident sweeps (n_scans x points_per_scan x 12 B):
 * ~790 MB + sweeps at the defaults.  ptl_batch_create checks the sum against
 * hipMemGetInfo and fails with PTL_ERR_CAPACITY and the numbers when it does not fit. */

/* Compile-time constants of the loaded build that a caller's byte model depends on (bench.py EXEC_COST): [0] candidates
 * per answer row, [1] doubles per answer row, [2] source positions a workgroup keeps in LDS, [3] / [4] points per thread
 * and pass of K1 + map update / K2-K4 in the free-running kernel, [5] threads per workgroup of the 8-lane kernels,
 * [6] lanes per point of the full search, [7] nearest other boxes in its first round, [8] voxels per survivor round,
 * [9] chunks phase A requests ahead, [10] 1000 x the pruning margin, [11] / [12] bytes per map-table / voxel-table
 * entry, [13] 1 when diagnostic clocks are compiled in, [14] reserved (0; rounds 5's movement-budget experiment is gone from the sources). */
int ptl_build_info(int32_t out[16]);
/* identity of what THIS library was built from: sha256 over the kernel sources, the Makefile and the experiment flags (12 hex digits,
 * csrc/Makefile CODE_ID).  bench.py records it in its line and tools/pmc_summary.py in every counter summary: a counter pass speaks
 * for the build it ran on, whatever the source tree looks like by then. */
const char *ptl_code_id(void);

/* ---- multi-GPU: the final trajectory gather.  The reference has no distributed layer; the path shards
 * across independent sequences only (one rank per GPU, no data-path collective) and its ONE collective is the all-gather of every
 * rank's NC-GT rows [t, x, y, z, qx, qy, qz, qw] (reference utils.py:191-252 writes such rows) after the run: ncclAllGather straight
 * from librccl (opened on first use; PTL_RCCL_PATH overrides the search), RCCL over xGMI on the node.
 * The library does no bootstrap: ONE rank calls ptl_comm_unique_id, the caller carries the PTL_COMM_ID_BYTES bytes to the other ranks
 * (environment, file, gloo / MPI / TCP - its business), EVERY rank then calls ptl_comm_create (collective: returns when all have). */
#define PTL_COMM_ID_BYTES 128
typedef struct ptl_comm ptl_comm;
int ptl_comm_unique_id(uint8_t id[PTL_COMM_ID_BYTES]);
int ptl_comm_create(const uint8_t id[PTL_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device_id, ptl_comm **out);
int ptl_comm_destroy(ptl_comm *c);
/* Every rank: d_rows = S x T x 8 doubles in DEVICE memory (sequence-major, rows beyond a sequence's count are padding),
 * counts = S valid row counts (host); S and T equal on every rank.  On return, on every rank, rows_out [world][S][T][8] and
 * counts_out [world][S] (host) hold everybody's. */
int ptl_gather_trajectories(ptl_comm *c, const double *d_rows, int64_t S, int64_t T, const int64_t *counts,
                            double *rows_out, int64_t *counts_out);
/* ... the rows a batch's filter kernel wrote on device (what ptl_batch_copy_traj copies), T = the batch's n_scans:
 * rows_out [world][n_sequences][n_scans][8], counts_out [world][n_sequences] */
int ptl_batch_gather_trajectories(ptl_batch *b, ptl_comm *c, double *rows_out, int64_t *counts_out);

#ifdef __cplusplus
}
#endif
#endif
