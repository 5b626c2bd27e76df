// icp_kernels.h -- HIP kernels (gfx950) of the scan-to-local-map registration path.
//
// Replaces, on device, what reference src/ptudes/kiss.py:83-131 drives inside kiss-icp 0.2.10:
//   K0 scan_prologue   deskew twist, adaptive threshold, initial guess          (kiss.py:90,99,102-105)
//   K1 k_deskew_vds1   DeSkewScan + Preprocess + VoxelDownsample(0.5 vs) claim  (kiss.py:90,93,96)
//   K2 k_count_w1      winners of pass 1 counted per block (only the drivers with one launch per stage; the free-running kernel looks back in K3)
//   K3 k_compact_fd    ordered compaction -> frame_downsample
//   K3b k_vds2_fd      frame_downsample (compact) claims VoxelDownsample(1.5 vs)   [+ k_count_w2: its winners per block, as K2]
//   K4 k_compact_src   ordered compaction -> source, pass-2 slots released
//   K5 k_gn_loop       persistent Gauss-Newton loop: 27-voxel NN + robust 6x6 system + solve (kiss.py:108-114)
//      (tail of K5)    new pose, innovation log, threshold model deviation      (kiss.py:116-130)
//   K7-9 k_map_*       VoxelHashMap::AddPoints, order-preserving, cap per voxel (kiss.py:129)
//   K10 k_map_prune    RemovePointsFarFromLocation
//   K11 k_map_rebuild  hash-table rebuild (drops tombstones)
//
// Data layout in HBM (all fp64 coordinates, as the reference):
//   points            AoS double[3] per point (24 B), scan order (row-major beam-outer)
//   VDS hash tables   16-byte entries {u64 packed voxel key, u32 smallest claiming point index}, 2x2x2 voxel bricks per 128-byte line
//   local map table   16-B entries {u64 key, i32 block, i32 batch list head}, open addressing, linear probe, 2x2x2 voxel bricks per line
//   local map blocks  one 128-B-aligned block per voxel: {i32 count, i32 slot, pad} | (x, y, z)[P]
//                     (P = 20 -> 512 B = four cache lines; a typical voxel's 7-8 points + header: its first two)
// Deterministic semantics shared with the CPU oracle: first point per voxel in scan order, voxel keeps
// its first P points in scan order, nearest neighbour = strictly-smaller distance in (voxel i,j,k
// ascending, insertion order) candidate order, voxel index by truncation toward zero.
#pragma once
#include "devmath.h"
#include <stdint.h>
#include <type_traits>

#define EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define TOMB_KEY 0xFFFFFFFFFFFFFFFEull
#define KEY_OFF (1 << 20)

#define ERR_KEY_RANGE 1
#define ERR_POOL 2
#define ERR_TABLE 4
#define ERR_VDS_TABLE 8
#define ERR_GN_TIMEOUT 16
// make EXTRA=-DMAP_DIAG: every site that raises ERR_TABLE counts itself in st->dbg_sums[24 + site] and leaves two values of its first
// occurrence in dbg_sums[28..31] (tools/um_diag.py reads them)
#ifdef MAP_DIAG
#define MAP_DIAG_AT(st_, site, a_, b_) do { if (atomicAdd(&(st_)->dbg_sums[24 + (site)], 1.0) == 0.0) { (st_)->dbg_sums[28] = (double)(site); (st_)->dbg_sums[29] = (double)(a_); (st_)->dbg_sums[30] = (double)(b_); } } while (0)
#else
#define MAP_DIAG_AT(st_, site, a_, b_) do { } while (0)
#endif

struct TabEnt {
    unsigned long long key;
    int blk;   // -1 = none, else block id (low 24 bits) | stored point count (high 8 bits)
    int head;  // head of this batch's insertion list (map update only)
};
#define BLK_ID_MASK 0x00FFFFFF

// per-scan voxel-downsampling table: the voxel and the smallest point index that maps to it, one line for claim and bid
struct VdsEnt {
    unsigned long long key;
    unsigned vmin;
    unsigned pad;
};
struct ScanStats {  // mirrors ptl_icp_stats
    double sigma, err_dt, err_drot;
    int iterations, n_corr_last;
    long long n_in, n_valid, n_down, n_src, sum_cand, map_voxels, map_points;
};

struct DevState {
    // per-scan counters
    int n_in, n_valid, n_down, n_src;
    unsigned gn_epoch;  // bumped before every single-sequence GN launch: part of the flag of its published words
    int gn_iters, gn_ncorr;
    int gn_abort;       // set by the first Gauss-Newton workgroup whose exchange poll ran out: everybody leaves the loop
    int dbg_dead_wg;    // test hook (ptl_icp_debug_stall_workgroup): this workgroup of the next GN launches returns at once; -1 = none
    long long gn_cand;
    double gn_max_dist, gn_kernel;
    double T_icp[16];
    double dbg_sums[32];
    long long gn_phase_clk[8];  // accumulated s_memtime ticks of WG0: nn, wg-reduce, barrier, grid-reduce, solve
    long long seq_clk[8];       // free-running kernel, 100 MHz wall-clock ticks summed over scans: workgroup 0's K0-K4 | wait before GN |
                                // GN | wait after GN | map update; [5] the filter workgroup's step; [6] scans; [7] stage-internal barrier waits of workgroup 0
    // pose history (kiss_icp.KissICP.poses: only first / last two are ever read)
    int n_poses, has_ext_guess;
    double pose_first[16], pose_prev[16], pose_last[16];
    double guess[16], new_pose[16];
    double xi[6];
    int do_deskew, pad1;
    // Threshold.cpp AdaptiveThreshold
    double sigma, sse;
    long long n_samples;
    double model_dev[16];
    int prev_n_in, pending_finish;
    int pro_next;       // free-running kernel: 1 + the scan whose prologue (K0) has already been run by the filter workgroup beside the previous scan's map update, 0 = none
    int n_src_last;     // n_src of the last registered scan (n_src itself belongs to the next scan's K0-K4 already)
    int n_down_ins;     // frame_down points of the scan whose map update is in flight (n_down belongs to the next scan's K3 already)
    int stats_pending;  // scan whose map_voxels / map_points are still to be recorded once its map update is complete, or -1
    // map
    int pool_hw, pool_hw_s;   // high-water marks: full blocks (an id bound, >= n_small) | small blocks
    int free_top, n_live;
    int free_top_s, mig_n;    // free small blocks | small blocks waiting for their move to a full block (this scan)
    long long map_points;
    unsigned tab_used;
    int err_flags;
    // executed-work counters, cumulative over the scans of the sequence (never reset by a scan; ptl_*_exec_counters):
    // what the kernels actually requested from memory, as opposed to the brute-force figures of SURVEY 8(d) -
    //   [0] full 27-voxel searches (8-lane kernel: points the answer cache did not settle)   [1] probe rows rebuilt (27 hash probes each)
    //   [2] stored map points read by those searches   [3] Gauss-Newton iterations   [4] voxel claims of VDS pass 1 (run heads)   [5] of pass 2
    //   [6] source point-iterations (phase A evaluations)   [7] scans
    unsigned long long exec_cnt[8];
    // free-running kernel, cumulative since the cold start (ptl_batch_sched_counters): [0] scans run by a team of another XCD than the
    // sequence's home XCD, [1] scans that ran on another XCD than the scan before (cross-XCD hand-overs), [2] XCC id of the last scan + 1
    unsigned long long sched_cnt[4];
    // 8-lane Gauss-Newton loop, cumulative like exec_cnt: point-iterations settled as "nothing in reach" (the point's last search found its 27
    // voxels empty and it has not left its voxel since: no row evaluation, no search)
    unsigned long long empty_cnt;
};

struct Ctx {
    // algorithm parameters
    double max_range, min_range, vs, vds1, vds2, init_thr, min_motion, conv;
    int P, deskew, max_iter, W;
    // scan input
    const float* in_f32;
    const double* in_f64;
    // range-image input (ouster client.XYZLut equivalent, reference kiss.py:28-29,59-60): xyz = range * dir + off
    const unsigned* in_range;       // [H*W] millimetres, 0 = no return
    const double* lut_dir;          // [H*W][3] unit direction * 0.001 (metres per mm)
    const double* lut_off;          // [H*W][3] metres
    const unsigned char* row_mask;  // [H] 1 = active beam (reduce_active_beams, reference utils.py:328-341), or null
    const double* t01;
    int n_in, n_max;
    // scan work buffers
    double* pts;
    int *slot1, *slot2;
    VdsEnt *vtab1, *vtab2;
    unsigned vmask, vmask2;      // slots - 1 of the pass-1 / pass-2 per-scan voxel tables
    int *bcnt1, *bcnt2;
    double *fd, *src0, *src_cur, *fdw;
    double* coltab;  // [12][W] per-column deskew transforms (R row-major 9, t 3), entry-major
    int *pslot, *nxt, *prank, *plen;
    // map
    TabEnt* tab;
    unsigned tmask;
    unsigned char* blocks;
    unsigned char* big_base;  // blocks + n_small * (SMALL_BYTES - bstride): full block b (>= n_small) at big_base + b * bstride
    int n_small;              // small blocks (ids below it); 0 = every block is a full one
    int* free_stack_s;        // free small ids
    int* mig_list;            // [n_small] small blocks a batch has outgrown (prune -> d_map_migrate)
    int* bhdr;        // [pool_cap][4]  block directory: count | table slot | batch points pending | pad
    double* bfirst;   // [pool_cap][3]  ... and the voxel's first point
    int bstride, pool_cap;
    int* free_stack;
    // GN
    int G;
    // state / outputs
    DevState* st;
    double* traj;        // [T][16] kiss poses
    ScanStats* sstats;   // [T]
    const double* ext_guess;  // device 4x4 or null
    unsigned long long* pc_key;  // [n_max]      multi-pass probe cache: voxel key of source point i at its last probe
    int* pc_pb;                  // [n_max][32]  and the 27 probe results (block id | count << 24, -1 = absent)
    double* pc_ans;              // [n_max][GN8_ANS_ROW]  8-lane kernel: exact answer cache (s0 | the K nearest candidates | distance bound of the others | order ids, counts)
    unsigned long long* gn_rows_ll;  // [2][G][64]      per-workgroup sums as (32 data bits | 32 flag bits) words
    unsigned long long* gn_xsum_ll;  // [2][8][8][64]   per-group sums, one copy per consumer slot
    int overlap_pre;          // the next scan's K0-K4 run beside this scan's map update (own stream): see flush_map_stats
    long long* wg_clk;        // [2 G] diagnostic: ticks each GN workgroup spent in the search phase (all waves / first wave)
    int traj_cap;
};

// One block per voxel: the stored points as (x, y, z) triples, point j at blk_x()[3 j .. 3 j + 2], and nothing else.  HBM serves whole
// 128-byte lines (tools/hip/gather_probe.hip: a scattered 8-byte read costs as much bandwidth as a full line), so a block's first line
// carries its first five points.  What the map update needs per voxel beside the points lives in a DIRECTORY indexed by block id
// (round 5): bhdr[b] = {i32 count, i32 table slot, i32 points of the current batch not yet counted in (free-running kernel:
// insert b -> prune), pad} and bfirst[b] = the voxel's first point (a copy of blk_x()[0..2]).  The prune pass, which looks at the first
// point of EVERY live voxel every scan (VoxelHashMap::RemovePointsFarFromLocation), streams 40 coalesced bytes per voxel from the
// directory instead of pulling one scattered 128-byte line per voxel out of the pool: BASELINE config 5 holds 1.6 M voxels -
// 200 MB of lines per scan (a sixth of its HBM traffic) against 63 MB.  The search never looks at a header (the stored count comes with
// the table entry), so its lines hold points only.
// (Rounds 3-4: header + triples in the block; rounds 1-2: x[P] | y[P] | z[P] | header, three to four lines per visit.)
// Two block classes by id range (round 5; n_small = 0: one class, as before): ids [0, n_small) are SMALL blocks - one 128-byte line,
// SMALL_CAP = 5 points - ids [n_small, pool_cap) are full blocks of bstride bytes (P points).  A voxel starts in a small block and
// moves to a full one when a batch takes it past five points (d_map_migrate, free-running kernel only).  BASELINE config 5 (0.1 m
// voxels) holds 3.6 points per voxel, 77 % of the voxels <= 4: the pool shrinks to about a third and a sparse voxel is ONE line.
#define SMALL_CAP 5
#define SMALL_BYTES 128
template <class CT>
__device__ __forceinline__ double* blk_x(const CT& c, int b) {
    return (double*)(b < c.n_small ? c.blocks + (size_t)b * SMALL_BYTES : c.big_base + (size_t)b * c.bstride);
}
template <class CT>
__device__ __forceinline__ int blk_cap(const CT& c, int b) { return b < c.n_small ? SMALL_CAP : c.P; }
template <class CT>
__device__ __forceinline__ int* blk_hdr(const CT& c, int b) { return c.bhdr + 4 * (size_t)b; }
template <class CT>
__device__ __forceinline__ double* blk_first(const CT& c, int b) { return c.bfirst + 3 * (size_t)b; }

__device__ __forceinline__ unsigned long long mix64(unsigned long long h) {
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
    return h;
}
// Slot of a voxel in an open-addressing table of 16-byte entries: the 2 x 2 x 2 brick of voxels that agree in all but the
// lowest bit of each (offset) index occupies 8 consecutive entries = one 128-byte line, and the bricks are scattered by the
// hash.  A 27-voxel neighbourhood then touches at most 8 lines instead of 27, consecutive returns of a beam (neighbouring
// voxels) meet in the same lines, and every line that arrives carries up to 8 useful entries - HBM moves whole lines
// whatever a lane asks for.  Linear probing walks on from there as before.
__device__ __forceinline__ unsigned brick_slot(unsigned long long key, unsigned mask) {
    const unsigned long long low = (1ull << 42) | (1ull << 21) | 1ull;
    const unsigned fine = (unsigned)(((key >> 42) & 1ull) << 2 | ((key >> 21) & 1ull) << 1 | (key & 1ull));
    return (((unsigned)mix64(key & ~low) << 3) | fine) & mask;
}
// (int)(x / vs) - the voxel index of VoxelHashMap / VoxelDownsample (C truncation) - without the fp64 division in the
// common case: q = x * (1 / vs) differs from the correctly rounded quotient by a few ulp, so both truncate alike
// unless an integer lies within 1e-9 (|q| + 1) of q; only then is the real division evaluated.
__device__ __forceinline__ int voxel_index(double x, double vs, double inv_vs) {
    const double q = x * inv_vs;
    int k = (int)q;
    if (fabs(q - rint(q)) <= 1e-9 * (fabs(q) + 1.0)) k = (int)(x / vs);
    return k;
}

// voxel index by C truncation toward zero, exactly (int)(coordinate / voxel_size)
__device__ __forceinline__ bool vox_key(V3 p, double size, unsigned long long& key, int& kx, int& ky, int& kz) {
    const double inv = 1.0 / size;  // (loop-invariant for the callers: one division per thread, not three per point)
    kx = voxel_index(p.x, size, inv); ky = voxel_index(p.y, size, inv); kz = voxel_index(p.z, size, inv);
    const int ox = kx + KEY_OFF, oy = ky + KEY_OFF, oz = kz + KEY_OFF;
    const bool ok = ((unsigned)ox | (unsigned)oy | (unsigned)oz) < (1u << 21);
    key = ((unsigned long long)ox << 42) | ((unsigned long long)oy << 21) | (unsigned long long)oz;
    return ok;
}
__device__ __forceinline__ unsigned long long pack_key(int kx, int ky, int kz) {
    return ((unsigned long long)(kx + KEY_OFF) << 42) | ((unsigned long long)(ky + KEY_OFF) << 21) |
           (unsigned long long)(kz + KEY_OFF);
}

// ------------------------------------------------------------------------------------------------ K0
// One thread.  kiss_icp.KissICP: deskew twist from its own last two poses, get_adaptive_threshold()
// (Threshold.cpp ComputeThreshold, stateful), initial guess (reference kiss.py:102-105 or the caller's).
// bookkeeping that closes a scan (KissICP.poses.append, kiss.py:130); runs at the start of the next
// scan's prologue or from k_finish_scan when the host wants to read state
// map size after a scan's map update: recorded by the first single-threaded step that is ordered after that update -
// the next prologue (one stream), or the next GN launch / k_finish_scan when the prologue overlaps the update
__device__ __forceinline__ void flush_map_stats(const Ctx& c, DevState* st) {
    const int k = st->stats_pending;
    if (k < 0) return;
    st->stats_pending = -1;
    if (k < c.traj_cap) { c.sstats[k].map_voxels = st->n_live; c.sstats[k].map_points = st->map_points; }
}
__device__ __forceinline__ void finish_pending(const Ctx& c, DevState* st) {
    if (!st->pending_finish) return;
    st->pending_finish = 0;
    const int k = st->n_poses;
    if (!c.overlap_pre) flush_map_stats(c, st);  // single stream: the map update of scan k is complete here
    if (k == 0) for (int i = 0; i < 16; ++i) st->pose_first[i] = st->new_pose[i];
    for (int i = 0; i < 16; ++i) { st->pose_prev[i] = st->pose_last[i]; st->pose_last[i] = st->new_pose[i]; }
    st->n_poses = k + 1;
}
__global__ void k_finish_scan(Ctx c) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    finish_pending(c, c.st);
    flush_map_stats(c, c.st);
}

// the exchange buffers of the GN kernel are zeroed again whenever the 22-bit launch epoch wraps (flags restart)
__device__ __forceinline__ void gn_ll_clear_on_wrap(const Ctx& c) {
    if (c.st->gn_epoch != 0u || !c.gn_rows_ll) return;
    const size_t n_rows = (size_t)2 * c.G * 64, n_xsum = (size_t)2 * 8 * 8 * 64;
    for (size_t i = threadIdx.x; i < n_rows; i += blockDim.x) c.gn_rows_ll[i] = 0ull;
    for (size_t i = threadIdx.x; i < n_xsum; i += blockDim.x) c.gn_xsum_ll[i] = 0ull;
}
// keep_abort: the caller runs beside workgroups that may still poll the sequence's abort word (the free-running kernel's filter
// workgroup runs scan k + 1's prologue during scan k's map update): an abort raised there must not be erased
__device__ __forceinline__ void d_scan_prologue(const Ctx& c, const bool keep_abort = false) {
    DevState* st = c.st;
    // The pose block of the state (first, previous, last, new, model deviation: 80 doubles) is brought into LDS by 80
    // threads at once, one thread does the bookkeeping on it, and the three poses that changed go back in parallel:
    // two memory round trips instead of a chain of about a hundred dependent global accesses by one thread.
    __shared__ double sp[5][16];
    __shared__ int s_closed, s_nposes, s_moved;
    __shared__ double s_err;
    if (threadIdx.x < 80) {
        const int a = threadIdx.x >> 4, k = threadIdx.x & 15;
        const double* src = a == 0 ? st->pose_first : a == 1 ? st->pose_prev : a == 2 ? st->pose_last : a == 3 ? st->new_pose : st->model_dev;
        sp[a][k] = src[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
    // finish_pending() on the LDS copy: KissICP.poses.append of the previous scan (kiss.py:130)
    int n_poses = st->n_poses;
    s_closed = 0;
    if (st->pending_finish) {
        st->pending_finish = 0;
        if (!c.overlap_pre) flush_map_stats(c, st);  // single stream: the map update of that scan is complete here
        if (n_poses == 0) for (int i = 0; i < 16; ++i) sp[0][i] = sp[3][i];
        for (int i = 0; i < 16; ++i) { sp[1][i] = sp[2][i]; sp[2][i] = sp[3][i]; }
        s_closed = (n_poses == 0) ? 2 : 1;
        n_poses += 1;
        st->n_poses = n_poses;
    }
    st->prev_n_in = st->n_in;
    st->n_in = c.n_in;
    st->n_valid = 0; st->n_down = 0; st->n_src = 0;
    st->gn_epoch = (st->gn_epoch + 1u) & 0x3FFFFFu;
    st->gn_iters = 0; st->gn_ncorr = 0; st->gn_cand = 0;
    if (!keep_abort) st->gn_abort = 0;
    st->do_deskew = (c.deskew && n_poses >= 2) ? 1 : 0;
    s_nposes = n_poses;
    __threadfence_block();
    }
    __syncthreads();
    // four independent chains of single-lane fp64 transcendentals, one wavefront each instead of one after the other
    const int n_poses = s_nposes;
    if (threadIdx.x == 0 && st->do_deskew) {  // deskew: xi = Log(P[-2]^-1 P[-1])  (Deskew.cpp)
        double xi[6];
        se3_log(rt_mul(rt_inv(rt_from16(sp[1])), rt_from16(sp[2])), xi);
        for (int i = 0; i < 6; ++i) st->xi[i] = xi[i];
    }
    if (threadIdx.x == 64) {  // has the sensor moved away from its first pose?
        bool moved = false;
        if (n_poses >= 1) {
            Rt d = rt_mul(rt_inv(rt_from16(sp[0])), rt_from16(sp[2]));
            moved = sqrt(d.t[0] * d.t[0] + d.t[1] * d.t[1] + d.t[2] * d.t[2]) > 5.0 * c.min_motion;
        }
        s_moved = moved ? 1 : 0;
    }
    if (threadIdx.x == 128) {  // model error of the last registration (Threshold.cpp ComputeModelError)
        Rt dev = rt_from16(sp[4]);
        const double theta = rot_angle(dev.R);
        s_err = sqrt(dev.t[0] * dev.t[0] + dev.t[1] * dev.t[1] + dev.t[2] * dev.t[2]) + 2.0 * c.max_range * sin(0.5 * theta);
    }
    if (threadIdx.x == 192 && !c.ext_guess) {
        // initial guess: the constant-velocity model here; a caller-supplied one (c.ext_guess) is read by the GN kernel
        // itself, so that whatever produces it (the EKF predict on its own stream) may still be running during K0-K4
        Rt last = rt_from16(sp[2]);
        Rt pred = (n_poses >= 2) ? rt_mul(rt_inv(rt_from16(sp[1])), last) : rt_identity();
        Rt lp = (n_poses >= 1) ? last : rt_identity();
        rt_to16(rt_mul(lp, pred), st->guess);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // adaptive threshold
        double sigma = c.init_thr;
        if (s_moved) {
            const double err = s_err;
            if (err > c.min_motion) { st->sse += err * err; st->n_samples += 1; }
            sigma = (st->n_samples < 1) ? c.init_thr : sqrt(st->sse / (double)st->n_samples);
        }
        st->sigma = sigma;
        st->gn_max_dist = 3.0 * sigma;
        st->gn_kernel = sigma / 3.0;
    }
    __threadfence_block();
    if (s_closed && threadIdx.x < 48) {  // the poses the bookkeeping changed
        const int a = threadIdx.x >> 4, k = threadIdx.x & 15;
        double* dst = a == 0 ? st->pose_first : a == 1 ? st->pose_prev : st->pose_last;
        if (a > 0 || s_closed == 2) dst[k] = sp[a][k];
    }
    __syncthreads();
    gn_ll_clear_on_wrap(c);
    // per-column deskew transforms Exp((j/W - 0.5) xi): only W distinct times exist in a sweep (kiss.py:34-35)
    if (st->do_deskew && c.t01 == nullptr) {
        double xi[6];
        for (int k = 0; k < 6; ++k) xi[k] = st->xi[k];
        for (int j = threadIdx.x; j < c.W; j += blockDim.x) {
            const double sft = (double)j * (1.0 / (double)c.W) - 0.5;
            double x[6];
            for (int k = 0; k < 6; ++k) x[k] = sft * xi[k];
            const Rt M = se3_exp(x);
            double* o = c.coltab + (size_t)j;  // entry k of column j at [k W + j]: neighbouring lanes of K1 read neighbouring words
            for (int k = 0; k < 9; ++k) o[(size_t)k * c.W] = M.R[k];
            for (int k = 0; k < 3; ++k) o[(size_t)(9 + k) * c.W] = M.t[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------ XYZLut
// ouster-sdk client.XYZLut (third-party, [UPSTREAM-KNOWLEDGE]): for pixel (row u, col v)
//   theta_enc = 2 pi (1 - v / W), theta_az = -az_u, phi = alt_u (degrees -> radians), n = beam origin offset (mm)
//   dir = (cos(theta_enc + theta_az) cos phi, sin(theta_enc + theta_az) cos phi, sin phi)
//   off = (n (cos theta_enc - dir.x), n (sin theta_enc - dir.y), -n dir.z)
// then the rigid transform T (lidar->sensor, optionally followed by the extrinsic) rotates both and adds its
// translation to the offset.  Stored scaled to metres: xyz_m = range_mm * dir_m + off_m.
__global__ __launch_bounds__(256) void k_build_lut(int H, int W, const double* alt_deg, const double* az_deg, double n_mm,
                                                   const double* T16_mm, double* dir_out, double* off_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int u = i / W, v = i % W;
    const double kPi = 3.14159265358979323846;
    const double enc = 2.0 * kPi * (1.0 - (double)v / (double)W);
    const double az = -az_deg[u] * kPi / 180.0, phi = alt_deg[u] * kPi / 180.0;
    const double dx = cos(enc + az) * cos(phi), dy = sin(enc + az) * cos(phi), dz = sin(phi);
    const double ox = n_mm * (cos(enc) - dx), oy = n_mm * (sin(enc) - dy), oz = -n_mm * dz;
    const double* T = T16_mm;
    for (int r = 0; r < 3; ++r) {
        dir_out[3 * (size_t)i + r] = (T[4 * r] * dx + T[4 * r + 1] * dy + T[4 * r + 2] * dz) * 1e-3;
        off_out[3 * (size_t)i + r] = (T[4 * r] * ox + T[4 * r + 1] * oy + T[4 * r + 2] * oz + T[4 * r + 3]) * 1e-3;
    }
}

// A stage body works on block `b` of `nb` blocks of U * blockDim.x consecutive points (thread t holds the points
// base + u blockDim.x + t, u < U): a launch of its own passes (blockIdx.x, gridDim.x) and U = 1; the free-running sequence
// kernel (kx_seq_run) walks the blocks of a scan with the workgroups of one sequence and U = 4 - a workgroup that owns a
// CU alone hides memory latency only with independent accesses of its own, so the U points of a thread go through every
// step together: U loads in flight, then U hash probes, then U atomics.  Every thread of the workgroup calls the body (it
// contains workgroup barriers).
struct Slice { int b, nb, clk; };  // clk: this caller's thread 0 records the stage-internal clocks (make STAGES=1)
__device__ __forceinline__ Slice launch_slice() { Slice s; s.b = (int)blockIdx.x; s.nb = (int)gridDim.x; s.clk = 0; return s; }
#define STAGE_MAX_WAVES 16  /* blockDim.x <= 1024 */
#define STAGE_MAX_U 16

// U find-or-claims in a per-scan VDS table in flight together: the compare-and-swap on the hashed slot straight away (it
// returns what was there: empty -> claimed, the key -> found; a look at the slot first would be one more dependent memory
// round trip per pass).  What that does not settle (the slot holds another voxel) walks on by linear probing IN ROUNDS: every
// unsettled point of the thread sends its next compare-and-swap, then all are looked at - a round trip per probe distance.
// (Point by point it was three dependent round trips per collision and u: some lane of a wavefront collides
// for nearly every u, and the claims of a K1 pass took 58 us.)
template <int U>
__device__ __forceinline__ void vds_claim_u(VdsEnt* tab, unsigned mask, const unsigned long long (&key)[U], const bool (&want)[U],
                                            int (&slot)[U]) {
    unsigned sp[U];
    unsigned long long old[U];
    bool pend[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        sp[u] = brick_slot(key[u], mask);
        slot[u] = -1;
        old[u] = key[u];
        if (want[u]) old[u] = atomicCAS(&tab[sp[u]].key, EMPTY_KEY, key[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        pend[u] = want[u] && !(old[u] == EMPTY_KEY || old[u] == key[u]);
        if (want[u] && !pend[u]) slot[u] = (int)sp[u];
    }
    for (unsigned probe = 1; probe <= mask; ++probe) {
        bool anyp = false;
#pragma unroll
        for (int u = 0; u < U; ++u) anyp = anyp || pend[u];
        if (!anyp) break;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (pend[u]) { sp[u] = (sp[u] + 1) & mask; old[u] = atomicCAS(&tab[sp[u]].key, EMPTY_KEY, key[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (pend[u] && (old[u] == EMPTY_KEY || old[u] == key[u])) { slot[u] = (int)sp[u]; pend[u] = false; }
    }
}
// the claimed slots' bids: slot value = the smallest point index that maps to the voxel (no result is needed: the
// atomics are sent and not waited for)
template <int U>
__device__ __forceinline__ void vds_bid_u(VdsEnt* tab, const int (&slot)[U], const bool (&want)[U], const int (&idx)[U], int* err_flags) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!want[u]) continue;
        if (slot[u] < 0) atomicOr(err_flags, ERR_VDS_TABLE);
        else atomicMin(&tab[slot[u]].vmin, (unsigned)idx[u]);
    }
}
// number of set flags over the workgroup's U * blockDim.x points (every thread gets it)
template <int U>
__device__ __forceinline__ int block_count_u(const bool (&f)[U]) {
    __shared__ int cnt[STAGE_MAX_WAVES];
    int mine = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) mine += __popcll(__ballot(f[u]));
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    int tot = 0;
    for (int k = 0; k < ((int)blockDim.x >> 6); ++k) tot += cnt[k];
    __syncthreads();
    return tot;
}

// ------------------------------------------------------------------------------------------------ K1
#ifdef SEQ_STAGE_CLOCKS
#define K1_CLK(i) do { __builtin_amdgcn_s_waitcnt(0); const long long n_ = (long long)wall_clock64(); if (k1_me) st->dbg_sums[i] += (double)(n_ - k1_t); k1_t = n_; } while (0)
#else
#define K1_CLK(i) do { } while (0)
#endif
// the raw f32 points of one block, as a pass of K1 wants them: the free-running kernel requests the NEXT pass's while the
// current pass waits for its hash atomics (one dependent memory round trip per pass less)
template <int U> struct RawF32 { float v[U][3]; };
template <int U>
__device__ __forceinline__ RawF32<U> k1_load_raw(const Ctx& c, const Slice sl) {
    RawF32<U> r;
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = base + u * BS;
        const bool in = i < c.n_in;
        r.v[u][0] = in ? c.in_f32[3 * (size_t)i] : 0.0f; r.v[u][1] = in ? c.in_f32[3 * (size_t)i + 1] : 0.0f; r.v[u][2] = in ? c.in_f32[3 * (size_t)i + 2] : 0.0f;
    }
    return r;
}
template <int U>
__device__ __forceinline__ void d_deskew_vds1(const Ctx& c, const Slice sl, const RawF32<U>* pre = nullptr, int* wave_valid = nullptr) {
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    DevState* st = c.st;
#ifdef SEQ_STAGE_CLOCKS
    long long k1_t = (long long)wall_clock64();
    const bool k1_me = sl.clk != 0 && threadIdx.x == 0;  // (workgroup 0 of the team)
#endif
    const int do_deskew = st->do_deskew;
    int idx[U];
    bool valid[U], keyed[U];
    unsigned long long key[U];
    V3 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = base + u * BS;
    // (the pass-2 voxel slots of the previous scan are released by their winners in K4 since round 4 - until then this pass began
    // with a read of every point's slot word and the stores that release the winners' entries: 34 of K1's 343 us)
    K1_CLK(20);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = idx[u];
        p[u] = v3(0.0, 0.0, 0.0);
        if (i < c.n_in) {
            if (c.in_range) {
                unsigned rg = c.in_range[i];
                if (c.row_mask && !c.row_mask[i / c.W]) rg = 0;
                const double r = (double)rg;
                const double* d = c.lut_dir + 3 * (size_t)i;
                const double* o = c.lut_off + 3 * (size_t)i;
                p[u] = rg ? v3(r * d[0] + o[0], r * d[1] + o[1], r * d[2] + o[2]) : v3(0.0, 0.0, 0.0);
            } else if (c.in_f32) {
                if (pre) p[u] = v3((double)pre->v[u][0], (double)pre->v[u][1], (double)pre->v[u][2]);
                else p[u] = v3((double)c.in_f32[3 * (size_t)i], (double)c.in_f32[3 * (size_t)i + 1], (double)c.in_f32[3 * (size_t)i + 2]);
            } else {
                p[u] = v3(c.in_f64[3 * (size_t)i], c.in_f64[3 * (size_t)i + 1], c.in_f64[3 * (size_t)i + 2]);
            }
        }
    }
    Rt Mcol[2];
    const bool col_period2 = U > 2 && ((2 * BS) % c.W) == 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = idx[u];
        valid[u] = false; keyed[u] = false; key[u] = EMPTY_KEY;
        if (i < c.n_in) {
            if (do_deskew) {
                if (c.t01) {
                    const double s = c.t01[i] - 0.5;
                    double x[6];
                    for (int k = 0; k < 6; ++k) x[k] = s * st->xi[k];
                    p[u] = rt_apply(se3_exp(x), p[u]);
                } else {
                    // (2 blockDim) % W == 0 (512 threads, 1024 columns): the thread's points u, u + 2, ... lie in one column - its
                    // transform is loaded once for the even and once for the odd u instead of 12 words per point
                    Rt& M = (u & 1) ? Mcol[1] : Mcol[0];
                    if (u < 2 || !col_period2) {
                        const double* m = c.coltab + (size_t)(i % c.W);
                        for (int k = 0; k < 9; ++k) M.R[k] = m[(size_t)k * c.W];
                        for (int k = 0; k < 3; ++k) M.t[k] = m[(size_t)(9 + k) * c.W];
                    }
                    p[u] = rt_apply(M, p[u]);
                }
            }
            const double r = sqrt(p[u].x * p[u].x + p[u].y * p[u].y + p[u].z * p[u].z);
            valid[u] = (r < c.max_range) && (r > c.min_range);
            if (valid[u]) {
                c.pts[3 * (size_t)i] = p[u].x; c.pts[3 * (size_t)i + 1] = p[u].y; c.pts[3 * (size_t)i + 2] = p[u].z;
                int kx, ky, kz;
                keyed[u] = vox_key(p[u], c.vds1, key[u], kx, ky, kz);
                if (!keyed[u]) atomicOr(&st->err_flags, ERR_KEY_RANGE);
            }
        }
    }
    // Neighbours along a beam fall into the same voxel in long runs (hundreds of returns close to the sensor), and
    // same-address atomics serialise at the memory side.  Only the first lane of each run of equal keys within the
    // wavefront - the lowest index of the run, the only one that can win - claims the slot and bids for it; the others
    // take the slot from it.
    K1_CLK(21);
    {
        const int lane = threadIdx.x & 63;
        bool head[U];
        int slot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long prev = __shfl_up(key[u], 1);
            const bool prev_keyed = __shfl_up(keyed[u] ? 1 : 0, 1) != 0;
            head[u] = keyed[u] && (lane == 0 || !prev_keyed || prev != key[u]);
        }
        vds_claim_u<U>(c.vtab1, c.vmask, key, head, slot);
        {   // executed-work counter: voxel claims (one per run head)
            int nh = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) nh += __popcll(__ballot(head[u]));
            if (lane == 0 && nh) atomicAdd(&st->exec_cnt[4], (unsigned long long)nh);
        }
        K1_CLK(22);
        vds_bid_u<U>(c.vtab1, slot, head, idx, &st->err_flags);
        K1_CLK(23);
        // Only the HEAD of a run of equal keys can be its voxel's first point in scan order (every other point of the run has a lower
        // index of the same voxel right beside it in this wavefront), so only heads keep their slot: K3 then reads the slot's winner
        // for 46 k points of a sweep instead of 120 k (scattered 16-byte reads), and the heads' slots need not be handed down the runs
        // (a ballot, a count and a shuffle per point).  Round 6.
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (idx[u] < c.n_in) c.slot1[idx[u]] = head[u] ? slot[u] : -1;
    }
    K1_CLK(24);
    if (wave_valid) {
        // the caller walks many blocks (free-running kernel): this wavefront's valid points are summed in a register over all of them and
        // added to the scan's counter once at the end - no workgroup-wide count (two barriers) and no atomic per block (55 of K1's 343 us)
        int nvw = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) nvw += __popcll(__ballot(valid[u]));
        *wave_valid += nvw;
    } else {
        const int nv = block_count_u<U>(valid);
        if (threadIdx.x == 0 && nv) atomicAdd(&st->n_valid, nv);
    }
    K1_CLK(25);
}

// the per-point word a pass of K2 / K3 / K4 starts from (slot1 / slot2), requested a pass ahead like K1's raw points
template <int U> struct PreI32 { int v[U]; };
template <int U>
__device__ __forceinline__ PreI32<U> stage_load_i32(const int* arr, int n, const Slice sl) {
    PreI32<U> r;
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = base + u * BS; r.v[u] = (i < n) ? arr[i] : -1; }
    return r;
}

// exclusive prefix of per-block counts + in-block rank (scan order preserved)
// (integer sums: any grouping gives the same value)
// (in two steps: the reads of the block counts are requested at the top of a pass, together with its other loads, and summed
// where the offset is needed - one dependent memory round trip per pass less)
__device__ __forceinline__ int block_offset_part(const int* bcnt, int b) {
    int s = 0;
    for (int k = threadIdx.x; k < b; k += (int)blockDim.x) s += bcnt[k];
    return s;
}
__device__ __forceinline__ int block_offset(int s) {
    __shared__ int red[STAGE_MAX_WAVES];
    const int nw = (int)blockDim.x >> 6;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    int tot = 0;
    for (int k = 0; k < nw; ++k) tot += red[k];
    __syncthreads();
    return tot;
}
// rank of each flagged point among the block's flagged points in point order (sub-block u before u + 1, threads in order)
template <int U>
__device__ __forceinline__ void block_rank_u(const bool (&f)[U], int (&rk)[U], int& total) {
    __shared__ int red[STAGE_MAX_U][STAGE_MAX_WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    int within[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long long m = __ballot(f[u]);
        within[u] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) red[u][wv] = __popcll(m);
    }
    __syncthreads();
    int running = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int off = 0, tot = 0;
        for (int k = 0; k < nw; ++k) { const int v = red[u][k]; if (k < wv) off += v; tot += v; }
        rk[u] = running + off + within[u];
        running += tot;
    }
    total = running;
    __syncthreads();
}
// Ordered compaction in ONE pass over the blocks (free-running kernel): a block publishes its own count as (tag << 16 | count) -
// tag = the scan's number, so whatever an earlier scan left in the word is not taken for this scan's - and then adds up the counts
// of the blocks before it, waiting for those that are not there yet.  The team's workgroups walk their blocks in ascending
// order (block b of workgroup w: w, w + n, ...) and every workgroup is resident, so whoever is waited for is at work on an
// earlier block of its own: no cycle.  The separate counting pass over all points (and its team barrier) is gone.
// Bounded like every wait of the kernel: a budget of polls, then the sequence's time-out flag (the caller's next team barrier
// sees the abort word and the team leaves).
#define LOOKBACK_POLLS (1u << 22)
__device__ __forceinline__ void lookback_publish(int* bcnt, int b, unsigned tag, int count) {
    if (threadIdx.x == 0) __hip_atomic_store(&bcnt[b], (int)((tag << 16) | (unsigned)count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int lookback_offset(const int* bcnt, int b, unsigned tag, DevState* st) {
    int s = 0;
    bool bad = false;
    for (int k = threadIdx.x; k < b; k += (int)blockDim.x) {
        unsigned polls = 0;
        for (;;) {
            const unsigned v = (unsigned)__hip_atomic_load(&bcnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 16) == tag) { s += (int)(v & 0xFFFFu); break; }
            if (++polls > LOOKBACK_POLLS || ((polls & 1023u) == 0u && __hip_atomic_load(&st->gn_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { bad = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (bad) { atomicOr(&st->err_flags, ERR_GN_TIMEOUT); __hip_atomic_store(&st->gn_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return block_offset(s);
}
__device__ __forceinline__ unsigned lookback_tag(int scan_k) { return ((unsigned)scan_k % 0xFFFFu) + 1u; }  // 1 .. 65535 (0 = never published: the arrays are zeroed at a reset); a block holds <= 16 x 1024 points: 16 bits of count
static_assert(STAGE_MAX_U * 1024 < (1 << 16), "the look-back word keeps a block's count in 16 bits");

// ------------------------------------------------------------------------------------------------ K2
// VoxelDownsample(0.5 vs) decided: the winner of a voxel is the point whose index stands in its slot.  The drivers that launch
// every stage as a kernel of its own count the winners per block here (the prefix of K3); the free-running kernel does not run
// this pass at all (K3 counts for itself and looks back).
template <int U>
__device__ __forceinline__ void d_count_w1(const Ctx& c, const Slice sl) {
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    int idx[U], s1[U];
    bool w1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { idx[u] = base + u * BS; s1[u] = (idx[u] < c.n_in) ? c.slot1[idx[u]] : -1; }
    {
        unsigned vm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vm[u] = (s1[u] >= 0) ? c.vtab1[s1[u]].vmin : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) w1[u] = (s1[u] >= 0) && (vm[u] == (unsigned)idx[u]);
    }
    const int n1 = block_count_u<U>(w1);
    if (threadIdx.x == 0) c.bcnt1[sl.b] = n1;
}

// ------------------------------------------------------------------------------------------------ K3
// frame_downsample = the pass-1 winners in scan order.  LB: one pass with look-back (tag = lookback_tag(scan)); else the block
// counts of d_count_w1.
template <int U, bool LB>
__device__ __forceinline__ void d_compact_fd(const Ctx& c, const Slice sl, const PreI32<U>* pre = nullptr, unsigned tag = 0u) {
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    int idx[U], s1[U], rk[U];
    bool w1[U];
    int off_part = 0;
    if (!LB) off_part = block_offset_part(c.bcnt1, sl.b);
#pragma unroll
    for (int u = 0; u < U; ++u) { idx[u] = base + u * BS; s1[u] = pre ? pre->v[u] : ((idx[u] < c.n_in) ? c.slot1[idx[u]] : -1); }
    {
        unsigned vm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vm[u] = (s1[u] >= 0) ? c.vtab1[s1[u]].vmin : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) w1[u] = (s1[u] >= 0) && (vm[u] == (unsigned)idx[u]);
    }
    double q[U][3];  // every winner's point requested before anything waits or is stored
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = (size_t)idx[u];
        if (w1[u]) { q[u][0] = c.pts[3 * i]; q[u][1] = c.pts[3 * i + 1]; q[u][2] = c.pts[3 * i + 2]; }
    }
    int total;
    block_rank_u<U>(w1, rk, total);
    int off;
    if (LB) {
        lookback_publish(c.bcnt1, sl.b, tag, total);
        off = lookback_offset(c.bcnt1, sl.b, tag, c.st);
    } else {
        off = block_offset(off_part);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!w1[u]) continue;
        const size_t o = (size_t)(off + rk[u]) * 3;
        c.fd[o] = q[u][0]; c.fd[o + 1] = q[u][1]; c.fd[o + 2] = q[u][2];
        // release the pass-1 slot for the next scan (only its winner touches it)
        c.vtab1[s1[u]].key = EMPTY_KEY;
        c.vtab1[s1[u]].vmin = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0 && sl.b == sl.nb - 1) c.st->n_down = off + total;
}

// ------------------------------------------------------------------------------------------------ K3b
// VoxelDownsample(1.5 vs) of frame_downsample, on the COMPACT array (N_d ~ 35 k entries instead of a pass over the 131 k raw
// indices of which a quarter are winners): entry j claims its coarse voxel, the smallest j in a voxel wins - the first in
// scan order, as before (frame_downsample is in scan order).  slot2[j] = the voxel's slot.  Runs as in pass 1: consecutive
// entries of a wavefront mostly share the coarser voxel; only the first of a run of equal keys claims and bids.
template <int U>
__device__ __forceinline__ void d_vds2_fd(const Ctx& c, const Slice sl) {
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    const int nd = c.st->n_down;
    int idx[U];
    bool act[U];
    unsigned long long key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { idx[u] = base + u * BS; act[u] = idx[u] < nd; key[u] = EMPTY_KEY; }
    {
        double q[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = (size_t)idx[u];
            q[u][0] = q[u][1] = q[u][2] = 0.0;
            if (act[u]) { q[u][0] = c.fd[3 * j]; q[u][1] = c.fd[3 * j + 1]; q[u][2] = c.fd[3 * j + 2]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!act[u]) continue;
            int kx, ky, kz;
            vox_key(v3(q[u][0], q[u][1], q[u][2]), c.vds2, key[u], kx, ky, kz);
        }
    }
    const int lane = threadIdx.x & 63;
    bool head[U];
    int slot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long long prev = __shfl_up(key[u], 1);
        const bool prev_act = __shfl_up(act[u] ? 1 : 0, 1) != 0;
        head[u] = act[u] && (lane == 0 || !prev_act || prev != key[u]);
    }
    vds_claim_u<U>(c.vtab2, c.vmask2, key, head, slot);
    {
        int nh = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) nh += __popcll(__ballot(head[u]));
        if (lane == 0 && nh) atomicAdd(&c.st->exec_cnt[5], (unsigned long long)nh);
    }
    vds_bid_u<U>(c.vtab2, slot, head, idx, &c.st->err_flags);
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (act[u]) c.slot2[idx[u]] = head[u] ? slot[u] : -1;  // (only a run's head can win its voxel: see K1)
}
// ... and its winners counted per block (the drivers with one launch per stage; the free-running kernel looks back in K4)
template <int U>
__device__ __forceinline__ void d_count_w2(const Ctx& c, const Slice sl) {
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    const int nd = c.st->n_down;
    bool w2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = base + u * BS;
        const int s2 = (j < nd) ? c.slot2[j] : -1;
        w2[u] = s2 >= 0 && c.vtab2[s2].vmin == (unsigned)j;
    }
    const int n2 = block_count_u<U>(w2);
    if (threadIdx.x == 0) c.bcnt2[sl.b] = n2;
}

// ------------------------------------------------------------------------------------------------ K4
// source = the pass-2 winners in scan order, over the compact frame_downsample; a winner releases its voxel slot for the next scan
template <int U, bool LB>
__device__ __forceinline__ void d_compact_src(const Ctx& c, const Slice sl, unsigned tag = 0u) {
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    const int nd = c.st->n_down;
    int idx[U], s2[U], rk[U];
    bool w2[U];
    int off_part = 0;
    if (!LB) off_part = block_offset_part(c.bcnt2, sl.b);
#pragma unroll
    for (int u = 0; u < U; ++u) { idx[u] = base + u * BS; s2[u] = (idx[u] < nd) ? c.slot2[idx[u]] : -1; }
    {
        unsigned vm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vm[u] = (s2[u] >= 0) ? c.vtab2[s2[u]].vmin : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) w2[u] = (s2[u] >= 0) && (vm[u] == (unsigned)idx[u]);
    }
    double q[U][3];  // every winner's point requested before anything waits or is stored
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = (size_t)idx[u];
        if (w2[u]) { q[u][0] = c.fd[3 * j]; q[u][1] = c.fd[3 * j + 1]; q[u][2] = c.fd[3 * j + 2]; }
    }
    int total;
    block_rank_u<U>(w2, rk, total);
    int off;
    if (LB) {
        lookback_publish(c.bcnt2, sl.b, tag, total);
        off = lookback_offset(c.bcnt2, sl.b, tag, c.st);
    } else {
        off = block_offset(off_part);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!w2[u]) continue;
        const size_t o = (size_t)(off + rk[u]) * 3;
        c.src0[o] = q[u][0]; c.src0[o + 1] = q[u][1]; c.src0[o + 2] = q[u][2];
        c.vtab2[s2[u]].key = EMPTY_KEY;
        c.vtab2[s2[u]].vmin = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0 && sl.b == sl.nb - 1) c.st->n_src = off + total;
}
// ------------------------------------------------------------------------------------------------ K5
// map probe: block id of voxel `key` or -1
template <class CT>
__device__ __forceinline__ int map_find(const CT& c, unsigned long long key) {
    unsigned s = brick_slot(key, c.tmask);
    for (unsigned probe = 0; probe <= c.tmask; ++probe) {
        const TabEnt e = c.tab[s];
        if (e.key == key) return e.blk;  // packed: block id | count << 24
        if (e.key == EMPTY_KEY) return -1;
        s = (s + 1) & c.tmask;
    }
    return -1;
}

// The correspondence gate of Registration.cpp is `norm < max_correspondence_distance`, i.e. sqrt(d2) < M on the squared
// distance the search returns.  sqrt is correctly rounded and monotone, so the distances that pass are exactly those
// below T = min { x : sqrt(x) >= M }: the gate becomes one comparison per point and iteration instead of an fp64
// square root, with the same outcome for every input.  T is M * M up to a few ulp; it is found once per launch.
__device__ __forceinline__ double sqrt_gate(double M) {
    if (!(M > 0.0) || !(M < 1.0e150)) return (M > 0.0) ? M : 0.0;  // +inf: everything passes; <= 0 or NaN: nothing does
    double T = M * M;
    for (int k = 0; k < 64 && sqrt(__longlong_as_double(__double_as_longlong(T) - 1)) >= M; ++k)
        T = __longlong_as_double(__double_as_longlong(T) - 1);
    for (int k = 0; k < 64 && sqrt(T) < M; ++k) T = __longlong_as_double(__double_as_longlong(T) + 1);
    return T;
}

// 32 lanes cooperate on one source point.  Lanes 0..26 probe the 27 neighbour voxels in (i,j,k) ascending order
// (upstream's visiting order); a probe returns block id and point count together.  `ck` / `cblk` carry the centre
// voxel and this lane's probe result from the previous call: the map is constant during a Gauss-Newton loop, so
// when the point has not left its voxel the probes are skipped.  The point's own voxel is scanned first (lane l <->
// its l-th stored point, one coalesced read per coordinate); a neighbour voxel whose box lies farther than the
// best distance found there cannot hold a strictly closer point and is dropped - exact, because a candidate
// replaces the incumbent only when strictly closer or equally close and earlier in visiting order, and a dropped
// voxel can offer neither.  The surviving voxels are scanned two at a time.
// Returns in every lane of the group: best squared distance, best target, found flag; adds this lane's probe
// count to ncand (the caller sums the lanes).
// Minimum over the 32 lanes of a group, result in every lane.  Four of the five butterfly steps are DPP moves inside the
// VALU (quad permutes, then the half-row / row mirrors: every lane of a quad already holds the quad's minimum, so
// pairing mirrored lanes is as good as xor 4 / xor 8); only the step across the two 16-lane rows goes through the
// LDS crossbar.  min is exact and commutative: the value is the one the plain xor butterfly gives.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double group_min32(double v) {
    v = fmin(v, dpp_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmin(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmin(v, dpp_f64<0x141>(v));  // row_half_mirror
    v = fmin(v, dpp_f64<0x140>(v));  // row_mirror
    return fmin(v, __shfl_xor(v, 16));
}
__device__ __forceinline__ unsigned group_min32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
    return min(v, (unsigned)__shfl_xor((int)v, 16));
}

__device__ __forceinline__ int group_sum32(int v) {  // integer sum over the 32 lanes of a group (order-free), same moves
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return v + __shfl_xor(v, 16);
}

// probe of this lane's neighbour voxel (lanes 0..26, (i,j,k) ascending): packed block id | count << 24, or -1
template <class CT>
__device__ __forceinline__ int nn_probe32(const CT& c, int kx, int ky, int kz, int lane32) {
    if (lane32 >= 27) return -1;
    const int di = lane32 / 9 - 1, dj = (lane32 / 3) % 3 - 1, dk = lane32 % 3 - 1;
    return map_find(c, pack_key(kx + di, ky + dj, kz + dk));
}
// The scan proper, given the probe results.  First round: the point's own voxel together with `lastv`, the voxel
// (lane index 0..26) that held the winner the last time this point was searched - between two Gauss-Newton iterations
// the point moves by a fraction of a voxel, so the nearest neighbour is nearly always in one of the two and the bound
// they give drops every other voxel: one memory round trip on the dependent chain instead of two or three.  The
// result does not depend on the order voxels are looked at: candidates compare by (distance, visiting order), and a
// voxel is dropped only when its box lies strictly farther than a distance already found.
template <int PC, class CT>
__device__ __forceinline__ bool nn_scan32(const CT& c, V3 s, int kx, int ky, int kz, int pb, int lane32, int gbase,
                                          V3& best, double& best_d2, int& lastv) {
    const int di = lane32 / 9 - 1, dj = (lane32 / 3) % 3 - 1, dk = lane32 % 3 - 1;
    const int cnt0 = (pb < 0) ? 0 : (int)((unsigned)pb >> 24);
    double bd = 1.7976931348623157e308;
    unsigned border = 0xFFFFFFFFu;  // visiting order of a candidate: voxel lane * 32 + slot
    V3 bp = v3(0, 0, 0);
    const int lv = (lastv != 13) ? lastv : -1;  // uniform over the group; -1: none
    bool keep = lane32 != 13 && lane32 != lv && cnt0 > 0;
    const int pbc = __shfl(pb, gbase + 13);
    const int pbl = (lv >= 0) ? __shfl(pb, gbase + lv) : -1;
    if (pbc >= 0 || pbl >= 0) {  // uniform over the group (a stored voxel holds at least one point)
        const bool ac = pbc >= 0 && lane32 < (int)((unsigned)pbc >> 24), al = pbl >= 0 && lane32 < (int)((unsigned)pbl >> 24);
        double cx = 0.0, cy = 0.0, cz = 0.0, lx = 0.0, ly = 0.0, lz = 0.0;
        if (ac) { const double* X = blk_x(c, pbc & BLK_ID_MASK); cx = X[3 * lane32]; cy = X[3 * lane32 + 1]; cz = X[3 * lane32 + 2]; }
        if (al) { const double* X = blk_x(c, pbl & BLK_ID_MASK); lx = X[3 * lane32]; ly = X[3 * lane32 + 1]; lz = X[3 * lane32 + 2]; }
        if (ac) {
            const double dx = cx - s.x, dy = cy - s.y, dz = cz - s.z;
            bd = dx * dx + dy * dy + dz * dz;
            border = 13u * 32u + (unsigned)lane32;
            bp = v3(cx, cy, cz);
        }
        if (al) {
            const double dx = lx - s.x, dy = ly - s.y, dz = lz - s.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            const unsigned id = (unsigned)(lv * 32 + lane32);
            if (d2 < bd || (d2 == bd && id < border)) { bd = d2; border = id; bp = v3(lx, ly, lz); }
        }
        // distance from the point to this lane's voxel box.  Under truncation toward zero index 0 spans (-vs, vs),
        // v > 0 spans [v vs, (v+1) vs), v < 0 spans ((v-1) vs, v vs]; 1 nm of slack covers the rounding of x / vs
        const int d[3] = {di, dj, dk}, k[3] = {kx, ky, kz};
        const double x[3] = {s.x, s.y, s.z};
        double gap2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int v = k[a] + d[a];
            const int bi = (d[a] > 0) ? (v > 0 ? v : v - 1) : (v < 0 ? v : v + 1);  // near face of the box
            double g = ((double)bi * c.vs - x[a]) * (double)d[a] - 1e-9;
            g = (d[a] == 0) ? 0.0 : fmax(g, 0.0);
            gap2 += g * g;
        }
        const double m = group_min32(bd);
        if (keep && gap2 > m) keep = false;  // a dropped voxel: every point in it is at least this far
    }
    unsigned todo = (unsigned)(__ballot(keep) >> gbase);  // this group's surviving voxels, visited in ascending order
    while (todo) {
        const int v0 = __ffs(todo) - 1;
        todo &= todo - 1;
        const int v1 = todo ? __ffs(todo) - 1 : v0;
        const bool two = todo != 0;
        todo &= todo - 1;
        const int pb0 = __shfl(pb, gbase + v0), pb1 = __shfl(pb, gbase + v1);
        const bool a0 = lane32 < (int)((unsigned)pb0 >> 24), a1 = two && lane32 < (int)((unsigned)pb1 >> 24);
        double q0x = 0.0, q0y = 0.0, q0z = 0.0, q1x = 0.0, q1y = 0.0, q1z = 0.0;
        if (a0) { const double* X = blk_x(c, pb0 & BLK_ID_MASK); q0x = X[3 * lane32]; q0y = X[3 * lane32 + 1]; q0z = X[3 * lane32 + 2]; }
        if (a1) { const double* X = blk_x(c, pb1 & BLK_ID_MASK); q1x = X[3 * lane32]; q1y = X[3 * lane32 + 1]; q1z = X[3 * lane32 + 2]; }
        if (a0) {
            const double dx = q0x - s.x, dy = q0y - s.y, dz = q0z - s.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            const unsigned id = (unsigned)(v0 * 32 + lane32);
            if (d2 < bd || (d2 == bd && id < border)) { bd = d2; border = id; bp = v3(q0x, q0y, q0z); }
        }
        if (a1) {
            const double dx = q1x - s.x, dy = q1y - s.y, dz = q1z - s.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            const unsigned id = (unsigned)(v1 * 32 + lane32);
            if (d2 < bd || (d2 == bd && id < border)) { bd = d2; border = id; bp = v3(q1x, q1y, q1z); }
        }
    }
    // lexicographic (d2, visiting order) minimum over the 32 lanes: the distance, then - only if two candidates are
    // exactly equally far - the order among the lanes that hold it; the winner lane hands out its point
    const double m = group_min32(bd);
    const bool found = m < 1.7976931348623157e308;  // some lane holds a candidate
    unsigned win = (unsigned)(__ballot(bd == m) >> gbase);
    if (found && (win & (win - 1u))) {
        const unsigned bo = group_min32((bd == m) ? border : 0xFFFFFFFFu);
        win = (unsigned)(__ballot(bd == m && border == bo) >> gbase);
    }
    const int wl = gbase + (win ? __ffs(win) - 1 : 0);
    best = v3(__shfl(bp.x, wl), __shfl(bp.y, wl), __shfl(bp.z, wl));
    best_d2 = m;
    lastv = found ? (int)((unsigned)__shfl((int)border, wl) >> 5) : -1;
    return found;
}
// probe (or reuse this lane's previous probe while the point has not left its voxel) + scan.  `lastv` (the winner's
// voxel of the previous call, -1 = none) is meaningful only together with the cached probes: it is dropped when the
// point changes voxel, and callers that cannot carry it per point pass -1.
template <int PC, class CT>
__device__ __forceinline__ bool nn_search32(const CT& c, V3 s, int lane32, int gbase, V3& best, double& best_d2,
                                            int& ncand, unsigned long long& ck, int& cblk, bool use_cache, int& lastv) {
    const double inv_vs = 1.0 / c.vs;   // loop-invariant
    const int kx = voxel_index(s.x, c.vs, inv_vs), ky = voxel_index(s.y, c.vs, inv_vs), kz = voxel_index(s.z, c.vs, inv_vs);
    const unsigned long long key = pack_key(kx, ky, kz);
    const bool hit = use_cache && key == ck;
    const int pb = hit ? cblk : nn_probe32(c, kx, ky, kz, lane32);
    if (!hit) lastv = -1;
    ck = key;
    cblk = pb;
    ncand += (pb < 0) ? 0 : (int)((unsigned)pb >> 24);
    return nn_scan32<PC>(c, s, kx, ky, kz, pb, lane32, gbase, best, best_d2, lastv);
}

// column `idx` of the 3x7 matrix [ I | -hat(s) | r ]: the Jacobian J = [I | -hat(s)] of Registration.cpp
// BuildLinearSystem plus the residual, so that JTJ(a,b) = w col(a).col(b) and JTr(a) = w col(a).col(6)
__device__ __forceinline__ V3 jcol(int idx, V3 s, V3 r) {
    switch (idx) {
        case 0: return v3(1.0, 0.0, 0.0);
        case 1: return v3(0.0, 1.0, 0.0);
        case 2: return v3(0.0, 0.0, 1.0);
        case 3: return v3(0.0, -s.z, s.y);
        case 4: return v3(s.z, 0.0, -s.x);
        case 5: return v3(-s.y, s.x, 0.0);
        default: return r;
    }
}

// solve6_ldlt() spread over a wavefront: lane i < 6 owns row i of the factorisation; pivots, the pivot row and the
// finished unknowns travel by v_readlane.  Every product and difference is the one solve6_ldlt() forms, in the same
// order, so dx is bit-identical to the single-lane version - with about half the instructions on the critical
// path of each Gauss-Newton iteration.  All 64 lanes must call it; all return dx.
__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <class T> __device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long u = (unsigned long long)(size_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return (T*)(size_t)(((unsigned long long)hi << 32) | lo);
}
// a value that is the same in every lane, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ void solve6_ldlt_wave(const double* s, int lane, double dx[6]) {
    const int i = lane < 6 ? lane : 5;  // lanes >= 6 shadow row 5 (their results are never read)
    double A[6], L[6], D[6];
    // row i of the symmetric matrix from the packed upper triangle: entry (r, c), r <= c, sits at r (11 - r) / 2 + c
#pragma unroll
    for (int cidx = 0; cidx < 6; ++cidx) {
        const int r = i < cidx ? i : cidx, cc = i < cidx ? cidx : i;
        A[cidx] = s[r * (11 - r) / 2 + cc];
        L[cidx] = 0.0;
    }
    double Dmine = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v = A[j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < j) v -= L[k] * lane_bcast(L[k], j) * D[k];
        const double d = lane_bcast(v, j);
        D[j] = d;
        if (i == j) Dmine = d;
        L[j] = (i > j) ? ((d != 0.0) ? v / d : 0.0) : ((i == j) ? 1.0 : 0.0);
    }
    double y = -s[21 + i];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double yk = lane_bcast(y, k);
        if (i > k) y -= L[k] * yk;
    }
    y = (Dmine != 0.0) ? y / Dmine : 0.0;
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double v = lane_bcast(y, r);
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > r) v -= lane_bcast(L[r], k) * dx[k];
        dx[r] = v;
    }
}

// Tsh <- Esh * Tsh (both 3x3 row-major + translation in LDS)
__device__ __forceinline__ void gn_compose(const double* Esh, double* Tsh) {
    Rt e, T;
    for (int k = 0; k < 9; ++k) { e.R[k] = Esh[k]; T.R[k] = Tsh[k]; }
    for (int k = 0; k < 3; ++k) { e.t[k] = Esh[9 + k]; T.t[k] = Tsh[9 + k]; }
    T = rt_mul(e, T);
    for (int k = 0; k < 9; ++k) Tsh[k] = T.R[k];
    for (int k = 0; k < 3; ++k) Tsh[9 + k] = T.t[k];
}

// Source point -> workgroup assignment.  Workgroups with the same (wg & 7) run on one XCD (observed dispatch,
// used for speed only) and share its 4 MB L2, so each such set gets one CONTIGUOUS eighth of the scan-ordered
// source points (a band of beams): the map voxels it probes then stay resident in that XCD's L2.
struct PointWalk {
    int first, last, step;
};
__device__ __forceinline__ PointWalk point_walk(int n, int G, int wg, int NG, int grp) {
    PointWalk w;
    if ((G & 7) == 0) {
        const int chunk = (n + 7) >> 3, x = wg & 7, j = wg >> 3, J = G >> 3;
        w.last = (x + 1) * chunk < n ? (x + 1) * chunk : n;
        if (J > 1 && chunk <= (J - 1) * NG) {
            // the share fits without the group leaders (wg < 8, j == 0): they take no points, so they are already
            // waiting for their members' rows when the first one arrives (see the exchange in k_gn_loop)
            w.first = (j == 0) ? w.last : x * chunk + (j - 1) * NG + grp;
            w.step = (J - 1) * NG;
        } else {
            w.first = x * chunk + j * NG + grp;
            w.step = J * NG;
        }
    } else {
        w.first = wg * NG + grp; w.last = n; w.step = G * NG;
    }
    return w;
}

// kiss.py:116-130 after the ICP: new pose, innovation (err_dt / err_drot), threshold model deviation, stats
// row.  full != 0 => a real scan (trajectory + stats row, closes with finish_pending); 0 => ptl_icp_align.
__device__ __forceinline__ const double* guess_src(const Ctx& c) { return c.ext_guess ? c.ext_guess : c.st->guess; }
__device__ __forceinline__ void gn_post(const Ctx& c, DevState* st, bool map_empty, int full) {
    double graw[16];  // the guess as the caller handed it over: kiss.py:116, :128 invert THAT
    {
        const double* g16 = guess_src(c);
        for (int i = 0; i < 16; ++i) graw[i] = g16[i];
    }
    const Rt guess = rt_project(rt_from16(graw));  // Sophus::SE3d initial_guess(T_guess): as the kernel registered it
    rt_to16(guess, st->guess);
    const Rt np = map_empty ? guess : rt_project(rt_mul(rt_from16(st->T_icp), guess));  // (T_icp * initial_guess).matrix()
    rt_to16(np, st->new_pose);
    if (!full) return;
    // pose_gain = np.linalg.inv(initial_guess) @ new_pose (kiss.py:116, :128): a general 4x4 inverse of the raw guess and a 4x4
    // product.  err_dt is the norm of its translation column; err_drot (scipy's Rotation.from_matrix) and the model deviation
    // (a Sophus::SE3d on the C++ side) see its rotation block through a unit quaternion.
    Rt gain;
    double drot;
    {
        double gi[16], np16[16], g16[16];
        mat4_inv(graw, gi);
        rt_to16(np, np16);
        mat4_mul(gi, np16, g16);
        gain = rt_from16(g16);
        double Qg[9];
        mat3_polar(gain.R, Qg);   // kiss.py:119-120: scipy orthogonalises (the polar factor), then the rotation vector's norm
        drot = rot_angle(Qg);
        gain = rt_project(gain);  // kiss.py:128: the Sophus::SE3d of it (matrix -> quaternion, normalised)
    }
    rt_to16(gain, st->model_dev);
    const int k = st->n_poses;
    if (k < c.traj_cap) {
        for (int i = 0; i < 16; ++i) c.traj[(size_t)k * 16 + i] = st->new_pose[i];
        ScanStats s;
        s.sigma = st->sigma;
        s.err_dt = sqrt(gain.t[0] * gain.t[0] + gain.t[1] * gain.t[1] + gain.t[2] * gain.t[2]);
        s.err_drot = drot;
        s.iterations = st->gn_iters; s.n_corr_last = st->gn_ncorr;
        s.n_in = st->n_in; s.n_valid = st->n_valid; s.n_down = st->n_down; s.n_src = st->n_src;
        s.sum_cand = st->gn_cand; s.map_voxels = 0; s.map_points = 0;  // filled by finish_pending
        c.sstats[k] = s;
    }
    st->pending_finish = 1;
    st->n_down_ins = st->n_down;
    st->n_src_last = st->n_src;
    st->stats_pending = k;
}

// ---- flag-in-word exchange of the 29 sums (single-sequence GN kernel) -------------------------------------------
// A double travels as two 8-byte words, each (32 data bits | 32-bit flag); an aligned 8-byte store is single-copy
// atomic, so a reader that sees the flag sees the data - no acknowledgement wait, no counter, no release word.
// The flag is unique per (launch, iteration): gn_epoch << 10 | iteration + 1 (never 0; the buffers start zeroed and
// are zeroed again when the epoch wraps).  Rows are double-buffered by iteration parity: nobody can be two
// iterations ahead of a reader (a workgroup leaves iteration i only after every leader published i, and a leader
// publishes only after every one of its members did).
#define GN_LL_WORDS 64   /* words per row: 58 used */
#ifndef GN_XSUM_COPIES
#define GN_XSUM_COPIES 8  /* copies of every group sum (power of two <= 8): consumers are spread over them */
#endif
#define GN_LL_SPINS (1u << 22)
__device__ __forceinline__ unsigned gn_flag(unsigned epoch, int it) { return (epoch << 10) | (unsigned)(it + 1); }
// word `w` (0..57) of a row holding the doubles vals[0..28]
__device__ __forceinline__ unsigned long long gn_ll_word(const double* vals, int w, unsigned flag) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(vals[w >> 1]);
    const unsigned half = (w & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
    return (unsigned long long)half | ((unsigned long long)flag << 32);
}
// A poll that runs out of spins (a workgroup that is not resident, a co-tenant holding CUs) must not be repeated
// max_iter times by every workgroup: whoever runs out first raises ERR_GN_TIMEOUT and sets st->gn_abort; every
// spinning wavefront looks at that word every 1024 polls and gives up too, and a workgroup that gave up leaves the loop
// at the end of the iteration.  The host sees the error flag at its next wait.
__device__ __forceinline__ bool gn_abort_seen(const int* abort_word) {
    return __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void gn_raise_abort(DevState* st) {
    atomicOr(&st->err_flags, ERR_GN_TIMEOUT);
    __hip_atomic_store(&st->gn_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// spin until both words of one row carry `flag` (both loads in flight together); *ok = false on a timeout / abort
__device__ __forceinline__ void gn_ll_wait2(const unsigned long long* p0, const unsigned long long* p1, bool second,
                                            unsigned flag, unsigned* h0, unsigned* h1, bool* ok, const int* abort_word) {
    unsigned spins = 0;
    for (;;) {
        const unsigned long long a = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = second ? __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : ((unsigned long long)flag << 32);
        if ((unsigned)(a >> 32) == flag && (unsigned)(b >> 32) == flag) { *h0 = (unsigned)a; *h1 = (unsigned)b; return; }
        ++spins;
        if (spins > GN_LL_SPINS || ((spins & 1023u) == 0u && gn_abort_seen(abort_word))) { *ok = false; *h0 = 0u; *h1 = 0u; return; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// lanes (2e, 2e + 1) of a wavefront hold the low / high half of entry e: the even lane returns the double
__device__ __forceinline__ double gn_ll_join(unsigned half) {
    const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)half, 0xB1, 0xF, 0xF, true);  // lane ^ 1
    return __longlong_as_double((long long)(((unsigned long long)other << 32) | half));  // meaningful in even lanes
}

// Persistent Gauss-Newton loop (Registration.cpp RegisterFrame).  G workgroups, all resident; per iteration the
// workgroups' 29 sums meet through the flag-in-word exchange above (rows -> 8 group leaders -> group sums to everybody,
// fixed summation order), and every workgroup solves the 6x6 system redundantly from the same totals, so all agree
// bit-for-bit on dx and on convergence.  Within a 32-lane group lane k accumulates entry k of the 27 sums (21 JTJ
// upper triangle + 6 JTr); lane 27 counts pairs.
// mode 0: a scan: source = guess * src0, loop to convergence, then the post-ICP bookkeeping.
// mode 1: src_cur given in world frame, one pass, sums exported to st->dbg_sums (teacher-forced entry).
// mode 2: like 0 without touching the trajectory (ptl_icp_align).
// In-kernel phase clocks (tools_phase.py, ptl_icp_gn_phases / ptl_icp_gn_wg_clocks) are compiled in only with
// -DGN_PHASE_CLOCKS (make PHASES=1): eight 64-bit accumulators and the counter reads otherwise compete with the loop's
// live values for scalar and vector registers (the kernel runs at its 128-VGPR cap and spills).
#if defined(GN_PHASE_CLOCKS) || defined(GN_IT0_CLOCK)  /* (make IT0=1: only the split first iteration / other iterations of the point loop) */
#define GN_CLK() ((long long)__builtin_readcyclecounter())
#else
#define GN_CLK() (0ll)
#endif
#ifndef GN_MAX_THREADS
#define GN_MAX_THREADS 512
#endif
#define GN_MAX_GROUPS (GN_MAX_THREADS / 32)
// MC (compile time): keep the probe results of every source point in memory when a group serves several points per
// iteration (dense scans: N_s > workgroups x groups).  The host picks the variant from the previous scan's N_s (a hint
// copied back asynchronously; both variants are correct for any N_s), so the single-pass code carries none of it.
// XL (compile time): how the workgroups' rows meet.  false: the two-hop exchange above (rows -> 8 group leaders -> group
// sums to every consumer slot), for one sequence spread over the whole chip.  true: ONE hop - every workgroup polls all G
// rows of its sequence itself and forms the same sums (members of a group in member order, then the groups in group
// order: bit-identical to the two-hop result for the same G); meant for G <= 64 workgroups that sit on one XCD (the
// batched kernel kx_gn_loop: sequence s <-> the workgroups with blockIdx & 7 == s), where a store becomes visible to
// its neighbours after one memory-side round trip instead of two hops across the chip.
// G / wg: number and index of the workgroups that cooperate on THIS sequence (the grid, or one XCD's share of it).
template <int PC, bool MC, bool XL>
__device__ __forceinline__ void gn_loop_body(const Ctx& c, int mode, const int G, const int wg) {
    __shared__ double red[GN_MAX_GROUPS][32];
    __shared__ double tot[32];
    __shared__ double Esh2[2][12];  // the increment of iteration i lives in buffer i & 1 (the guess in buffer 1): the solver
                                    // of iteration i + 1 writes the other one, so nobody has to wait before it may
    __shared__ double Tsh[12];
    __shared__ double scur[GN_MAX_GROUPS][4];
    __shared__ double redL[64][32];  // a leader's member rows (G <= 512)
    __shared__ int flag_done2[2];   // likewise
    DevState* st = c.st;
    const int tid = threadIdx.x, lane32 = tid & 31, grp = tid >> 5, gbase = (tid & 63) & 32;
    const int NG = blockDim.x >> 5;
    const int n = st->n_src;
    const unsigned epoch = st->gn_epoch;
    if (wg == 0 && tid == 0 && mode == 0) flush_map_stats(c, st);  // the previous scan's map update precedes this launch
    if (st->n_live == 0 && mode != 1) {  // voxel_map.Empty() => return initial_guess
        if (wg == 0 && tid == 0) {
            Rt I = rt_identity();
            rt_to16(I, st->T_icp);
            st->gn_iters = 0; st->gn_ncorr = 0; st->gn_cand = 0;
            gn_post(c, st, true, mode == 0);
        }
        return;
    }
    if (wg == st->dbg_dead_wg) return;  // test hook: a workgroup that never arrives (exercises the time-out / abort path)
    const double kern = st->gn_kernel, k2 = kern * kern;
    const double gate2 = sqrt_gate(st->gn_max_dist);  // sqrt(d2) < max_dist  <=>  d2 < gate2
    const double conv2 = sqrt_gate(c.conv);           // likewise for the convergence test on |dx|
    const int max_iter = (mode == 1) ? 1 : c.max_iter;
    // which of the 27 sums this lane owns
    int ia = 0, ib = 0;
    {
        int o = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) { if (o == lane32) { ia = a; ib = b; } ++o; }
        if (lane32 >= 21) { ia = lane32 - 21; ib = 6; }
    }
    if (tid < 12) Tsh[tid] = (tid < 9) ? ((tid % 4 == 0) ? 1.0 : 0.0) : 0.0;
    if (tid == 64) {
        // the guess as a Sophus::SE3d would hold it (rotation through a unit quaternion): one lane of the second wavefront
        Rt g = rt_identity();
        if (mode != 1) g = rt_project(rt_from16(guess_src(c)));
        for (int k = 0; k < 9; ++k) Esh2[1][k] = g.R[k];
        for (int k = 0; k < 3; ++k) Esh2[1][9 + k] = g.t[k];
    }
    __syncthreads();
    __shared__ long long cand_total_sh;  // (kept out of the registers: only workgroup 0 reads it, after the loop)
    if (tid == (int)blockDim.x - 1) cand_total_sh = 0;
    int iters = 0;
    long long ph[5] = {0, 0, 0, 0, 0}, ph_wait = 0, ph_x1 = 0, ph_x2 = 0;
    // one point per 32-lane group for the whole loop => its probe results can be cached across iterations
    const bool single_pass = ((G & 7) == 0) ? (((n + 7) >> 3) <= (G >> 3) * NG) : (n <= G * NG);  // (also true when the leaders idle)
    unsigned long long ckey = EMPTY_KEY;
    int cblk = -1, lastv = -1;
    for (int it = 0; it < max_iter; ++it) {
        const long long c0 = GN_CLK();
        // T_icp <- e T_icp for the previous iteration's increment, off the serial tail: one lane of the second
        // wavefront does it while everybody searches (Esh is not rewritten before the next solve)
        const double* Esh = Esh2[(it + 1) & 1];  // what the previous iteration (or the prologue) left
        if (tid == 64 && it > 0) gn_compose(Esh, Tsh);
        double acc = 0.0;
        int ncand = 0;
        const PointWalk pw = point_walk(n, G, wg, NG, grp);
        for (int i = pw.first; i < pw.last; i += pw.step) {
            // lazily apply the previous iteration's increment (TransformPoints(estimation, source)); the increment is
            // re-read from LDS per point so that it does not occupy 24 registers during the search
            Rt E;
            for (int k = 0; k < 9; ++k) E.R[k] = Esh[k];
            for (int k = 0; k < 3; ++k) E.t[k] = Esh[9 + k];
            V3 s;
            if (it == 0 && mode != 1) s = rt_apply(E, v3(c.src0[3 * (size_t)i], c.src0[3 * (size_t)i + 1], c.src0[3 * (size_t)i + 2]));
            else if (it == 0) s = v3(c.src_cur[3 * (size_t)i], c.src_cur[3 * (size_t)i + 1], c.src_cur[3 * (size_t)i + 2]);
            else if (single_pass) s = rt_apply(E, v3(scur[grp][0], scur[grp][1], scur[grp][2]));  // the group's only point
            else s = rt_apply(E, v3(c.src_cur[3 * (size_t)i], c.src_cur[3 * (size_t)i + 1], c.src_cur[3 * (size_t)i + 2]));
            if (single_pass) {
                // one point per group for the whole loop: its current position stays in LDS (a global round trip at
                // the head of every iteration's dependent chain otherwise)
                __builtin_amdgcn_wave_barrier();
                if (lane32 == 0) { scur[grp][0] = s.x; scur[grp][1] = s.y; scur[grp][2] = s.z; }
            } else if (lane32 == 0 && !(it == 0 && mode == 1)) {
                c.src_cur[3 * (size_t)i] = s.x; c.src_cur[3 * (size_t)i + 1] = s.y; c.src_cur[3 * (size_t)i + 2] = s.z;
            }
            V3 t;
            double d2;
            // several points per group: one 128-byte row of probe results per source point in memory (the map is constant
            // during the loop, so a row stays valid while the point keeps its voxel) - a coalesced read instead of 27
            // dependent random reads of the hash table per point and iteration
            unsigned long long key_before = EMPTY_KEY;
            int lastv_before = -1;
            if (!single_pass) lastv = -1;  // several points per group: carried per point in the probe row, if there is one
            if (MC && !single_pass) {
                ckey = (it > 0) ? c.pc_key[i] : EMPTY_KEY;
                cblk = (it > 0) ? c.pc_pb[32 * (size_t)i + lane32] : -1;
                if (it > 0) lastv = c.pc_pb[32 * (size_t)i + 27];  // entry 27 of the row (no voxel) holds the winner's voxel
                if (lane32 == 27) cblk = -1;
                key_before = ckey;
                lastv_before = lastv;
            }
            const bool found = nn_search32<PC>(c, s, lane32, gbase, t, d2, ncand, ckey, cblk, (single_pass || MC) && it > 0, lastv);
            if (MC && !single_pass) {
                if (ckey != key_before) {
                    c.pc_pb[32 * (size_t)i + lane32] = cblk;
                    if (lane32 == 0) c.pc_key[i] = ckey;
                }
                if (lane32 == 27 && (lastv != lastv_before || ckey != key_before)) c.pc_pb[32 * (size_t)i + 27] = lastv;
            }
            if (found && d2 < gate2) {  // uniform over the group
                const V3 r = v3(s.x - t.x, s.y - t.y, s.z - t.z);
                const double den = kern + (r.x * r.x + r.y * r.y + r.z * r.z);
                const double w = k2 / (den * den);
                if (lane32 < 27) acc += w * dot(jcol(ia, s, r), jcol(ib, s, r));
                else if (lane32 == 27) acc += 1.0;
            }
        }
        const long long c1 = GN_CLK();
        // workgroup reduction in fixed order (column 28 carries the candidate count)
        const int ncand0 = group_sum32(ncand);  // sum of the per-lane probe counts over the 32 lanes of the group
        {
            // workgroup reduction, fixed tree: the two groups of a wavefront, then 4 segments of wavefronts, then 4 -> 1
            const double mine = (lane32 == 28) ? (double)ncand0 : acc;
            const double pair = mine + __shfl_xor(mine, 32);
            if ((tid & 63) < 32) red[tid >> 6][lane32] = pair;
        }
        __syncthreads();
        const long long c1b = GN_CLK();
        // ---- exchange: workgroup row -> the leader of its group (wg & 7) -> group sums to everybody.  The leader adds
        // its members' rows in member order and everybody adds the group sums in group order: the association the
        // counter-barrier version used (8 strided partial sums, then 8 -> 1), bit for bit.
        const unsigned flag = gn_flag(epoch, it);
        const int par = it & 1, ngroups = G < 8 ? G : 8;
        bool ok = true;
        if (tid < 58) {
            // lanes 2e, 2e + 1 of the first wavefront both form entry e of the workgroup's row - the 4 segments of
            // wavefronts, then 4 -> 1, the tree the batched kernel builds through LDS - and store one half each
            const int e = tid >> 1, NW = NG >> 1, per = (NW + 3) >> 2;
            double sg[4];
            if (NW == 16) {  // 1024 threads: all 16 reads in flight, then the same additions
                double r[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) r[g] = red[g][e];
#pragma unroll
                for (int seg = 0; seg < 4; ++seg) sg[seg] = (((0.0 + r[4 * seg]) + r[4 * seg + 1]) + r[4 * seg + 2]) + r[4 * seg + 3];
            } else {
#pragma unroll
                for (int seg = 0; seg < 4; ++seg) {
                    double v = 0.0;
                    for (int g = seg * per; g < (seg + 1) * per && g < NW; ++g) v += red[g][e];
                    sg[seg] = v;
                }
            }
            const double rowv = ((sg[0] + sg[1]) + sg[2]) + sg[3];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(rowv);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            __hip_atomic_store(&c.gn_rows_ll[((size_t)par * G + wg) * GN_LL_WORDS + tid],
                               (unsigned long long)half | ((unsigned long long)flag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long c2 = GN_CLK();
        if (XL) {
            // one hop: every 32-lane group polls one row of the sequence (all its workgroups, this one included); the
            // first 29 lanes then add them in the two-hop association and hand the totals to the solve
            for (int j = grp; j < G; j += NG) {
                const unsigned long long* row = c.gn_rows_ll + ((size_t)par * G + j) * GN_LL_WORDS;
                unsigned h0, h1;
                gn_ll_wait2(row + lane32, row + lane32 + 32, lane32 + 32 < 58, flag, &h0, &h1, &ok, &st->gn_abort);
                const double v0 = gn_ll_join(h0), v1 = gn_ll_join(h1);
                if ((lane32 & 1) == 0) {
                    redL[j][lane32 >> 1] = v0;
                    if (16 + (lane32 >> 1) < 29) redL[j][16 + (lane32 >> 1)] = v1;
                }
            }
            if (!ok) gn_raise_abort(st);
            ok = __syncthreads_or(ok ? 0 : 1) == 0;
            ph_x1 += GN_CLK() - c2;
            if (tid < 29) {
                double t = 0.0;
                if (G == 32) {  // all reads in flight, then the additions in (group, member) order
                    double r[32];
#pragma unroll
                    for (int j = 0; j < 32; ++j) r[j] = redL[j][tid];
#pragma unroll
                    for (int g = 0; g < 8; ++g) t += (((0.0 + r[g]) + r[g + 8]) + r[g + 16]) + r[g + 24];
                } else {
                    for (int g = 0; g < ngroups; ++g) {
                        double sg = 0.0;
                        for (int j = g; j < G; j += 8) sg += redL[j][tid];
                        t += sg;
                    }
                }
                tot[tid] = t;
            }
            ph_x2 += GN_CLK() - c2;
        } else {
            if (wg < ngroups) {  // leader of group wg: members wg, wg + 8, ... ; every wavefront takes two of their rows
                const int nmem = (G - wg + 7) / 8;
                for (int j = grp; j < nmem; j += NG) {
                    const unsigned long long* row = c.gn_rows_ll + ((size_t)par * G + (wg + 8 * j)) * GN_LL_WORDS;
                    unsigned h0, h1;
                    gn_ll_wait2(row + lane32, row + lane32 + 32, lane32 + 32 < 58, flag, &h0, &h1, &ok, &st->gn_abort);
                    const double v0 = gn_ll_join(h0), v1 = gn_ll_join(h1);
                    if ((lane32 & 1) == 0) {
                        redL[j][lane32 >> 1] = v0;
                        if (16 + (lane32 >> 1) < 29) redL[j][16 + (lane32 >> 1)] = v1;
                    }
                }
                if (!ok) gn_raise_abort(st);
                __syncthreads();
                ph_x1 += GN_CLK() - c2;
                if (tid < 29) {  // sum in member order; each summing lane publishes both halves of its entry, one copy per consumer slot
                    double s = 0.0;
                    if (nmem == 32) {  // 256 workgroups: all reads in flight, then the same additions in member order
                        double r[32];
    #pragma unroll
                        for (int j = 0; j < 32; ++j) r[j] = redL[j][tid];
    #pragma unroll
                        for (int j = 0; j < 32; ++j) s += r[j];
                    } else {
                        for (int j = 0; j < nmem; ++j) s += redL[j][tid];
                    }
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(s), fl = (unsigned long long)flag << 32;
                    const unsigned long long lo = (bits & 0xFFFFFFFFull) | fl, hi = (bits >> 32) | fl;
    #pragma unroll
                    for (int r = 0; r < GN_XSUM_COPIES; ++r) {
                        unsigned long long* dst = c.gn_xsum_ll + (((size_t)par * 8 + r) * 8 + wg) * GN_LL_WORDS + 2 * tid;
                        __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            // The group sums are collected by ONE wavefront (lane l <-> word l of a row; lanes 2e, 2e + 1 hold the halves of
            // entry e): nothing between the arrival of the last word and the solve needs LDS traffic but the 29 totals.
            if (tid < 64) {
                const bool mine = tid < 58, even = (tid & 1) == 0;
                ok = true;
                ph_x2 += GN_CLK() - c2;
                // everybody: the group sums addressed to this workgroup's slot, added in group order
                unsigned h[8];
                unsigned spins = 0;
                for (;;) {
                    unsigned long long vv[8];
    #pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        vv[g] = (unsigned long long)flag << 32;
                        if (mine && g < ngroups)
                            vv[g] = __hip_atomic_load(c.gn_xsum_ll + (((size_t)par * 8 + (wg & (GN_XSUM_COPIES - 1))) * 8 + g) * GN_LL_WORDS + tid,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    unsigned bad = 0u;
    #pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        bad |= (unsigned)(vv[g] >> 32) ^ flag;
                        h[g] = (unsigned)vv[g];
                    }
                    if (__all(bad == 0u)) break;
                    if (++spins > GN_LL_SPINS) { ok = false; break; }
                    /* no sleep: nothing else runs in this workgroup while its first wavefront waits for the totals */
                }
                double t = 0.0;
    #pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const double v = gn_ll_join(h[g]);
                    if (g < ngroups) t += v;
                }
                if (mine && even) tot[tid >> 1] = t;
                if (!ok && tid == 0) gn_raise_abort(st);
            }
        }
        long long c3 = 0, c4 = 0;
        if (tid < 64) {  // the wavefront that collected the totals solves: its own LDS writes are ordered before its reads,
            c3 = c4 = GN_CLK();  // so no workgroup barrier stands between the last arrival and the factorisation
            __builtin_amdgcn_wave_barrier();
            double dx[6];        // every lane ends up with the same dx, lane 0 publishes
            solve6_ldlt_wave(tot, tid, dx);
          if (tid == 0) {
            const Rt e = se3_exp_gn(dx);
            for (int k = 0; k < 9; ++k) Esh2[it & 1][k] = e.R[k];
            for (int k = 0; k < 3; ++k) Esh2[it & 1][9 + k] = e.t[k];
            double nn = 0.0;
            for (int k = 0; k < 6; ++k) nn += dx[k] * dx[k];
            flag_done2[it & 1] = (nn < conv2 || !ok) ? 1 : 0;  // sqrt(nn) < conv, without the square root on the serial path; !ok: the exchange gave up (time-out / abort)
          }
        }
        __syncthreads();
        const long long c5 = GN_CLK();
        ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3; ph[4] += c5 - c4;
        ph_wait += c1b - c1;
        if (tid == (int)blockDim.x - 1) cand_total_sh += (long long)tot[28];  // (a lane of the last wavefront: off the serial path)
        iters = it + 1;
        const int done = flag_done2[it & 1];
        if (mode == 1 && wg == 0 && tid < 29) st->dbg_sums[tid] = tot[tid];
        if (done) break;
    }
    if (tid == 64 && iters > 0) gn_compose(Esh2[(iters - 1) & 1], Tsh);  // the last increment
    __syncthreads();
    if (tid == 0 && c.wg_clk) { c.wg_clk[wg] += ph[0] + ph_wait; c.wg_clk[G + wg] += ph[0]; }  // diagnostic: search phase per workgroup
    if (wg == 0 && tid == 0) {
        Rt T;
        for (int k = 0; k < 9; ++k) T.R[k] = Tsh[k];
        for (int k = 0; k < 3; ++k) T.t[k] = Tsh[9 + k];
        rt_to16(T, st->T_icp);
        st->gn_iters = iters;
        st->gn_ncorr = iters > 0 ? (int)tot[27] : 0;  // the last iteration's totals are still in LDS
        st->gn_cand = cand_total_sh;
        for (int k = 0; k < 5; ++k) st->gn_phase_clk[k] += ph[k];
        st->gn_phase_clk[5] += iters;
        st->gn_phase_clk[6] += ph_x1; st->gn_phase_clk[7] += ph_x2;  // part of phase 1 spent waiting for the workgroup's other wavefronts
        if (mode != 1) gn_post(c, st, false, mode == 0);
    }
}
// one sequence over the whole grid
template <int PC, bool MC>
__global__ __launch_bounds__(GN_MAX_THREADS) void k_gn_loop(Ctx c, int mode) {
    gn_loop_body<PC, MC, false>(c, mode, (int)gridDim.x, (int)blockIdx.x);
}


// ================================================================================================ K5, 8 lanes per point
// Throughput form of the Gauss-Newton loop for workgroups that walk MANY source points per iteration (the batched
// runner: one, two or four sequences per XCD, 200-800 points per workgroup and iteration; dense scans).  The 32-lanes-per-
// point search above is VALU-issue-bound there - every lane executes the whole search for voxels that hold 7-8 points on
// average.  Per iteration (gn8_body):
//   * phase A, one LANE per point, one pass: apply the increment, test the point's answer row (exact answer cache: the
//     GN8_KCAND nearest candidates of the last full search and a bound on everybody else - whichever of them is nearest
//     now provably is what a search would return while the point has moved less than the bound allows); settled points
//     go straight into the lane's sums;
//   * the others are compacted in point order (deterministic) and get the full 27-voxel search with 8 lanes per point
//     (gn8_search): 128-byte probe row per point in memory (lane l holds entries 4 l .. 4 l + 3; entry 27 = the voxel of
//     the last winner, entry 28 = the candidate count of the 27 voxels), box distances from the point alone, first round
//     = own + last winner's + the two nearest other voxels with every load in flight together (lane l <-> stored points
//     l, l + 8, l + 16), the rest exactly pruned as in nn_scan32, group reductions by DPP;
//   * the SAME correspondence, distance, gate and weight for every point as the 32-lane kernel; the linear system is
//     accumulated as 16 moments sum w z_a z_b over z = (1, s, r, s x r) per lane instead of 27 products picked per lane:
//     JTJ = [[ W I, -hat(Ws) ], [ ., tr(S) I - S ]] with W = sum w, Ws = sum w s, S = sum w s s^T, and
//     JTr = [ sum w r ; sum w s x r ]  (J = [I | -hat(s)], Registration.cpp BuildLinearSystem) - the sums differ from the
//     32-lane kernel's in association only (poses agree far below the 1e-9 m bar, integer statistics are identical);
//   * wavefront reduction by DPP, workgroup reduction through LDS in wavefront order, one 18-entry row per workgroup;
//     exchange in one hop (<= 64 workgroups: the first wavefront polls every row itself and solves straight away) or in
//     two hops through 8 group leaders (one sequence over the whole chip).
#ifndef GN8_KEEP
#define GN8_KEEP 1.5          /* a voxel is dropped from the search when its box lies farther than sqrt(GN8_KEEP) x the best distance so far.  1 = exactly what
                                 cannot hold a closer point; > 1 also scans voxels that cannot win, and what is found there replaces the box distance in
                                 the bound of the answer row: fewer repeated searches later.  1 / 1.5 / 2.25 / 3: Gauss-Newton 2566 / 2493 / 2520 / 2564 us
                                 per scan (make KEEP=x) */
#endif
#ifndef GN8_SURV
#define GN8_SURV 2            /* voxels per survivor round of the search (make SURV=n) */
#endif
#ifndef GN8_KCAND
#define GN8_KCAND 4           /* candidates an answer row keeps (2..5); a CPU simulation of the policy on the bench's sweeps: repeated searches per scan
                                 32 k (1) | 11.5 k (2) | 6.5 k (3) | 4.4 k (4) | 2.4 k (6) for ~6800 source points x 36 iterations */
#endif
#define GN8_ANS_ROW ((3 + 3 * GN8_KCAND + 3 + 1) & ~1)  /* doubles per answer row: s0 (3) | the K nearest candidates (3 each) | bound | ids, counts (packed) | voxel key of s0 [| pad]
                                                          (K = 4: 18 doubles = 144 B as before - the key sits where the padding was, and phase A no longer reads pc_key[]) */
#define GN8_ANS_D (3 + 3 * GN8_KCAND)
#define GN8_NOTHING_IN_REACH (-2.0)  /* bound slot of an answer row: the search found the point's 27 voxels empty (gn8_search -> phase A of gn8_body) */
#ifndef GN8_SPEC
#define GN8_SPEC 2            /* how many of the nearest other boxes the first search round takes along, their loads in flight with the own voxel's (0 .. 3; make SPEC=n) */
#endif
// ---- group primitives for LP = 8 or 4 lanes per point: DPP steps inside the VALU (quad permutes, for 8 lanes the half-row
// mirror on top - every quad holds its result already, so pairing mirrored lanes is as good as xor 4); nothing goes through
// the LDS crossbar.  group_bcastL: the value one lane of the group holds (everybody else passes 0.0; x + 0 = x exactly)
template <int LP> __device__ __forceinline__ double group_minL(double v) {
    v = fmin(v, dpp_f64<0xB1>(v));
    v = fmin(v, dpp_f64<0x4E>(v));
    return LP == 8 ? fmin(v, dpp_f64<0x141>(v)) : v;
}
template <int LP> __device__ __forceinline__ unsigned group_minL(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    return LP == 8 ? min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true)) : v;
}
template <int LP> __device__ __forceinline__ int group_sumL(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    return LP == 8 ? v + __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true) : v;
}
template <int LP> __device__ __forceinline__ double group_bcastL(double v) {  // one lane's value, the others pass 0.0
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    return LP == 8 ? v + dpp_f64<0x141>(v) : v;
}
// entry q (group-uniform) of this lane's RE row entries
// (the entries pass through an empty asm first: without it the compiler turns the chain of selects over VALUES into one load
// through a select over ADDRESSES, the row then lives in scratch memory and every pick is a memory round trip in the search's
// dependent chain - 16-byte store + up to five dependent loads per search in the round-3 build)
template <int RE> __device__ __forceinline__ int sel_entry(const int (&r)[RE], int q) {
    int a[RE];
#pragma unroll
    for (int k = 0; k < RE; ++k) { a[k] = r[k]; asm("" : "+v"(a[k])); }
    int v = a[0];
#pragma unroll
    for (int k = 1; k < RE; ++k) v = (q == k) ? a[k] : v;
    return v;
}
// NV stored voxels against the point with LP lanes: lane l <-> stored points l, l + LP, ... - every load of the call in
// flight before the first distance is formed.  (bd, border, bp) / (b2d, b2o, b2p): the best and the second-best candidate
// this lane has seen; sd = the smallest distance it has seen lose to both.
template <int PC, int LP, int NV, class CT>
__device__ __forceinline__ void scan_voxelsL(const CT& c, const int (&pb)[NV], const int (&vx)[NV], V3 s, int laneL, double& bd, double& sd,
                                             unsigned& border, V3& bp, double& b2d, unsigned& b2o, V3& b2p) {
    const int P = (PC > 0) ? PC : c.P;
    constexpr int NCH = (PC > 0) ? (PC + LP - 1) / LP : 24 / LP;  // chunks of LP stored points (P <= 24 when not compile-time)
    if (P <= 24) {
        double q[NV][NCH][3];
        bool act[NV][NCH];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int cnt = (pb[v] < 0) ? 0 : (int)((unsigned)pb[v] >> 24);
            const double* X = blk_x(c, pb[v] & BLK_ID_MASK);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = LP * k + laneL;
                act[v][k] = idx < cnt;
                q[v][k][0] = act[v][k] ? X[3 * idx] : 0.0;
                q[v][k][1] = act[v][k] ? X[3 * idx + 1] : 0.0;
                q[v][k][2] = act[v][k] ? X[3 * idx + 2] : 0.0;
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (act[v][k]) {
                    const double dx = q[v][k][0] - s.x, dy = q[v][k][1] - s.y, dz = q[v][k][2] - s.z;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    const unsigned id = (unsigned)(vx[v] * 32 + LP * k + laneL);
                    if (d2 < bd || (d2 == bd && id < border)) {
                        sd = fmin(sd, b2d); b2d = bd; b2o = border; b2p = bp;
                        bd = d2; border = id; bp = v3(q[v][k][0], q[v][k][1], q[v][k][2]);
                    } else if (d2 < b2d || (d2 == b2d && id < b2o)) {
                        sd = fmin(sd, b2d); b2d = d2; b2o = id; b2p = v3(q[v][k][0], q[v][k][1], q[v][k][2]);
                    } else sd = fmin(sd, d2);
                }
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int cnt = (pb[v] < 0) ? 0 : (int)((unsigned)pb[v] >> 24);
            const double* X = blk_x(c, pb[v] & BLK_ID_MASK);
            for (int base = 0; __any(base < cnt); base += LP) {
                const int idx = base + laneL;
                if (idx < cnt) {
                    const double qx = X[3 * idx], qy = X[3 * idx + 1], qz = X[3 * idx + 2];
                    const double dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    const unsigned id = (unsigned)(vx[v] * 32 + idx);
                    if (d2 < bd || (d2 == bd && id < border)) {
                        sd = fmin(sd, b2d); b2d = bd; b2o = border; b2p = bp;
                        bd = d2; border = id; bp = v3(qx, qy, qz);
                    } else if (d2 < b2d || (d2 == b2d && id < b2o)) {
                        sd = fmin(sd, b2d); b2d = d2; b2o = id; b2p = v3(qx, qy, qz);
                    } else sd = fmin(sd, d2);
                }
            }
        }
    }
}
// The full 27-voxel search of source point i at s by a group of LP lanes (all of them call it): probe row (re-probed when
// the point changed voxel), box distances, first round = own voxel + last winner's voxel (+ the two nearest other boxes
// when LP == 8), the remaining voxels exactly pruned and scanned two at a time, the new answer row.  Returns in every
// lane the neighbour, its squared distance, found, and the candidate count of the 27 voxels.
// xc: the workgroup's executed-work counters in LDS ([1] probe rows rebuilt, [2] stored points read; DevState::exec_cnt)
__device__ __forceinline__ unsigned pb_count(int pb) { return pb < 0 ? 0u : (unsigned)pb >> 24; }
template <int PC, int LP>
__device__ __forceinline__ void gn8_search(const Ctx& c, int i, int it, V3 s, double inv_vs, int laneL, int gb, V3& t, double& m, bool& found, int& ctot,
                                           unsigned* xc) {
    constexpr int RE = 32 / LP;  // row entries per lane
#ifdef GN_PHASE_CLOCKS
#define SRCH_CLK(k) do { __builtin_amdgcn_s_waitcnt(0); const long long n_ = GN_CLK(); if (srch_me) atomicAdd((unsigned long long*)&c.wg_clk[61 + (k)], (unsigned long long)(n_ - srch_t)); srch_t = n_; } while (0)
    const bool srch_me = it > 0 && blockIdx.x < 8 && threadIdx.x == 0;  // (thread 0 of workgroup 0 of the sequences on the first teams)
    long long srch_t = GN_CLK();
#else
#define SRCH_CLK(k) do { } while (0)
#endif
    const int kx = voxel_index(s.x, c.vs, inv_vs), ky = voxel_index(s.y, c.vs, inv_vs), kz = voxel_index(s.z, c.vs, inv_vs);
    const unsigned long long key = pack_key(kx, ky, kz);
    int r[RE];
#pragma unroll
    for (int k = 0; k < RE; ++k) r[k] = -1;
    if (it > 0) {  // (uniform; a scan's first iteration rebuilds every row: nothing to read, nothing to wait for)
        const int4* rp = (const int4*)(c.pc_pb + 32 * (size_t)i + RE * laneL);  // (requested together with the key that validates it)
#pragma unroll
        for (int k = 0; k < RE / 4; ++k) { const int4 v = rp[k]; r[4 * k] = v.x; r[4 * k + 1] = v.y; r[4 * k + 2] = v.z; r[4 * k + 3] = v.w; }
    }
    const bool same_voxel = it > 0 && c.pc_key[i] == key;
    SRCH_CLK(0);  // row + key
    if (!same_voxel) {  // uniform over the group: probe this lane's neighbour voxels, rebuild the row
        int cs = 0;
        {   // this lane's RE probes with their first table reads in flight together; a collision walks on by itself
            unsigned long long kq[RE];
            unsigned sq[RE];
            TabEnt eq[RE];
#pragma unroll
            for (int q = 0; q < RE; ++q) {
                const int ee = RE * laneL + q;
                kq[q] = pack_key(kx + ee / 9 - 1, ky + (ee / 3) % 3 - 1, kz + ee % 3 - 1);
                sq[q] = brick_slot(kq[q], c.tmask);
                eq[q].key = EMPTY_KEY; eq[q].blk = -1; eq[q].head = -1;
                if (ee < 27) eq[q] = c.tab[sq[q]];
            }
            bool pend[RE];
#pragma unroll
            for (int q = 0; q < RE; ++q) {
                r[q] = -1;
                pend[q] = false;
                if (eq[q].key == kq[q]) r[q] = eq[q].blk;
                else if (eq[q].key != EMPTY_KEY) pend[q] = true;  // occupied by another voxel (or a tombstone): linear probing from the next slot
            }
            for (unsigned probe = 1; probe <= c.tmask; ++probe) {  // ... in rounds: the next slots of all unsettled probes travel together
                bool anyp = false;
#pragma unroll
                for (int q = 0; q < RE; ++q) anyp = anyp || pend[q];
                if (!anyp) break;
#pragma unroll
                for (int q = 0; q < RE; ++q)
                    if (pend[q]) { sq[q] = (sq[q] + 1) & c.tmask; eq[q] = c.tab[sq[q]]; }
#pragma unroll
                for (int q = 0; q < RE; ++q) {
                    if (!pend[q]) continue;
                    if (eq[q].key == kq[q]) { r[q] = eq[q].blk; pend[q] = false; }
                    else if (eq[q].key == EMPTY_KEY) pend[q] = false;
                }
            }
#pragma unroll
            for (int q = 0; q < RE; ++q) cs += (r[q] < 0) ? 0 : (int)((unsigned)r[q] >> 24);
        }
        cs = group_sumL<LP>(cs);
        if (laneL == 28 / RE) r[28 % RE] = cs;  // entry 28: candidates of the 27 voxels (entry 27: no last winner yet = -1)
        int4* wp = (int4*)(c.pc_pb + 32 * (size_t)i + RE * laneL);
#pragma unroll
        for (int k = 0; k < RE / 4; ++k) wp[k] = make_int4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
        if (laneL == 0) { c.pc_key[i] = key; atomicAdd(xc + 1, 1u); }
    }
    SRCH_CLK(1);  // rebuild (some group of the wavefront changed voxel)
    const int lv_raw = __shfl(r[27 % RE], gb + 27 / RE);
    ctot = __shfl(r[28 % RE], gb + 28 / RE);
    const int lv = (lv_raw != 13) ? lv_raw : -1;
    const int pbc = __shfl(r[13 % RE], gb + 13 / RE);  // the point's own voxel
    const int pbl = (lv >= 0) ? __shfl(sel_entry<RE>(r, lv % RE), gb + lv / RE) : -1;
    double bd = 1.7976931348623157e308, sd = 1.7976931348623157e308, b2d = 1.7976931348623157e308;
    unsigned border = 0xFFFFFFFFu, b2o = 0xFFFFFFFFu;
    V3 bp = v3(0, 0, 0), b2p = v3(0, 0, 0);
    // squared distance from the point to the box of each of this lane's (stored) neighbour voxels: known before any load
    double gap2[RE];
    {
        double g2m[3], g2p[3];
        const int kk[3] = {kx, ky, kz};
        const double xx[3] = {s.x, s.y, s.z};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int vm = kk[ax] - 1, vp = kk[ax] + 1;
            const int bim = (vm < 0) ? vm : vm + 1, bip = (vp > 0) ? vp : vp - 1;
            const double gm = fmax(-((double)bim * c.vs - xx[ax]) - 1e-9, 0.0), gp = fmax(((double)bip * c.vs - xx[ax]) - 1e-9, 0.0);
            g2m[ax] = gm * gm; g2p[ax] = gp * gp;
        }
#pragma unroll
        for (int q = 0; q < RE; ++q) {
            const int ee = RE * laneL + q;
            gap2[q] = 1.7976931348623157e308;  // not a candidate voxel: absent, the own one, the last winner's, no voxel
            if (ee >= 27 || ee == 13 || ee == lv || r[q] < 0) continue;
            const int cx = ee / 9, cy = (ee / 3) % 3, cz = ee % 3;
            const double gx = cx == 0 ? g2m[0] : cx == 1 ? 0.0 : g2p[0];
            const double gy = cy == 0 ? g2m[1] : cy == 1 ? 0.0 : g2p[1];
            const double gz = cz == 0 ? g2m[2] : cz == 1 ? 0.0 : g2p[2];
            gap2[q] = (gx + gy) + gz;
        }
    }
    if (LP == 8 && GN8_SPEC > 0) {
        // first round: the own voxel, the last winner's and the two nearest other boxes, all loads in flight together (a
        // voxel scanned although its box turns out to lie beyond the best distance is harmless: it is one of the 27)
        constexpr int NS = GN8_SPEC > 0 ? GN8_SPEC : 1;  // nearest other boxes taken along
        int pbN[2 + NS], vxN[2 + NS];
        pbN[0] = pbc; pbN[1] = pbl; vxN[0] = 13; vxN[1] = lv;
        unsigned npts = pb_count(pbc) + pb_count(pbl);
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            double gl = 1.7976931348623157e308;
            int ql = 0;
#pragma unroll
            for (int q = 0; q < RE; ++q) if (gap2[q] < gl) { gl = gap2[q]; ql = q; }
            const double gmin = group_minL<LP>(gl);
            const unsigned v = group_minL<LP>((gl == gmin && gmin < 1.0e300) ? (unsigned)(RE * laneL + ql) : 0xFFu);
            vxN[2 + u] = (v == 0xFFu) ? -1 : (int)v;
            pbN[2 + u] = (v == 0xFFu) ? -1 : __shfl(sel_entry<RE>(r, (int)(v % RE)), gb + (int)((v / RE) & (LP - 1)));
            npts += pb_count(pbN[2 + u]);
            if (v != 0xFFu && (int)(v / RE) == laneL) {
#pragma unroll
                for (int q = 0; q < RE; ++q) if (q == (int)(v % RE)) gap2[q] = 1.7976931348623157e308;
            }
        }
        if (laneL == 0) atomicAdd(xc + 2, npts);
        scan_voxelsL<PC, LP, 2 + NS>(c, pbN, vxN, s, laneL, bd, sd, border, bp, b2d, b2o, b2p);
    } else {
        const int pb2[2] = {pbc, pbl}, vx2[2] = {13, lv};
        if (laneL == 0) atomicAdd(xc + 2, pb_count(pbc) + pb_count(pbl));
        scan_voxelsL<PC, LP, 2>(c, pb2, vx2, s, laneL, bd, sd, border, bp, b2d, b2o, b2p);
    }
    SRCH_CLK(2);  // box distances + first round
    // the other voxels: dropped when their box lies farther than the best distance so far (exact, see nn_scan32)
    unsigned mine = 0u;
    double gdrop = 1.7976931348623157e308;  // smallest squared box distance among the voxels this lane dropped
    {
        const double m0 = group_minL<LP>(bd);
#pragma unroll
        for (int q = 0; q < RE; ++q) {
            if (!(gap2[q] < 1.0e300)) continue;
            if (gap2[q] > GN8_KEEP * m0) gdrop = fmin(gdrop, gap2[q]); else mine |= 1u << q;
        }
    }
    for (;;) {  // the surviving voxels, GN8_SURV at a time (their loads in flight together)
        int pbS[GN8_SURV], vxS[GN8_SURV];
        unsigned npts = 0u;
#pragma unroll
        for (int j = 0; j < GN8_SURV; ++j) {
            const unsigned cand = mine ? (unsigned)(RE * laneL + __ffs(mine) - 1) : 0xFFu;
            const unsigned v = group_minL<LP>(cand);  // uniform over the group; 0xFF: nothing left
            if (v != 0xFFu && (int)(v / RE) == laneL) mine &= mine - 1u;
            vxS[j] = (int)v;
            pbS[j] = (v != 0xFFu) ? __shfl(sel_entry<RE>(r, (int)(v % RE)), gb + (int)((v / RE) & (LP - 1))) : -1;
            npts += pb_count(pbS[j]);
        }
        if (vxS[0] == 0xFF) break;
#ifdef GN_PHASE_CLOCKS
        if (laneL == 0) atomicAdd((unsigned long long*)&c.wg_clk[41], 1ull);  // survivor rounds (wg_clk[40]: searches)
#endif
        if (laneL == 0) atomicAdd(xc + 2, npts);
        scan_voxelsL<PC, LP, GN8_SURV>(c, pbS, vxS, s, laneL, bd, sd, border, bp, b2d, b2o, b2p);
    }
#ifdef GN_PHASE_CLOCKS
    if (laneL == 0) { atomicAdd((unsigned long long*)&c.wg_clk[40], 1ull); if (!same_voxel) atomicAdd((unsigned long long*)&c.wg_clk[42], 1ull); }
#endif
    SRCH_CLK(3);  // survivors
    // ---- the group's GN8_KCAND nearest candidates in (distance, visiting order) order.  Every lane holds its own two nearest
    // (bd / b2d) and the smallest distance that lost to both (sd): K rounds of "every lane offers the nearest it has not
    // given yet, the group's minimum takes it"; what is left - the lanes' next offers, sd, the boxes of the dropped voxels -
    // is at least sqrt(D2) away from s and bounds everybody who is not in the row.  (A lane that holds three of the true K
    // nearest contributes two, its third goes into the bound: the row is then a little less useful, never wrong.)  The first
    // is the answer of the search; the winning lanes store their points straight into the row.
    const double INF = 1.7976931348623157e308;
    int taken = 0;
    unsigned long long meta = 0ull;
    int nvalid = 0;
    unsigned bo = 0xFFFFFFFFu;
    m = INF;
    t = v3(0.0, 0.0, 0.0);
    double* arow = c.pc_ans + GN8_ANS_ROW * (size_t)i;
#pragma unroll
    for (int r4 = 0; r4 < GN8_KCAND; ++r4) {
        const double od = taken == 0 ? bd : taken == 1 ? b2d : INF;
        const unsigned oo = taken == 0 ? border : taken == 1 ? b2o : 0xFFFFFFFFu;
        const double mr = group_minL<LP>(od);
        const unsigned orr = group_minL<LP>((od == mr) ? oo : 0xFFFFFFFFu);
        const bool has = mr < INF;
        const bool win = has && od == mr && oo == orr;
        const V3 pp = taken == 0 ? bp : b2p;
        if (r4 == 0) {
            m = mr; bo = orr;
            t = v3(group_bcastL<LP>(win ? pp.x : 0.0), group_bcastL<LP>(win ? pp.y : 0.0), group_bcastL<LP>(win ? pp.z : 0.0));
        }
        if (win) { arow[3 + 3 * r4] = pp.x; arow[4 + 3 * r4] = pp.y; arow[5 + 3 * r4] = pp.z; ++taken; }
        meta |= (unsigned long long)(has ? (orr & 0x3FFu) : 0x3FFu) << (10 * r4);
        nvalid += has ? 1 : 0;
    }
    found = m < INF;
    const int lv_new = found ? (int)(bo >> 5) : -1;
    if (laneL == 27 / RE && lv_new != lv_raw) c.pc_pb[32 * (size_t)i + 27] = lv_new;
    const double rest = taken == 0 ? bd : taken == 1 ? b2d : sd;
    const double D2 = group_minL<LP>(fmin(rest, gdrop));
#ifdef GN_PHASE_CLOCKS
    {   // what bounds everybody else: a candidate that was scanned or the box of a dropped voxel?
        const double Dc = group_minL<LP>(rest), Dg = group_minL<LP>(gdrop);
        if (laneL == 0 && found) {
            atomicAdd((unsigned long long*)&c.wg_clk[44 + (Dc <= Dg ? 0 : 1)], 1ull);
            const double q = sqrt(D2 / m);  // bound / distance of the winner: [1,1.2) [1.2,1.5) [1.5,2) [2,3) [3,..)
            atomicAdd((unsigned long long*)&c.wg_clk[46 + (q < 1.2 ? 0 : q < 1.5 ? 1 : q < 2.0 ? 2 : q < 3.0 ? 3 : 4)], 1ull);
        }
    }
#endif
    double sl2 = -1.0;
    if (found && D2 < 1.0e300) sl2 = sqrt(D2) * (1.0 - 1e-6) - 1e-9;  // (a little less is stored: the margin of the test)
    else if (found) sl2 = 1.0e300;  // nobody else in reach of this voxel
    else if (ctot == 0) sl2 = GN8_NOTHING_IN_REACH;  // the 27 voxels around this key hold no point at all: true for as long as the point keeps the voxel
    if (laneL == 0) {  // s0 | ... | bound | order ids (10 bits each), candidate count of the 27 voxels, candidates in the row
        arow[0] = s.x; arow[1] = s.y; arow[2] = s.z;
        arow[GN8_ANS_D] = sl2;
        meta |= (unsigned long long)(unsigned)ctot << 50 | (unsigned long long)(unsigned)nvalid << 60;
        arow[GN8_ANS_D + 1] = __longlong_as_double((long long)meta);
        arow[GN8_ANS_D + 2] = __longlong_as_double((long long)key);  // (= pc_key[i]: the voxel this search looked around)
    }
    SRCH_CLK(4);  // reductions + answer row
#ifdef GN_PHASE_CLOCKS
    if (srch_me) atomicAdd((unsigned long long*)&c.wg_clk[66], 1ull);
#endif
#undef SRCH_CLK
}

// packed upper triangle of JTJ (21) + JTr (6) from the 16 moments
//   M: 0 W | 1..3 W s | 4 xx 5 xy 6 xz 7 yy 8 yz 9 zz | 10..12 sum w r | 13..15 sum w s x r
__device__ __forceinline__ double sums_from_moments(int e, const double* M) {
    // a chain of selects over constant indices: M stays in registers.  (As a switch with M[10 + (e - 21)] in it the array
    // went to scratch memory - nine 16-byte stores and a load back per iteration, 6.5 k ticks of every iteration's serial tail.)
    double r = 0.0;                                     // (0,1) (0,2) (0,3) (1,2) (1,4) (2,5)
#define SFM(k, v) r = (e == (k)) ? (v) : r
    SFM(0, M[0]); SFM(6, M[0]); SFM(11, M[0]);          // (0,0) (1,1) (2,2)
    SFM(4, M[3]);                                       // (0,4) = +W sz
    SFM(5, -M[2]);                                      // (0,5) = -W sy
    SFM(8, -M[3]);                                      // (1,3)
    SFM(10, M[1]);                                      // (1,5) = +W sx
    SFM(12, M[2]);                                      // (2,3) = +W sy
    SFM(13, -M[1]);                                     // (2,4)
    SFM(15, M[7] + M[9]);                               // (3,3) = yy + zz
    SFM(16, -M[5]);                                     // (3,4) = -xy
    SFM(17, -M[6]);                                     // (3,5) = -xz
    SFM(18, M[4] + M[9]);                               // (4,4) = xx + zz
    SFM(19, -M[8]);                                     // (4,5) = -yz
    SFM(20, M[4] + M[7]);                               // (5,5) = xx + yy
    SFM(21, M[10]); SFM(22, M[11]); SFM(23, M[12]);
    SFM(24, M[13]); SFM(25, M[14]); SFM(26, M[15]);
#undef SFM
    return r;
}

#define GN8_ROW_ENTRIES 18   /* 16 moments, pair count, candidate count */
#define GN8_PREFETCH 1        /* chunks phase A requests ahead (ptl_build_info) */
#ifndef GN8_LDS_PTS
#define GN8_LDS_PTS (6 * 512) /* source-point positions a workgroup keeps in LDS (72 KB) */
#endif
#ifndef GN8_QWAVE
#define GN8_QWAVE 512         /* queue of points awaiting the full search: one region per wavefront of phase A (8 chunks of 64 at most between two
                                 flushes), filled from both ends by kind - no cross-wavefront prefix, no workgroup barrier per chunk */
#endif
#define GN8_QCHUNKS (GN8_QWAVE / 64)
#ifndef GN8_LPB
#define GN8_LPB 8             /* lanes per point of the full search (8 or 4) */
#endif
#ifndef GN8_IT0_LP
#define GN8_IT0_LP 8          /* ... in the first iteration of a scan, where every point searches (make IT0LP=4: tried) */
#endif
#ifndef GN8_MAX_THREADS
#define GN8_MAX_THREADS 512  /* 8 wavefronts per workgroup: 256 VGPRs per lane, nothing spills (at 768 / 168 VGPRs the search spills ~50) */
#endif
// one correspondence into a lane's 16 moments (+ pair count): w z_a z_b over z = (1, s, r, s x r)
__device__ __forceinline__ void gn8_accumulate(double (&M)[GN8_ROW_ENTRIES], V3 s, V3 t, double kern, double k2) {
    const V3 rr = v3(s.x - t.x, s.y - t.y, s.z - t.z);
    const double den = kern + (rr.x * rr.x + rr.y * rr.y + rr.z * rr.z);
    const double w = k2 / (den * den);  // Registration.cpp: square(kernel) / square(kernel + residual2)
    const double cx = s.y * rr.z - s.z * rr.y, cy = s.z * rr.x - s.x * rr.z, cz = s.x * rr.y - s.y * rr.x;
    M[0] += w;
    M[1] += w * s.x; M[2] += w * s.y; M[3] += w * s.z;
    M[4] += w * (s.x * s.x); M[5] += w * (s.x * s.y); M[6] += w * (s.x * s.z);
    M[7] += w * (s.y * s.y); M[8] += w * (s.y * s.z); M[9] += w * (s.z * s.z);
    M[10] += w * rr.x; M[11] += w * rr.y; M[12] += w * rr.z;
    M[13] += w * cx; M[14] += w * cy; M[15] += w * cz;
    M[16] += 1.0;
}
// GC (compile time): the number of cooperating workgroups when it is 32 / 16 / 8 (the batched runner's shares of an XCD:
// the exchange then unrolls over exactly that many rows), 0 = any (run-time G).
// What phase A needs from memory for one point: its previous position and (after the first iteration) its voxel key and
// answer row.  None of it depends on the increment of the iteration, so the loads are issued early - the next chunk's
// before the current chunk is evaluated, the next iteration's first chunk before the sums are exchanged and the system
// is solved - and phase A itself does not wait for memory.
struct Gn8Pre {
    double px, py, pz;
    unsigned long long key;
    double2 r[GN8_ANS_ROW / 2];  // the answer row
};
// want_pos: the point's previous position comes from memory (the scan's first iteration: src0; later: src_cur for the points
// that do not fit the workgroup's LDS copy, gn8_body)
__device__ __forceinline__ Gn8Pre gn8_preload(const Ctx& c, int i, bool valid, bool first, bool want_pos) {
    Gn8Pre p;
    p.px = p.py = p.pz = 0.0; p.key = EMPTY_KEY;
#pragma unroll
    for (int k = 0; k < GN8_ANS_ROW / 2; ++k) p.r[k] = make_double2(-1.0, -1.0);  // (bound < 0: no answer)
    if (valid) {
        const double* sp0 = first ? c.src0 : c.src_cur;
        if (want_pos) { p.px = sp0[3 * (size_t)i]; p.py = sp0[3 * (size_t)i + 1]; p.pz = sp0[3 * (size_t)i + 2]; }
        if (!first) {  // (the voxel key of the row's s0 travels in the row: no separate read of pc_key[])
            const double2* row = (const double2*)(c.pc_ans + GN8_ANS_ROW * (size_t)i);
#pragma unroll
            for (int k = 0; k < GN8_ANS_ROW / 2; ++k) p.r[k] = row[k];
            const double kd = ((GN8_ANS_D + 2) & 1) ? p.r[(GN8_ANS_D + 2) / 2].y : p.r[(GN8_ANS_D + 2) / 2].x;
            p.key = (unsigned long long)__double_as_longlong(kd);
        }
    }
    return p;
}
template <int PC, int GC>
__device__ __forceinline__ void gn8_body(const Ctx& c_in, int mode, const int G_rt, const int wg) {
    // the base pointers of the loop's arrays in scalar registers (the free-running kernel's accesses are flat_*: left alone, each
    // base sits in a vector register pair for the whole loop)
    Ctx c = c_in;
    c.pc_ans = uniform_ptr(c.pc_ans); c.pc_key = uniform_ptr(c.pc_key); c.pc_pb = uniform_ptr(c.pc_pb);
    c.tab = uniform_ptr(c.tab); c.blocks = uniform_ptr(c.blocks);
    c.src0 = uniform_ptr(c.src0); c.src_cur = uniform_ptr(c.src_cur);
    c.gn_rows_ll = uniform_ptr(c.gn_rows_ll);
    const int G = GC > 0 ? GC : G_rt;
    __shared__ int2 missq[(GN8_MAX_THREADS / 64) * GN8_QWAVE];  // points whose answer row did not settle them: (index, slot of its position in posL or -1)
    __shared__ int qcount[2][GN8_MAX_THREADS / 64];   // per wavefront: entries at the front / at the back of its region
    __shared__ double part[GN8_ROW_ENTRIES][16];
    __shared__ double redL8[64][GN8_ROW_ENTRIES];     // everybody's rows (G != 32)
    __shared__ double mom[GN8_ROW_ENTRIES];
    __shared__ double tot[32];
    __shared__ double Esh2[2][12];
    __shared__ double Tsh[12];
    __shared__ int flag_done2[2];
    __shared__ long long cand_total_sh;
    __shared__ unsigned xcnt[4];  // executed-work counters of this workgroup and scan: searches | rows rebuilt | stored points read | points settled as 'nothing in reach'
    // The current positions of the workgroup's source points live in LDS for the whole loop (every iteration moves every point:
    // 24 B read + 24 B written per point-iteration otherwise, a sixth of the loop's memory traffic): chunk q of the workgroup
    // (blockDim points) at [q blockDim, (q + 1) blockDim).  Chunks beyond the array (a team of 2 on a scan of > 6144 source
    // points) keep theirs in src_cur.
    __shared__ double posL[3][GN8_LDS_PTS];
    DevState* st = c.st;
    const int tid = threadIdx.x;
    const int NT = blockDim.x, NG32 = blockDim.x >> 5, NW = blockDim.x >> 6;
    if (tid < 4) xcnt[tid] = 0u;
    const int n = __builtin_amdgcn_readfirstlane(st->n_src);
    const unsigned epoch = (unsigned)__builtin_amdgcn_readfirstlane((int)st->gn_epoch);
    if (wg == 0 && tid == 0 && mode == 0) flush_map_stats(c, st);
    if (st->n_live == 0) {  // voxel_map.Empty() => return initial_guess
        if (wg == 0 && tid == 0) {
            Rt I = rt_identity();
            rt_to16(I, st->T_icp);
            st->gn_iters = 0; st->gn_ncorr = 0; st->gn_cand = 0;
            gn_post(c, st, true, mode == 0);
        }
        return;
    }
    if (wg == st->dbg_dead_wg) return;  // test hook: a workgroup that never arrives
    // the loop's uniform values sit in scalar registers (they arrive through vector loads: left alone, each would hold a vector
    // register pair for the whole loop - at the 256-register cap that is what gets spilled)
    const double kern = uniform_f64(st->gn_kernel), k2 = uniform_f64(kern * kern);
    const double gate2 = uniform_f64(sqrt_gate(st->gn_max_dist));
    const double conv2 = uniform_f64(sqrt_gate(c.conv));
    const double inv_vs = uniform_f64(1.0 / c.vs);
    if (tid < 12) Tsh[tid] = (tid < 9) ? ((tid % 4 == 0) ? 1.0 : 0.0) : 0.0;
    if (tid == 64) {  // the guess as a Sophus::SE3d would hold it (rotation through a unit quaternion)
        const Rt g = rt_project(rt_from16(guess_src(c)));
        for (int k = 0; k < 9; ++k) Esh2[1][k] = g.R[k];
        for (int k = 0; k < 3; ++k) Esh2[1][9 + k] = g.t[k];
    }
    if (tid == NT - 1) cand_total_sh = 0;
    __syncthreads();
    // The scan's source points are dealt to the workgroups in blocks of 64 consecutive points, round robin: block q of this
    // workgroup is global block q G + wg, a wavefront of phase A takes one block (its loads coalesce), and every workgroup
    // gets an even sample of the scan - the number of points that need a full search is then the same everywhere (with
    // one contiguous share per workgroup it ranged from 48 to 73 of 203 per iteration, and a workgroup with more than
    // blockDim / 8 of them pays a second search pass while the others wait).  The sequence's workgroups share one L2, so
    // the voxels they touch are resident there either way.
    const int nblk = (n + 63) >> 6;
    const int my_blocks = (nblk - wg + G - 1) / G;  // blocks wg, wg + G, ...
    int iters = 0;
    [[maybe_unused]] long long ph_miss = 0, ph_a = 0, pb_t[5] = {0, 0, 0, 0, 0};
    long long ph[5] = {0, 0, 0, 0, 0};  // (only with -DGN_PHASE_CLOCKS) point loop | wg reduce + publish | exchange | - | totals + solve
    // this lane's point of chunk qb: block qb + wavefront of this workgroup, i.e. global block (qb + wavefront) G + wg
    // tl = tid, made opaque at the top of every iteration: what the loop derives from the lane number (LDS and row addresses of the
    // reduction, the exchange and the solve, the first chunk's addresses) is then computed where it is used - a few integer
    // instructions - instead of once before the loop.  Hoisted, those values outlived the point loop's 256 registers in scratch
    // memory, and the serial tail of every iteration fetched them back one dependent round trip at a time (about 25 of them).
    int tl = tid;
    auto chunk_point = [&](int qb, int& i) -> bool { const int q = qb + (tl >> 6); i = ((q * G + wg) << 6) + (tl & 63); return q < my_blocks && i < n; };
    auto pos_in_lds = [&](int qb) -> bool { return (qb / NW + 1) * NT <= GN8_LDS_PTS; };
    // (two chunks ahead - make PF=2 of rounds 3-5 - was no faster: the second row buffer spills; profiles/r03_b_geometry_and_variants.txt)
    auto preload_chunk = [&](int qb, bool first) -> Gn8Pre {
        int i2;
        const bool v2 = chunk_point(qb, i2);
        return gn8_preload(c, i2, v2, first, first || !pos_in_lds(qb));
    };
    Gn8Pre pre = preload_chunk(0, true);
    for (int it = 0; it < c.max_iter; ++it) {
        asm volatile("" : "+v"(tl));
        [[maybe_unused]] const int lane32 = tl & 31, grp32 = tl >> 5;
        const long long c0 = GN_CLK();
        const double* Esh = Esh2[(it + 1) & 1];
        if (tl == 64 && it > 0) gn_compose(Esh, Tsh);
        double M[GN8_ROW_ENTRIES];
#pragma unroll
        for (int e = 0; e < GN8_ROW_ENTRIES; ++e) M[e] = 0.0;
        int nq = 0, nqr = 0;  // this WAVEFRONT's queued points: row valid (from the front of its region) | row to be rebuilt (from the back)
        int nemp = 0;         // this WAVEFRONT's points settled as "nothing in reach" in this iteration
        for (int qb = 0; qb < my_blocks; qb += NW) {
            int missA = -1;  // this lane's point of phase A when the answer row did not settle it
            int nempty = 0;  // ... 1 when it was settled as "nothing in reach" (executed-work counter)
            bool rebA = true;  // ... and whether its probe row has to be rebuilt (first iteration, or the point changed voxel)
            int liA = -1;      // ... and where its position lives: slot in posL, or -1 = src_cur
#ifdef GN_PHASE_CLOCKS
            const long long pa0 = GN_CLK();
#endif
            // ---- phase A, ONE LANE PER POINT (one pass, one memory round trip): apply the increment, look at the point's
            // answer row.  The last full search of the point (at s0, same voxel => same 27-voxel candidate set) found t and
            // every other candidate - scanned, or inside a dropped voxel's box - at >= D from s0.  After a move by
            // delta = |s - s0| every other candidate is still >= D - delta away (triangle inequality), so t is the strict
            // winner while |s - t| < D - delta, and then a full search would return exactly t: the pair goes straight into
            // this lane's sums.  The test uses the CURRENT distance to t: Gauss-Newton moves a point towards its neighbour
            // more often than away from it, and a move towards t costs no margin at all (with the distance at s0 instead,
            // d0 + 2 delta < D, 88 % of the repeated searches returned the neighbour they already had).  The row keeps the
            // GN8_KCAND nearest candidates of the search, "t" is whichever of them is nearest NOW (the search's own comparison,
            // order ids included) and D bounds everybody else.  Row: s0 (3) | the candidates (3 each) | D less a safety
            // margin, < 0 = no answer | order ids, candidate count of the 27 voxels, candidates in the row (packed).  The
            // distance, the gate and the weight come from the current s either way - same values as after a search.
            {
                int i;
                const bool valid = chunk_point(qb, i);
                const Gn8Pre cur = pre;
                const bool lds = pos_in_lds(qb);
                const int li = (qb / NW) * NT + tl;
                if (qb + NW < my_blocks) pre = preload_chunk(qb + NW, it == 0);  // in flight while this chunk is evaluated
                int miss = -1;
                if (valid) {
                    Rt E;
                    for (int k = 0; k < 9; ++k) E.R[k] = Esh[k];
                    for (int k = 0; k < 3; ++k) E.t[k] = Esh[9 + k];
                    const V3 p0 = (it == 0 || !lds) ? v3(cur.px, cur.py, cur.pz) : v3(posL[0][li], posL[1][li], posL[2][li]);
                    const V3 s = rt_apply(E, p0);
                    if (lds) { posL[0][li] = s.x; posL[1][li] = s.y; posL[2][li] = s.z; liA = li; }
                    else { c.src_cur[3 * (size_t)i] = s.x; c.src_cur[3 * (size_t)i + 1] = s.y; c.src_cur[3 * (size_t)i + 2] = s.z; }
                    miss = i;
                    if (it > 0) {
                        const unsigned long long old_key = cur.key;
                        const int kx = voxel_index(s.x, c.vs, inv_vs), ky = voxel_index(s.y, c.vs, inv_vs), kz = voxel_index(s.z, c.vs, inv_vs);
                        if (old_key == pack_key(kx, ky, kz)) {
                            rebA = false;
                            double row[GN8_ANS_ROW];
#pragma unroll
                            for (int k = 0; k < GN8_ANS_ROW / 2; ++k) { row[2 * k] = cur.r[k].x; row[2 * k + 1] = cur.r[k].y; }
                            const double ex = s.x - row[0], ey = s.y - row[1], ez = s.z - row[2];
                            const double delta2 = ex * ex + ey * ey + ez * ez;
                            const unsigned long long meta = (unsigned long long)__double_as_longlong(row[GN8_ANS_D + 1]);
                            const int nvalid = (int)(meta >> 60), ctot_row = (int)((meta >> 50) & 0x3FFull);
                            // whichever of the row's candidates is nearest NOW, by the search's own comparison (distance, then visiting order)
                            double m = 1.7976931348623157e308, m2 = 1.7976931348623157e308;  // ... and the runner-up's distance (the budget below)
                            unsigned mo = 0xFFFFFFFFu;
                            V3 t = v3(0.0, 0.0, 0.0);
#pragma unroll
                            for (int j = 0; j < GN8_KCAND; ++j) {
                                const V3 cj = v3(row[3 + 3 * j], row[4 + 3 * j], row[5 + 3 * j]);
                                const double dx = cj.x - s.x, dy = cj.y - s.y, dz = cj.z - s.z;
                                const double d2 = dx * dx + dy * dy + dz * dz;  // the expression the search evaluates for this candidate
                                const unsigned oj = (unsigned)((meta >> (10 * j)) & 0x3FFull);
                                if (j < nvalid && (d2 < m || (d2 == m && oj < mo))) { m2 = m; m = d2; mo = oj; t = cj; }
                                else if (j < nvalid && d2 < m2) m2 = d2;
                            }
                            const double dm = sqrt(m), dd = sqrt(delta2);
                            if (nvalid > 0 && dm + dd < row[GN8_ANS_D]) {  // (bound < 0: no answer stored)
                                M[17] += (double)ctot_row;
                                if (m < gate2) gn8_accumulate(M, s, t, kern, k2);
                                miss = -1;
                            } else if (row[GN8_ANS_D] == GN8_NOTHING_IN_REACH) {
                                // The last search of this point found its 27 voxels EMPTY, and it still sits in the same voxel: the same 27
                                // voxels, and the map does not change inside a registration - a search would find nothing again
                                // (GetCorrespondences: no neighbour, no pair, no candidate).  Settled without one.  With 0.1 m voxels
                                // (BASELINE config 5) 5-30 % of a sweep's source points look into empty neighbourhoods, at every one of
                                // a scan's ~50 iterations: most of its repeated searches were these.
                                miss = -1;
                                nempty = 1;
                            }
                        }
                    }
                }
                missA = miss;
            }
#ifdef GN_PHASE_CLOCKS
            const long long pa1 = GN_CLK();
#endif
            // The noted points join the queue: every wavefront has its own region and fills it in point order - points whose probe
            // row is still valid from the front, points that need the 27 hash probes first from the back (a wavefront of the
            // search serves 8 points and pays for whatever ONE of them needs: sorted by kind only the wavefronts at the seams
            // are mixed).  Positions come from a ballot: no shuffle, no LDS traffic, no workgroup barrier per chunk - a fast
            // wavefront goes on to its next chunk while a slow one still waits for its loads (the barrier cost 3 k ticks per
            // chunk, mostly that wait).  The order of the queue - wavefront by wavefront, chunk by chunk - depends on the
            // data alone: deterministic.
            {
                int tq = tl;  // (opaque per chunk: the region's base and the lane mask are two instructions, not a spilled register each)
                asm volatile("" : "+v"(tq));
                const unsigned long long bm = __ballot(missA >= 0), br = __ballot(missA >= 0 && rebA);
                const unsigned long long bf = bm & ~br, below = (1ull << (tq & 63)) - 1ull;
                if (missA >= 0) {
                    const int base = (tq >> 6) * GN8_QWAVE;
                    const int pos = rebA ? base + GN8_QWAVE - 1 - (nqr + __popcll(br & below)) : base + nq + __popcll(bf & below);
                    missq[pos] = make_int2(missA, liA);
                }
                nq += __popcll(bf);   // (this wavefront's counts)
                nqr += __popcll(br);
                nemp += __popcll(__ballot(nempty != 0));
#ifdef GN_PHASE_CLOCKS
                ph_miss += __popcll(bm); ph_a += GN_CLK() - c0;
#endif
            }
#ifdef GN_PHASE_CLOCKS
            const long long pa2 = GN_CLK();
            if (wg == 0 && tl == 0 && it > 0) { atomicAdd((unsigned long long*)&c.wg_clk[56], (unsigned long long)(pa1 - pa0)); atomicAdd((unsigned long long*)&c.wg_clk[57], (unsigned long long)(pa2 - pa1)); atomicAdd((unsigned long long*)&c.wg_clk[59], 1ull); }
#endif
            if (qb + NW < my_blocks && ((qb / NW + 1) % GN8_QCHUNKS) != 0) continue;  // (a region holds GN8_QCHUNKS chunks whatever they bring)
            if ((tl & 63) == 0) { qcount[0][tl >> 6] = nq; qcount[1][tl >> 6] = nqr; }
            nq = 0; nqr = 0;
            __syncthreads();
            // search slots: the fronts of the regions, wavefront by wavefront, then - from the next wavefront's first slot on - the backs
            int cf[GN8_MAX_THREADS / 64], cb[GN8_MAX_THREADS / 64];
            int nfront = 0, nback = 0;
#pragma unroll
            for (int w = 0; w < GN8_MAX_THREADS / 64; ++w) {
                cf[w] = w < NW ? qcount[0][w] : 0; cb[w] = w < NW ? qcount[1][w] : 0;
                nfront += cf[w]; nback += cb[w];
            }
            const int kback0 = (nfront + 7) & ~7, nmiss = kback0 + nback;
            if (tl == 0) xcnt[0] += (unsigned)(nfront + nback);
#ifdef GN_PHASE_CLOCKS
            if (wg == 0 && tl == 0 && it < 24) st->dbg_sums[8 + it] += (double)(nfront + nback);  // misses by iteration index
#endif
            // queue slot of search slot k
            auto qslot = [&](int k) -> int {
                int j = k < nfront ? k : k - kback0, q = -1;
#pragma unroll
                for (int w = 0; w < GN8_MAX_THREADS / 64; ++w) {
                    const int cw = k < nfront ? cf[w] : cb[w];
                    if (q < 0 && j < cw) q = k < nfront ? w * GN8_QWAVE + j : w * GN8_QWAVE + GN8_QWAVE - 1 - j;
                    j -= cw;
                }
                return q;
            };
            // ---- phase B, GN8_LPB LANES PER POINT: the full search of the queued points (gn8_search)
            auto phaseB = [&](auto lp_tag) {
                constexpr int LPB = decltype(lp_tag)::value;
                const int laneL = tl & (LPB - 1), gb = (tl & 63) & ~(LPB - 1);
                for (int k = tl / LPB; __any(k < nmiss); k += NT / LPB) {
                  if (k < nfront || (k >= kback0 && k < nmiss)) {
                    [[maybe_unused]] const long long b0 = GN_CLK();
                    const int2 qe = missq[qslot(k)];
                    const int i = qe.x;
                    const V3 s = qe.y >= 0 ? v3(posL[0][qe.y], posL[1][qe.y], posL[2][qe.y])  // (the point's position now: from the LDS copy,
                                           : v3(c.src_cur[3 * (size_t)i], c.src_cur[3 * (size_t)i + 1], c.src_cur[3 * (size_t)i + 2]);  // or what phase A has just written)
                    V3 t;
                    double m;
                    bool found;
                    int ctot;
#ifdef GN_PHASE_CLOCKS
                    // why did the answer row not settle this point?  (wg 0, iterations > 0): voxel changed | same neighbour again | another one
                    double old_t[3] = {0, 0, 0}, old_sl = -1.0;
                    bool old_same_key = false;
                    if (wg == 0 && it > 0 && laneL == 0) {
                        const int kx0 = voxel_index(s.x, c.vs, inv_vs), ky0 = voxel_index(s.y, c.vs, inv_vs), kz0 = voxel_index(s.z, c.vs, inv_vs);
                        old_same_key = c.pc_key[i] == pack_key(kx0, ky0, kz0);
                        old_t[0] = c.pc_ans[GN8_ANS_ROW * (size_t)i + 3]; old_t[1] = c.pc_ans[GN8_ANS_ROW * (size_t)i + 4]; old_t[2] = c.pc_ans[GN8_ANS_ROW * (size_t)i + 5];
                        old_sl = c.pc_ans[GN8_ANS_ROW * (size_t)i + GN8_ANS_D];
                    }
#endif
                    gn8_search<PC, LPB>(c, i, it, s, inv_vs, laneL, gb, t, m, found, ctot, xcnt);
#ifdef GN_PHASE_CLOCKS
                    if (wg == 0 && it > 0 && laneL == 0) {
                        const int cat = !old_same_key ? 5 : (old_sl < 0.0) ? 6 : (found && t.x == old_t[0] && t.y == old_t[1] && t.z == old_t[2]) ? 7 : 4;
                        atomicAdd(&st->dbg_sums[cat], 1.0);  // 5 voxel changed | 6 no answer stored | 7 same neighbour | 4 another neighbour
                    }
#endif
                    if (laneL == 0) {
                        M[17] += (double)ctot;
                        if (found && m < gate2) gn8_accumulate(M, s, t, kern, k2);
                    }
#ifdef GN_PHASE_CLOCKS
                    { pb_t[0] += GN_CLK() - b0; pb_t[4] += 1; }
#endif
                  }
                }
            };
            if (GN8_IT0_LP != GN8_LPB && it == 0) phaseB(std::integral_constant<int, GN8_IT0_LP>{});  // (every point searches: more points per pass)
            else phaseB(std::integral_constant<int, GN8_LPB>{});
            __syncthreads();  // the queue is reused by the next chunk
#ifdef GN_PHASE_CLOCKS
            if (wg == 0 && tl == 0 && it > 0) { atomicAdd((unsigned long long*)&c.wg_clk[58], (unsigned long long)(GN_CLK() - pa2)); atomicAdd((unsigned long long*)&c.wg_clk[60], (unsigned long long)((nmiss + NT / GN8_LPB - 1) / (NT / GN8_LPB))); }
#endif
            {   // (queue flushed before the last chunk: its loads are requested again rather than carried across the search;
                // after the last chunk nothing is loaded - chunk_point says so - and the stale values are dead for the compiler too)
                pre = preload_chunk(qb + NW, it == 0);
            }
        }
        // the next iteration's first chunk: positions and answer rows as this iteration leaves them (the searches above have
        // written theirs), requested now - they arrive while the sums are exchanged and the system is solved
        pre = preload_chunk(0, false);
        if ((tl & 63) == 0 && nemp) atomicAdd(&xcnt[3], (unsigned)nemp);
        const long long c1 = GN_CLK();
        // ---- workgroup reduction, fixed tree: the 64 lanes of a wavefront (DPP inside the rows, two crossbar steps across
        // them), then the wavefronts in order
#pragma unroll
        for (int e = 0; e < GN8_ROW_ENTRIES; ++e) {
            double v = M[e];
            v += dpp_f64<0xB1>(v);   // lane ^ 1
            v += dpp_f64<0x4E>(v);   // lane ^ 2
            v += dpp_f64<0x141>(v);  // half-row mirror
            v += dpp_f64<0x140>(v);  // row mirror
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            M[e] = v;
        }
        if ((tl & 63) == 0) {
#pragma unroll
            for (int e = 0; e < GN8_ROW_ENTRIES; ++e) part[e][tl >> 6] = M[e];
        }
        __syncthreads();
        const unsigned flag = gn_flag(epoch, it);
        const int par = it & 1, ngroups = G < 8 ? G : 8;
        bool ok = true;
        if (tl < 2 * GN8_ROW_ENTRIES) {
            const int e = tl >> 1;
            double rowv = 0.0;
            for (int j = 0; j < NW; ++j) rowv += part[e][j];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(rowv);
            const unsigned half = (tl & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            __hip_atomic_store(&c.gn_rows_ll[((size_t)par * G + wg) * GN_LL_WORDS + tl],
                               (unsigned long long)half | ((unsigned long long)flag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long c2 = GN_CLK();
        // ---- one-hop exchange.  G == 32 / 16 / 8 (the batched runner's share of an XCD for 1 / 2 / 4 sequences per XCD): the FIRST WAVEFRONT alone polls all 32 rows -
        // lane w <-> word w of every row, the 32 loads of a round in flight together - adds them in (group, member) order in
        // registers and solves straight away: no LDS staging, no workgroup barrier between the last arrival and the
        // factorisation.  Other G: every 32-lane group polls one row into LDS, then the sums.
        long long c3 = c2;
        if (GC > 0) {
            if (tl < 64) {
                const bool mine = tl < 2 * GN8_ROW_ENTRIES;
                unsigned h[GC > 0 ? GC : 32];
                unsigned spins = 0;
                for (;;) {
                    unsigned bad = 0u;
#pragma unroll
                    for (int j = 0; j < (GC > 0 ? GC : 32); ++j) {
                        unsigned long long vv = (unsigned long long)flag << 32;
                        if (mine) vv = __hip_atomic_load(c.gn_rows_ll + ((size_t)par * G + j) * GN_LL_WORDS + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bad |= (unsigned)(vv >> 32) ^ flag;
                        h[j] = (unsigned)vv;
                    }
                    if (__all(bad == 0u)) break;
                    ++spins;
                    if (spins > GN_LL_SPINS || ((spins & 1023u) == 0u && gn_abort_seen(&st->gn_abort))) { ok = false; break; }
                }
                double t = 0.0;
#pragma unroll
                for (int g = 0; g < (GC < 8 ? GC : 8); ++g) {  // members g, g + 8, ... of group g in order, then the groups in order
                    double sg = 0.0 + gn_ll_join(h[g]);
                    if (GC > 8) sg += gn_ll_join(h[(g + 8) % (GC > 0 ? GC : 32)]);
                    if (GC > 16) { sg += gn_ll_join(h[(g + 16) % (GC > 0 ? GC : 32)]); sg += gn_ll_join(h[(g + 24) % (GC > 0 ? GC : 32)]); }
                    t += sg;
                }
                if (mine && (tl & 1) == 0) mom[tl >> 1] = t;
                if (!ok && tl == 0) gn_raise_abort(st);
                c3 = GN_CLK();
            }
        } else if (G <= 64) {
            for (int j = grp32; j < G; j += NG32) {
                const unsigned long long* row = c.gn_rows_ll + ((size_t)par * G + j) * GN_LL_WORDS;
                unsigned h0, h1;
                gn_ll_wait2(row + lane32, row + lane32 + 32, lane32 + 32 < 2 * GN8_ROW_ENTRIES, flag, &h0, &h1, &ok, &st->gn_abort);
                const double v0 = gn_ll_join(h0), v1 = gn_ll_join(h1);
                if ((lane32 & 1) == 0) {
                    redL8[j][lane32 >> 1] = v0;
                    if (16 + (lane32 >> 1) < GN8_ROW_ENTRIES) redL8[j][16 + (lane32 >> 1)] = v1;
                }
            }
            if (!ok) gn_raise_abort(st);
            ok = __syncthreads_or(ok ? 0 : 1) == 0;
            c3 = GN_CLK();
            if (tl < GN8_ROW_ENTRIES) {
                double t = 0.0;
                for (int g = 0; g < ngroups; ++g) {
                    double sg = 0.0;
                    for (int j = g; j < G; j += 8) sg += redL8[j][tl];
                    t += sg;
                }
                mom[tl] = t;
            }
        } else {
            // one sequence over the whole chip (G up to 512): the two-hop exchange of the 32-lane kernel - rows -> the
            // leader of each group (wg & 7) -> 8 group sums to every consumer slot - with this kernel's 18-entry rows
            if (wg < ngroups) {
                const int nmem = (G - wg + 7) / 8;
                for (int j = grp32; j < nmem; j += NG32) {
                    const unsigned long long* row = c.gn_rows_ll + ((size_t)par * G + (wg + 8 * j)) * GN_LL_WORDS;
                    unsigned h0, h1;
                    gn_ll_wait2(row + lane32, row + lane32 + 32, lane32 + 32 < 2 * GN8_ROW_ENTRIES, flag, &h0, &h1, &ok, &st->gn_abort);
                    const double v0 = gn_ll_join(h0), v1 = gn_ll_join(h1);
                    if ((lane32 & 1) == 0) {
                        redL8[j][lane32 >> 1] = v0;
                        if (16 + (lane32 >> 1) < GN8_ROW_ENTRIES) redL8[j][16 + (lane32 >> 1)] = v1;
                    }
                }
                if (!ok) gn_raise_abort(st);
                __syncthreads();
                if (tl < GN8_ROW_ENTRIES) {  // sum in member order; both halves of the entry to every consumer slot
                    double sm = 0.0;
                    for (int j = 0; j < nmem; ++j) sm += redL8[j][tl];
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(sm), fl = (unsigned long long)flag << 32;
                    const unsigned long long lo = (bits & 0xFFFFFFFFull) | fl, hi = (bits >> 32) | fl;
#pragma unroll
                    for (int r = 0; r < GN_XSUM_COPIES; ++r) {
                        unsigned long long* dst = c.gn_xsum_ll + (((size_t)par * 8 + r) * 8 + wg) * GN_LL_WORDS + 2 * tl;
                        __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (tl < 64) {  // everybody: the group sums addressed to this workgroup's slot, added in group order
                const bool mine = tl < 2 * GN8_ROW_ENTRIES;
                bool ok2 = true;
                unsigned h[8];
                unsigned spins = 0;
                for (;;) {
                    unsigned bad = 0u;
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        unsigned long long vv = (unsigned long long)flag << 32;
                        if (mine && g < ngroups)
                            vv = __hip_atomic_load(c.gn_xsum_ll + (((size_t)par * 8 + (wg & (GN_XSUM_COPIES - 1))) * 8 + g) * GN_LL_WORDS + tl,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bad |= (unsigned)(vv >> 32) ^ flag;
                        h[g] = (unsigned)vv;
                    }
                    if (__all(bad == 0u)) break;
                    ++spins;
                    if (spins > GN_LL_SPINS || ((spins & 1023u) == 0u && gn_abort_seen(&st->gn_abort))) { ok2 = false; break; }
                }
                double t = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const double v = gn_ll_join(h[g]);
                    if (g < ngroups) t += v;
                }
                if (mine && (tl & 1) == 0) mom[tl >> 1] = t;
                if (!ok2 && tl == 0) gn_raise_abort(st);
                ok = ok && ok2;
                c3 = GN_CLK();
            }
        }
        if (tl < 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {   // every lane takes all 18 moments into registers (18 broadcast reads in flight together) and picks its entry of the
                // 27 sums there: as a switch over LDS reads the 16 cases ran one after the other, each with its own wait
                // (6.5 k ticks of the iteration's serial tail, profiles/r03_d_gn_phase_clocks...)
                double mm[GN8_ROW_ENTRIES];
#pragma unroll
                for (int e = 0; e < GN8_ROW_ENTRIES; ++e) mm[e] = mom[e];
                if (tl < 27) tot[tl] = sums_from_moments(tl, mm);
                else if (tl == 27) tot[27] = mm[16];
                else if (tl == 28) tot[28] = mm[17];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#ifdef GN_PHASE_CLOCKS
            const long long s1 = GN_CLK();
#endif
            double dx[6];
            solve6_ldlt_wave(tot, tl, dx);
#ifdef GN_PHASE_CLOCKS
            const long long s2 = GN_CLK();
#endif
            if (tl == 0) {
                const Rt e = se3_exp_gn(dx);
                for (int k = 0; k < 9; ++k) Esh2[it & 1][k] = e.R[k];
                for (int k = 0; k < 3; ++k) Esh2[it & 1][9 + k] = e.t[k];
                double nn = 0.0;
                for (int k = 0; k < 6; ++k) nn += dx[k] * dx[k];
                flag_done2[it & 1] = (nn < conv2 || !ok) ? 1 : 0;
            }
#ifdef GN_PHASE_CLOCKS
            if (tl == 0 && wg == 0) { const long long s3 = GN_CLK(); atomicAdd((unsigned long long*)&c.wg_clk[67], (unsigned long long)(s1 - c3)); atomicAdd((unsigned long long*)&c.wg_clk[68], (unsigned long long)(s2 - s1));
                                       atomicAdd((unsigned long long*)&c.wg_clk[69], (unsigned long long)(s3 - s2)); atomicAdd((unsigned long long*)&c.wg_clk[70], 1ull); }
#endif
        }
        __syncthreads();
        const long long c5 = GN_CLK();
        ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[4] += c5 - c3;
#if defined(GN_PHASE_CLOCKS) || defined(GN_IT0_CLOCK)
        if (tl == 0 && wg == 0) atomicAdd((unsigned long long*)&c.wg_clk[52 + (it == 0 ? 0 : 1)], (unsigned long long)(c1 - c0));  // point loop of the first iteration / of the others
#endif
        if (tl == NT - 1) cand_total_sh += (long long)tot[28];
        iters = it + 1;
        if (flag_done2[it & 1]) break;
    }
    if (tid == 64 && iters > 0) gn_compose(Esh2[(iters - 1) & 1], Tsh);
    __syncthreads();
    if (tid == 0 && c.wg_clk) { c.wg_clk[wg] += ph[0]; c.wg_clk[G + wg] += ph_miss; }
    if (tid < 3 && xcnt[tid]) atomicAdd(&st->exec_cnt[tid], (unsigned long long)xcnt[tid]);  // (integer sums: any order gives the same totals)
    if (tid == 3 && xcnt[3]) atomicAdd(&st->empty_cnt, (unsigned long long)xcnt[3]);
    if (wg == 0 && tid == 0) {
        Rt T;
        for (int k = 0; k < 9; ++k) T.R[k] = Tsh[k];
        for (int k = 0; k < 3; ++k) T.t[k] = Tsh[9 + k];
        rt_to16(T, st->T_icp);
        st->gn_iters = iters;
        st->gn_ncorr = iters > 0 ? (int)tot[27] : 0;
        st->gn_cand = cand_total_sh;
        st->exec_cnt[3] += (unsigned long long)iters;
        st->exec_cnt[6] += (unsigned long long)iters * (unsigned long long)n;
        st->exec_cnt[7] += 1ull;
        for (int k = 0; k < 5; ++k) st->gn_phase_clk[k] += ph[k];
        st->gn_phase_clk[5] += iters;
        st->gn_phase_clk[6] += ph_miss; st->gn_phase_clk[7] += ph_a;
#ifdef GN_PHASE_CLOCKS
        for (int k = 0; k < 4; ++k) st->dbg_sums[k] += (double)pb_t[k];
#endif
        gn_post(c, st, false, mode == 0);
    }
}
// one sequence, 8 lanes per point (gn_workgroups <= 64: one-hop exchange; more: two hops through 8 group leaders)
// GC = gridDim.x when that is 32 / 16 / 8 (the host picks the instance), else 0.  The exchange's association is the same
// in every instance: a batch member equals the single run with as many workgroups.  (One body per kernel: each
// instance of gn8_body has its own 40 KB of LDS.)
// (make T256=1 / an occupancy experiment: the 8-lane kernels held to SEQ_WAVES_PER_EU wavefronts per SIMD explicitly - a launch bound
// alone lets the register allocator plan for fewer)
#ifdef SEQ_WAVES_PER_EU
#define SEQ_OCC __attribute__((amdgpu_waves_per_eu(SEQ_WAVES_PER_EU, SEQ_WAVES_PER_EU)))
#else
#define SEQ_OCC
#endif
template <int PC, int GC>
__global__ __launch_bounds__(GN8_MAX_THREADS) SEQ_OCC void k_gn_loop8(Ctx c, int mode) {
    gn8_body<PC, GC>(c, mode, (int)gridDim.x, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ K7-K9
// AddPoints, phase a: world transform, find-or-create the voxel's table entry, join its batch list.
// `pose` null => points are already in the world frame (stage-level API)
// (U points per thread as in K1-K4: the first table read of each, then the compare-and-swaps, then the list pushes travel together)
template <int U>
__device__ __forceinline__ void d_map_insert_a(const Ctx& c, const double* pts_in, const int* n_ptr, int n_fixed,
                                                      int use_pose, const Slice sl) {
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    DevState* st = c.st;
    int idx[U], slot[U];
    bool act[U], keyed[U];
    unsigned long long key[U], cur[U], old[U];
    unsigned s0[U];
    V3 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        idx[u] = base + u * BS; act[u] = idx[u] < n;
        p[u] = v3(0.0, 0.0, 0.0);
        if (act[u]) { const size_t i = (size_t)idx[u]; p[u] = v3(pts_in[3 * i], pts_in[3 * i + 1], pts_in[3 * i + 2]); }
    }
    Rt T = rt_identity();
    if (use_pose) T = rt_from16(st->new_pose);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        slot[u] = -1; keyed[u] = false; key[u] = EMPTY_KEY; s0[u] = 0u; cur[u] = 0ull;
        if (!act[u]) continue;
        const size_t i = (size_t)idx[u];
        if (use_pose) p[u] = rt_apply(T, p[u]);
        c.fdw[3 * i] = p[u].x; c.fdw[3 * i + 1] = p[u].y; c.fdw[3 * i + 2] = p[u].z;
        int kx, ky, kz;
        keyed[u] = vox_key(p[u], c.vs, key[u], kx, ky, kz);
        if (!keyed[u]) { atomicOr(&st->err_flags, ERR_KEY_RANGE); continue; }
        s0[u] = brick_slot(key[u], c.tmask);
        cur[u] = c.tab[s0[u]].key;  // plain (L2-cached) read: EMPTY -> key is the only change in this kernel, so the worst a stale line shows is EMPTY - and then the swap decides
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        old[u] = key[u];
        if (keyed[u] && cur[u] == EMPTY_KEY) old[u] = atomicCAS(&c.tab[s0[u]].key, EMPTY_KEY, key[u]);
    }
    bool created[U], pend[U];
    unsigned sp[U];
    // (The three outcomes are formed as predicates and selects, not as an if / else-if / else chain over slot[u]: with U = 4 the
    // compiler keeps slot[] in one 128-bit register tuple and - ROCm 7.2 hipcc, -O3 - generated for the chain a region in which the
    // lanes that LOST the swap to their own key (cur == EMPTY, old == key: every further point of a new voxel inside one wavefront)
    // had their slot reset to -1 together with the pending lanes and never set again: ERR_TABLE on the first scan, 60 % of the
    // points dropped.  U = 8 compiled correctly.  profiles/r05_a_seq_um4_root_cause.txt has the ISA; the test matrix over U guards it.)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        sp[u] = s0[u];
        const bool was_empty = cur[u] == EMPTY_KEY;
        const bool found = keyed[u] && (cur[u] == key[u] || (was_empty && old[u] == key[u]));
        created[u] = keyed[u] && was_empty && old[u] == EMPTY_KEY;
        slot[u] = (found || created[u]) ? (int)s0[u] : -1;
        pend[u] = keyed[u] && !found && !created[u];  // occupied by another voxel or a tombstone (or the swap lost to another key): linear probing goes on
    }
    // ... in rounds: the next slot of every unsettled point is read, then the swaps on the empty ones are sent, then all are
    // looked at (point by point it was two to three dependent round trips per collision and u)
    for (unsigned probe = 1; probe <= c.tmask; ++probe) {
        bool anyp = false;
#pragma unroll
        for (int u = 0; u < U; ++u) anyp = anyp || pend[u];
        if (!anyp) break;
        unsigned long long ck[U], o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ck[u] = 0ull;
            if (pend[u]) { sp[u] = (sp[u] + 1) & c.tmask; ck[u] = c.tab[sp[u]].key; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o[u] = key[u];
            if (pend[u] && ck[u] == EMPTY_KEY) o[u] = atomicCAS(&c.tab[sp[u]].key, EMPTY_KEY, key[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {  // (predicates and selects, as above)
            const bool was_empty = ck[u] == EMPTY_KEY;
            const bool found = pend[u] && (ck[u] == key[u] || (was_empty && o[u] == key[u]));
            const bool made = pend[u] && was_empty && o[u] == EMPTY_KEY;
            slot[u] = (found || made) ? (int)sp[u] : slot[u];
            created[u] = created[u] || made;
            pend[u] = pend[u] && !found && !made;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (keyed[u] && slot[u] < 0) { atomicOr(&st->err_flags, ERR_TABLE); MAP_DIAG_AT(st, 0, idx[u], s0[u]); }
    // the list pushes need the slot only - sent before the creation chain below (pool pop -> free-stack read -> header), whose round trips
    // they then share instead of following them
    int nx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nx[u] = (slot[u] >= 0) ? atomicExch(&c.tab[slot[u]].head, idx[u]) : -1;
    {   // The new voxels take their blocks from the pool: every step for all U points at once (the pops, then the reads of the
        // free stack, then the headers) - step by step per point it is a chain of dependent memory round trips per u, and some
        // lane of a wavefront creates a voxel for nearly every u.
        // The pool's counters (free tops, live voxels, high-water marks, table entries used) sit in ONE line of the state: an atomic
        // per created voxel on each of them is a queue of same-address atomics at one L2 channel - BASELINE config 5 creates ~30 k
        // voxels per scan, 120 k serialised atomics, and its insert a took 1 007 us of a 1 870 us map update (profiles/
        // r06_a_config5_stage_clocks...).  Round 6: a wavefront counts its creations over all U points with ballots and sends ONE
        // atomic per counter and pass; lane l's r-th creation of point u takes pool index base - 1 - (creations before it in (u, lane)
        // order) - the indices the lanes would have got one by one in that order.
        const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
        const bool lead = (threadIdx.x & 63) == 0;
        int top[U], blk[U];
        bool small[U];
        unsigned long long mc[U];
        int pre[U], tot = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) { mc[u] = __ballot(created[u]); pre[u] = tot; tot += __popcll(mc[u]); top[u] = -1; blk[u] = -1; small[u] = false; }
        if (tot > 0) {  // (uniform over the wavefront)
            int* first_top = c.n_small > 0 ? &st->free_top_s : &st->free_top;  // two classes: a new voxel starts in a small block
            int base = 0;
            if (lead) base = atomicSub(first_top, tot);
            base = __shfl(base, 0);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (created[u]) { top[u] = base - 1 - (pre[u] + __popcll(mc[u] & below)); small[u] = c.n_small > 0; }
            // what the first pool could not serve (its top went below zero): given back in one piece; with two classes those voxels go to
            // the full pool (only when the small one has run out), with one class the pool is exhausted
            unsigned long long mf[U];
            int pre2[U], nf = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) { mf[u] = __ballot(created[u] && top[u] < 0); pre2[u] = nf; nf += __popcll(mf[u]); }
            if (nf > 0) {
                int base2 = 0;
                if (lead) {
                    atomicAdd(first_top, nf);  // (the prune pass pushes at the top: it must not find it negative)
                    if (c.n_small > 0) base2 = atomicSub(&st->free_top, nf);
                }
                base2 = __shfl(base2, 0);
                int back = 0;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (created[u] && top[u] < 0) {
                        small[u] = false;
                        if (c.n_small > 0) top[u] = base2 - 1 - (pre2[u] + __popcll(mf[u] & below));
                    }
                if (c.n_small > 0) {  // ... and the full pool's own shortfall
#pragma unroll
                    for (int u = 0; u < U; ++u) back += __popcll(__ballot(created[u] && top[u] < 0));
                    if (lead && back > 0) atomicAdd(&st->free_top, back);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (created[u] && top[u] >= 0) blk[u] = small[u] ? c.free_stack_s[top[u]] : c.free_stack[top[u]];
            int live = 0, hw_s = 0, hw_f = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!created[u]) continue;
                if (blk[u] >= 0) {
                    int* h = blk_hdr(c, blk[u]);
                    h[0] = 0;
                    h[1] = slot[u];
                    h[2] = 0;  // (points of a batch waiting to be counted in: FUSE form of insert b / prune)
                    ++live;
                    if (small[u]) hw_s = max(hw_s, blk[u] + 1); else hw_f = max(hw_f, blk[u] + 1);
                } else {
                    atomicOr(&st->err_flags, ERR_POOL);
                }
                c.tab[slot[u]].blk = blk[u];
            }
            // one atomic per counter and wavefront: live voxels, the high-water marks, table entries used
            for (int o = 32; o > 0; o >>= 1) {
                live += __shfl_xor(live, o);
                hw_s = max(hw_s, __shfl_xor(hw_s, o));
                hw_f = max(hw_f, __shfl_xor(hw_f, o));
            }
            if (lead) {
                if (live > 0) atomicAdd(&st->n_live, live);
                if (hw_s > 0) atomicMax(&st->pool_hw_s, hw_s);
                if (hw_f > 0) atomicMax(&st->pool_hw, hw_f);
                const unsigned used = atomicAdd(&st->tab_used, (unsigned)tot) + (unsigned)tot;
                if (used > (c.tmask + 1u) / 4u * 3u) { atomicOr(&st->err_flags, ERR_TABLE); MAP_DIAG_AT(st, 1, used, c.tmask); }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (act[u]) { c.pslot[idx[u]] = slot[u]; c.nxt[idx[u]] = nx[u]; }
}
// phase b: rank among this batch's points of the same voxel (by scan order) -> slot in the block
// FUSE (the free-running kernel, where the prune pass always follows): the voxel's first point of the batch leaves the batch's
// length in the block header, and the prune pass - which reads every live header anyway - counts it in, mirrors the count in
// the table entry and resets the list: phase c, its pass over the points with two dependent reads and its barrier are gone.
template <int U, bool FUSE = false>
__device__ __forceinline__ void d_map_insert_b(const Ctx& c, const int* n_ptr, int n_fixed, const Slice sl) {
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    int idx[U], slot[U], j[U], rank[U], len[U], pb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { idx[u] = base + u * BS; slot[u] = (idx[u] < n) ? c.pslot[idx[u]] : -1; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        rank[u] = 0; len[u] = 0; j[u] = -1; pb[u] = -1;
        if (slot[u] >= 0) { const TabEnt e = c.tab[slot[u]]; j[u] = e.head; pb[u] = e.blk; }
        // FUSE: a voxel without a block (the pool ran out: ERR_POOL) never shows up in the prune pass - its points need no rank, so each of
        // them resets the batch list itself (the others either saw the list and drop it here too, or see it empty).  A list left
        // standing would be pushed onto by the next scan and walked in circles.
        if (FUSE && slot[u] >= 0 && pb[u] < 0) { c.tab[slot[u]].head = -1; j[u] = -1; }
    }
    // (the voxel's stored count comes with the table entry - insert c / rebuild mirror it there, a new voxel's entry says 0 - and the point itself is
    // requested before the walk and arrives during it)
    int cnt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cnt[u] = (slot[u] >= 0 && pb[u] >= 0) ? (pb[u] >> 24) : 0;
    double w[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = (size_t)idx[u];
        w[u][0] = w[u][1] = w[u][2] = 0.0;
        if (idx[u] < n && slot[u] >= 0 && pb[u] >= 0) { w[u][0] = c.fdw[3 * i]; w[u][1] = c.fdw[3 * i + 1]; w[u][2] = c.fdw[3 * i + 2]; }
    }
    for (int steps = 0;; ++steps) {  // the U list walks step together: their reads of nxt[] are in flight at the same time
        if (steps > n) { atomicOr(&c.st->err_flags, ERR_TABLE); MAP_DIAG_AT(c.st, 2, idx[0], n); break; }  // (a list longer than the batch: corrupted links must not hang the launch)
        bool any = false;
#pragma unroll
        for (int u = 0; u < U; ++u) any = any || j[u] >= 0;
        if (!any) break;
        int nx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) nx[u] = (j[u] >= 0) ? c.nxt[j[u]] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j[u] >= 0) { ++len[u]; rank[u] += (j[u] < idx[u]) ? 1 : 0; j[u] = nx[u]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (idx[u] >= n) continue;
        const size_t i = (size_t)idx[u];
        if (!FUSE) {
            if (slot[u] < 0) { c.prank[i] = -1; continue; }
            c.prank[i] = rank[u];
            c.plen[i] = len[u];
        }
        if (slot[u] < 0 || pb[u] < 0) continue;
        if (FUSE && rank[u] == 0) blk_hdr(c, pb[u] & BLK_ID_MASK)[2] = len[u];
        if (FUSE && c.n_small > 0) c.prank[i] = rank[u];  // (d_map_migrate places the points a small block has no room for)
        const int pos = cnt[u] + rank[u];
        if (pos < blk_cap(c, pb[u] & BLK_ID_MASK)) {
            double* X = blk_x(c, pb[u] & BLK_ID_MASK);
            X[3 * pos] = w[u][0]; X[3 * pos + 1] = w[u][1]; X[3 * pos + 2] = w[u][2];
            if (pos == 0) { double* F = blk_first(c, pb[u] & BLK_ID_MASK); F[0] = w[u][0]; F[1] = w[u][1]; F[2] = w[u][2]; }  // the directory's copy (prune, rebuild)
        }
    }
}
// phase c: publish the new counts, reset the batch lists
template <int U>
__device__ __forceinline__ void d_map_insert_c(const Ctx& c, const int* n_ptr, int n_fixed, const Slice sl) {
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    int idx[U], slot[U], pb[U], pl[U], cnt[U];
    bool first[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // (rank, slot and list length of a point travel together: the latter two are used by the voxel's first point only)
        idx[u] = base + u * BS;
        const bool in = idx[u] < n;
        const int pr = in ? c.prank[idx[u]] : -1, ps = in ? c.pslot[idx[u]] : -1, pn = in ? c.plen[idx[u]] : 0;
        first[u] = in && pr == 0;
        slot[u] = first[u] ? ps : -1; pl[u] = first[u] ? pn : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) pb[u] = first[u] ? c.tab[slot[u]].blk : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) cnt[u] = (pb[u] >= 0) ? (pb[u] >> 24) : 0;  // the entry's mirror of the block's count: no dependent read of the header
    int added = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!first[u]) continue;
        c.tab[slot[u]].head = -1;
        if (pb[u] >= 0) {
            const int b = pb[u] & BLK_ID_MASK;
            int nc = cnt[u] + pl[u];
            if (nc > c.P) nc = c.P;
            blk_hdr(c, b)[0] = nc;
            c.tab[slot[u]].blk = b | (nc << 24);  // the table entry mirrors the count so a probe returns both
            added += nc - cnt[u];
        }
    }
    // one atomic per wavefront: tens of thousands of same-address atomics serialise at the memory side (225 us for 8
    // sequences in one launch, measured)
    for (int o = 32; o > 0; o >>= 1) added += __shfl_xor(added, o);
    if ((threadIdx.x & 63) == 0 && added) atomicAdd((unsigned long long*)&c.st->map_points, (unsigned long long)added);
}

// ------------------------------------------------------------------------------------------------ K10
// The passes over "every block" walk a VIRTUAL index v: the small blocks below their high-water mark, then the full blocks below
// theirs (block ids are handed out low-first in both classes: nothing lives above the marks).  One class: v = block id.
__device__ __forceinline__ int blk_virtual_count(const Ctx& c, const DevState* st) { return st->pool_hw_s + (st->pool_hw - c.n_small); }
__device__ __forceinline__ int blk_from_virtual(const Ctx& c, int hw_s, int v) { return v < hw_s ? v : c.n_small + (v - hw_s); }
// a block goes back to the free stack of its class
__device__ __forceinline__ void blk_free(const Ctx& c, DevState* st, int b) {
    if (b < c.n_small) {
        const int top = atomicAdd(&st->free_top_s, 1);
        if (top >= 0 && top < c.n_small) c.free_stack_s[top] = b;
    } else {
        const int top = atomicAdd(&st->free_top, 1);
        if (top >= 0 && top < c.pool_cap - c.n_small) c.free_stack[top] = b;
    }
}
// RemovePointsFarFromLocation: a voxel goes when its FIRST point is farther than max_range from the origin
// FUSE: ... and counts in what insert b left in the headers (see there).  With two block classes a small block whose batch takes it
// past its capacity is only NOTED here (mig_list): its move to a full block pops the full pool, which this pass pushes to - the
// migration pass runs behind a team barrier (d_map_migrate).
template <int U, bool FUSE = false>
__device__ __forceinline__ void d_map_prune(const Ctx& c, const double* origin_xyz, int use_new_pose, const Slice sl) {
    const int BS = (int)blockDim.x, base = sl.b * BS * U + (int)threadIdx.x;
    DevState* st = c.st;
    const int hw_s = st->pool_hw_s, nv = hw_s + (st->pool_hw - c.n_small);
    double ox, oy, oz;
    if (use_new_pose) { ox = st->new_pose[3]; oy = st->new_pose[7]; oz = st->new_pose[11]; }
    else { ox = origin_xyz[0]; oy = origin_xyz[1]; oz = origin_xyz[2]; }
    int cnt[U], hs[U], add[U], bid[U];
    double x0[U], y0[U], z0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int v = base + u * BS;
        cnt[u] = 0; hs[u] = -1; add[u] = 0; x0[u] = y0[u] = z0[u] = 0.0; bid[u] = -1;
        if (v < nv) {
            const int b = blk_from_virtual(c, hw_s, v);
            bid[u] = b;
            const int* h = blk_hdr(c, b);
            cnt[u] = h[0]; hs[u] = h[1];
            if (FUSE) add[u] = h[2];
            const double* X = blk_first(c, b);
            x0[u] = X[0]; y0[u] = X[1]; z0[u] = X[2];  // (read whether or not the block is live: no dependent round trip)
        }
    }
    long long added = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int nc = cnt[u];
        if (FUSE && add[u] > 0) { nc += add[u]; if (nc > c.P) nc = c.P; }
        if (nc <= 0) continue;
        const int b = bid[u];
        const double dx = x0[u] - ox, dy = y0[u] - oy, dz = z0[u] - oz;
        if (dx * dx + dy * dy + dz * dz > c.max_range * c.max_range) {
            c.tab[hs[u]].key = TOMB_KEY;
            c.tab[hs[u]].blk = -1;
            blk_hdr(c, b)[0] = 0;
            if (FUSE && add[u] > 0) { blk_hdr(c, b)[2] = 0; c.tab[hs[u]].head = -1; }
            blk_free(c, st, b);
            atomicSub(&st->n_live, 1);
            if (FUSE) added -= cnt[u];  // (what this batch brought was never counted)
            else atomicAdd((unsigned long long*)&st->map_points, (unsigned long long)(-(long long)cnt[u]));
        } else if (FUSE && add[u] > 0) {
            if (nc > blk_cap(c, b)) {  // a small block the batch has outgrown: its points move to a full block behind the next barrier
                const int w = atomicAdd(&st->mig_n, 1);
                if (w < c.n_small) c.mig_list[w] = b;
                continue;
            }
            // what insert c does in the other drivers: publish the new count, reset the batch list
            int* h = blk_hdr(c, b);
            h[0] = nc; h[2] = 0;
            c.tab[hs[u]].head = -1;
            c.tab[hs[u]].blk = b | (nc << 24);  // the table entry mirrors the count so a probe returns both
            added += nc - cnt[u];
        }
    }
    if (FUSE) {  // one atomic per wavefront (see insert c)
        for (int o = 32; o > 0; o >>= 1) added += __shfl_xor(added, o);
        if ((threadIdx.x & 63) == 0 && added != 0) atomicAdd((unsigned long long*)&c.st->map_points, (unsigned long long)added);
    }
}
// A small block the batch has outgrown (noted by the prune pass) moves to a full block: its stored points are copied, the batch's points
// that found no room (position >= SMALL_CAP; insert b left their ranks in prank[]) are placed from the world-frame copy of the batch
// along the voxel's batch list, the directory entry and the table entry follow, the small block goes back to its stack.  One thread per
// voxel; a scan of BASELINE config 5 moves a few thousand.  Runs behind a team barrier after the prune pass: it POPS the full pool,
// which the prune pass pushes to (and pushes the small pool, which nobody pops here).
__device__ __forceinline__ void d_map_migrate(const Ctx& c, const Slice sl) {
    DevState* st = c.st;
    const int t = sl.b * (int)blockDim.x + (int)threadIdx.x;
    const int nm = st->mig_n < c.n_small ? st->mig_n : c.n_small;
    if (t >= nm) return;
    const int b = c.mig_list[t];
    int* h = blk_hdr(c, b);
    const int cnt = h[0], slot = h[1], add = h[2];
    int nc = cnt + add;
    if (nc > c.P) nc = c.P;
    const int top = atomicSub(&st->free_top, 1) - 1;
    if (top < 0) {  // no full block left: the voxel keeps what its small block holds, the loss is flagged
        atomicAdd(&st->free_top, 1);
        atomicOr(&st->err_flags, ERR_POOL);
        h[0] = SMALL_CAP; h[2] = 0;
        c.tab[slot].head = -1;
        c.tab[slot].blk = b | (SMALL_CAP << 24);
        atomicAdd((unsigned long long*)&st->map_points, (unsigned long long)(long long)(SMALL_CAP - cnt));
        return;
    }
    const int nb = c.free_stack[top];
    const double* Xo = blk_x(c, b);
    double* Xn = blk_x(c, nb);
    const int held = (cnt + add < SMALL_CAP) ? cnt + add : SMALL_CAP;  // the stored points + the batch's that found room
    for (int j = 0; j < 3 * held; ++j) Xn[j] = Xo[j];
    int j = c.tab[slot].head;
    for (int steps = 0; j >= 0 && steps <= add; ++steps) {
        const int pos = cnt + c.prank[j];
        if (pos >= SMALL_CAP && pos < c.P) { Xn[3 * pos] = c.fdw[3 * (size_t)j]; Xn[3 * pos + 1] = c.fdw[3 * (size_t)j + 1]; Xn[3 * pos + 2] = c.fdw[3 * (size_t)j + 2]; }
        j = c.nxt[j];
    }
    int* hn = blk_hdr(c, nb);
    hn[0] = nc; hn[1] = slot; hn[2] = 0;
    const double* Fo = blk_first(c, b);
    double* Fn = blk_first(c, nb);
    Fn[0] = Fo[0]; Fn[1] = Fo[1]; Fn[2] = Fo[2];
    c.tab[slot].head = -1;
    c.tab[slot].blk = nb | (nc << 24);
    h[0] = 0; h[2] = 0;
    blk_free(c, st, b);
    atomicMax(&st->pool_hw, nb + 1);
    atomicAdd((unsigned long long*)&st->map_points, (unsigned long long)(long long)(nc - cnt));
}

// ------------------------------------------------------------------------------------------------ K11
// after the table has been reset to EMPTY: re-enter every live voxel
__device__ __forceinline__ void d_map_rebuild(const Ctx& c, const Slice sl) {
    const int v = sl.b * (int)blockDim.x + (int)threadIdx.x;
    const int hw_s = c.st->pool_hw_s;
    if (v >= hw_s + (c.st->pool_hw - c.n_small)) return;
    const int b = blk_from_virtual(c, hw_s, v);
    int* h = blk_hdr(c, b);
    if (h[0] <= 0) return;
    const double* X = blk_first(c, b);
    unsigned long long key; int kx, ky, kz;
    vox_key(v3(X[0], X[1], X[2]), c.vs, key, kx, ky, kz);
    unsigned s = brick_slot(key, c.tmask);
    for (unsigned probe = 0; probe <= c.tmask; ++probe) {
        const unsigned long long old = atomicCAS(&c.tab[s].key, EMPTY_KEY, key);
        if (old == EMPTY_KEY) { c.tab[s].blk = b | (h[0] << 24); h[1] = (int)s; atomicAdd(&c.st->tab_used, 1u); return; }
        s = (s + 1) & c.tmask;
    }
    atomicOr(&c.st->err_flags, ERR_TABLE);
    MAP_DIAG_AT(c.st, 3, b, s);
}

// export of the map points (KissICPWrapper.local_map_points)
__global__ __launch_bounds__(256) void k_map_export(Ctx c, double* out, int* counter, int max_points) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= c.pool_cap) return;
    const int cnt = blk_hdr(c, b)[0];
    if (cnt <= 0) return;
    const int o = atomicAdd(counter, cnt);
    const double* X = blk_x(c, b);
    for (int j = 0; j < cnt; ++j) {
        if (o + j >= max_points) break;
        out[3 * (size_t)(o + j)] = X[3 * j]; out[3 * (size_t)(o + j) + 1] = X[3 * j + 1]; out[3 * (size_t)(o + j) + 2] = X[3 * j + 2];
    }
}

// MotionCompensator.deskew_scan as a stand-alone call (KissICPWrapper.deskew, reference kiss.py:76-78): the same
// per-point transform K1 applies, without filtering; fewer than two poses => the frame is returned unchanged.
__global__ __launch_bounds__(256) void k_deskew_only(Ctx c, const double* xyz, const double* t01, int n, double* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const DevState* st = c.st;
    V3 p = v3(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
    if (c.deskew && st->n_poses >= 2) {
        double xi[6], x[6];
        se3_log(rt_mul(rt_inv(rt_from16(st->pose_prev)), rt_from16(st->pose_last)), xi);
        const double sft = t01[i] - 0.5;
        for (int k = 0; k < 6; ++k) x[k] = sft * xi[k];
        p = rt_apply(se3_exp(x), p);
    }
    out[3 * (size_t)i] = p.x; out[3 * (size_t)i + 1] = p.y; out[3 * (size_t)i + 2] = p.z;
}

// ------------------------------------------------------------------------------------------------ range statistics
// StreamStatsTracker.trackScan's per-scan moments (reference ins/data.py:286-308): over the non-zero ranges of the
// selected beam rows, converted to metres - count, mean, population variance (two passes, like np.mean / np.var),
// min, max.  One workgroup, fixed reduction tree: deterministic; min / max are selections and therefore exact.
// out[5] = {n, mean, var, min, max}
__global__ __launch_bounds__(1024) void k_range_stats(const unsigned* range, int H, int W, const unsigned char* row_mask,
                                                      double to_m, double* out) {
    __shared__ double sh[1024];
    __shared__ double shmin[1024], shmax[1024];
    __shared__ double s_mean, s_n;
    const int tid = threadIdx.x, n_all = H * W;
    for (int pass = 0; pass < 2; ++pass) {
        const double mean = pass ? s_mean : 0.0;
        double acc = 0.0, cnt = 0.0, lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
        for (int i = tid; i < n_all; i += 1024) {
            const unsigned r = range[i];
            if (r == 0 || (row_mask && !row_mask[i / W])) continue;
            const double x = (double)r * to_m;
            if (pass == 0) { acc += x; cnt += 1.0; lo = fmin(lo, x); hi = fmax(hi, x); }
            else { const double d = x - mean; acc += d * d; }
        }
        sh[tid] = acc;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) { if (tid < o) sh[tid] += sh[tid + o]; __syncthreads(); }
        const double total = sh[0];
        __syncthreads();
        if (pass == 0) {
            sh[tid] = cnt; shmin[tid] = lo; shmax[tid] = hi;
            __syncthreads();
            for (int o = 512; o > 0; o >>= 1) {
                if (tid < o) { sh[tid] += sh[tid + o]; shmin[tid] = fmin(shmin[tid], shmin[tid + o]); shmax[tid] = fmax(shmax[tid], shmax[tid + o]); }
                __syncthreads();
            }
            if (tid == 0) {
                s_n = sh[0];
                s_mean = sh[0] > 0.0 ? total / sh[0] : 0.0;
                out[0] = sh[0]; out[1] = s_mean; out[3] = shmin[0]; out[4] = shmax[0];
            }
            __syncthreads();
        } else if (tid == 0) {
            out[2] = s_n > 0.0 ? total / s_n : 0.0;
        }
    }
}

// ================================================================================================ launch wrappers
// Every stage exists as a device function d_*(ctx, ...); k_* runs it for one sequence (context by value), kb_* runs
// it for S sequences in one launch (blockIdx.y = sequence, contexts in device memory): the batched driver
// amortises every launch and, in the Gauss-Newton loop, every grid barrier over S scans.
struct EkfState;
struct SeqCtx {
    Ctx c;
    const float* scan_base;        // sweeps of this sequence, resident in HBM
    long long scan_stride_floats;  // floats between consecutive sweeps
    int input_is_range, n_scans;
    double* fd_buf[2];             // frame_down of scan k lives in buffer k & 1: the map update of scan k reads it while K3 of scan k + 1 writes the other
    // the free-running sequence kernel (seq_kernel.h) also steps the sequence's filter and writes its outputs
    EkfState* ekf;
    const double* imu;             // [n_imu][7]
    const int* imu_end;            // [n_scans] IMU samples that precede scan k
    double *res_poses, *res_t, *rows;
};
__device__ __forceinline__ Ctx load_seq_ctx(const SeqCtx* a, int s, int scan_k) {
    Ctx c = a[s].c;
    c.fd = a[s].fd_buf[scan_k & 1];
    const float* p = a[s].scan_base + (size_t)scan_k * (size_t)a[s].scan_stride_floats;
    if (a[s].input_is_range) { c.in_range = (const unsigned*)p; c.in_f32 = nullptr; }
    else { c.in_f32 = p; c.in_range = nullptr; }
    // the stage bodies' base pointers as explicit scalars (see gn8_body)
    c.in_f32 = uniform_ptr(c.in_f32); c.in_range = uniform_ptr(c.in_range);
    c.pts = uniform_ptr(c.pts); c.slot1 = uniform_ptr(c.slot1); c.slot2 = uniform_ptr(c.slot2);
    c.vtab1 = uniform_ptr(c.vtab1); c.vtab2 = uniform_ptr(c.vtab2); c.bcnt1 = uniform_ptr(c.bcnt1); c.bcnt2 = uniform_ptr(c.bcnt2);
    c.fd = uniform_ptr(c.fd); c.fdw = uniform_ptr(c.fdw); c.src0 = uniform_ptr(c.src0); c.coltab = uniform_ptr(c.coltab);
    c.pslot = uniform_ptr(c.pslot); c.nxt = uniform_ptr(c.nxt); c.prank = uniform_ptr(c.prank); c.plen = uniform_ptr(c.plen);
    c.tab = uniform_ptr(c.tab); c.blocks = uniform_ptr(c.blocks); c.free_stack = uniform_ptr(c.free_stack); c.st = uniform_ptr(c.st);
    c.bhdr = uniform_ptr(c.bhdr); c.bfirst = uniform_ptr(c.bfirst); c.big_base = uniform_ptr(c.big_base);
    c.free_stack_s = uniform_ptr(c.free_stack_s); c.mig_list = uniform_ptr(c.mig_list);
    return c;
}
__global__ __launch_bounds__(1024) void k_scan_prologue(Ctx c) { d_scan_prologue(c); }
__global__ __launch_bounds__(1024) void kb_scan_prologue(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_scan_prologue(c);
}
__global__ __launch_bounds__(256) void k_deskew_vds1(Ctx c) { d_deskew_vds1<1>(c, launch_slice()); }
__global__ __launch_bounds__(256) void kb_deskew_vds1(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_deskew_vds1<1>(c, launch_slice());
}
__global__ __launch_bounds__(256) void k_count_w1(Ctx c) { d_count_w1<1>(c, launch_slice()); }
__global__ __launch_bounds__(256) void kb_count_w1(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_count_w1<1>(c, launch_slice());
}
__global__ __launch_bounds__(256) void k_compact_fd(Ctx c) { d_compact_fd<1, false>(c, launch_slice()); }
__global__ __launch_bounds__(256) void kb_compact_fd(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_compact_fd<1, false>(c, launch_slice());
}
__global__ __launch_bounds__(256) void k_vds2_fd(Ctx c) { d_vds2_fd<1>(c, launch_slice()); }
__global__ __launch_bounds__(256) void kb_vds2_fd(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_vds2_fd<1>(c, launch_slice());
}
__global__ __launch_bounds__(256) void k_count_w2(Ctx c) { d_count_w2<1>(c, launch_slice()); }
__global__ __launch_bounds__(256) void kb_count_w2(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_count_w2<1>(c, launch_slice());
}
__global__ __launch_bounds__(256) void k_compact_src(Ctx c) { d_compact_src<1, false>(c, launch_slice()); }
__global__ __launch_bounds__(256) void kb_compact_src(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_compact_src<1, false>(c, launch_slice());
}
__global__ __launch_bounds__(256) void k_map_insert_a(Ctx c, const double* pts_in, const int* n_ptr, int n_fixed, int use_pose) { d_map_insert_a<1>(c, pts_in, n_ptr, n_fixed, use_pose, launch_slice()); }
__global__ __launch_bounds__(256) void kb_map_insert_a(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_map_insert_a<1>(c, c.fd, &c.st->n_down_ins, 0, 1, launch_slice());
}
__global__ __launch_bounds__(256) void k_map_insert_b(Ctx c, const int* n_ptr, int n_fixed) { d_map_insert_b<1>(c, n_ptr, n_fixed, launch_slice()); }
__global__ __launch_bounds__(256) void kb_map_insert_b(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_map_insert_b<1>(c, &c.st->n_down_ins, 0, launch_slice());
}
__global__ __launch_bounds__(256) void k_map_insert_c(Ctx c, const int* n_ptr, int n_fixed) { d_map_insert_c<1>(c, n_ptr, n_fixed, launch_slice()); }
__global__ __launch_bounds__(256) void kb_map_insert_c(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_map_insert_c<1>(c, &c.st->n_down_ins, 0, launch_slice());
}
__global__ __launch_bounds__(256) void k_map_prune(Ctx c, const double* origin_xyz, int use_new_pose) { d_map_prune<1>(c, origin_xyz, use_new_pose, launch_slice()); }
__global__ __launch_bounds__(256) void kb_map_prune(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_map_prune<1>(c, nullptr, 1, launch_slice());
}
__global__ __launch_bounds__(256) void k_map_rebuild(Ctx c) { d_map_rebuild(c, launch_slice()); }
__global__ __launch_bounds__(256) void kb_map_rebuild(const SeqCtx* a, int scan_k) {
    const Ctx c = load_seq_ctx(a, blockIdx.y, scan_k);
    d_map_rebuild(c, launch_slice());
}

// ------------------------------------------------------------------------------------------------ batched K5
// The Gauss-Newton loops of up to 8 scans (one per sequence) in ONE persistent launch, one sequence per XCD: sequence s
// is owned by the workgroups with blockIdx & 7 == s (what the dispatcher places on XCD s today; speed only - every
// exchanged word travels with agent-scope atomics).  Its map blocks, probe rows and source points stay in that XCD's L2,
// its workgroups meet through the one-hop exchange of gn_loop_body<XL = true> - nothing crosses the chip - and it leaves
// the loop on its own convergence.  Per sequence the point -> workgroup assignment, the reduction trees and the
// arithmetic are those of k_gn_loop launched alone with gridDim / 8 workgroups: bit-identical results.
#define GN_MAX_SEQ 256  /* lockstep: up to four sequences per XCD (32); free-running: up to 32 per XCD, served by its teams (seq_kernel.h) */
// S <= 8: sequence s <-> XCD s with all gridDim / 8 workgroups of that XCD.  S > 8: the XCD's workgroups are split evenly
// among the sequences s, s + 8, s + 16, ... it hosts (spx of them, a power of two): one sequence alone leaves an XCD
// latency-bound and mostly idle (it costs the same per iteration as eight on eight), a second and a fourth loop fill
// the gaps.
__device__ __forceinline__ bool kx_assign(int S, int& s, int& G, int& wg) {
    const int x = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3), J = (int)(gridDim.x >> 3);
    const int spx = S <= 8 ? 1 : (S <= 16 ? 2 : 4);
    G = J / spx;
    s = x + 8 * (j / G);
    wg = j % G;
    return s < S;
}
template <int PC, int GC>
__global__ __launch_bounds__(GN8_MAX_THREADS) SEQ_OCC void kx_gn_loop8(const SeqCtx* a, int S, int scan_k) {
    int s, G, wg;
    if (!kx_assign(S, s, G, wg)) return;
    const Ctx c = load_seq_ctx(a, s, scan_k);
    gn8_body<PC, GC>(c, 0, G, wg);
}
template <int PC>
__global__ __launch_bounds__(GN_MAX_THREADS) void kx_gn_loop(const SeqCtx* a, int S, int scan_k) {
    int s, G, wg;
    if (!kx_assign(S, s, G, wg)) return;
    const Ctx c = load_seq_ctx(a, s, scan_k);
    gn_loop_body<PC, true, true>(c, 0, G, wg);
}
