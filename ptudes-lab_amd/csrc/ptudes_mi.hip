// ptudes_mi.hip -- C-ABI (include/ptudes_mi.h) + host orchestration of the HIP kernels.
//
// One HIP stream per handle; all arithmetic runs in the kernels of icp_kernels.h / ekf_kernels.h.
// The host only sizes buffers, enqueues launches and copies results: there is no CPU compute path.
#include "../../include/ptudes_mi.h"
#include "ekf_kernels.h"
#include "icp_kernels.h"
#include "seq_kernel.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <vector>

static thread_local char g_err[512] = "";
static int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
// (a failed runtime call also leaves its code in the runtime's sticky "last error": it is cleared here, or a LATER, unrelated
// HIPCHK(hipGetLastError()) - behind a kernel launch of another handle - would report this call's failure as its own)
#define HIPCHK(x)                                                                                    \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) { (void)hipGetLastError(); return set_err(PTL_ERR_HIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } \
    } while (0)

extern "C" const char* ptl_last_error(void) { return g_err; }
extern "C" int ptl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int ptl_backend(void) { return ptl_device_count() > 0 ? 1 : 0; }
extern "C" int ptl_host_pin(int device_id, void* host_ptr, uint64_t bytes) {
    if (!host_ptr || bytes == 0) return set_err(PTL_ERR_ARG, "null argument");
    if (ptl_device_count() <= device_id || device_id < 0) return set_err(PTL_ERR_HIP, "no HIP device %d", device_id);
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipHostRegister(host_ptr, (size_t)bytes, hipHostRegisterDefault));
    return PTL_OK;
}
extern "C" int ptl_host_unpin(void* host_ptr) {
    if (!host_ptr) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipHostUnregister(host_ptr));
    return PTL_OK;
}

// ---- ABI guard (include/ptudes_mi.h): a configuration struct says how big its writer thinks it is and which ABI it was written against
extern "C" int ptl_abi_version(void) { return PTL_ABI_VERSION; }
extern "C" int64_t ptl_sizeof_cfg(int which) {
    switch (which) {
        case PTL_CFG_ICP: return (int64_t)sizeof(ptl_icp_cfg);
        case PTL_CFG_EKF: return (int64_t)sizeof(ptl_ekf_cfg);
        case PTL_CFG_SEQ: return (int64_t)sizeof(ptl_seq_cfg);
        case PTL_CFG_ICP_STATS: return (int64_t)sizeof(ptl_icp_stats);
    }
    return -1;
}
// (the two leading words are read only: on a mismatch nothing of the caller's memory is written)
static int abi_check(const char* what, uint32_t size_caller, uint32_t abi_caller, size_t size_lib) {
    if (size_caller == (uint32_t)size_lib && abi_caller == (uint32_t)PTL_ABI_VERSION) return PTL_OK;
    return set_err(PTL_ERR_ARG, "%s: the caller's struct_size = %u, abi_version = %u; this library's sizeof(%s) = %zu, PTL_ABI_VERSION = %d "
                   "- set both before the call (PTL_CFG_INIT) and rebuild / rewrite the binding against this library's include/ptudes_mi.h",
                   what, size_caller, abi_caller, what, size_lib, PTL_ABI_VERSION);
}
#define ABI_CHECK(what, p) do { int rc_ = abi_check(#what, (p)->struct_size, (p)->abi_version, sizeof(what)); if (rc_) return rc_; } while (0)

// ================================================================================================ ICP
struct ptl_icp {
    ptl_icp_cfg cfg;
    hipStream_t stream;
    bool own_stream;
    Ctx c;               // kernel context (device pointers)
    int64_t n_max;
    int nblk_scan;
    void* d_in;          // staging for host-provided scans (n_max * 3 * 8 bytes)
    double* d_t01;
    double* d_ext;       // staging for a host-provided guess
    int* d_counter;
    int64_t traj_cap;
    int64_t scans_done;
    unsigned char* d_row_mask;  // active-beam mask for range-image input, or null
    int64_t last_n;      // points of the previous scan
    bool lazy_map_stats; // per-call API: do not wait for the scan's map update before returning (ptl_icp_set_lazy_map_stats)
    // profiling of the dominant kernel
    bool prof;
    int prof_every;  // time every n-th GN launch (two event records cost ~18 us of command-processor time per scan)
    std::vector<hipEvent_t> ev;
    size_t ev_used;
    // set by the sequence runner around a scan: the GN kernel waits for `gn_wait` (the EKF stream produced the guess)
    hipEvent_t gn_wait, gn_done;  // ... and `gn_done` is recorded right after it (a second event: two streams waiting on one measured slower)
    hipEvent_t ev_step_guess, ev_step_gn;  // ptl_icp_ekf_step's own pair (created on first use)
    // ev_gn is recorded right after each GN launch: the map update (and the sequence runner's EKF step) start from it.
    // frame_down is double-buffered so that K3 of scan k+1 does not overwrite what K7 of scan k still reads.
    // the map update (K7-K10, rebuild) runs on `map_stream`: it needs the GN result (ev_gn) and must be complete before
    // the next GN launch (ev_map); beside it, on `stream`, run K0-K4 of the next scan
    hipStream_t map_stream;
    hipEvent_t ev_map;
    bool ev_map_valid;
    hipEvent_t ev_gn;
    bool ev_gn_valid;
    int* n_src_hint;  // pinned host copy of the last scan's N_s, written by an asynchronous copy on the map stream
    double* fd_buf[2];
    double gn_ms;
    int64_t gn_launches;
};

extern "C" int ptl_icp_default_cfg(ptl_icp_cfg* cfg, double max_range, double min_range) {
    if (!cfg) return set_err(PTL_ERR_ARG, "cfg is null");
    ABI_CHECK(ptl_icp_cfg, cfg);
    PTL_CFG_INIT(cfg);
    cfg->max_range = max_range;
    cfg->min_range = min_range;
    cfg->voxel_size = max_range / 100.0;  // kiss-icp load_config
    cfg->max_points_per_voxel = 20;
    cfg->initial_threshold = 2.0;
    cfg->min_motion_th = 0.1;
    cfg->deskew = 1;
    cfg->max_iterations = 500;
    cfg->convergence = 1e-4;
    cfg->device_id = 0;
    cfg->scan_cols = 1024;
    cfg->max_points_per_scan = 131072;
    cfg->map_block_capacity = 1 << 19;
    // The table is sparse on purpose: a 2 x 2 x 2 brick of voxels is one 128-byte line, and two bricks that hash to the same line push each other's
    // entries into dependent probe rounds (probe-row rebuilds, map insert).  ~30 k occupied bricks in 2 M lines hardly ever meet; in the 262 k lines of
    // round 3's 2^21 slots they did: Gauss-Newton 2 357 -> 2 196 us per scan, + 7 % scans/s (profiles/r04_y_map_table_and_rebuild.txt).  256 MB per
    // sequence - what 288 GB per GPU are for.  The rebuild (tombstones dropped, table cleared by the team) is correspondingly rarer.
    cfg->map_table_capacity = 1 << 24;
    cfg->gn_workgroups = 256;
    cfg->rebuild_every = 128;
    cfg->gn_threads = 1024;
    cfg->gn_lanes_per_point = 32;
    cfg->map_small_blocks = 0;
    return PTL_OK;
}

template <typename T>
static hipError_t dalloc(T** p, size_t n) { return hipMalloc((void**)p, n * sizeof(T)); }

static int icp_free(ptl_icp* h) {
    if (!h) return PTL_OK;
    (void)hipSetDevice(h->cfg.device_id);
    Ctx& c = h->c;
    void* ptrs[] = {c.pts, c.slot1, c.slot2, c.vtab1, c.vtab2, c.bcnt1, c.bcnt2, h->fd_buf[0], h->fd_buf[1], c.src0,
                    c.src_cur, c.fdw, c.coltab, c.pslot, c.nxt, c.prank, c.plen, c.tab, c.blocks, c.bhdr, c.bfirst, c.free_stack, c.free_stack_s, c.mig_list, c.wg_clk, c.pc_key, c.pc_pb, c.pc_ans, c.gn_rows_ll, c.gn_xsum_ll,
                    c.st, c.traj, c.sstats, h->d_in, h->d_t01, h->d_ext, h->d_counter, h->d_row_mask};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : {h->ev_gn, h->ev_map, h->ev_step_guess, h->ev_step_gn})
        if (e) (void)hipEventDestroy(e);
    if (h->n_src_hint) (void)hipHostFree(h->n_src_hint);
    if (h->map_stream) (void)hipStreamDestroy(h->map_stream);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PTL_OK;
}

__global__ void k_fill_free_stack(int* fs, int n, int base) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) fs[i] = base + n - 1 - i;  // pops hand out low block ids first
}
__global__ void k_state_init(DevState* st, int n_full, int n_small) {
    if (threadIdx.x || blockIdx.x) return;
    memset(st, 0, sizeof(DevState));
    Rt I = rt_identity();
    rt_to16(I, st->pose_first); rt_to16(I, st->pose_prev); rt_to16(I, st->pose_last);
    rt_to16(I, st->model_dev); rt_to16(I, st->guess); rt_to16(I, st->new_pose); rt_to16(I, st->T_icp);
    st->free_top = n_full;
    st->free_top_s = n_small;
    st->pool_hw = n_small;  // (an id bound: the full blocks' ids start behind the small ones)
    st->stats_pending = -1;
    st->dbg_dead_wg = -1;
}

// slots of a per-scan voxel table (power of two): make EXTRA="-DVDS1_SLOTS_PER_POINT=n -DVDS2_SLOTS_PER_POINT=m" for A/B runs
#ifndef VDS1_SLOTS_PER_POINT
#define VDS1_SLOTS_PER_POINT 64
#endif
#ifndef VDS2_SLOTS_PER_POINT
#define VDS2_SLOTS_PER_POINT 16
#endif
static size_t vds_table_slots(size_t per_point, size_t n) {
    size_t vcap = 1024;
    while (vcap < per_point * n) vcap <<= 1;
    return vcap;
}
static int icp_reset_device(ptl_icp* h) {
    Ctx& c = h->c;
    const size_t vcap = (size_t)c.vmask + 1, vcap2 = (size_t)c.vmask2 + 1;
    HIPCHK(hipMemsetAsync(c.vtab1, 0xFF, vcap * sizeof(VdsEnt), h->stream));  // (key = EMPTY, index = none)
    HIPCHK(hipMemsetAsync(c.vtab2, 0xFF, vcap2 * sizeof(VdsEnt), h->stream));
    HIPCHK(hipMemsetAsync(c.tab, 0xFF, ((size_t)c.tmask + 1) * sizeof(TabEnt), h->stream));
    HIPCHK(hipMemsetAsync(c.blocks, 0, (size_t)c.n_small * SMALL_BYTES + (size_t)(c.pool_cap - c.n_small) * c.bstride, h->stream));
    HIPCHK(hipMemsetAsync(c.bhdr, 0, (size_t)c.pool_cap * 16, h->stream));
    HIPCHK(hipMemsetAsync(c.bfirst, 0, (size_t)c.pool_cap * 24, h->stream));
    HIPCHK(hipMemsetAsync(c.bcnt1, 0, (size_t)h->nblk_scan * sizeof(int), h->stream));  // look-back words of the compactions: tag 0 = never published
    HIPCHK(hipMemsetAsync(c.bcnt2, 0, (size_t)h->nblk_scan * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(c.gn_rows_ll, 0, (size_t)2 * c.G * 64 * 8, h->stream));  // the launch epoch restarts with the state
    HIPCHK(hipMemsetAsync(c.gn_xsum_ll, 0, (size_t)2 * 8 * 8 * 64 * 8, h->stream));
    k_fill_free_stack<<<(c.pool_cap - c.n_small + 255) / 256, 256, 0, h->stream>>>(c.free_stack, c.pool_cap - c.n_small, c.n_small);
    if (c.n_small > 0) k_fill_free_stack<<<(c.n_small + 255) / 256, 256, 0, h->stream>>>(c.free_stack_s, c.n_small, 0);
    k_state_init<<<1, 64, 0, h->stream>>>(c.st, c.pool_cap - c.n_small, c.n_small);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->map_stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // the other streams must not start on half-reset tables
    h->ev_gn_valid = false; h->ev_map_valid = false;
    h->scans_done = 0;
    h->last_n = 0;
    return PTL_OK;
}

static int icp_create_impl(const ptl_icp_cfg* cfg, hipStream_t shared_stream, ptl_icp** out, bool batch_member = false) {
    if (!cfg || !out) return set_err(PTL_ERR_ARG, "null argument");
    ABI_CHECK(ptl_icp_cfg, cfg);
    // the search maps one lane of a 32-lane group to one stored point of a voxel: more than 32 per voxel would be stored and never examined
    if (cfg->max_points_per_voxel < 1 || cfg->max_points_per_voxel > 32) return set_err(PTL_ERR_ARG, "max_points_per_voxel must be in [1, 32]");
    if (cfg->map_block_capacity < 1 || cfg->map_small_blocks < 0 || cfg->map_block_capacity + cfg->map_small_blocks >= (1 << 24) - 1)
        return set_err(PTL_ERR_ARG, "map_block_capacity + map_small_blocks must be below 2^24 - 1");
    if (cfg->map_small_blocks > 0 && !batch_member)
        return set_err(PTL_ERR_ARG, "map_small_blocks (two block classes) is served by the free-running batch driver only: a voxel's move to a full block is part of its map update");
    if (cfg->map_small_blocks > 0 && cfg->max_points_per_voxel < SMALL_CAP) return set_err(PTL_ERR_ARG, "map_small_blocks needs max_points_per_voxel >= %d", SMALL_CAP);
    if (cfg->map_table_capacity & (cfg->map_table_capacity - 1)) return set_err(PTL_ERR_ARG, "map_table_capacity must be a power of two");
    if (cfg->max_points_per_scan < 1 || cfg->gn_workgroups < 1 || cfg->gn_workgroups > 512) return set_err(PTL_ERR_ARG, "bad capacity (gn_workgroups must be in [1, 512])");
    if (cfg->max_iterations < 1 || cfg->max_iterations > 1000) return set_err(PTL_ERR_ARG, "max_iterations must be in [1, 1000]");
    if (cfg->gn_threads < 256 || cfg->gn_threads > GN_MAX_THREADS || (cfg->gn_threads & 63)) return set_err(PTL_ERR_ARG, "gn_threads must be a multiple of 64 in [256, GN_MAX_THREADS]");
    if (cfg->gn_lanes_per_point != 32 && cfg->gn_lanes_per_point != 8) return set_err(PTL_ERR_ARG, "gn_lanes_per_point must be 32 or 8");
    if (cfg->gn_lanes_per_point == 8 && cfg->gn_threads > GN8_MAX_THREADS) return set_err(PTL_ERR_ARG, "gn_lanes_per_point = 8 runs at most %d threads per workgroup", GN8_MAX_THREADS);
    if (ptl_device_count() <= cfg->device_id) return set_err(PTL_ERR_HIP, "no HIP device %d (the HIP backend is the only backend)", cfg->device_id);
    HIPCHK(hipSetDevice(cfg->device_id));
    {
        // The Gauss-Newton kernel is persistent: its workgroups exchange sums every iteration, so ALL of them have to be
        // resident at once.  Check that against the occupancy the runtime reports for this kernel on this device
        // (partitioned GPU, fewer CUs, larger gn_threads) instead of finding out through a poll time-out.
        int per_cu = 0, cus = 0;
        const int P20 = cfg->max_points_per_voxel == 20;
        hipError_t e;
        if (cfg->gn_lanes_per_point == 8)
            e = P20 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gn_loop8<20, 0>, cfg->gn_threads, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gn_loop8<0, 0>, cfg->gn_threads, 0);  // (the instance with the largest footprint)
        else
            e = P20 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gn_loop<20, true>, cfg->gn_threads, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gn_loop<0, true>, cfg->gn_threads, 0);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id);
        if (e != hipSuccess) return set_err(PTL_ERR_HIP, "occupancy query failed: %s", hipGetErrorString(e));
        if ((int64_t)per_cu * cus < cfg->gn_workgroups)
            return set_err(PTL_ERR_STATE, "gn_workgroups = %d x %d threads cannot be co-resident on device %d (%d CUs x %d workgroups per CU): "
                           "lower gn_workgroups", cfg->gn_workgroups, cfg->gn_threads, cfg->device_id, cus, per_cu);
    }
    ptl_icp* h = new ptl_icp();
    h->cfg = *cfg;
    h->own_stream = shared_stream == nullptr;
    h->stream = shared_stream;
    h->prof = false; h->prof_every = 1; h->ev_used = 0; h->gn_ms = 0; h->gn_launches = 0;
    h->gn_wait = nullptr; h->gn_done = nullptr; h->ev_step_guess = nullptr; h->ev_step_gn = nullptr;
    h->lazy_map_stats = false;
    h->n_src_hint = nullptr;
    h->map_stream = nullptr; h->ev_map = nullptr; h->ev_map_valid = false;
    h->ev_gn = nullptr;
    h->ev_gn_valid = false; h->fd_buf[0] = h->fd_buf[1] = nullptr;
    h->d_in = nullptr; h->d_t01 = nullptr; h->d_ext = nullptr; h->d_counter = nullptr; h->d_row_mask = nullptr;
    memset(&h->c, 0, sizeof(Ctx));
    if (h->own_stream && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return set_err(PTL_ERR_HIP, "hipStreamCreate failed");
    }
    Ctx& c = h->c;
    const int64_t n = cfg->max_points_per_scan;
    h->n_max = n;
    c.max_range = cfg->max_range; c.min_range = cfg->min_range; c.vs = cfg->voxel_size;
    c.vds1 = cfg->voxel_size * 0.5;  // KissICP.voxelize
    c.vds2 = cfg->voxel_size * 1.5;
    c.init_thr = cfg->initial_threshold; c.min_motion = cfg->min_motion_th; c.conv = cfg->convergence;
    c.P = cfg->max_points_per_voxel; c.deskew = cfg->deskew; c.max_iter = cfg->max_iterations;
    c.W = cfg->scan_cols > 0 ? cfg->scan_cols : 1;
    c.n_max = (int)n;
    // per-scan voxel tables.  A table line holds a 2x2x2 brick of voxels (brick_slot), so what counts is how many LINES are taken:
    // a voxel of a brick that lands on a taken line walks on through dependent reads, and one such voxel among the 512 claims of a
    // wavefront's round costs the wavefront another round trip.  The lines a scan touches do not depend on the table's size, the
    // slots are released by the compaction that follows (no clear per scan): size costs footprint only.  Pass 1 (~46 k voxels of
    // ~120 k points): 64 slots per point (K0-K4 516 -> 485 us per scan against 8; 32: 493); pass 2 (~14 k voxels): 16.
    const size_t vcap = vds_table_slots(VDS1_SLOTS_PER_POINT, n), vcap2 = vds_table_slots(VDS2_SLOTS_PER_POINT, n);
    c.vmask = (unsigned)(vcap - 1);
    c.vmask2 = (unsigned)(vcap2 - 1);
    c.tmask = (unsigned)(cfg->map_table_capacity - 1);
    c.bstride = (int)(((size_t)c.P * 24 + 127) / 128 * 128);  // points only: the per-voxel header lives in the block directory (bhdr / bfirst)
    c.n_small = (int)cfg->map_small_blocks;
    c.pool_cap = (int)(cfg->map_block_capacity + cfg->map_small_blocks);  // block ids: the small blocks first, then map_block_capacity full ones
    c.G = cfg->gn_workgroups;
    h->nblk_scan = (int)((n + 255) / 256);
    h->traj_cap = 4096;
    c.traj_cap = (int)h->traj_cap;
    bool ok = true;
    ok &= dalloc(&c.pts, 3 * n) == hipSuccess;
    ok &= dalloc(&c.slot1, n) == hipSuccess && dalloc(&c.slot2, n) == hipSuccess;
    ok &= dalloc(&c.vtab1, vcap) == hipSuccess && dalloc(&c.vtab2, vcap2) == hipSuccess;
    ok &= dalloc(&c.bcnt1, h->nblk_scan) == hipSuccess && dalloc(&c.bcnt2, h->nblk_scan) == hipSuccess;
    ok &= dalloc(&c.fd, 3 * n) == hipSuccess && dalloc(&c.src0, 3 * n) == hipSuccess;
    h->fd_buf[0] = c.fd;
    ok &= dalloc(&h->fd_buf[1], 3 * n) == hipSuccess;
    ok &= hipStreamCreateWithFlags(&h->map_stream, hipStreamNonBlocking) == hipSuccess;
    ok &= hipEventCreateWithFlags(&h->ev_map, hipEventDisableTiming) == hipSuccess;
    ok &= hipEventCreateWithFlags(&h->ev_gn, hipEventDisableTiming) == hipSuccess;
    c.overlap_pre = 1;
    ok &= dalloc(&c.src_cur, 3 * n) == hipSuccess && dalloc(&c.fdw, 3 * n) == hipSuccess;
    ok &= dalloc(&c.coltab, (size_t)c.W * 12) == hipSuccess;
    ok &= dalloc(&c.pslot, n) == hipSuccess && dalloc(&c.nxt, n) == hipSuccess;
    ok &= dalloc(&c.prank, n) == hipSuccess && dalloc(&c.plen, n) == hipSuccess;
    ok &= dalloc(&c.tab, (size_t)cfg->map_table_capacity) == hipSuccess;
    ok &= hipMalloc((void**)&c.blocks, (size_t)c.n_small * SMALL_BYTES + (size_t)(c.pool_cap - c.n_small) * c.bstride) == hipSuccess;
    c.big_base = (unsigned char*)((uintptr_t)c.blocks + (uintptr_t)c.n_small * SMALL_BYTES - (uintptr_t)c.n_small * (uintptr_t)c.bstride);
    ok &= dalloc(&c.bhdr, (size_t)c.pool_cap * 4) == hipSuccess && dalloc(&c.bfirst, (size_t)c.pool_cap * 3) == hipSuccess;
    ok &= dalloc(&c.free_stack, c.pool_cap - c.n_small) == hipSuccess;
    ok &= dalloc(&c.free_stack_s, (size_t)(c.n_small > 0 ? c.n_small : 1)) == hipSuccess && dalloc(&c.mig_list, (size_t)(c.n_small > 0 ? c.n_small : 1)) == hipSuccess;
    ok &= dalloc(&c.pc_key, n) == hipSuccess && dalloc(&c.pc_pb, 32 * n) == hipSuccess && dalloc(&c.pc_ans, GN8_ANS_ROW * n) == hipSuccess;
    ok &= hipHostMalloc((void**)&h->n_src_hint, sizeof(int)) == hipSuccess;
    if (h->n_src_hint) *h->n_src_hint = 0;
    ok &= dalloc(&c.gn_rows_ll, (size_t)2 * c.G * 64) == hipSuccess && hipMemset(c.gn_rows_ll, 0, (size_t)2 * c.G * 64 * 8) == hipSuccess;
    ok &= dalloc(&c.gn_xsum_ll, (size_t)2 * 8 * 8 * 64) == hipSuccess && hipMemset(c.gn_xsum_ll, 0, (size_t)2 * 8 * 8 * 64 * 8) == hipSuccess;
    ok &= hipMalloc((void**)&c.wg_clk, (size_t)c.G * 16) == hipSuccess && hipMemset(c.wg_clk, 0, (size_t)c.G * 16) == hipSuccess;
    ok &= dalloc(&c.st, 1) == hipSuccess;
    ok &= dalloc(&c.traj, (size_t)h->traj_cap * 16) == hipSuccess;
    ok &= dalloc(&c.sstats, (size_t)h->traj_cap) == hipSuccess;
    ok &= hipMalloc(&h->d_in, (size_t)n * 3 * 8) == hipSuccess;
    ok &= dalloc(&h->d_t01, n) == hipSuccess;
    ok &= dalloc(&h->d_ext, 16) == hipSuccess;
    ok &= dalloc(&h->d_counter, 4) == hipSuccess;
    if (!ok) {
        icp_free(h);
        return set_err(PTL_ERR_HIP, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));
    }
    int rc = icp_reset_device(h);
    if (rc != PTL_OK) { icp_free(h); return rc; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { icp_free(h); return set_err(PTL_ERR_HIP, "init sync failed"); }
    *out = h;
    return PTL_OK;
}
// device bytes one registration handle allocates (icp_create_impl above, term by term): per point of max_points_per_scan the work
// buffers (deskewed points, slots, the two per-scan voxel tables, frame_down x 2, source x 2, world frame_down, list links,
// probe rows, answer rows, the per-call input), plus the map table, the block pool and its free stack
static size_t icp_footprint_bytes(const ptl_icp_cfg* cfg) {
    const size_t n = (size_t)cfg->max_points_per_scan;
    const size_t bstride = ((size_t)cfg->max_points_per_voxel * 24 + 127) / 128 * 128;
    const size_t per_point = 24 + 4 + 4 + 24 * 2 + 24 + 24 + 24 + 4 * 4 + 8 + 128 + GN8_ANS_ROW * 8 + 24 + 8;
    return n * per_point + (vds_table_slots(VDS1_SLOTS_PER_POINT, n) + vds_table_slots(VDS2_SLOTS_PER_POINT, n)) * sizeof(VdsEnt) + (size_t)cfg->map_table_capacity * sizeof(TabEnt) +
           (size_t)cfg->map_block_capacity * (bstride + 4 + 16 + 24) + (size_t)cfg->map_small_blocks * (SMALL_BYTES + 8 + 16 + 24) + (size_t)4096 * (128 + sizeof(ScanStats)) + (1u << 20);
}
extern "C" int ptl_icp_create(const ptl_icp_cfg* cfg, ptl_icp** out) { return icp_create_impl(cfg, nullptr, out); }
extern "C" int ptl_icp_destroy(ptl_icp* h) { return icp_free(h); }

static int icp_grow_traj(ptl_icp* h) {
    // doubles the trajectory / stats buffers (host-side decision: scans_done is known on the host)
    Ctx& c = h->c;
    const int64_t ncap = h->traj_cap * 2;
    double* nt = nullptr;
    ScanStats* ns = nullptr;
    HIPCHK(dalloc(&nt, (size_t)ncap * 16));
    HIPCHK(dalloc(&ns, (size_t)ncap));
    HIPCHK(hipMemcpyAsync(nt, c.traj, (size_t)h->traj_cap * 16 * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(ns, c.sstats, (size_t)h->traj_cap * sizeof(ScanStats), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->map_stream));
    HIPCHK(hipDeviceSynchronize());  // (a sequence runner's EKF stream may still read the old trajectory rows)
    (void)hipFree(c.traj); (void)hipFree(c.sstats);
    c.traj = nt; c.sstats = ns;
    h->traj_cap = ncap; c.traj_cap = (int)ncap;
    return PTL_OK;
}

static int map_rebuild(ptl_icp* h, hipStream_t s) {
    Ctx& c = h->c;
    HIPCHK(hipMemsetAsync(c.tab, 0xFF, ((size_t)c.tmask + 1) * sizeof(TabEnt), s));
    HIPCHK(hipMemsetAsync(&c.st->tab_used, 0, sizeof(unsigned), s));
    k_map_rebuild<<<(c.pool_cap + 255) / 256, 256, 0, s>>>(c);
    return PTL_OK;
}

// Enqueue one whole scan on the handle's stream (no host synchronisation):
// in_f32 / in_f64 / t01 / ext_guess are DEVICE pointers (one of in_* non-null).
struct ptl_lut {
    int device_id, H, W;
    double *dir, *off;
};
static int icp_enqueue_scan(ptl_icp* h, const float* in_f32, const double* in_f64, const double* t01, int64_t n,
                            const double* ext_guess, const unsigned* in_range = nullptr, const ptl_lut* lut = nullptr) {
    if (n > h->n_max) return set_err(PTL_ERR_CAPACITY, "scan has %lld points, capacity %lld", (long long)n, (long long)h->n_max);
    if (h->scans_done >= h->traj_cap) { int rc = icp_grow_traj(h); if (rc) return rc; }
    Ctx c = h->c;
    c.in_f32 = in_f32; c.in_f64 = in_f64; c.t01 = t01; c.n_in = (int)n; c.ext_guess = ext_guess;
    c.in_range = in_range;
    if (in_range) {
        if (!lut || (int64_t)lut->H * lut->W != n || lut->W != c.W) return set_err(PTL_ERR_ARG, "range image needs a LUT of the same H x W (W = cfg.scan_cols)");
        c.lut_dir = lut->dir; c.lut_off = lut->off;
        c.row_mask = h->d_row_mask;
    }
    const int nb = (int)((n + 255) / 256) > 0 ? (int)((n + 255) / 256) : 1;
    hipStream_t s = h->stream;
    const int nb1 = (int)(((n > h->last_n ? n : h->last_n) + 255) / 256) > 0 ? (int)(((n > h->last_n ? n : h->last_n) + 255) / 256) : 1;
    // frame_down of this scan goes to the buffer the previous scan's map update is not reading
    c.fd = h->fd_buf[h->scans_done & 1];
    h->c.fd = c.fd;
    // K0-K4 on the main stream, right behind the previous GN kernel (no hand-over: a stream that has to wait for another
    // pays ~20 us of wake-up latency); beside them run the previous scan's map update (map stream) and, in the sequence
    // runner, the EKF step (its own stream) - both are done before this chain is, so the waits below do not block
    k_scan_prologue<<<1, 1024, 0, s>>>(c);
    k_deskew_vds1<<<nb1, 256, 0, s>>>(c);
    k_count_w1<<<nb, 256, 0, s>>>(c);
    k_compact_fd<<<nb, 256, 0, s>>>(c);
    k_vds2_fd<<<nb, 256, 0, s>>>(c);  // (over the compact frame_downsample: blocks beyond N_d find nothing to do)
    k_count_w2<<<nb, 256, 0, s>>>(c);
    k_compact_src<<<nb, 256, 0, s>>>(c);
    if (h->ev_map_valid) HIPCHK(hipStreamWaitEvent(s, h->ev_map, 0));
    if (h->gn_wait) HIPCHK(hipStreamWaitEvent(s, h->gn_wait, 0));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = h->prof && (h->scans_done % h->prof_every) == 0;
    if (timed) {
        if (h->ev_used + 2 > h->ev.size()) {
            for (int k = 0; k < 2; ++k) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
        }
        e0 = h->ev[h->ev_used++]; e1 = h->ev[h->ev_used++];
        HIPCHK(hipEventRecord(e0, s));
    }
    // ends with the post-ICP bookkeeping (kiss.py:116-128).  Dense scans (more source points than 32-lane groups) run the
    // variant that keeps probe results in memory; the previous scan's N_s, copied back without a wait, is the hint
    const bool dense = *h->n_src_hint > (int64_t)c.G * (h->cfg.gn_threads / 32);
    if (h->cfg.gn_lanes_per_point == 8) {
#define KG8(GC) do { if (c.P == 20) k_gn_loop8<20, GC><<<c.G, h->cfg.gn_threads, 0, s>>>(c, 0); else k_gn_loop8<0, GC><<<c.G, h->cfg.gn_threads, 0, s>>>(c, 0); } while (0)
        if (c.G == 32) KG8(32); else if (c.G == 16) KG8(16); else if (c.G == 8) KG8(8); else KG8(0);
#undef KG8
    }
    else if (c.P == 20) { if (dense) k_gn_loop<20, true><<<c.G, h->cfg.gn_threads, 0, s>>>(c, 0); else k_gn_loop<20, false><<<c.G, h->cfg.gn_threads, 0, s>>>(c, 0); }
    else { if (dense) k_gn_loop<0, true><<<c.G, h->cfg.gn_threads, 0, s>>>(c, 0); else k_gn_loop<0, false><<<c.G, h->cfg.gn_threads, 0, s>>>(c, 0); }
    if (timed) HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventRecord(h->ev_gn, s));
    h->ev_gn_valid = true;
    if (h->gn_done) HIPCHK(hipEventRecord(h->gn_done, s));
    // local_map.update(frame_downsample, new_pose)  (kiss.py:129), on the map stream
    hipStream_t sm = h->map_stream;
    HIPCHK(hipStreamWaitEvent(sm, h->ev_gn, 0));
    k_map_insert_a<<<nb, 256, 0, sm>>>(c, c.fd, &c.st->n_down_ins, 0, 1);
    k_map_insert_b<<<nb, 256, 0, sm>>>(c, &c.st->n_down_ins, 0);
    k_map_insert_c<<<nb, 256, 0, sm>>>(c, &c.st->n_down_ins, 0);
    k_map_prune<<<(c.pool_cap + 255) / 256, 256, 0, sm>>>(c, nullptr, 1);
    h->last_n = n;
    h->scans_done++;
    if (h->cfg.rebuild_every > 0 && (h->scans_done % h->cfg.rebuild_every) == 0) { int rc = map_rebuild(h, sm); if (rc) return rc; }
    HIPCHK(hipMemcpyAsync(h->n_src_hint, &c.st->n_src_last, sizeof(int), hipMemcpyDeviceToHost, sm));  // hint for the next launches
    HIPCHK(hipEventRecord(h->ev_map, sm));
    h->ev_map_valid = true;
    HIPCHK(hipGetLastError());
    return PTL_OK;
}

static void icp_collect_profile(ptl_icp* h) {
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) == hipSuccess) { h->gn_ms += ms; h->gn_launches++; }
    }
    h->ev_used = 0;
}

// Every entry point that touches the map, the device state or the handle's staging buffers (d_in, d_ext, fdw) on h->stream goes through
// this first: with lazy_map_stats a registration returns while its scan's map update (K7-K10, the rebuild) still runs on map_stream, and
// the sequence runner leaves one in flight too - work enqueued on h->stream must be ordered behind it (map_add / align /
// linear_system used to start beside it: two inserts on one table and pool, or a search over a half-inserted batch).
static int icp_join_map(ptl_icp* h) {
    if (h->ev_map_valid) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_map, 0));
    return PTL_OK;
}
static int icp_check_flags(ptl_icp* h) {
    int flags = 0;
    HIPCHK(hipStreamSynchronize(h->map_stream));  // the last scan's map update
    k_finish_scan<<<1, 64, 0, h->stream>>>(h->c);  // closes the last scan's bookkeeping (no-op when none is pending)
    HIPCHK(hipMemcpyAsync(&flags, &h->c.st->err_flags, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->prof) icp_collect_profile(h);
    if (flags)
        return set_err(PTL_ERR_CAPACITY, "device capacity/error flags 0x%x (1 key range, 2 block pool, 4 map table, 8 vds table, 16 gn barrier timeout)", flags);
    return PTL_OK;
}

__global__ void k_finish_pose(Ctx c) {  // KissICP.poses.append of the scan just registered, WITHOUT the map statistics (its map update may still be running)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    finish_pending(c, c.st);
}
static void stats_out(const ScanStats& s, ptl_icp_stats* o);
// what follows the enqueue of a per-call registration: close the scan, bring pose and statistics row back, check the error flags.
// Default: after the scan's map update (map size is part of the row).  lazy_map_stats: as soon as the Gauss-Newton kernel is done -
// the map update (K7-K10, ~80 us) then runs beside the caller's next steps and the next call's upload and K0-K4; the row's map_voxels /
// map_points are -1 (ptl_icp_map_size gives them on demand) and an error flag raised by that update is reported by the next call.
static int icp_percall_finish(ptl_icp* h, double out_pose[16], ptl_icp_stats* stats) {
    const bool lazy = h->lazy_map_stats;
    if (!lazy && h->ev_map_valid) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_map, 0));  // map size is part of the row
    if (lazy) k_finish_pose<<<1, 64, 0, h->stream>>>(h->c);
    else k_finish_scan<<<1, 64, 0, h->stream>>>(h->c);  // close the scan before its stats row is read back
    const int64_t k = h->scans_done - 1;
    double pose[16];
    ScanStats ss;
    int flags = 0;
    HIPCHK(hipMemcpyAsync(pose, h->c.traj + 16 * k, sizeof pose, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&ss, h->c.sstats + k, sizeof ss, hipMemcpyDeviceToHost, h->stream));
    int rc = PTL_OK;
    if (lazy) {
        HIPCHK(hipMemcpyAsync(&flags, &h->c.st->err_flags, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->prof) icp_collect_profile(h);
        if (flags) rc = set_err(PTL_ERR_CAPACITY, "device capacity/error flags 0x%x (1 key range, 2 block pool, 4 map table, 8 vds table, 16 gn barrier timeout)", flags);
        ss.map_voxels = -1; ss.map_points = -1;
    } else {
        rc = icp_check_flags(h);
    }
    if (rc) return rc;
    if (out_pose) memcpy(out_pose, pose, sizeof pose);
    if (stats) stats_out(ss, stats);
    return PTL_OK;
}
extern "C" int ptl_icp_set_lazy_map_stats(ptl_icp* h, int32_t on) {
    if (!h) return set_err(PTL_ERR_ARG, "null argument");
    h->lazy_map_stats = on != 0;
    return PTL_OK;
}
static void stats_out(const ScanStats& s, ptl_icp_stats* o) {
    o->sigma = s.sigma; o->err_dt = s.err_dt; o->err_drot = s.err_drot;
    o->iterations = s.iterations; o->n_corr_last = s.n_corr_last; o->n_in = s.n_in; o->n_valid = s.n_valid;
    o->n_down = s.n_down; o->n_src = s.n_src; o->sum_cand = s.sum_cand; o->map_voxels = s.map_voxels;
    o->map_points = s.map_points;
}

extern "C" int ptl_icp_register_frame(ptl_icp* h, const void* xyz, int dtype, int64_t n, const double* t01,
                                      double scan_ts, const double* guess, double out_pose[16], ptl_icp_stats* stats) {
    (void)scan_ts;
    if (!h || (!xyz && n > 0) || n < 0) return set_err(PTL_ERR_ARG, "bad argument");
    if (dtype != PTL_F32 && dtype != PTL_F64) return set_err(PTL_ERR_ARG, "dtype must be PTL_F32 or PTL_F64");
    if (n > h->n_max) return set_err(PTL_ERR_CAPACITY, "scan has %lld points, capacity %lld", (long long)n, (long long)h->n_max);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const size_t esz = dtype == PTL_F32 ? 4 : 8;
    if (n) HIPCHK(hipMemcpyAsync(h->d_in, xyz, (size_t)n * 3 * esz, hipMemcpyHostToDevice, h->stream));
    if (t01 && n) HIPCHK(hipMemcpyAsync(h->d_t01, t01, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    if (guess) HIPCHK(hipMemcpyAsync(h->d_ext, guess, 16 * 8, hipMemcpyHostToDevice, h->stream));
    int rc = icp_enqueue_scan(h, dtype == PTL_F32 ? (const float*)h->d_in : nullptr,
                              dtype == PTL_F64 ? (const double*)h->d_in : nullptr, t01 ? h->d_t01 : nullptr, n,
                              guess ? h->d_ext : nullptr);
    if (rc) return rc;
    return icp_percall_finish(h, out_pose, stats);
}

extern "C" int ptl_icp_num_poses(ptl_icp* h, int64_t* n) {
    if (!h || !n) return set_err(PTL_ERR_ARG, "null argument");
    *n = h->scans_done;
    return PTL_OK;
}
extern "C" int ptl_icp_get_poses(ptl_icp* h, double* out, int64_t max_poses, int64_t* n_written) {
    if (!h || !out) return set_err(PTL_ERR_ARG, "null argument");
    const int64_t n = h->scans_done < max_poses ? h->scans_done : max_poses;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    if (n > 0) HIPCHK(hipMemcpyAsync(out, h->c.traj, (size_t)n * 16 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n_written) *n_written = n;
    return PTL_OK;
}
__global__ void k_prediction(const DevState* st, double* out) {
    if (threadIdx.x || blockIdx.x) return;
    Rt p = (st->n_poses >= 2) ? rt_mul(rt_inv(rt_from16(st->pose_prev)), rt_from16(st->pose_last)) : rt_identity();
    rt_to16(p, out);
}
extern "C" int ptl_icp_get_prediction(ptl_icp* h, double out[16]) {
    if (!h || !out) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    { int rcj = icp_join_map(h); if (rcj) return rcj; }
    k_prediction<<<1, 64, 0, h->stream>>>(h->c.st, h->d_ext);
    HIPCHK(hipMemcpyAsync(out, h->d_ext, 16 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}
extern "C" int ptl_icp_map_size(ptl_icp* h, int64_t* voxels, int64_t* points) {
    if (!h) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    DevState st;
    HIPCHK(hipStreamSynchronize(h->map_stream));  // (a map update may still be under way: lazy_map_stats, the sequence runner)
    HIPCHK(hipMemcpyAsync(&st, h->c.st, sizeof st, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (voxels) *voxels = st.n_live;
    if (points) *points = st.map_points;
    return PTL_OK;
}
extern "C" int ptl_icp_map_points(ptl_icp* h, double* xyz_out, int64_t max_points, int64_t* n_written) {
    if (!h || !xyz_out) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    double* d_out = nullptr;
    HIPCHK(hipStreamSynchronize(h->map_stream));  // (a map update may still be under way)
    HIPCHK(dalloc(&d_out, (size_t)(max_points > 0 ? max_points : 1) * 3));
    HIPCHK(hipMemsetAsync(h->d_counter, 0, sizeof(int), h->stream));
    k_map_export<<<(h->c.pool_cap + 255) / 256, 256, 0, h->stream>>>(h->c, d_out, h->d_counter, (int)max_points);
    int cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, h->d_counter, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int64_t n = cnt < max_points ? cnt : max_points;
    if (n > 0) HIPCHK(hipMemcpy(xyz_out, d_out, (size_t)n * 3 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_out);
    if (n_written) *n_written = n;
    return PTL_OK;
}
static int copy_cloud(ptl_icp* h, const double* d_src, const int* d_n, double* out, int64_t max_points, int64_t* n_written) {
    if (!h || !out) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    int n = 0;
    HIPCHK(hipMemcpyAsync(&n, d_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int64_t m = n < max_points ? n : max_points;
    if (m > 0) HIPCHK(hipMemcpy(out, d_src, (size_t)m * 3 * 8, hipMemcpyDeviceToHost));
    if (n_written) *n_written = m;
    return PTL_OK;
}
extern "C" int ptl_icp_last_frame_down(ptl_icp* h, double* out, int64_t max_points, int64_t* n_written) {
    return copy_cloud(h, h ? h->c.fd : nullptr, h ? &h->c.st->n_down_ins : nullptr, out, max_points, n_written);
}
extern "C" int ptl_icp_last_source(ptl_icp* h, double* out, int64_t max_points, int64_t* n_written) {
    return copy_cloud(h, h ? h->c.src0 : nullptr, h ? &h->c.st->n_src_last : nullptr, out, max_points, n_written);
}

extern "C" int ptl_icp_deskew(ptl_icp* h, const double* xyz, const double* t01, int64_t n, double* out) {
    if (!h || (!xyz && n > 0) || (!t01 && n > 0) || !out || n < 0) return set_err(PTL_ERR_ARG, "bad argument");
    if (n > h->n_max) return set_err(PTL_ERR_CAPACITY, "too many points");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    if (n == 0) return PTL_OK;
    { int rcj = icp_join_map(h); if (rcj) return rcj; }  // (the result is staged in fdw, which insert b of a running map update still reads)
    k_finish_scan<<<1, 64, 0, h->stream>>>(h->c);
    HIPCHK(hipMemcpyAsync(h->d_in, xyz, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_t01, t01, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    k_deskew_only<<<(int)((n + 255) / 256), 256, 0, h->stream>>>(h->c, (const double*)h->d_in, h->d_t01, (int)n, h->c.fdw);
    HIPCHK(hipMemcpyAsync(out, h->c.fdw, (size_t)n * 24, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}
extern "C" int ptl_icp_map_add(ptl_icp* h, const double* xyz_world, int64_t n, const double origin[3], int prune) {
    if (!h || (!xyz_world && n > 0)) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    Ctx& c = h->c;
    hipStream_t s = h->stream;
    { int rcj = icp_join_map(h); if (rcj) return rcj; }
    for (int64_t off = 0; off < n; off += h->n_max) {  // batches keep scan order: earlier batch = earlier points
        const int m = (int)((n - off) < h->n_max ? (n - off) : h->n_max);
        HIPCHK(hipMemcpyAsync(h->d_in, xyz_world + 3 * off, (size_t)m * 24, hipMemcpyHostToDevice, s));
        const int nb = (m + 255) / 256;
        k_map_insert_a<<<nb, 256, 0, s>>>(c, (const double*)h->d_in, nullptr, m, 0);
        k_map_insert_b<<<nb, 256, 0, s>>>(c, nullptr, m);
        k_map_insert_c<<<nb, 256, 0, s>>>(c, nullptr, m);
    }
    if (prune && origin) {
        HIPCHK(hipMemcpyAsync(h->d_ext, origin, 24, hipMemcpyHostToDevice, s));
        k_map_prune<<<(c.pool_cap + 255) / 256, 256, 0, s>>>(c, h->d_ext, 0);
    }
    return icp_check_flags(h);
}

__global__ void k_set_gn(Ctx c, int n_src, double max_dist, double kernel, const double* guess) {
    DevState* st = c.st;
    if (threadIdx.x == 0) {
        st->n_src = n_src; st->gn_max_dist = max_dist; st->gn_kernel = kernel;
        st->gn_epoch = (st->gn_epoch + 1u) & 0x3FFFFFu;
        st->gn_abort = 0;
        if (guess) for (int i = 0; i < 16; ++i) st->guess[i] = guess[i];
        __threadfence_block();
    }
    __syncthreads();
    gn_ll_clear_on_wrap(c);
}
extern "C" int ptl_icp_linear_system(ptl_icp* h, const double* src_world, int64_t n, double max_dist, double kernel,
                                     double sums[27], int64_t* n_corr, int64_t* n_cand) {
    if (!h || !src_world || !sums) return set_err(PTL_ERR_ARG, "null argument");
    if (n > h->n_max) return set_err(PTL_ERR_CAPACITY, "too many points");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    Ctx& c = h->c;
    { int rcj = icp_join_map(h); if (rcj) return rcj; }
    HIPCHK(hipMemcpyAsync(c.src_cur, src_world, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
    k_set_gn<<<1, 64, 0, h->stream>>>(c, (int)n, max_dist, kernel, nullptr);
    if (c.P == 20) k_gn_loop<20, false><<<c.G, h->cfg.gn_threads, 0, h->stream>>>(c, 1); else k_gn_loop<0, false><<<c.G, h->cfg.gn_threads, 0, h->stream>>>(c, 1);
    double out[32];
    HIPCHK(hipMemcpyAsync(out, (char*)c.st + offsetof(DevState, dbg_sums), sizeof out, hipMemcpyDeviceToHost, h->stream));
    int rc = icp_check_flags(h);
    if (rc) return rc;
    memcpy(sums, out, 27 * 8);
    if (n_corr) *n_corr = (int64_t)out[27];
    if (n_cand) *n_cand = (int64_t)out[28];
    return PTL_OK;
}
extern "C" int ptl_icp_align(ptl_icp* h, const double* frame, int64_t n, const double guess[16], double max_dist,
                             double kernel, double out_pose[16], int32_t* iterations) {
    if (!h || !frame || !guess || !out_pose) return set_err(PTL_ERR_ARG, "null argument");
    if (n > h->n_max) return set_err(PTL_ERR_CAPACITY, "too many points");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    Ctx& c = h->c;
    { int rcj = icp_join_map(h); if (rcj) return rcj; }
    HIPCHK(hipMemcpyAsync(c.src0, frame, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_ext, guess, 128, hipMemcpyHostToDevice, h->stream));
    k_set_gn<<<1, 64, 0, h->stream>>>(c, (int)n, max_dist, kernel, h->d_ext);
    if (c.P == 20) k_gn_loop<20, false><<<c.G, h->cfg.gn_threads, 0, h->stream>>>(c, 2); else k_gn_loop<0, false><<<c.G, h->cfg.gn_threads, 0, h->stream>>>(c, 2);  // new_pose = T_icp * guess, trajectory untouched
    DevState st;
    HIPCHK(hipMemcpyAsync(&st, c.st, sizeof st, hipMemcpyDeviceToHost, h->stream));
    int rc = icp_check_flags(h);
    if (rc) return rc;
    memcpy(out_pose, st.new_pose, 128);
    if (iterations) *iterations = st.gn_iters;
    return PTL_OK;
}

// test hook: set the launch epoch of the Gauss-Newton exchange (its 22-bit wrap-around is otherwise 4 million scans away)
__global__ void k_set_epoch(DevState* st, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) st->gn_epoch = v & 0x3FFFFFu; }
extern "C" int ptl_icp_debug_set_epoch(ptl_icp* h, uint32_t epoch) {
    if (!h) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    int rc = icp_check_flags(h);
    if (rc) return rc;
    k_set_epoch<<<1, 64, 0, h->stream>>>(h->c.st, epoch);
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}

// test hook: workgroup `wg` of the following Gauss-Newton launches returns at once (-1: back to normal).  The others run into
// the exchange time-out, raise the abort word and leave; the call that synchronises next reports the error flag.
__global__ void k_set_dead_wg(DevState* st, int wg) { if (threadIdx.x == 0 && blockIdx.x == 0) { st->dbg_dead_wg = wg; st->err_flags &= ~ERR_GN_TIMEOUT; } }
extern "C" int ptl_icp_debug_stall_workgroup(ptl_icp* h, int32_t wg) {
    if (!h) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->map_stream));
    k_set_dead_wg<<<1, 64, 0, h->stream>>>(h->c.st, wg);
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}

// test hook: leave the handle's block pool with at most `free_blocks` free blocks (< 0: unchanged) and / or declare `table_used`
// entries of its map table taken (< 0: unchanged) - the next map updates then run into ERR_POOL / ERR_TABLE, which the call
// that synchronises next reports as PTL_ERR_CAPACITY
__global__ void k_limit_capacity(DevState* st, int free_blocks, long long table_used) {
    if (threadIdx.x || blockIdx.x) return;
    if (free_blocks >= 0 && st->free_top > free_blocks) st->free_top = free_blocks;
    if (table_used >= 0) st->tab_used = (unsigned)table_used;
}
extern "C" int ptl_icp_debug_limit_capacity(ptl_icp* h, int32_t free_blocks, int64_t table_used) {
    if (!h) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->map_stream));
    k_limit_capacity<<<1, 64, 0, h->stream>>>(h->c.st, free_blocks, (long long)table_used);
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}

// diagnostic: ticks every GN workgroup spent in the search phase since creation: out[0..G) until its last wavefront
// finished, out[G..2G) its first wavefront
extern "C" int ptl_icp_gn_wg_clocks(ptl_icp* h, int64_t* out, int32_t max_wgs) {
    if (!h || !out || max_wgs < h->c.G) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(out, h->c.wg_clk, (size_t)h->c.G * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}

// diagnostic: the 32 debug sums of the handle's device state (phase-clock builds park sub-step clocks there)
extern "C" int ptl_icp_debug_sums(ptl_icp* h, double out[32]) {
    if (!h || !out) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(out, (char*)h->c.st + offsetof(DevState, dbg_sums), 256, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}
// diagnostic: accumulated per-phase clock ticks of workgroup 0 in the GN loop since the handle was reset
// out[0..4] = nn, wg-reduce+publish, barrier, grid-reduce, solve; out[5] = iterations
extern "C" int ptl_icp_gn_phases(ptl_icp* h, int64_t out[8]) {
    if (!h || !out) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(out, (char*)h->c.st + offsetof(DevState, gn_phase_clk), 64, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}

extern "C" int ptl_icp_profile(ptl_icp* h, int enable, double* gn_ms_total, int64_t* gn_launches, int reset) {
    if (!h) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    icp_collect_profile(h);
    if (gn_ms_total) *gn_ms_total = h->gn_ms;
    if (gn_launches) *gn_launches = h->gn_launches;
    if (reset) { h->gn_ms = 0; h->gn_launches = 0; }
    h->prof = enable != 0;
    h->prof_every = enable > 1 ? enable : 1;
    return PTL_OK;
}

// ------------------------------------------------------------------------------------------------ range-image input
extern "C" int ptl_lut_create(int device_id, int32_t H, int32_t W, const double* beam_altitude_deg,
                              const double* beam_azimuth_deg, double lidar_origin_to_beam_origin_mm,
                              const double* lidar_to_sensor16_mm, const double* extrinsic16_m, ptl_lut** out) {
    if (!beam_altitude_deg || !beam_azimuth_deg || !lidar_to_sensor16_mm || !out || H < 1 || W < 1) return set_err(PTL_ERR_ARG, "bad argument");
    if (ptl_device_count() <= device_id) return set_err(PTL_ERR_HIP, "no HIP device %d (the HIP backend is the only backend)", device_id);
    HIPCHK(hipSetDevice(device_id));
    // total transform in millimetres: extrinsic (metres -> mm) * lidar_to_sensor
    double T[16];
    memcpy(T, lidar_to_sensor16_mm, sizeof T);
    if (extrinsic16_m) {
        double E[16], R[16];
        memcpy(E, extrinsic16_m, sizeof E);
        E[3] *= 1e3; E[7] *= 1e3; E[11] *= 1e3;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += E[4 * i + k] * T[4 * k + j];
                R[4 * i + j] = s;
            }
        memcpy(T, R, sizeof T);
    }
    ptl_lut* l = new ptl_lut();
    l->device_id = device_id; l->H = H; l->W = W; l->dir = nullptr; l->off = nullptr;
    double* d_tmp = nullptr;
    const size_t n = (size_t)H * W;
    if (dalloc(&l->dir, 3 * n) != hipSuccess || dalloc(&l->off, 3 * n) != hipSuccess || dalloc(&d_tmp, (size_t)2 * H + 16) != hipSuccess) {
        if (l->dir) (void)hipFree(l->dir);
        if (l->off) (void)hipFree(l->off);
        delete l;
        return set_err(PTL_ERR_HIP, "LUT allocation failed");
    }
    HIPCHK(hipMemcpy(d_tmp, beam_altitude_deg, (size_t)H * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tmp + H, beam_azimuth_deg, (size_t)H * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tmp + 2 * H, T, sizeof T, hipMemcpyHostToDevice));
    k_build_lut<<<(int)((n + 255) / 256), 256>>>(H, W, d_tmp, d_tmp + H, lidar_origin_to_beam_origin_mm, d_tmp + 2 * H, l->dir, l->off);
    HIPCHK(hipDeviceSynchronize());
    (void)hipFree(d_tmp);
    *out = l;
    return PTL_OK;
}
extern "C" int ptl_lut_destroy(ptl_lut* l) {
    if (!l) return PTL_OK;
    (void)hipSetDevice(l->device_id);
    (void)hipFree(l->dir); (void)hipFree(l->off);
    delete l;
    return PTL_OK;
}
// XYZLut.__call__: range image -> (H*W, 3) f64 xyz in metres (RANGE == 0 -> (0,0,0))
__global__ __launch_bounds__(256) void k_lut_apply(int n, const unsigned* range, const double* dir, const double* off, double* xyz) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned rg = range[i];
    const double r = (double)rg;
    for (int k = 0; k < 3; ++k) xyz[3 * (size_t)i + k] = rg ? r * dir[3 * (size_t)i + k] + off[3 * (size_t)i + k] : 0.0;
}
extern "C" int ptl_lut_apply(ptl_lut* l, const uint32_t* range_mm, double* xyz_out) {
    if (!l || !range_mm || !xyz_out) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(l->device_id));
    const size_t n = (size_t)l->H * l->W;
    unsigned* d_r = nullptr;
    double* d_x = nullptr;
    HIPCHK(dalloc(&d_r, n));
    HIPCHK(dalloc(&d_x, 3 * n));
    HIPCHK(hipMemcpy(d_r, range_mm, n * 4, hipMemcpyHostToDevice));
    k_lut_apply<<<(int)((n + 255) / 256), 256>>>((int)n, d_r, l->dir, l->off, d_x);
    HIPCHK(hipMemcpy(xyz_out, d_x, n * 24, hipMemcpyDeviceToHost));
    (void)hipFree(d_r); (void)hipFree(d_x);
    return PTL_OK;
}
// StreamStatsTracker.trackScan's reduction (reference ins/data.py:284-308) on a range image given by the host
extern "C" int ptl_range_stats(int device_id, const uint32_t* range, int32_t H, int32_t W, int32_t beams_num,
                               double range_to_m, double out5[5]) {
    if (!range || !out5 || H < 1 || W < 1) return set_err(PTL_ERR_ARG, "bad argument");
    if (ptl_device_count() <= device_id) return set_err(PTL_ERR_HIP, "no HIP device %d (the HIP backend is the only backend)", device_id);
    HIPCHK(hipSetDevice(device_id));
    const size_t n = (size_t)H * W;
    unsigned* d_r = nullptr;
    unsigned char* d_m = nullptr;
    double* d_o = nullptr;
    HIPCHK(dalloc(&d_r, n));
    HIPCHK(dalloc(&d_o, 5));
    HIPCHK(hipMemcpy(d_r, range, n * 4, hipMemcpyHostToDevice));
    if (beams_num > 0) {  // np.linspace(0, H, num, endpoint=False, dtype=int) (:288-293)
        std::vector<unsigned char> m((size_t)H, 0);
        const double step = (double)H / (double)beams_num;
        for (int i = 0; i < beams_num; ++i) {
            const int r = (int)((double)i * step);
            if (r >= 0 && r < H) m[(size_t)r] = 1;
        }
        HIPCHK(hipMalloc((void**)&d_m, (size_t)H));
        HIPCHK(hipMemcpy(d_m, m.data(), (size_t)H, hipMemcpyHostToDevice));
    }
    k_range_stats<<<1, 1024>>>(d_r, H, W, d_m, range_to_m, d_o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out5, d_o, 40, hipMemcpyDeviceToHost));
    (void)hipFree(d_r); (void)hipFree(d_o);
    if (d_m) (void)hipFree(d_m);
    return PTL_OK;
}
// reduce_active_beams (reference utils.py:328-341): rows np.linspace(0, H, beams, endpoint=False, dtype=int) stay
// active, every other row's RANGE is treated as 0.  beams_num <= 0 re-enables all rows.
extern "C" int ptl_icp_set_active_beams(ptl_icp* h, int32_t H, int32_t beams_num) {
    if (!h || H < 1) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_row_mask) { (void)hipFree(h->d_row_mask); h->d_row_mask = nullptr; }
    if (beams_num <= 0) return PTL_OK;
    std::vector<unsigned char> m((size_t)H, 0);
    const double step = (double)H / (double)beams_num;  // numpy linspace(0, H, num, endpoint=False): i * (H / num)
    for (int i = 0; i < beams_num; ++i) {
        const int r = (int)((double)i * step);
        if (r >= 0 && r < H) m[(size_t)r] = 1;
    }
    HIPCHK(hipMalloc((void**)&h->d_row_mask, (size_t)H));
    HIPCHK(hipMemcpy(h->d_row_mask, m.data(), (size_t)H, hipMemcpyHostToDevice));
    return PTL_OK;
}
// KissICPWrapper.register_frame on a raw range image (reference kiss.py:54-74): XYZLut + RANGE != 0 mask + pipeline
extern "C" int ptl_icp_register_range(ptl_icp* h, ptl_lut* lut, const uint32_t* range_mm, double scan_ts,
                                      const double* guess, double out_pose[16], ptl_icp_stats* stats) {
    (void)scan_ts;
    if (!h || !lut || !range_mm) return set_err(PTL_ERR_ARG, "null argument");
    const int64_t n = (int64_t)lut->H * lut->W;
    if (n > h->n_max) return set_err(PTL_ERR_CAPACITY, "scan has %lld points, capacity %lld", (long long)n, (long long)h->n_max);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(h->d_in, range_mm, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    if (guess) HIPCHK(hipMemcpyAsync(h->d_ext, guess, 16 * 8, hipMemcpyHostToDevice, h->stream));
    int rc = icp_enqueue_scan(h, nullptr, nullptr, nullptr, n, guess ? h->d_ext : nullptr, (const unsigned*)h->d_in, lut);
    if (rc) return rc;
    return icp_percall_finish(h, out_pose, stats);
}

// ------------------------------------------------------------------------------------------------ posed scans (flyby)
// Poses along a time-stamped trajectory == ouster.sdk.pose_util.TrajectoryEvaluator as the reference uses it
// (utils.py:344-392 pose_scans_from_nc_gt, time_bounds = 1.5; cli/ekf_bench.py:489, :537 --use-gt-guess, time_bounds = 1.0;
// third-party, [UPSTREAM-KNOWLEDGE]): between the knots (t_i, P_i) that bracket ts the pose is the SE(3) geodesic
//   P(ts) = P_i Exp(alpha Log(P_i^-1 P_i+1)),  alpha = (ts - t_i) / (t_i+1 - t_i);
// up to `before` / `after` seconds outside the knots the first / last segment is extended (alpha < 0 / > 1); further out
// is an error (the reference skips such scans, utils.py:382-384).  One thread per timestamp, binary search for the segment.
__global__ __launch_bounds__(256) void k_traj_poses_at(const double* kt, const double* kp, int n, double before, double after,
                                                       const double* ts, int m, double* out, int* n_outside) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double t = ts[j];
    if (!(t >= kt[0] - before) || !(t <= kt[n - 1] + after)) {
        atomicAdd(n_outside, 1);
        for (int k = 0; k < 16; ++k) out[16 * (size_t)j + k] = (k % 5 == 0) ? 1.0 : 0.0;
        return;
    }
    int lo = 0, hi = n - 1;  // largest i with kt[i] <= t, clamped to a valid segment start
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (kt[mid] <= t) lo = mid; else hi = mid;
    }
    const int i = lo < n - 1 ? lo : n - 2;
    const double alpha = (t - kt[i]) / (kt[i + 1] - kt[i]);
    const Rt P0 = rt_from16(kp + 16 * (size_t)i), P1 = rt_from16(kp + 16 * (size_t)(i + 1));
    double xi[6];
    se3_log(rt_mul(rt_inv(P0), P1), xi);
    for (int k = 0; k < 6; ++k) xi[k] *= alpha;
    rt_to16(rt_mul(P0, se3_exp(xi)), out + 16 * (size_t)j);
}
extern "C" int ptl_traj_poses_at(int device_id, const double* knot_ts, const double* knot_poses16, int64_t n_knots,
                                 double bound_before, double bound_after, const double* ts, int64_t n, double* poses16_out,
                                 int64_t* n_outside) {
    if (!knot_ts || !knot_poses16 || !ts || !poses16_out || n_knots < 2 || n < 0) return set_err(PTL_ERR_ARG, "bad argument (a trajectory needs >= 2 knots)");
    for (int64_t i = 1; i < n_knots; ++i)
        if (!(knot_ts[i] > knot_ts[i - 1])) return set_err(PTL_ERR_ARG, "knot timestamps must increase strictly (knot %lld)", (long long)i);
    if (ptl_device_count() <= device_id) return set_err(PTL_ERR_HIP, "no HIP device %d (the HIP backend is the only backend)", device_id);
    HIPCHK(hipSetDevice(device_id));
    if (n_outside) *n_outside = 0;
    if (n == 0) return PTL_OK;
    double *d_kt = nullptr, *d_kp = nullptr, *d_ts = nullptr, *d_out = nullptr;
    int* d_cnt = nullptr;
    hipError_t e = dalloc(&d_kt, (size_t)n_knots);
    if (e == hipSuccess) e = dalloc(&d_kp, (size_t)n_knots * 16);
    if (e == hipSuccess) e = dalloc(&d_ts, (size_t)n);
    if (e == hipSuccess) e = dalloc(&d_out, (size_t)n * 16);
    if (e == hipSuccess) e = dalloc(&d_cnt, 1);
    if (e == hipSuccess) e = hipMemcpy(d_kt, knot_ts, (size_t)n_knots * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_kp, knot_poses16, (size_t)n_knots * 128, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_ts, ts, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof(int));
    int outside = 0;
    if (e == hipSuccess) {
        k_traj_poses_at<<<(int)((n + 255) / 256), 256>>>(d_kt, d_kp, (int)n_knots, bound_before, bound_after, d_ts, (int)n, d_out, d_cnt);
        e = hipMemcpy(poses16_out, d_out, (size_t)n * 128, hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess) e = hipMemcpy(&outside, d_cnt, sizeof(int), hipMemcpyDeviceToHost);
    for (void* q : {(void*)d_kt, (void*)d_kp, (void*)d_ts, (void*)d_out, (void*)d_cnt})
        if (q) (void)hipFree(q);
    if (e != hipSuccess) return set_err(PTL_ERR_HIP, "ptl_traj_poses_at: %s", hipGetErrorString(e));
    if (n_outside) *n_outside = outside;
    return PTL_OK;
}
// ouster client.dewarp(XYZLut(scan), column_poses = scan.pose) (what ScansAccumulator does with a posed scan, reference
// fly.py:75-86): pixel (u, v) -> R_v (range dir + off) + t_v with the pose of ITS column.  As upstream, pixels without a
// return (range 0 -> xyz 0) land on their column's sensor origin; n_valid counts the others.
__global__ __launch_bounds__(256) void k_dewarp(int H, int W, const unsigned* range, const double* dir, const double* off,
                                                const double* col_poses, double* xyz, int* n_valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool valid = false;
    if (i < H * W) {
        const unsigned rg = range[i];
        const double r = (double)rg;
        const double* P = col_poses + 16 * (size_t)(i % W);
        double p[3];
        for (int k = 0; k < 3; ++k) p[k] = rg ? r * dir[3 * (size_t)i + k] + off[3 * (size_t)i + k] : 0.0;
        for (int k = 0; k < 3; ++k) xyz[3 * (size_t)i + k] = ((P[4 * k] * p[0] + P[4 * k + 1] * p[1]) + P[4 * k + 2] * p[2]) + P[4 * k + 3];
        valid = rg != 0;
    }
    const int nv = __syncthreads_count(valid ? 1 : 0);
    if (threadIdx.x == 0 && nv) atomicAdd(n_valid, nv);
}
extern "C" int ptl_lut_dewarp(ptl_lut* l, const uint32_t* range_mm, const double* col_poses16, double* xyz_out, int64_t* n_valid) {
    if (!l || !range_mm || !col_poses16 || !xyz_out) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(l->device_id));
    const size_t n = (size_t)l->H * l->W;
    unsigned* d_r = nullptr;
    double *d_x = nullptr, *d_p = nullptr;
    int* d_c = nullptr;
    hipError_t e = dalloc(&d_r, n);
    if (e == hipSuccess) e = dalloc(&d_x, 3 * n);
    if (e == hipSuccess) e = dalloc(&d_p, (size_t)l->W * 16);
    if (e == hipSuccess) e = dalloc(&d_c, 1);
    if (e == hipSuccess) e = hipMemcpy(d_r, range_mm, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_p, col_poses16, (size_t)l->W * 128, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_c, 0, sizeof(int));
    int cnt = 0;
    if (e == hipSuccess) {
        k_dewarp<<<(int)((n + 255) / 256), 256>>>(l->H, l->W, d_r, l->dir, l->off, d_p, d_x, d_c);
        e = hipMemcpy(xyz_out, d_x, n * 24, hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess) e = hipMemcpy(&cnt, d_c, sizeof(int), hipMemcpyDeviceToHost);
    for (void* q : {(void*)d_r, (void*)d_x, (void*)d_p, (void*)d_c})
        if (q) (void)hipFree(q);
    if (e != hipSuccess) return set_err(PTL_ERR_HIP, "ptl_lut_dewarp: %s", hipGetErrorString(e));
    if (n_valid) *n_valid = cnt;
    return PTL_OK;
}

// ================================================================================================ EKF
struct ptl_ekf {
    ptl_ekf_cfg cfg;
    hipStream_t stream;
    bool own_stream;
    EkfState* st;
    double* d_buf;  // staging: imu rows / pose / cov
    int64_t buf_rows;
};

extern "C" int ptl_ekf_default_cfg(ptl_ekf_cfg* cfg) {
    if (!cfg) return set_err(PTL_ERR_ARG, "cfg is null");
    ABI_CHECK(ptl_ekf_cfg, cfg);
    PTL_CFG_INIT(cfg);
    cfg->init_grav[2] = -9.782940329221166;  // GRAV * DOWN (es_ekf.py:75, ins/data.py:10)
    return PTL_OK;
}
static int ekf_reset(ptl_ekf* h) {
    HIPCHK(hipMemcpyAsync(h->d_buf, h->cfg.init_grav, 24, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_buf + 3, h->cfg.init_bacc, 24, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_buf + 6, h->cfg.init_bgyr, 24, hipMemcpyHostToDevice, h->stream));
    k_ekf_init<<<1, 64, 0, h->stream>>>(h->st, h->d_buf, h->d_buf + 3, h->d_buf + 6);
    HIPCHK(hipGetLastError());
    return PTL_OK;
}
static int ekf_create_impl(const ptl_ekf_cfg* cfg, hipStream_t shared, ptl_ekf** out) {
    if (!cfg || !out) return set_err(PTL_ERR_ARG, "null argument");
    ABI_CHECK(ptl_ekf_cfg, cfg);
    if (ptl_device_count() <= cfg->device_id) return set_err(PTL_ERR_HIP, "no HIP device %d (the HIP backend is the only backend)", cfg->device_id);
    HIPCHK(hipSetDevice(cfg->device_id));
    ptl_ekf* h = new ptl_ekf();
    h->cfg = *cfg;
    h->own_stream = shared == nullptr;
    h->stream = shared;
    h->st = nullptr; h->d_buf = nullptr; h->buf_rows = 1024;
    if (h->own_stream && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return set_err(PTL_ERR_HIP, "stream"); }
    if (dalloc(&h->st, 1) != hipSuccess || dalloc(&h->d_buf, (size_t)h->buf_rows * 7 + 64) != hipSuccess) {
        if (h->st) (void)hipFree(h->st);
        delete h;
        return set_err(PTL_ERR_HIP, "device allocation failed");
    }
    int rc = ekf_reset(h);
    if (rc == PTL_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = set_err(PTL_ERR_HIP, "sync");
    if (rc) { (void)hipFree(h->st); (void)hipFree(h->d_buf); delete h; return rc; }
    *out = h;
    return PTL_OK;
}
extern "C" int ptl_ekf_create(const ptl_ekf_cfg* cfg, ptl_ekf** out) { return ekf_create_impl(cfg, nullptr, out); }
extern "C" int ptl_ekf_destroy(ptl_ekf* h) {
    if (!h) return PTL_OK;
    (void)hipSetDevice(h->cfg.device_id);
    (void)hipFree(h->st); (void)hipFree(h->d_buf);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PTL_OK;
}
extern "C" int ptl_ekf_process_imu_batch(ptl_ekf* h, const double* imu, int64_t n) {
    if (!h || (!imu && n > 0)) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    for (int64_t off = 0; off < n; off += h->buf_rows) {
        const int m = (int)((n - off) < h->buf_rows ? (n - off) : h->buf_rows);
        HIPCHK(hipMemcpyAsync(h->d_buf, imu + 7 * off, (size_t)m * 56, hipMemcpyHostToDevice, h->stream));
        k_ekf_step<<<1, EKF_THREADS, 0, h->stream>>>(h->st, h->d_buf, 0, m, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        HIPCHK(hipStreamSynchronize(h->stream));  // staging buffer reuse
    }
    HIPCHK(hipGetLastError());
    return PTL_OK;
}
extern "C" int ptl_ekf_process_imu(ptl_ekf* h, const double lacc[3], const double avel[3], double ts) {
    if (!h || !lacc || !avel) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const EkfRow7 row = {{ts, lacc[0], lacc[1], lacc[2], avel[0], avel[1], avel[2]}};
    k_ekf_imu_value<<<1, 384, 0, h->stream>>>(h->st, row);  // asynchronous: the getters synchronise
    HIPCHK(hipGetLastError());
    return PTL_OK;
}
extern "C" int ptl_ekf_process_pose(ptl_ekf* h, const double pose[16], const double* meas_cov36) {
    if (!h || !pose) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    EkfPoseArg a;
    memcpy(a.pose, pose, sizeof a.pose);
    a.has_cov = meas_cov36 ? 1 : 0;
    if (meas_cov36) memcpy(a.cov, meas_cov36, sizeof a.cov); else memset(a.cov, 0, sizeof a.cov);
    k_ekf_pose_value<<<1, 384, 0, h->stream>>>(h->st, a);  // asynchronous: the getters synchronise
    HIPCHK(hipGetLastError());
    return PTL_OK;
}
extern "C" int ptl_ekf_get_state(ptl_ekf* h, double nav[19], double cov[324]) {
    if (!h) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    EkfState st;
    HIPCHK(hipMemcpyAsync(&st, h->st, sizeof st, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nav) {
        memcpy(nav, st.nav.pos, 24); memcpy(nav + 3, st.nav.q, 32); memcpy(nav + 7, st.nav.vel, 24);
        memcpy(nav + 10, st.nav.bg, 24); memcpy(nav + 13, st.nav.ba, 24); memcpy(nav + 16, st.nav.grav, 24);
    }
    if (cov) memcpy(cov, st.P, sizeof st.P);
    return PTL_OK;
}
extern "C" int ptl_ekf_pose_mat(ptl_ekf* h, double T[16]) {
    if (!h || !T) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(T, (char*)h->st + offsetof(EkfState, pose), 128, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}
extern "C" int ptl_ekf_ts(ptl_ekf* h, double* ts) {
    if (!h || !ts) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(ts, (char*)h->st + (offsetof(EkfState, nav) + offsetof(EkfNav, cur_ts)), 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTL_OK;
}

// ================================================================================================ one scan of the driver loop, per-call handles
// The reference's loop body (cli/ekf_bench.py:493-563) for host-fed data in ONE host round trip: the IMU samples that precede the scan
// (processImu each), the registration with the filter's pose as its guess (--use-imu-prediction) or the caller's / the constant-velocity
// one, processPose with the new pose.  Object by object (ESEKF.processImu x n, ekf.nav.pose_mat(), register_frame, processPose, nav again)
// that is n + 2 launches and THREE synchronising read-backs per scan; here the filter's launches go to its own stream, the Gauss-Newton
// kernel waits for the predicts on the device and reads the guess there, the update waits for the Gauss-Newton kernel on the device and
// reads the pose there, and the host waits once.  Same kernels, same order of operations per handle: same bits.
extern "C" int ptl_icp_ekf_step(ptl_icp* h, ptl_ekf* e, const double* imu_rows, int64_t n_imu, const void* xyz, int dtype, int64_t n,
                                const double* t01, const double* guess, int32_t use_imu_prediction, double kiss_pose[16],
                                double ekf_pose[16], double* ekf_ts, ptl_icp_stats* stats) {
    if (!h || !e || (!imu_rows && n_imu > 0) || (!xyz && n > 0) || n < 0 || n_imu < 0) return set_err(PTL_ERR_ARG, "bad argument");
    if (dtype != PTL_F32 && dtype != PTL_F64) return set_err(PTL_ERR_ARG, "dtype must be PTL_F32 or PTL_F64");
    if (n > h->n_max) return set_err(PTL_ERR_CAPACITY, "scan has %lld points, capacity %lld", (long long)n, (long long)h->n_max);
    if (h->cfg.device_id != e->cfg.device_id) return set_err(PTL_ERR_ARG, "registration on device %d, filter on device %d", h->cfg.device_id, e->cfg.device_id);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    if (!h->ev_step_guess) {
        HIPCHK(hipEventCreateWithFlags(&h->ev_step_guess, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_step_gn, hipEventDisableTiming));
    }
    hipStream_t es = e->stream;
    // processImu x n_imu (ptl_ekf_process_imu_batch's kernel).  Any count is served, like the call-by-call loop (a long IMU-only prefix,
    // dropped lidar frames, a 1 kHz IMU): what does not fit the staging buffer goes first, buf_rows samples at a time with a wait for
    // the buffer's reuse; the last buf_rows or fewer ride on this step's single wait.
    int64_t off = 0;
    for (; n_imu - off > e->buf_rows; off += e->buf_rows) {
        HIPCHK(hipMemcpyAsync(e->d_buf, imu_rows + 7 * off, (size_t)e->buf_rows * 56, hipMemcpyHostToDevice, es));
        k_ekf_step<<<1, EKF_THREADS, 0, es>>>(e->st, e->d_buf, 0, (int)e->buf_rows, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        HIPCHK(hipStreamSynchronize(es));
    }
    if (n_imu > off) {
        HIPCHK(hipMemcpyAsync(e->d_buf, imu_rows + 7 * off, (size_t)(n_imu - off) * 56, hipMemcpyHostToDevice, es));
        k_ekf_step<<<1, EKF_THREADS, 0, es>>>(e->st, e->d_buf, 0, (int)(n_imu - off), nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    }
    const size_t esz = dtype == PTL_F32 ? 4 : 8;
    if (n) HIPCHK(hipMemcpyAsync(h->d_in, xyz, (size_t)n * 3 * esz, hipMemcpyHostToDevice, h->stream));
    if (t01 && n) HIPCHK(hipMemcpyAsync(h->d_t01, t01, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    const double* guess_ptr = nullptr;
    if (use_imu_prediction) {  // ekf.nav.pose_mat() as the guess, read on the device once the predicts are through
        HIPCHK(hipEventRecord(h->ev_step_guess, es));
        h->gn_wait = h->ev_step_guess;
        guess_ptr = (const double*)((char*)e->st + offsetof(EkfState, pose));
    } else if (guess) {
        HIPCHK(hipMemcpyAsync(h->d_ext, guess, 16 * 8, hipMemcpyHostToDevice, h->stream));
        guess_ptr = h->d_ext;
    }
    h->gn_done = h->ev_step_gn;
    int rc = icp_enqueue_scan(h, dtype == PTL_F32 ? (const float*)h->d_in : nullptr, dtype == PTL_F64 ? (const double*)h->d_in : nullptr,
                              t01 ? h->d_t01 : nullptr, n, guess_ptr);
    h->gn_wait = nullptr; h->gn_done = nullptr;
    if (rc) return rc;
    // processPose(kiss pose): behind the Gauss-Newton kernel, on the filter's stream; its outputs (pose after the update, timestamp) land
    // in the tail of the filter's staging buffer
    double* d_out = e->d_buf + (size_t)e->buf_rows * 7;
    HIPCHK(hipStreamWaitEvent(es, h->ev_step_gn, 0));
    k_ekf_step<<<1, EKF_THREADS, 0, es>>>(e->st, e->d_buf, 0, 0, h->c.traj + 16 * (h->scans_done - 1), nullptr, d_out, d_out + 16, nullptr, 1);
    double out[17];
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, es));
    rc = icp_percall_finish(h, kiss_pose, stats);  // (synchronises the registration's stream: pose, statistics row, error flags)
    HIPCHK(hipStreamSynchronize(es));
    if (rc) return rc;
    if (ekf_pose) memcpy(ekf_pose, out, 128);
    if (ekf_ts) *ekf_ts = out[16];
    return PTL_OK;
}

// ================================================================================================ sequence runner
struct ptl_seq {
    ptl_seq_cfg cfg;
    hipStream_t stream;
    // The EKF launches run on their own stream: the update with scan k's pose and the IMU predicts up to scan k+1
    // overlap scan k's map update and scan k+1's deskew / downsampling; only the GN kernel waits for the EKF
    // (its guess), and the EKF waits for the GN kernel (the pose measurement).
    hipStream_t ekf_stream;
    hipEvent_t ev_guess, ev_gn;
    ptl_icp* icp;
    ptl_ekf* ekf;
    float* d_scans;      // [n_scans][pps][3] f32
    double* d_imu;       // [n_imu][7]
    std::vector<int64_t> imu_end;
    double* d_res_poses; // [n_scans][16]
    double* d_res_t;     // [n_scans]
    double* d_rows;      // [n_scans][8]
    std::vector<int64_t> scan_of_out;  // scan index of each processed output
    int64_t n_out;
    int64_t next_scan, imu_pos, imus_per_scan;  // driver-loop position (ekf_bench.py:491-518)
    ptl_lut* lut;                    // non-null => sweeps were uploaded as u32 range images
    std::vector<unsigned char> is_range;
};

extern "C" int ptl_seq_destroy(ptl_seq* s) {
    if (!s) return PTL_OK;
    (void)hipSetDevice(s->cfg.icp.device_id);
    if (s->icp) icp_free(s->icp);
    if (s->ekf) ptl_ekf_destroy(s->ekf);
    if (s->d_scans) (void)hipFree(s->d_scans);
    if (s->d_imu) (void)hipFree(s->d_imu);
    if (s->d_res_poses) (void)hipFree(s->d_res_poses);
    if (s->d_res_t) (void)hipFree(s->d_res_t);
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->ev_guess) (void)hipEventDestroy(s->ev_guess);
    if (s->ev_gn) (void)hipEventDestroy(s->ev_gn);
    if (s->ekf_stream) (void)hipStreamDestroy(s->ekf_stream);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return PTL_OK;
}
extern "C" int ptl_seq_create(const ptl_seq_cfg* cfg, ptl_seq** out) {
    if (!cfg || !out) return set_err(PTL_ERR_ARG, "null argument");
    ABI_CHECK(ptl_seq_cfg, cfg);
    ABI_CHECK(ptl_icp_cfg, &cfg->icp);
    ABI_CHECK(ptl_ekf_cfg, &cfg->ekf);
    if (cfg->n_scans < 1 || cfg->points_per_scan < 1) return set_err(PTL_ERR_ARG, "empty sequence");
    if (ptl_device_count() <= cfg->icp.device_id) return set_err(PTL_ERR_HIP, "no HIP device %d (the HIP backend is the only backend)", cfg->icp.device_id);
    HIPCHK(hipSetDevice(cfg->icp.device_id));
    ptl_seq* s = new ptl_seq();
    s->cfg = *cfg;
    s->icp = nullptr; s->ekf = nullptr; s->d_scans = nullptr; s->d_imu = nullptr;
    s->d_res_poses = nullptr; s->d_res_t = nullptr; s->d_rows = nullptr; s->n_out = 0; s->stream = nullptr;
    s->ekf_stream = nullptr; s->ev_guess = nullptr; s->ev_gn = nullptr;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->ekf_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_guess, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_gn, hipEventDisableTiming) != hipSuccess) {
        ptl_seq_destroy(s);
        return set_err(PTL_ERR_HIP, "stream / event creation failed");
    }
    ptl_icp_cfg ic = cfg->icp;
    if (ic.max_points_per_scan < cfg->points_per_scan) ic.max_points_per_scan = cfg->points_per_scan;
    s->cfg.icp = ic;
    ptl_ekf_cfg ec = cfg->ekf;
    ec.device_id = ic.device_id;
    int rc = icp_create_impl(&ic, s->stream, &s->icp);
    if (rc == PTL_OK) rc = ekf_create_impl(&ec, s->ekf_stream, &s->ekf);
    const size_t nim = cfg->n_imu > 0 ? (size_t)cfg->n_imu : 1;
    if (rc == PTL_OK &&
        (hipMalloc((void**)&s->d_scans, (size_t)cfg->n_scans * cfg->points_per_scan * 12) != hipSuccess ||
         dalloc(&s->d_imu, nim * 7) != hipSuccess || dalloc(&s->d_res_poses, (size_t)cfg->n_scans * 16) != hipSuccess ||
         dalloc(&s->d_res_t, (size_t)cfg->n_scans) != hipSuccess || dalloc(&s->d_rows, (size_t)cfg->n_scans * 8) != hipSuccess))
        rc = set_err(PTL_ERR_HIP, "sequence allocation failed (%.1f MB of scans)", cfg->n_scans * cfg->points_per_scan * 12 / 1e6);
    if (rc) { ptl_seq_destroy(s); return rc; }
    s->imu_end.assign((size_t)cfg->n_scans, 0);
    s->is_range.assign((size_t)cfg->n_scans, 0);
    s->lut = nullptr;
    *out = s;
    return PTL_OK;
}
extern "C" int ptl_seq_upload_scan(ptl_seq* s, int64_t k, const float* xyz) {
    if (!s || !xyz || k < 0 || k >= s->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    const size_t bytes = (size_t)s->cfg.points_per_scan * 12;
    HIPCHK(hipMemcpy((char*)s->d_scans + (size_t)k * bytes, xyz, bytes, hipMemcpyHostToDevice));
    return PTL_OK;
}
// sweep k as a raw range image (H*W u32 mm); needs ptl_seq_set_lut before running
extern "C" int ptl_seq_upload_range(ptl_seq* s, int64_t k, const uint32_t* range_mm) {
    if (!s || !range_mm || k < 0 || k >= s->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    const size_t slot = (size_t)s->cfg.points_per_scan * 12;
    HIPCHK(hipMemcpy((char*)s->d_scans + (size_t)k * slot, range_mm, (size_t)s->cfg.points_per_scan * 4, hipMemcpyHostToDevice));
    s->is_range[(size_t)k] = 1;
    return PTL_OK;
}
extern "C" int ptl_seq_set_lut(ptl_seq* s, ptl_lut* lut, int32_t active_beams) {
    if (!s || !lut) return set_err(PTL_ERR_ARG, "null argument");
    if ((int64_t)lut->H * lut->W != s->cfg.points_per_scan) return set_err(PTL_ERR_ARG, "LUT size does not match points_per_scan");
    s->lut = lut;
    return ptl_icp_set_active_beams(s->icp, lut->H, active_beams);
}
extern "C" int ptl_seq_upload_imu(ptl_seq* s, const double* imu, const int64_t* imu_end) {
    if (!s || !imu_end || (!imu && s->cfg.n_imu > 0)) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    if (s->cfg.n_imu > 0) HIPCHK(hipMemcpy(s->d_imu, imu, (size_t)s->cfg.n_imu * 56, hipMemcpyHostToDevice));
    for (int64_t k = 0; k < s->cfg.n_scans; ++k) {
        if (imu_end[k] < 0 || imu_end[k] > s->cfg.n_imu || (k && imu_end[k] < imu_end[k - 1])) return set_err(PTL_ERR_ARG, "imu_end must be non-decreasing within [0, n_imu]");
        s->imu_end[(size_t)k] = imu_end[k];
    }
    return PTL_OK;
}
static int seq_reset(ptl_seq* s) {
    int rc = icp_reset_device(s->icp);
    if (rc) return rc;
    rc = ekf_reset(s->ekf);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->ekf_stream));  // EKF launches may go to either stream afterwards
    s->scan_of_out.clear();
    s->n_out = 0;
    s->next_scan = 0;
    s->imu_pos = 0;
    s->imus_per_scan = 1;  // ekf_bench.py:491
    return PTL_OK;
}
// The reference's driver loop (cli/ekf_bench.py:493-563) enqueued for the next n scans: IMU samples up to
// the scan -> EKF predict; scan -> guess (EKF nav pose when use_imu_prediction) -> ICP -> EKF update.
// Nothing here waits for the device.
extern "C" int ptl_seq_enqueue(ptl_seq* s, int64_t n) {
    if (!s || n < 0 || s->next_scan + n > s->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    const bool with_ekf = s->cfg.with_ekf != 0;
    const double* guess_ptr = (with_ekf && s->cfg.use_imu_prediction) ? (const double*)((char*)s->ekf->st + offsetof(EkfState, pose)) : nullptr;
    const size_t pps = (size_t)s->cfg.points_per_scan;
    const int64_t end = s->next_scan + n;
    hipStream_t es = s->ekf_stream;  // used when the EKF does not feed the registration (no IMU prediction)
    for (int64_t k = s->next_scan; k < end; ++k) {
        // IMU samples that precede scan k and were not consumed yet (only before the very first scan of a run,
        // or when the previous scan was skipped: otherwise the previous scan's EKF launch already ran them)
        const int64_t e = with_ekf ? s->imu_end[(size_t)k] : s->imu_pos;
        if (e > s->imu_pos) {
            k_ekf_step<<<1, EKF_THREADS, 0, es>>>(s->ekf->st, s->d_imu, (int)s->imu_pos, (int)e, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
            s->imus_per_scan += e - s->imu_pos;
            s->imu_pos = e;
        }
        s->next_scan = k + 1;
        if (with_ekf && !s->imus_per_scan) continue;  // ekf_bench.py:512-518
        s->imus_per_scan = 0;
        const int64_t o = s->n_out;
        const int64_t e2 = (k + 1 < s->cfg.n_scans) ? s->imu_end[(size_t)k + 1] : s->imu_pos;
        if (with_ekf && guess_ptr) {
            // the EKF step (update with scan k-1's pose, IMU predicts up to scan k) runs on its own stream beside K0-K4 and
            // the map update; the GN launch of scan k waits for its pose - by then (K0-K4 take longer) it is there
            HIPCHK(hipEventRecord(s->ev_guess, es));
            s->icp->gn_wait = s->ev_guess;
        }
        if (with_ekf) s->icp->gn_done = s->ev_gn;
        int rc;
        if (s->is_range[(size_t)k]) {
            if (!s->lut) rc = set_err(PTL_ERR_STATE, "scan %lld is a range image but no LUT was set", (long long)k);
            else rc = icp_enqueue_scan(s->icp, nullptr, nullptr, nullptr, (int64_t)pps, guess_ptr,
                                       (const unsigned*)(s->d_scans + (size_t)k * pps * 3), s->lut);
        } else {
            rc = icp_enqueue_scan(s->icp, s->d_scans + (size_t)k * pps * 3, nullptr, nullptr, (int64_t)pps, guess_ptr);
        }
        s->icp->gn_wait = nullptr; s->icp->gn_done = nullptr;
        if (rc) return rc;
        const double* kiss_pose = s->icp->c.traj + 16 * (s->icp->scans_done - 1);
        if (with_ekf) {
            HIPCHK(hipStreamWaitEvent(es, s->ev_gn, 0));  // recorded right after the GN launch
            k_ekf_step<<<1, EKF_THREADS, 0, es>>>(s->ekf->st, s->d_imu, (int)s->imu_pos, (int)e2, kiss_pose, nullptr,
                                                 s->d_res_poses + 16 * o, s->d_res_t + o, s->d_rows + 8 * o, 1);
        }
        if (with_ekf) {
            s->imus_per_scan += e2 - s->imu_pos;
            s->imu_pos = e2;
        }
        s->scan_of_out.push_back(k);
        s->n_out++;
    }
    HIPCHK(hipGetLastError());
    return PTL_OK;
}
extern "C" int ptl_seq_wait(ptl_seq* s) {
    if (!s) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    int rc = icp_check_flags(s->icp);
    HIPCHK(hipStreamSynchronize(s->ekf_stream));
    return rc;
}
extern "C" int ptl_seq_run(ptl_seq* s, int64_t n) {
    if (!s || n < 0 || n > s->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    int rc = seq_reset(s);
    if (rc) return rc;
    rc = ptl_seq_enqueue(s, n);
    if (rc) return rc;
    return ptl_seq_wait(s);
}
extern "C" int ptl_seq_advance(ptl_seq* s, int64_t n) {
    int rc = ptl_seq_enqueue(s, n);
    if (rc) return rc;
    return ptl_seq_wait(s);
}
extern "C" int ptl_seq_copy_traj(ptl_seq* s, void* dst_device, int64_t max_rows, int64_t* rows) {
    if (!s || !dst_device) return set_err(PTL_ERR_ARG, "null argument");
    if (!s->cfg.with_ekf) return set_err(PTL_ERR_STATE, "trajectory rows need with_ekf");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    const int64_t n = s->n_out < max_rows ? s->n_out : max_rows;
    HIPCHK(hipStreamSynchronize(s->ekf_stream));
    if (n > 0) HIPCHK(hipMemcpyAsync(dst_device, s->d_rows, (size_t)n * 64, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (rows) *rows = n;
    return PTL_OK;
}
extern "C" int ptl_device_sync(int device_id) {
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipDeviceSynchronize());
    return PTL_OK;
}
extern "C" int ptl_seq_results(ptl_seq* s, double* res_poses, double* res_t, double* kiss_poses, ptl_icp_stats* stats,
                               int64_t max_n, int64_t* n_out) {
    if (!s) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(s->cfg.icp.device_id));
    const int64_t n = s->n_out < max_n ? s->n_out : max_n;
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipStreamSynchronize(s->ekf_stream));
    if (n > 0) {
        if (res_poses && s->cfg.with_ekf) HIPCHK(hipMemcpy(res_poses, s->d_res_poses, (size_t)n * 128, hipMemcpyDeviceToHost));
        if (res_t && s->cfg.with_ekf) HIPCHK(hipMemcpy(res_t, s->d_res_t, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (kiss_poses) HIPCHK(hipMemcpy(kiss_poses, s->icp->c.traj, (size_t)n * 128, hipMemcpyDeviceToHost));
        if (stats) {
            std::vector<ScanStats> tmp((size_t)n);
            HIPCHK(hipMemcpy(tmp.data(), s->icp->c.sstats, (size_t)n * sizeof(ScanStats), hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; ++i) stats_out(tmp[(size_t)i], &stats[i]);
        }
    }
    if (n_out) *n_out = n;
    return PTL_OK;
}
extern "C" int ptl_seq_traj_device(ptl_seq* s, void** dev_ptr, int64_t* rows) {
    if (!s || !dev_ptr || !rows) return set_err(PTL_ERR_ARG, "null argument");
    *dev_ptr = s->d_rows;
    *rows = s->n_out;
    return PTL_OK;
}
extern "C" int ptl_seq_icp(ptl_seq* s, ptl_icp** icp) {
    if (!s || !icp) return set_err(PTL_ERR_ARG, "null argument");
    *icp = s->icp;
    return PTL_OK;
}
extern "C" int ptl_seq_profile(ptl_seq* s, int enable, double* gn_ms_total, int64_t* gn_launches, int reset) {
    if (!s) return set_err(PTL_ERR_ARG, "null argument");
    return ptl_icp_profile(s->icp, enable, gn_ms_total, gn_launches, reset);
}

// ================================================================================================ batched runner
// S independent sequences on ONE GPU.  Two drivers over the same stage bodies:
//   free-running (default with gn_lanes_per_point = 8): one persistent launch (kx_seq_run, seq_kernel.h) carries up to
//     scans_per_launch scans of every sequence; each sequence's workgroups walk its pipeline at their own pace;
//   lockstep: every stage is one launch for all S scans (blockIdx.y = sequence), the S Gauss-Newton loops share one
//     persistent launch, and a step lasts as long as its slowest sequence.
struct ptl_batch {
    ptl_seq_cfg cfg;
    int S;
    hipStream_t stream;
    // the map update and the filter step of scan k run on `side` beside K0-K4 of scan k + 1 (both need only the GN result);
    // the GN launch of scan k + 1 waits for them
    hipStream_t side;
    hipEvent_t ev_gn, ev_side;
    bool ev_side_valid;
    ptl_icp* icp[GN_MAX_SEQ];
    ptl_ekf* ekf[GN_MAX_SEQ];
    float* d_scans[GN_MAX_SEQ];
    double* d_imu[GN_MAX_SEQ];
    int* d_imu_end[GN_MAX_SEQ];
    unsigned* d_bar;            // [SEQ_MAX_TEAMS][64] barrier counters and job words of the free-running kernel's teams
    SeqSched* d_sched;          // [8] its per-XCD scan schedulers
    int *d_flags, *h_flags;     // [GN_MAX_SEQ + 1] error flags of every sequence + the status word, gathered by k_finish_all (h_flags: pinned)
    unsigned* d_status;         // one sticky word: why teams of the free-running kernel left a launch early (SEQ_EXIT_*), cleared by a reset
    int force_agent;            // PTL_TEAM_SYNC=agent in the environment when the batch was created: no XCD-local barrier shortcut
    int balance_margin;         // PTL_SCHED_MARGIN (default 1; -1 = a team looks abroad only when its own XCD has nothing for it): seq_kernel.h SeqRun
    int dbg_dead_block, dbg_dead_round;  // test hook: ptl_batch_debug_stall_block
    int um_alt;                 // test hook: ptl_batch_debug_set_map_points_per_thread
    bool free_running;
    int team_wgs;               // workgroups per team of the free-running kernel (0 = by the number of sequences, batch_gseq)
    bool seq_run_checked;
    int64_t scans_per_launch;
    std::vector<int64_t> imu_end[GN_MAX_SEQ];
    double *d_res_poses[GN_MAX_SEQ], *d_res_t[GN_MAX_SEQ], *d_rows[GN_MAX_SEQ];
    unsigned char is_range;
    ptl_lut* lut;
    SeqCtx* d_ctx;
    int64_t next_scan, n_out;
    // Sweep ring (ptl_seq_cfg.resident_scans, round 6): the batch keeps `ring` sweep slots per sequence (ring == n_scans: every sweep resident, as
    // before); scan k lives in slot k % ring.  ring_lap = the lap d_ctx's scan_base is set for (a launch never crosses a lap: enqueue splits there).
    // up_hi[s] = sweeps of sequence s uploaded so far (in order); done_scans = scans whose launches are known complete (ptl_batch_wait).  Both are
    // atomics: uploads may run on another host thread while ptl_batch_wait blocks.
    int64_t ring, ring_lap;
    std::atomic<int64_t> up_hi[GN_MAX_SEQ];
    std::atomic<int64_t> done_scans;
    int64_t imu_pos[GN_MAX_SEQ];
    bool ctx_dirty;
    bool prof;
    std::vector<hipEvent_t> ev;
    size_t ev_used;
    double gn_ms;
    int64_t gn_launches;
};

// launch geometry: workgroups per team.  Lockstep (kx_assign): one team per sequence, 1 / 2 / 4 per XCD.  Free-running: 1 / 2 / 4
// teams per XCD for up to 8 / 16 / 64 sequences as before (a team of 8 workgroups of 512 threads at the default 256 workgroups),
// 8 teams up to 128 sequences, 16 beyond; ptl_batch_set_team_workgroups overrides.  Smaller teams pay the per-iteration
// fixed costs (workgroup reduction, exchange, solve) over more points each and need more sequences to stay busy.
static int batch_gseq(const ptl_batch* b) {
    if (b->free_running && b->team_wgs > 0) return b->team_wgs;
    const int J = b->cfg.icp.gn_workgroups / 8;
    if (!b->free_running || b->S <= 64) return J / (b->S <= 8 ? 1 : b->S <= 16 ? 2 : 4);
    const int g = J / (b->S <= 128 ? 8 : 16);
    return g < 1 ? 1 : g;
}
template <int GC>
static hipError_t seq_run_occupancy(bool p20, int threads, int* per_cu) {
    return p20 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kx_seq_run<20, GC>, threads, 0)
               : hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kx_seq_run<0, GC>, threads, 0);
}
// the instance of a template <int PC, int GC> kernel for `gseq` workgroups per team: GC = gseq when that is 32 / 16 / 8 / 4 / 2 / 1, else 0
#define GC_DISPATCH(gseq, CALL) do { switch (gseq) { case 32: CALL(32); break; case 16: CALL(16); break; case 8: CALL(8); break; \
    case 4: CALL(4); break; case 2: CALL(2); break; case 1: CALL(1); break; default: CALL(0); break; } } while (0)
// every workgroup of the persistent launch has to be resident (they wait for each other); a team needs a workgroup
// beside the filter's
static int batch_check_seq_run(ptl_batch* b) {
    const ptl_icp_cfg& ic = b->cfg.icp;
    if (ic.gn_lanes_per_point != 8) return set_err(PTL_ERR_ARG, "free-running batches use the 8-lane Gauss-Newton kernel (gn_lanes_per_point = 8)");
    const int gseq = batch_gseq(b), J = ic.gn_workgroups / 8;
    if (gseq < 1 || gseq > J || gseq > 64) return set_err(PTL_ERR_ARG, "a team needs 1 .. min(64, gn_workgroups / 8 = %d) workgroups, not %d", J, gseq);
    if (b->S > 8 * SEQ_SLOTS) return set_err(PTL_ERR_ARG, "at most %d sequences", 8 * SEQ_SLOTS);
    // the look-back words of the one-pass compactions tell a stale count from this scan's by a 16-bit tag (scan % 65535 + 1, lookback_tag):
    // a run must not come back to a tag it has used
    if (b->cfg.n_scans >= 65535) return set_err(PTL_ERR_ARG, "free-running batches run at most 65534 scans per sequence (look-back tags of the compaction passes)");
    if (b->cfg.with_ekf && ic.gn_threads < 384) return set_err(PTL_ERR_ARG, "free-running batches with a filter need gn_threads >= 384 (the filter step maps one thread to each of the 324 covariance cells)");
    int per_cu = 0, cus = 0;
    const bool p20 = ic.max_points_per_voxel == 20;
    hipError_t e = hipSuccess;
#define OCC(GC) e = seq_run_occupancy<GC>(p20, ic.gn_threads, &per_cu)
    GC_DISPATCH(gseq, OCC);
#undef OCC
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ic.device_id);
    if (e != hipSuccess) return set_err(PTL_ERR_HIP, "occupancy query failed: %s", hipGetErrorString(e));
    if ((int64_t)per_cu * cus < ic.gn_workgroups)
        return set_err(PTL_ERR_STATE, "free-running batch: %d workgroups x %d threads cannot be co-resident on device %d (%d CUs x %d per CU): lower gn_workgroups",
                       ic.gn_workgroups, ic.gn_threads, ic.device_id, cus, per_cu);
    return PTL_OK;
}
extern "C" int ptl_batch_destroy(ptl_batch* b) {
    if (!b) return PTL_OK;
    (void)hipSetDevice(b->cfg.icp.device_id);
    for (int s = 0; s < b->S; ++s) {
        if (b->icp[s]) icp_free(b->icp[s]);
        if (b->ekf[s]) ptl_ekf_destroy(b->ekf[s]);
        void* ptrs[] = {b->d_scans[s], b->d_imu[s], b->d_res_poses[s], b->d_res_t[s], b->d_rows[s], b->d_imu_end[s]};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    }
    if (b->d_ctx) (void)hipFree(b->d_ctx);
    if (b->d_bar) (void)hipFree(b->d_bar);
    if (b->d_sched) (void)hipFree(b->d_sched);
    if (b->d_status) (void)hipFree(b->d_status);
    if (b->d_flags) (void)hipFree(b->d_flags);
    if (b->h_flags) (void)hipHostFree(b->h_flags);
    for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
    if (b->ev_gn) (void)hipEventDestroy(b->ev_gn);
    if (b->ev_side) (void)hipEventDestroy(b->ev_side);
    if (b->side) (void)hipStreamDestroy(b->side);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return PTL_OK;
}
extern "C" int ptl_batch_create(const ptl_seq_cfg* cfg, int32_t n_sequences, ptl_batch** out) {
    if (!cfg || !out) return set_err(PTL_ERR_ARG, "null argument");
    ABI_CHECK(ptl_seq_cfg, cfg);
    ABI_CHECK(ptl_icp_cfg, &cfg->icp);
    ABI_CHECK(ptl_ekf_cfg, &cfg->ekf);
    if (n_sequences < 1 || n_sequences > GN_MAX_SEQ) return set_err(PTL_ERR_ARG, "n_sequences must be in [1, %d]", GN_MAX_SEQ);
    if (cfg->n_scans < 1 || cfg->points_per_scan < 1) return set_err(PTL_ERR_ARG, "empty sequence");
    if ((cfg->icp.gn_workgroups & 31) || cfg->icp.gn_workgroups > 512) return set_err(PTL_ERR_ARG, "batched runs need gn_workgroups = a multiple of 32 (8 XCDs x up to 4 sequences each), at most 512");
    if (ptl_device_count() <= cfg->icp.device_id) return set_err(PTL_ERR_HIP, "no HIP device %d (the HIP backend is the only backend)", cfg->icp.device_id);
    HIPCHK(hipSetDevice(cfg->icp.device_id));
    {   // the whole batch has to fit the device: say so here, with numbers, instead of failing part-way through the allocations
        ptl_icp_cfg ic0 = cfg->icp;
        if (ic0.max_points_per_scan < cfg->points_per_scan) ic0.max_points_per_scan = cfg->points_per_scan;
        const size_t slot0 = cfg->range_input ? 4 : 12;  // resident bytes per pixel of a sweep
        if (cfg->resident_scans < 0 || cfg->resident_scans == 1) return set_err(PTL_ERR_ARG, "resident_scans must be 0 (every sweep resident) or >= 2 sweep slots per sequence");
        const size_t res0 = (cfg->resident_scans > 0 && cfg->resident_scans < cfg->n_scans) ? (size_t)cfg->resident_scans : (size_t)cfg->n_scans;
        const size_t per_seq = icp_footprint_bytes(&ic0) + res0 * (size_t)cfg->points_per_scan * slot0 +
                               (size_t)(cfg->n_imu > 0 ? cfg->n_imu : 1) * 56 + (size_t)cfg->n_scans * (128 + 8 + 64 + 4) + (64u << 10);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && per_seq * (size_t)n_sequences > free_b)
            return set_err(PTL_ERR_CAPACITY, "batch of %d sequences needs %.1f GB of device memory (%.0f MB per sequence: %.0f MB of work buffers, map table "
                           "and block pool, %.0f MB of sweeps), %.1f GB are free of %.1f GB: fewer sequences, fewer resident sweeps or a smaller map_block_capacity",
                           n_sequences, per_seq * (double)n_sequences / 1e9, per_seq / 1e6, icp_footprint_bytes(&ic0) / 1e6,
                           (double)res0 * cfg->points_per_scan * slot0 / 1e6, free_b / 1e9, total_b / 1e9);
    }
    ptl_batch* b = new ptl_batch();
    b->cfg = *cfg;
    b->S = n_sequences;
    b->stream = nullptr; b->d_ctx = nullptr; b->lut = nullptr; b->is_range = 0; b->d_bar = nullptr; b->d_sched = nullptr; b->d_status = nullptr; b->d_flags = nullptr; b->h_flags = nullptr;
    b->dbg_dead_block = -1; b->dbg_dead_round = -1; b->um_alt = 0;
    {   // PTL_TEAM_SYNC=agent: the teams of the free-running kernel keep the agent-scope release at every barrier (seq_kernel.h team_sync)
        const char* e = getenv("PTL_TEAM_SYNC");
        b->force_agent = (e && strcmp(e, "agent") == 0) ? 1 : 0;
        const char* m = getenv("PTL_SCHED_MARGIN");
        b->balance_margin = (m && *m) ? atoi(m) : 1;
        if (b->balance_margin < -1 || b->balance_margin > 4096) b->balance_margin = 1;
    }
    b->free_running = cfg->icp.gn_lanes_per_point == 8;
    if (!b->free_running && cfg->icp.map_small_blocks > 0) { ptl_batch_destroy(b); return set_err(PTL_ERR_ARG, "map_small_blocks (two block classes) needs the free-running driver (gn_lanes_per_point = 8)"); }
    b->team_wgs = 0; b->seq_run_checked = false;
    b->scans_per_launch = 256;
    b->next_scan = 0; b->n_out = 0; b->ctx_dirty = true; b->prof = false; b->ev_used = 0; b->gn_ms = 0; b->gn_launches = 0;
    b->ring = (cfg->resident_scans > 0 && cfg->resident_scans < cfg->n_scans) ? cfg->resident_scans : cfg->n_scans;
    b->ring_lap = 0; b->done_scans = 0;
    for (int s = 0; s < GN_MAX_SEQ; ++s) b->up_hi[s] = 0;
    if (b->ring < cfg->n_scans && !b->free_running) { delete b; return set_err(PTL_ERR_ARG, "resident_scans (the sweep ring) is served by the free-running driver (gn_lanes_per_point = 8)"); }
    for (int s = 0; s < GN_MAX_SEQ; ++s) {
        b->icp[s] = nullptr; b->ekf[s] = nullptr; b->d_scans[s] = nullptr; b->d_imu[s] = nullptr; b->d_imu_end[s] = nullptr;
        b->d_res_poses[s] = nullptr; b->d_res_t[s] = nullptr; b->d_rows[s] = nullptr; b->imu_pos[s] = 0;
    }
    int rc = PTL_OK;
    b->side = nullptr; b->ev_gn = nullptr; b->ev_side = nullptr; b->ev_side_valid = false;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { delete b; return set_err(PTL_ERR_HIP, "stream"); }
    if (hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&b->ev_gn, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_side, hipEventDisableTiming) != hipSuccess) { ptl_batch_destroy(b); return set_err(PTL_ERR_HIP, "stream / event creation failed"); }
    ptl_icp_cfg ic = cfg->icp;
    if (ic.max_points_per_scan < cfg->points_per_scan) ic.max_points_per_scan = cfg->points_per_scan;
    b->cfg.icp = ic;
    ptl_ekf_cfg ec = cfg->ekf;
    ec.device_id = ic.device_id;
    const size_t nim = cfg->n_imu > 0 ? (size_t)cfg->n_imu : 1;
    for (int s = 0; s < b->S && rc == PTL_OK; ++s) {
        rc = icp_create_impl(&ic, b->stream, &b->icp[s], true);
        if (rc == PTL_OK) rc = ekf_create_impl(&ec, b->stream, &b->ekf[s]);
        while (rc == PTL_OK && b->icp[s]->traj_cap < cfg->n_scans) rc = icp_grow_traj(b->icp[s]);
        if (rc == PTL_OK &&
            (hipMalloc((void**)&b->d_scans[s], (size_t)b->ring * cfg->points_per_scan * (cfg->range_input ? 4 : 12)) != hipSuccess ||
             dalloc(&b->d_imu[s], nim * 7) != hipSuccess || dalloc(&b->d_res_poses[s], (size_t)cfg->n_scans * 16) != hipSuccess ||
             dalloc(&b->d_res_t[s], (size_t)cfg->n_scans) != hipSuccess || dalloc(&b->d_rows[s], (size_t)cfg->n_scans * 8) != hipSuccess ||
             dalloc(&b->d_imu_end[s], (size_t)cfg->n_scans) != hipSuccess))
            rc = set_err(PTL_ERR_HIP, "batch allocation failed");
        if (rc == PTL_OK && hipMemset(b->d_imu_end[s], 0, (size_t)cfg->n_scans * sizeof(int)) != hipSuccess) rc = set_err(PTL_ERR_HIP, "memset failed");
        b->imu_end[s].assign((size_t)cfg->n_scans, 0);
    }
    if (rc == PTL_OK && (dalloc(&b->d_ctx, (size_t)GN_MAX_SEQ) != hipSuccess || dalloc(&b->d_bar, (size_t)SEQ_MAX_TEAMS * 64) != hipSuccess || dalloc(&b->d_sched, (size_t)8) != hipSuccess ||
                         dalloc(&b->d_status, (size_t)16) != hipSuccess || hipMemset(b->d_status, 0, 16 * sizeof(unsigned)) != hipSuccess ||
                         dalloc(&b->d_flags, (size_t)GN_MAX_SEQ + 1) != hipSuccess || hipHostMalloc((void**)&b->h_flags, (size_t)(GN_MAX_SEQ + 1) * sizeof(int)) != hipSuccess))
        rc = set_err(PTL_ERR_HIP, "batch allocation failed");
    // (the free-running driver's own limits - co-residency of the persistent grid, team size - are checked when that driver is
    // chosen or first used: a batch that is switched to lockstep right after creation must not be refused on them)
    if (rc) { ptl_batch_destroy(b); return rc; }
    *out = b;
    return PTL_OK;
}
// Sweep ring (resident_scans < n_scans): sweeps arrive in order, and sweep k may take its slot (k % ring) only when the scan that used the slot
// before - k - ring - is known to be complete (ptl_batch_wait has returned for it).  Uploads of later sweeps may run while a launch works on
// earlier ones - from the thread that enqueued it, between ptl_batch_enqueue and ptl_batch_wait, or from another host thread.
static int batch_ring_slot_free(ptl_batch* b, int32_t s, int64_t k) {
    if (b->ring >= b->cfg.n_scans) return PTL_OK;
    const int64_t hi = b->up_hi[s], done = b->done_scans;
    if (k != hi) return set_err(PTL_ERR_STATE, "sweep ring (resident_scans = %lld): sweeps are uploaded in order - sequence %d expects sweep %lld, got %lld",
                                (long long)b->ring, s, (long long)hi, (long long)k);
    if (k >= done + b->ring) return set_err(PTL_ERR_STATE, "sweep ring (resident_scans = %lld): slot of sweep %lld still holds sweep %lld, which is not known to be done "
                                            "(%lld scans complete): ptl_batch_wait first", (long long)b->ring, (long long)k, (long long)(k - b->ring), (long long)done);
    return PTL_OK;
}
extern "C" int ptl_batch_upload_scan(ptl_batch* b, int32_t s, int64_t k, const float* xyz) {
    if (!b || !xyz || s < 0 || s >= b->S || k < 0 || k >= b->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    if (b->is_range || b->cfg.range_input) return set_err(PTL_ERR_STATE, "this batch holds range images");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    int rc = batch_ring_slot_free(b, s, k);
    if (rc) return rc;
    const size_t bytes = (size_t)b->cfg.points_per_scan * 12;
    HIPCHK(hipMemcpy((char*)b->d_scans[s] + (size_t)(k % b->ring) * bytes, xyz, bytes, hipMemcpyHostToDevice));
    if (b->ring < b->cfg.n_scans) b->up_hi[s] = k + 1;
    return PTL_OK;
}
extern "C" int ptl_batch_upload_range(ptl_batch* b, int32_t s, int64_t k, const uint32_t* range_mm) {
    if (!b || !range_mm || s < 0 || s >= b->S || k < 0 || k >= b->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    if (!b->lut) return set_err(PTL_ERR_STATE, "set the LUT first (ptl_batch_set_lut)");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    int rc = batch_ring_slot_free(b, s, k);
    if (rc) return rc;
    const size_t slot = (size_t)b->cfg.points_per_scan * (b->cfg.range_input ? 4 : 12);
    HIPCHK(hipMemcpy((char*)b->d_scans[s] + (size_t)(k % b->ring) * slot, range_mm, (size_t)b->cfg.points_per_scan * 4, hipMemcpyHostToDevice));
    if (b->ring < b->cfg.n_scans) b->up_hi[s] = k + 1;
    return PTL_OK;
}
extern "C" int ptl_batch_set_lut(ptl_batch* b, ptl_lut* lut, int32_t active_beams) {
    if (!b || !lut) return set_err(PTL_ERR_ARG, "null argument");
    if ((int64_t)lut->H * lut->W != b->cfg.points_per_scan || lut->W != b->cfg.icp.scan_cols) return set_err(PTL_ERR_ARG, "LUT size does not match the batch");
    b->lut = lut;
    b->is_range = 1;
    b->ctx_dirty = true;
    for (int s = 0; s < b->S; ++s) { int rc = ptl_icp_set_active_beams(b->icp[s], lut->H, active_beams); if (rc) return rc; }
    return PTL_OK;
}
extern "C" int ptl_batch_upload_imu(ptl_batch* b, int32_t s, const double* imu, const int64_t* imu_end) {
    if (!b || !imu_end || s < 0 || s >= b->S || (!imu && b->cfg.n_imu > 0)) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    if (b->cfg.n_imu > 0) HIPCHK(hipMemcpy(b->d_imu[s], imu, (size_t)b->cfg.n_imu * 56, hipMemcpyHostToDevice));
    for (int64_t k = 0; k < b->cfg.n_scans; ++k) {
        if (imu_end[k] < 0 || imu_end[k] > b->cfg.n_imu || (k && imu_end[k] < imu_end[k - 1])) return set_err(PTL_ERR_ARG, "imu_end must be non-decreasing within [0, n_imu]");
        // lockstep batching cannot skip a scan of one sequence only (ekf_bench.py:512-518): demand >= 1 IMU per scan
        if (b->cfg.with_ekf && k && imu_end[k] == imu_end[k - 1]) return set_err(PTL_ERR_ARG, "batched runs need at least one IMU sample between consecutive scans (scan %lld)", (long long)k);
        b->imu_end[s][(size_t)k] = imu_end[k];
    }
    std::vector<int> e32((size_t)b->cfg.n_scans);
    for (int64_t k = 0; k < b->cfg.n_scans; ++k) e32[(size_t)k] = (int)imu_end[k];
    HIPCHK(hipMemcpy(b->d_imu_end[s], e32.data(), e32.size() * sizeof(int), hipMemcpyHostToDevice));
    return PTL_OK;
}
static int batch_push_ctx(ptl_batch* b) {
    std::vector<SeqCtx> hv((size_t)GN_MAX_SEQ);
    SeqCtx* h = hv.data();
    memset(h, 0, sizeof(SeqCtx) * GN_MAX_SEQ);
    const bool with_ekf = b->cfg.with_ekf != 0;
    for (int s = 0; s < b->S; ++s) {
        Ctx c = b->icp[s]->c;
        c.in_f32 = nullptr; c.in_f64 = nullptr; c.in_range = nullptr; c.t01 = nullptr;
        c.n_in = (int)b->cfg.points_per_scan;
        c.overlap_pre = 1;  // the prologue of scan k + 1 runs beside the map update of scan k (side stream)
        c.ext_guess = (with_ekf && b->cfg.use_imu_prediction) ? (const double*)((char*)b->ekf[s]->st + offsetof(EkfState, pose)) : nullptr;
        if (b->is_range) { c.lut_dir = b->lut->dir; c.lut_off = b->lut->off; c.row_mask = b->icp[s]->d_row_mask; }
        h[s].c = c;
        h[s].scan_stride_floats = (long long)b->cfg.points_per_scan * (b->cfg.range_input ? 1 : 3);
        // (sweep ring: scan k of lap L = k / ring sits in slot k - L ring; the kernel addresses scan_base + k stride, so the base moves back a ring per lap)
        h[s].scan_base = b->d_scans[s] - (long long)b->ring_lap * (long long)b->ring * h[s].scan_stride_floats;
        h[s].input_is_range = b->is_range;
        h[s].n_scans = (int)b->cfg.n_scans;
        h[s].fd_buf[0] = b->icp[s]->fd_buf[0]; h[s].fd_buf[1] = b->icp[s]->fd_buf[1];
        h[s].ekf = b->ekf[s]->st; h[s].imu = b->d_imu[s]; h[s].imu_end = b->d_imu_end[s];
        h[s].res_poses = b->d_res_poses[s]; h[s].res_t = b->d_res_t[s]; h[s].rows = b->d_rows[s];
    }
    HIPCHK(hipMemcpyAsync(b->d_ctx, h, sizeof(SeqCtx) * GN_MAX_SEQ, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->ctx_dirty = false;
    return PTL_OK;
}
static int batch_reset(ptl_batch* b) {
    for (int s = 0; s < b->S; ++s) {
        int rc = icp_reset_device(b->icp[s]);
        if (rc) return rc;
        rc = ekf_reset(b->ekf[s]);
        if (rc) return rc;
        b->imu_pos[s] = 0;
    }
    b->next_scan = 0;
    b->n_out = 0;
    b->done_scans = 0;
    if (b->ring_lap != 0) { b->ring_lap = 0; b->ctx_dirty = true; }
    if (b->ring < b->cfg.n_scans)  // a ring that has been lapped no longer holds the run's first sweeps: they are uploaded again; else what is there stays
        for (int s = 0; s < b->S; ++s) if (b->up_hi[s] > b->ring) b->up_hi[s] = 0;
    HIPCHK(hipMemsetAsync(b->d_status, 0, sizeof(unsigned), b->stream));
    HIPCHK(hipStreamSynchronize(b->side));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->ev_side_valid = false;
    return PTL_OK;
}
__global__ void k_ring_rebase(SeqCtx* a, int S, long long delta_floats) {
    const int s = (int)threadIdx.x;
    if (s < S) a[s].scan_base += delta_floats;
}
// free-running driver: the scans [next_scan, next_scan + n) of every sequence in launches of up to scans_per_launch scans
static int batch_enqueue_free(ptl_batch* b, int64_t n) {
    const bool with_ekf = b->cfg.with_ekf != 0;
    const int S = b->S;
    const ptl_icp_cfg& ic = b->cfg.icp;
    hipStream_t st = b->stream;
    const int64_t end = b->next_scan + n;
    if (!b->seq_run_checked) { int rc = batch_check_seq_run(b); if (rc) return rc; b->seq_run_checked = true; }
    if (b->ring < b->cfg.n_scans)
        for (int s = 0; s < S; ++s)
            if (b->up_hi[s] < end) return set_err(PTL_ERR_STATE, "sweep ring: sequence %d has sweeps [0, %lld) uploaded, the launch needs [%lld, %lld)", s, (long long)b->up_hi[s].load(),
                                                  (long long)b->next_scan, (long long)end);
    while (b->next_scan < end) {
        const int64_t k0 = b->next_scan;
        int64_t k1 = (end - k0 > b->scans_per_launch) ? k0 + b->scans_per_launch : end;
        if (b->ring < b->cfg.n_scans) {  // a launch stays inside one lap of the ring (its sweeps are contiguous there)
            const int64_t lap = k0 / b->ring, lap_end = (lap + 1) * b->ring;
            if (k1 > lap_end) k1 = lap_end;
            if (lap != b->ring_lap) {
                k_ring_rebase<<<1, GN_MAX_SEQ, 0, st>>>(b->d_ctx, S, -(long long)(lap - b->ring_lap) * (long long)b->ring * (long long)b->cfg.points_per_scan * (b->cfg.range_input ? 1 : 3));
                b->ring_lap = lap;
            }
        }
        if (with_ekf) {  // IMU samples before the first scan of the launch that the filter has not seen (a run's first scan)
            for (int s0 = 0; s0 < S; s0 += EKF_MAX_SEQ) {  // (the argument block holds EKF_MAX_SEQ sequences)
                EkfBatchArgs ea;
                memset(&ea, 0, sizeof ea);
                bool any = false;
                const int ns = S - s0 < EKF_MAX_SEQ ? S - s0 : EKF_MAX_SEQ;
                for (int q = 0; q < ns; ++q) {
                    const int s = s0 + q;
                    ea.e[q] = b->ekf[s]->st; ea.imu[q] = b->d_imu[s];
                    ea.i0[q] = (int)b->imu_pos[s]; ea.i1[q] = (int)b->imu_end[s][(size_t)k0];
                    any = any || ea.i1[q] > ea.i0[q];
                    if (b->imu_end[s][(size_t)k0] > b->imu_pos[s]) b->imu_pos[s] = b->imu_end[s][(size_t)k0];
                }
                if (any) kb_ekf_step<<<ns, 384, 0, st>>>(ea);
            }
        }
        HIPCHK(hipMemsetAsync(b->d_bar, 0, (size_t)SEQ_MAX_TEAMS * 64 * sizeof(unsigned), st));
        k_sched_init<<<1, 8 * SEQ_SLOTS, 0, st>>>(b->d_sched, S, (int)k0, (int)k1);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (b->prof) {
            if (b->ev_used + 2 > b->ev.size())
                for (int q = 0; q < 2; ++q) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); b->ev.push_back(e); }
            e0 = b->ev[b->ev_used++]; e1 = b->ev[b->ev_used++];
            HIPCHK(hipEventRecord(e0, st));
        }
        SeqRun r;
        r.S = S; r.k0 = (int)k0; r.k1 = (int)k1; r.with_ekf = with_ekf ? 1 : 0; r.rebuild_every = ic.rebuild_every;
        const int gseq = batch_gseq(b);
        r.G = gseq;
        r.um_alt = b->um_alt;
        r.balance_margin = b->balance_margin;
        r.force_agent = b->force_agent; r.dbg_dead_block = b->dbg_dead_block; r.dbg_dead_round = b->dbg_dead_round;
        const bool p20 = ic.max_points_per_voxel == 20;
#define KXR(GC) do { if (p20) kx_seq_run<20, GC><<<ic.gn_workgroups, ic.gn_threads, 0, st>>>(b->d_ctx, r, b->d_sched, b->d_bar, b->d_status); \
                     else kx_seq_run<0, GC><<<ic.gn_workgroups, ic.gn_threads, 0, st>>>(b->d_ctx, r, b->d_sched, b->d_bar, b->d_status); } while (0)
        GC_DISPATCH(gseq, KXR);
#undef KXR
        if (b->prof) HIPCHK(hipEventRecord(e1, st));
        // did every sequence reach scan k1?  (a team may have left the launch while holding one: no exit is silent)
        k_sched_check<<<1, 8 * SEQ_SLOTS, 0, st>>>(b->d_ctx, b->d_sched, S, (int)k1, b->d_status);
        for (int s = 0; s < S; ++s) {
            if (with_ekf) b->imu_pos[s] = (k1 < b->cfg.n_scans) ? b->imu_end[s][(size_t)k1] : b->imu_end[s][(size_t)k1 - 1];
            b->icp[s]->scans_done += (k1 - k0);
            b->icp[s]->last_n = b->cfg.points_per_scan;
        }
        b->n_out += k1 - k0;
        b->next_scan = k1;
    }
    HIPCHK(hipGetLastError());
    return PTL_OK;
}
// back to the cold start (empty maps, fresh filters, scan 0 next) without running anything: what ptl_batch_run does first; after it the
// driver and the team size may be chosen again
extern "C" int ptl_batch_reset(ptl_batch* b) {
    if (!b) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    return batch_reset(b);
}
extern "C" int ptl_batch_set_driver(ptl_batch* b, int32_t free_running, int64_t scans_per_launch) {
    if (!b || scans_per_launch < 0) return set_err(PTL_ERR_ARG, "bad argument");
    if (b->next_scan != 0) return set_err(PTL_ERR_STATE, "choose the driver before the first scan of a run");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    if (scans_per_launch > 4096) return set_err(PTL_ERR_ARG, "scans_per_launch: at most 4096 (every wait inside the persistent kernel has a poll budget worth seconds, and a launch must stay well below it)");
    if (scans_per_launch > 0) b->scans_per_launch = scans_per_launch;
    if (!free_running && b->cfg.icp.map_small_blocks > 0) return set_err(PTL_ERR_ARG, "map_small_blocks (two block classes) needs the free-running driver");
    if (!free_running && b->ring < b->cfg.n_scans) return set_err(PTL_ERR_ARG, "resident_scans (the sweep ring) needs the free-running driver");
    const bool was = b->free_running;
    b->free_running = free_running != 0;
    b->seq_run_checked = false;
    if (free_running) { int rc = batch_check_seq_run(b); if (rc) { b->free_running = was; return rc; } b->seq_run_checked = true; }
    return PTL_OK;
}
// workgroups per team of the free-running kernel (0 = the default for the number of sequences, batch_gseq); before the first scan of a run
extern "C" int ptl_batch_set_team_workgroups(ptl_batch* b, int32_t team_workgroups) {
    if (!b || team_workgroups < 0) return set_err(PTL_ERR_ARG, "bad argument");
    if (b->next_scan != 0) return set_err(PTL_ERR_STATE, "choose the team size before the first scan of a run");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    const int was = b->team_wgs;
    b->team_wgs = team_workgroups;
    b->seq_run_checked = false;
    if (b->free_running) { int rc = batch_check_seq_run(b); if (rc) { b->team_wgs = was; return rc; } b->seq_run_checked = true; }
    return PTL_OK;
}
extern "C" int ptl_batch_team_workgroups(ptl_batch* b, int32_t* team_workgroups, int32_t* teams) {
    if (!b || !team_workgroups) return set_err(PTL_ERR_ARG, "null argument");
    const int g = batch_gseq(b);
    *team_workgroups = g;
    if (teams) {  // teams that can get work: per XCD min(workgroups / team size, its sequences)
        int n = 0;
        for (int x = 0; x < 8 && x < b->S; ++x) { const int per = (b->cfg.icp.gn_workgroups / 8) / (g > 0 ? g : 1), ns = (b->S - x + 7) / 8; n += per < ns ? per : ns; }
        *teams = n;
    }
    return PTL_OK;
}
extern "C" int ptl_batch_enqueue(ptl_batch* b, int64_t n) {
    if (!b || n < 0 || b->next_scan + n > b->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    if (b->ctx_dirty) { int rc = batch_push_ctx(b); if (rc) return rc; }
    if (b->free_running) return batch_enqueue_free(b, n);
    if (b->S > 32) return set_err(PTL_ERR_ARG, "the lockstep driver serves up to 32 sequences (four per XCD); %d need the free-running one", b->S);
    if (b->cfg.icp.map_small_blocks > 0) return set_err(PTL_ERR_ARG, "map_small_blocks (two block classes) needs the free-running driver");
    const bool with_ekf = b->cfg.with_ekf != 0;
    const int S = b->S;
    const int64_t pps = b->cfg.points_per_scan;
    const int nb = (int)((pps + 255) / 256);
    const ptl_icp_cfg& ic = b->cfg.icp;
    hipStream_t st = b->stream;
    const int64_t end = b->next_scan + n;
    for (int64_t k = b->next_scan; k < end; ++k) {
        if (with_ekf) {  // IMU samples before the first scan of a run (later ones ride on the previous scan's EKF launch)
            EkfBatchArgs ea;
            memset(&ea, 0, sizeof ea);
            bool any = false;
            for (int s = 0; s < S; ++s) {
                ea.e[s] = b->ekf[s]->st; ea.imu[s] = b->d_imu[s];
                ea.i0[s] = (int)b->imu_pos[s]; ea.i1[s] = (int)b->imu_end[s][(size_t)k];
                any = any || ea.i1[s] > ea.i0[s];
                b->imu_pos[s] = b->imu_end[s][(size_t)k] > b->imu_pos[s] ? b->imu_end[s][(size_t)k] : b->imu_pos[s];
            }
            if (any) {
                if (b->ev_side_valid) HIPCHK(hipStreamWaitEvent(st, b->ev_side, 0));
                kb_ekf_step<<<S, 384, 0, st>>>(ea);
            }
        }
        const int ki = (int)k;
        kb_scan_prologue<<<dim3(1, S), 1024, 0, st>>>(b->d_ctx, ki);
        kb_deskew_vds1<<<dim3(nb, S), 256, 0, st>>>(b->d_ctx, ki);
        kb_count_w1<<<dim3(nb, S), 256, 0, st>>>(b->d_ctx, ki);
        kb_compact_fd<<<dim3(nb, S), 256, 0, st>>>(b->d_ctx, ki);
        kb_vds2_fd<<<dim3(nb, S), 256, 0, st>>>(b->d_ctx, ki);
        kb_count_w2<<<dim3(nb, S), 256, 0, st>>>(b->d_ctx, ki);
        kb_compact_src<<<dim3(nb, S), 256, 0, st>>>(b->d_ctx, ki);
        if (b->ev_side_valid) HIPCHK(hipStreamWaitEvent(st, b->ev_side, 0));  // the previous scan's map update and filter step
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (b->prof) {
            if (b->ev_used + 2 > b->ev.size())
                for (int q = 0; q < 2; ++q) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); b->ev.push_back(e); }
            e0 = b->ev[b->ev_used++]; e1 = b->ev[b->ev_used++];
            HIPCHK(hipEventRecord(e0, st));
        }
        // one sequence per XCD: workgroups with blockIdx & 7 == s run sequence s's loop with gn_workgroups / 8 workgroups
        if (ic.gn_lanes_per_point == 8) {
            const int gseq = (ic.gn_workgroups / 8) / (S <= 8 ? 1 : S <= 16 ? 2 : 4);  // workgroups per sequence
            const bool p20 = ic.max_points_per_voxel == 20;
#define KX8(GC) do { if (p20) kx_gn_loop8<20, GC><<<ic.gn_workgroups, ic.gn_threads, 0, st>>>(b->d_ctx, S, ki); \
                     else kx_gn_loop8<0, GC><<<ic.gn_workgroups, ic.gn_threads, 0, st>>>(b->d_ctx, S, ki); } while (0)
            GC_DISPATCH(gseq, KX8);
#undef KX8
        } else if (ic.max_points_per_voxel == 20) kx_gn_loop<20><<<ic.gn_workgroups, ic.gn_threads, 0, st>>>(b->d_ctx, S, ki);
        else kx_gn_loop<0><<<ic.gn_workgroups, ic.gn_threads, 0, st>>>(b->d_ctx, S, ki);
        if (b->prof) HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventRecord(b->ev_gn, st));
        hipStream_t sd = b->side;
        HIPCHK(hipStreamWaitEvent(sd, b->ev_gn, 0));
        const int64_t o = b->n_out;
        if (with_ekf) {  // first: its pose is the next scan's guess
            EkfBatchArgs ea;
            memset(&ea, 0, sizeof ea);
            for (int s = 0; s < S; ++s) {
                const int64_t e2 = (k + 1 < b->cfg.n_scans) ? b->imu_end[s][(size_t)k + 1] : b->imu_pos[s];
                ea.e[s] = b->ekf[s]->st; ea.imu[s] = b->d_imu[s];
                ea.i0[s] = (int)b->imu_pos[s]; ea.i1[s] = (int)e2;
                ea.pose[s] = b->icp[s]->c.traj + 16 * o;
                ea.out_pose[s] = b->d_res_poses[s] + 16 * o; ea.out_t[s] = b->d_res_t[s] + o; ea.out_row8[s] = b->d_rows[s] + 8 * o;
                b->imu_pos[s] = e2;
            }
            ea.update_first = 1;
            kb_ekf_step<<<S, 384, 0, sd>>>(ea);
        }
        kb_map_insert_a<<<dim3(nb, S), 256, 0, sd>>>(b->d_ctx, ki);
        kb_map_insert_b<<<dim3(nb, S), 256, 0, sd>>>(b->d_ctx, ki);
        kb_map_insert_c<<<dim3(nb, S), 256, 0, sd>>>(b->d_ctx, ki);
        kb_map_prune<<<dim3((unsigned)((ic.map_block_capacity + 255) / 256), S), 256, 0, sd>>>(b->d_ctx, ki);
        for (int s = 0; s < S; ++s) { b->icp[s]->scans_done++; b->icp[s]->last_n = pps; }
        b->n_out++;
        b->next_scan = k + 1;
        if (ic.rebuild_every > 0 && (b->n_out % ic.rebuild_every) == 0) {
            for (int s = 0; s < S; ++s) {
                HIPCHK(hipMemsetAsync(b->icp[s]->c.tab, 0xFF, ((size_t)b->icp[s]->c.tmask + 1) * sizeof(TabEnt), sd));
                HIPCHK(hipMemsetAsync(&b->icp[s]->c.st->tab_used, 0, sizeof(unsigned), sd));
            }
            kb_map_rebuild<<<dim3((unsigned)((ic.map_block_capacity + 255) / 256), S), 256, 0, sd>>>(b->d_ctx, ki);
        }
        HIPCHK(hipEventRecord(b->ev_side, sd));
        b->ev_side_valid = true;
    }
    HIPCHK(hipGetLastError());
    return PTL_OK;
}
// closes the last scan of EVERY sequence (what k_finish_scan does for one) and collects the error flags and the batch's status word in one
// array: one launch and one copy per wait instead of S launches and S + 1 copies (240 + 241 of them sat inside the bench's timed region:
// 3 % of a 20-step run)
__global__ void k_finish_all(const SeqCtx* a, int S, const unsigned* status, int* out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) out[GN_MAX_SEQ] = (int)*status;
    if (s >= S) return;
    const Ctx& c = a[s].c;
    finish_pending(c, c.st);
    flush_map_stats(c, c.st);
    out[s] = c.st->err_flags;
}
extern "C" int ptl_batch_wait(ptl_batch* b) {
    if (!b) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    if (b->ctx_dirty) { int rc = batch_push_ctx(b); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(b->side));
    k_finish_all<<<(b->S + 63) / 64, 64, 0, b->stream>>>(b->d_ctx, b->S, b->d_status, b->d_flags);
    HIPCHK(hipMemcpyAsync(b->h_flags, b->d_flags, (size_t)(GN_MAX_SEQ + 1) * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->done_scans = b->next_scan;  // (every launch enqueued so far is through: their sweep slots may be overwritten)
    const int* flags = b->h_flags;
    const unsigned status = (unsigned)b->h_flags[GN_MAX_SEQ];
    for (size_t i = 0; i + 1 < b->ev_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]) == hipSuccess) { b->gn_ms += ms; b->gn_launches++; }
    }
    b->ev_used = 0;
    for (int s = 0; s < b->S; ++s)
        if (flags[s]) return set_err(PTL_ERR_CAPACITY, "sequence %d: device capacity/error flags 0x%x (1 key range, 2 block pool, 4 map table, 8 vds table, 16 gn barrier timeout); batch status 0x%x", s, flags[s], status);
    // no sequence carries an error, but a team of the free-running kernel left a launch early (a workgroup that never reached a
    // barrier; teams that never got work): the scans are complete, the launch was not healthy - said, not swallowed
    // (a team that only ran out of its idle budget while the last sequences were in other teams' hands - bit 4 alone, every sequence at its last
    // scan - did no harm: the bit stays readable through ptl_batch_status, the wait does not fail on it)
    if (status & ~SEQ_EXIT_IDLE) return set_err(PTL_ERR_STATE, "free-running launch: teams left early, status 0x%x (1 head-of-launch barrier, 2 job barrier, 4 idle, 8 gave up on a sequence, 16 scans incomplete)", status);
    return PTL_OK;
}
extern "C" int ptl_batch_status(ptl_batch* b, uint32_t* status) {
    if (!b || !status) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(status, b->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PTL_OK;
}
// test hook: workgroup `block` of the free-running grid returns right before the job barrier of its `round`-th job (0 = the first
// of every launch from now on); block < 0 = none
extern "C" int ptl_batch_debug_stall_block(ptl_batch* b, int32_t block, int32_t round) {
    if (!b) return set_err(PTL_ERR_ARG, "null argument");
    b->dbg_dead_block = block; b->dbg_dead_round = block < 0 ? -1 : round;
    return PTL_OK;
}
// test hook: points per thread and pass of the free-running kernel's map update - the build's default (SEQ_UM, 8) or its second instance
// (SEQ_UM_ALT, 4); anything else is refused.  The result must not depend on it (tests/test_gpu_batch.py).  points_out: the value in effect.
extern "C" int ptl_batch_debug_set_map_points_per_thread(ptl_batch* b, int32_t points, int32_t* points_out) {
    if (!b) return set_err(PTL_ERR_ARG, "null argument");
    if (points != 0 && points != SEQ_UM && points != SEQ_UM_ALT) return set_err(PTL_ERR_ARG, "this build carries the map update at %d and %d points per thread", SEQ_UM, SEQ_UM_ALT);
    if (points != 0) b->um_alt = (points == SEQ_UM_ALT && SEQ_UM_ALT != SEQ_UM) ? 1 : 0;
    if (points_out) *points_out = b->um_alt ? SEQ_UM_ALT : SEQ_UM;
    return PTL_OK;
}
// where the scans of sequence s ran (free-running kernel), cumulative since the cold start: out[0] scans run by a team of another XCD than
// the sequence's home XCD (s & 7), out[1] scans that ran on another XCD than the sequence's previous scan, out[2] XCC id of the last scan + 1
extern "C" int ptl_batch_sched_counters(ptl_batch* b, int32_t s, uint64_t out[4]) {
    if (!b || !out || s < 0 || s >= b->S) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    HIPCHK(hipStreamSynchronize(b->side));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, (char*)b->icp[s]->c.st + offsetof(DevState, sched_cnt), 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PTL_OK;
}
// compile-time constants of this build that callers' byte models depend on (bench.py EXEC_COST):
//   [0] GN8_KCAND  [1] doubles per answer row  [2] GN8_LDS_PTS  [3] SEQ_U  [4] SEQ_U2  [5] GN8_MAX_THREADS  [6] GN8_LPB  [7] GN8_SPEC
//   [8] GN8_SURV  [9] GN8_PREFETCH  [10] 1000 x GN8_KEEP  [11] bytes per map table entry  [12] bytes per VDS entry  [13] diagnostics compiled in
#ifndef PTL_CODE_ID
#define PTL_CODE_ID "unknown"
#endif
extern "C" const char* ptl_code_id(void) { return PTL_CODE_ID; }
extern "C" int ptl_build_info(int32_t out[16]) {
    if (!out) return set_err(PTL_ERR_ARG, "null argument");
    for (int i = 0; i < 16; ++i) out[i] = 0;
    out[0] = GN8_KCAND; out[1] = GN8_ANS_ROW; out[2] = GN8_LDS_PTS; out[3] = SEQ_U; out[4] = SEQ_U2; out[5] = GN8_MAX_THREADS;
    out[6] = GN8_LPB; out[7] = GN8_SPEC; out[8] = GN8_SURV; out[9] = GN8_PREFETCH; out[10] = (int32_t)(1000.0 * GN8_KEEP + 0.5);
    out[11] = (int32_t)sizeof(TabEnt); out[12] = (int32_t)sizeof(VdsEnt); out[14] = 0;
#if defined(GN_PHASE_CLOCKS) || defined(SEQ_STAGE_CLOCKS) || defined(GN_IT0_CLOCK)
    out[13] = 1;
#endif
    return PTL_OK;
}
extern "C" int ptl_batch_run(ptl_batch* b, int64_t n) {
    if (!b || n < 0 || n > b->cfg.n_scans) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    int rc = batch_reset(b);
    if (rc) return rc;
    rc = ptl_batch_enqueue(b, n);
    if (rc) return rc;
    return ptl_batch_wait(b);
}
extern "C" int ptl_batch_results(ptl_batch* b, int32_t s, double* res_poses, double* res_t, double* kiss_poses,
                                 ptl_icp_stats* stats, int64_t max_n, int64_t* n_out) {
    if (!b || s < 0 || s >= b->S) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    const int64_t n = b->n_out < max_n ? b->n_out : max_n;
    HIPCHK(hipStreamSynchronize(b->side));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (n > 0) {
        if (res_poses && b->cfg.with_ekf) HIPCHK(hipMemcpy(res_poses, b->d_res_poses[s], (size_t)n * 128, hipMemcpyDeviceToHost));
        if (res_t && b->cfg.with_ekf) HIPCHK(hipMemcpy(res_t, b->d_res_t[s], (size_t)n * 8, hipMemcpyDeviceToHost));
        if (kiss_poses) HIPCHK(hipMemcpy(kiss_poses, b->icp[s]->c.traj, (size_t)n * 128, hipMemcpyDeviceToHost));
        if (stats) {
            std::vector<ScanStats> tmp((size_t)n);
            HIPCHK(hipMemcpy(tmp.data(), b->icp[s]->c.sstats, (size_t)n * sizeof(ScanStats), hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; ++i) stats_out(tmp[(size_t)i], &stats[i]);
        }
    }
    if (n_out) *n_out = n;
    return PTL_OK;
}
extern "C" int ptl_batch_copy_traj(ptl_batch* b, int32_t s, void* dst_device, int64_t max_rows, int64_t* rows) {
    if (!b || !dst_device || s < 0 || s >= b->S) return set_err(PTL_ERR_ARG, "bad argument");
    if (!b->cfg.with_ekf) return set_err(PTL_ERR_STATE, "trajectory rows need with_ekf");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    const int64_t n = b->n_out < max_rows ? b->n_out : max_rows;
    HIPCHK(hipStreamSynchronize(b->side));
    if (n > 0) HIPCHK(hipMemcpyAsync(dst_device, b->d_rows[s], (size_t)n * 64, hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (rows) *rows = n;
    return PTL_OK;
}
extern "C" int ptl_batch_icp(ptl_batch* b, int32_t s, ptl_icp** icp) {
    if (!b || !icp || s < 0 || s >= b->S) return set_err(PTL_ERR_ARG, "bad argument");
    *icp = b->icp[s];
    return PTL_OK;
}
extern "C" int ptl_batch_seq_clocks(ptl_batch* b, int32_t s, int64_t out[8]) {
    if (!b || !out || s < 0 || s >= b->S) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    HIPCHK(hipStreamSynchronize(b->stream));
    long long h[8];
    HIPCHK(hipMemcpy(h, (char*)b->icp[s]->c.st + offsetof(DevState, seq_clk), sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out[i] = h[i];
    return PTL_OK;
}
// executed-work counters of sequence s (DevState::exec_cnt), cumulative since the cold start
extern "C" int ptl_batch_exec_counters(ptl_batch* b, int32_t s, uint64_t out[16]) {
    if (!b || !out || s < 0 || s >= b->S) return set_err(PTL_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    HIPCHK(hipStreamSynchronize(b->side));
    HIPCHK(hipStreamSynchronize(b->stream));
    memset(out, 0, 16 * sizeof(uint64_t));
    HIPCHK(hipMemcpy(out, (char*)b->icp[s]->c.st + offsetof(DevState, exec_cnt), 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out + 8, (char*)b->icp[s]->c.st + offsetof(DevState, empty_cnt), sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PTL_OK;
}
extern "C" int ptl_batch_gn_phases(ptl_batch* b, int64_t out[8]) {
    if (!b || !out) return set_err(PTL_ERR_ARG, "null argument");
    return ptl_icp_gn_phases(b->icp[0], out);
}
extern "C" int ptl_batch_profile(ptl_batch* b, int enable, double* gn_ms_total, int64_t* gn_launches, int reset) {
    if (!b) return set_err(PTL_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (gn_ms_total) *gn_ms_total = b->gn_ms;
    if (gn_launches) *gn_launches = b->gn_launches;
    if (reset) { b->gn_ms = 0; b->gn_launches = 0; }
    b->prof = enable != 0;
    return PTL_OK;
}

// ================================================================================================ multi-GPU: the trajectory gather (RCCL)
// The path shards across sequences only; its one collective is the all-gather of every rank's (S, T, 8) NC-GT rows
// [t, x, y, z, qx, qy, qz, qw] after the run (ptl_gather_trajectories) - ncclAllGather straight from librccl, over
// xGMI on the node.  The library does NO bootstrap: rank 0 makes the 128-byte ncclUniqueId (ptl_comm_unique_id), the caller carries the
// bytes to the other ranks however it likes (environment, a file, a gloo / MPI / TCP broadcast), every rank calls ptl_comm_create.
// librccl is opened on first use (dlopen): a process that never gathers needs no RCCL, and a process that has torch loaded gets the
// librccl that is already mapped.
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
extern char g_rccl_why[320];
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char why[256] = "";
};
RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    // RCCL has to sit on the SAME HIP runtime as this library: a process may hold two ROCm stacks (torch bundles its own libamdhip64 /
    // librccl with the same sonames; whichever HIP runtime was mapped first serves everybody).  So the first choice is the librccl that
    // lives beside the HIP runtime this library is actually bound to; a bare soname comes last (it would return whatever is already mapped).
    const char* override_path = getenv("PTL_RCCL_PATH");
    char beside[2][600] = {"", ""};
    {
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            const char* slash = strrchr(info.dli_fname, '/');
            if (slash) {
                const int dir = (int)(slash - info.dli_fname);
                snprintf(beside[0], sizeof beside[0], "%.*s/librccl.so.1", dir, info.dli_fname);
                snprintf(beside[1], sizeof beside[1], "%.*s/librccl.so", dir, info.dli_fname);
            }
        }
    }
    const char* names[] = {override_path, beside[0], beside[1], "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        if (override_path && *override_path && n != override_path) break;  // PTL_RCCL_PATH is exclusive: that library or none
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
        snprintf(api.why, sizeof api.why, "%s", dlerror());
    }
    if (!api.lib) { snprintf(g_rccl_why, sizeof g_rccl_why, "librccl could not be opened (%s); PTL_RCCL_PATH overrides the search", api.why); return nullptr; }
#define RSYM(f) api.f = (decltype(api.f))dlsym(api.lib, "nccl" #f)
    RSYM(GetUniqueId); RSYM(CommInitRank); RSYM(AllGather); RSYM(CommDestroy); RSYM(CommAbort); RSYM(GetErrorString);
#undef RSYM
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.GetErrorString) {
        snprintf(g_rccl_why, sizeof g_rccl_why, "the librccl that was opened lacks a needed symbol");
        dlclose(api.lib);
        api.lib = nullptr;
        return nullptr;
    }
    return &api;
}
char g_rccl_why[320] = "";
const char* rccl_why() { return g_rccl_why; }
}  // namespace
#define RCCLCHK(api, x)                                                                                          \
    do {                                                                                                         \
        ncclResult_t r_ = (x);                                                                                   \
        if (r_ != ncclSuccess) return set_err(PTL_ERR_HIP, "%s: %s (%s:%d)", #x, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct ptl_comm {
    ncclComm_t comm;
    int world, rank, device_id;
    hipStream_t stream;
    // staging of the gather, kept between calls (grow-only): send [per_rank], recv [world x per_rank] on the device, the host mirror of recv
    double *d_send, *d_recv;
    size_t cap_send, cap_recv;
    std::vector<double> host;
    bool broken;  // a local failure aborted the communicator (ncclCommAbort): the peers' collective fails instead of hanging, every later call is refused
};
extern "C" int ptl_comm_unique_id(uint8_t id[PTL_COMM_ID_BYTES]) {
    if (!id) return set_err(PTL_ERR_ARG, "null argument");
    static_assert(PTL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the ABI's id size is RCCL's");
    RcclApi* api = rccl_api();
    if (!api) return set_err(PTL_ERR_STATE, "%s", rccl_why());
    ncclUniqueId u;
    RCCLCHK(api, api->GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return PTL_OK;
}
extern "C" int ptl_comm_create(const uint8_t id[PTL_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device_id, ptl_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return set_err(PTL_ERR_ARG, "bad argument");
    if (ptl_device_count() <= device_id || device_id < 0) return set_err(PTL_ERR_HIP, "no HIP device %d", device_id);
    RcclApi* api = rccl_api();
    if (!api) return set_err(PTL_ERR_STATE, "%s", rccl_why());
    HIPCHK(hipSetDevice(device_id));
    ptl_comm* c = new ptl_comm();
    c->comm = nullptr; c->world = world; c->rank = rank; c->device_id = device_id; c->stream = nullptr;
    c->d_send = nullptr; c->d_recv = nullptr; c->cap_send = 0; c->cap_recv = 0; c->broken = false;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return set_err(PTL_ERR_HIP, "stream creation failed"); }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = api->CommInitRank(&c->comm, world, u, rank);  // (collective: returns when every rank has called it)
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return set_err(PTL_ERR_HIP, "ncclCommInitRank(world %d, rank %d, device %d): %s", world, rank, device_id, api->GetErrorString(r));
    }
    *out = c;
    return PTL_OK;
}
extern "C" int ptl_comm_destroy(ptl_comm* c) {
    if (!c) return PTL_OK;
    RcclApi* api = rccl_api();
    (void)hipSetDevice(c->device_id);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (api && c->comm && !c->broken) (void)api->CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PTL_OK;
}
// every rank: S sequences x T rows x 8 doubles on the DEVICE (padded with anything) and S row counts on the host; same S and T on every rank.
// rows_out [world][S][T][8] and counts_out [world][S] on the host, filled on every rank.  One ncclAllGather: the counts ride behind the rows
// in the same send buffer (as doubles - exact for any count below 2^53).
// A failure on THIS rank between communicator creation and the collective must not leave the peers waiting in ncclAllGather for ever
// (they sit in hipStreamSynchronize without a time-out): the communicator is aborted, which fails their collective.
static int comm_fail(ptl_comm* c, RcclApi* api, int rc) {
    if (api && api->CommAbort && c->comm && !c->broken) (void)api->CommAbort(c->comm);
    c->broken = true;
    return rc;
}
extern "C" int ptl_gather_trajectories(ptl_comm* c, const double* d_rows, int64_t S, int64_t T, const int64_t* counts,
                                       double* rows_out, int64_t* counts_out) {
    // arguments first: nothing is allocated, staged or sent for a call that is wrong in itself (the peers of a rank that returns HERE
    // are the caller's to tell - every rank validates the same S and T, and a bad count aborts the communicator below)
    if (!c || !d_rows || !counts || !rows_out || !counts_out || S < 1 || T < 1) return set_err(PTL_ERR_ARG, "bad argument");
    if (c->broken) return set_err(PTL_ERR_STATE, "the communicator was aborted after an earlier failure on this rank: create a new one");
    RcclApi* api = rccl_api();
    if (!api) return set_err(PTL_ERR_STATE, "%s", rccl_why());
    for (int64_t s = 0; s < S; ++s)
        if (counts[s] < 0 || counts[s] > T) return comm_fail(c, api, set_err(PTL_ERR_ARG, "count of sequence %lld outside [0, T]", (long long)s));
    HIPCHK(hipSetDevice(c->device_id));
    const size_t per_rank = (size_t)S * (size_t)T * 8 + (size_t)S, all = per_rank * (size_t)c->world;
    if (c->cap_send < per_rank) {
        if (c->d_send) (void)hipFree(c->d_send);
        c->d_send = nullptr; c->cap_send = 0;
        if (dalloc(&c->d_send, per_rank) != hipSuccess) return comm_fail(c, api, set_err(PTL_ERR_HIP, "gather buffer allocation failed (%.1f MB)", per_rank * 8 / 1e6));
        c->cap_send = per_rank;
    }
    if (c->cap_recv < all) {
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_recv = nullptr; c->cap_recv = 0;
        if (dalloc(&c->d_recv, all) != hipSuccess) return comm_fail(c, api, set_err(PTL_ERR_HIP, "gather buffer allocation failed (%.1f MB)", all * 8 / 1e6));
        c->cap_recv = all;
    }
    if (c->host.size() < all + (size_t)S) c->host.resize(all + (size_t)S);
    double* cnt = c->host.data() + all;  // (the counts as doubles, staged behind the mirror: exact below 2^53)
    for (int64_t s = 0; s < S; ++s) cnt[s] = (double)counts[s];
    if (hipMemcpyAsync(c->d_send, d_rows, (size_t)S * T * 64, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(c->d_send + (size_t)S * T * 8, cnt, (size_t)S * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return comm_fail(c, api, set_err(PTL_ERR_HIP, "gather staging failed: %s", hipGetErrorString(hipGetLastError())));
    ncclResult_t r = api->AllGather(c->d_send, c->d_recv, per_rank, ncclDouble, c->comm, c->stream);
    if (r != ncclSuccess) return comm_fail(c, api, set_err(PTL_ERR_HIP, "ncclAllGather: %s", api->GetErrorString(r)));
    if (hipMemcpyAsync(c->host.data(), c->d_recv, all * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return comm_fail(c, api, set_err(PTL_ERR_HIP, "gather read-back failed: %s", hipGetErrorString(hipGetLastError())));
    for (int rk = 0; rk < c->world; ++rk) {
        const double* p = c->host.data() + per_rank * (size_t)rk;
        memcpy(rows_out + (size_t)rk * S * T * 8, p, (size_t)S * T * 64);
        for (int64_t s = 0; s < S; ++s) counts_out[(size_t)rk * S + s] = (int64_t)p[(size_t)S * T * 8 + s];
    }
    return PTL_OK;
}
// ICP-only batches have no filter and so no NC-GT rows of its making: the registration's own poses (KissICP.poses, reference kiss.py:130,
// what `ekf-bench ouster` writes when no filter corrects them) as rows [scan index, x, y, z, qx, qy, qz, qw]
__global__ void k_kiss_rows(const double* traj16, int n, double* rows8) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= n) return;
    const double* T = traj16 + 16 * (size_t)k;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    double q[4];
    R_to_quat(R, q);
    double* o = rows8 + 8 * (size_t)k;
    o[0] = (double)k; o[1] = T[3]; o[2] = T[7]; o[3] = T[11];
    o[4] = q[0]; o[5] = q[1]; o[6] = q[2]; o[7] = q[3];
}
// ... of a batch: the rows its filter kernel wrote for every sequence (ptl_batch_copy_traj's source), T = the batch's n_scans
extern "C" int ptl_batch_gather_trajectories(ptl_batch* b, ptl_comm* c, double* rows_out, int64_t* counts_out) {
    if (!b || !c || !rows_out || !counts_out) return set_err(PTL_ERR_ARG, "null argument");
    if (c->device_id != b->cfg.icp.device_id) return set_err(PTL_ERR_ARG, "communicator on device %d, batch on device %d", c->device_id, b->cfg.icp.device_id);
    HIPCHK(hipSetDevice(b->cfg.icp.device_id));
    HIPCHK(hipStreamSynchronize(b->side));
    HIPCHK(hipStreamSynchronize(b->stream));
    const int64_t S = b->S, T = b->cfg.n_scans;
    double* d_rows = nullptr;
    HIPCHK(dalloc(&d_rows, (size_t)S * T * 8));
    std::vector<int64_t> counts((size_t)S, b->n_out < T ? b->n_out : T);
    hipError_t e = hipMemsetAsync(d_rows, 0, (size_t)S * T * 64, b->stream);
    for (int64_t s = 0; s < S && e == hipSuccess; ++s) {
        const int64_t n = counts[(size_t)s];
        if (n <= 0) continue;
        if (b->cfg.with_ekf) e = hipMemcpyAsync(d_rows + (size_t)s * T * 8, b->d_rows[s], (size_t)n * 64, hipMemcpyDeviceToDevice, b->stream);
        else { k_kiss_rows<<<(int)((n + 63) / 64), 64, 0, b->stream>>>(b->icp[s]->c.traj, (int)n, d_rows + (size_t)s * T * 8); e = hipGetLastError(); }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) { (void)hipFree(d_rows); return set_err(PTL_ERR_HIP, "row staging failed: %s", hipGetErrorString(e)); }
    const int rc = ptl_gather_trajectories(c, d_rows, S, T, counts.data(), rows_out, counts_out);
    (void)hipFree(d_rows);
    return rc;
}
