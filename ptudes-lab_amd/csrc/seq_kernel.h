// seq_kernel.h -- the free-running sequence kernel: ONE persistent launch carries a run of scans of every sequence of a
// batch, and every sequence advances at its own pace.
//
// The lockstep driver (ptl_batch_enqueue, one launch per stage for all sequences) makes every step wait for the
// sequence whose Gauss-Newton loop takes longest: over 16 sequences the slowest loop of a step runs 1.8 x the mean
// number of iterations (tools/lockstep.py), and the workgroups of the other sequences idle meanwhile.  Here the
// workgroups of an XCD form TEAMS (of 32, 16, 8, 4, 2 workgroups or one), and a team walks the whole per-scan pipeline of
// one sequence by itself - reference cli/ekf_bench.py:493-563 loop body = kiss.py:83-131 + ESEKF.processPose / processImu:
//
//     whole team:          K0 prologue | K1 deskew + vds1 | K3 compact fd (look-back) | K3b vds2 on the compact frame_downsample | K4 compact src (look-back)
//     whole team:          --- barrier ---  K5 Gauss-Newton loop (gn8_body)  --- barrier ---
//     filter wg:           ES-EKF: update with the scan's pose, predict through the IMU samples before the next scan, K0 of the next scan
//     whole team:          K7-K9 map insert a | b | c | K10 prune [| table reset | K11 rebuild]   (the filter workgroup joins when it is done)
//
// with team barriers where the lockstep driver has kernel boundaries.  Nothing is shared between teams, so a
// sequence whose loop converges early simply starts its next scan.  Stage bodies, reduction trees and the exchange are
// the ones of the per-stage kernels: results are bit-identical to the lockstep run and to the single-sequence run with
// as many Gauss-Newton workgroups.
//
// Teams are not tied to sequences: the sequences s = x (mod 8) belong to XCD x, whose teams serve them scan by scan - a team that
// finishes a scan takes the next scan of the sequence of its XCD that is furthest behind and not being worked on (SeqSched), and
// when its XCD has nothing for it, one of another XCD's.  With more sequences than teams the sequences advance
// evenly although their scans cost up to 40 % more or less than the average, and the run does not end with most
// of the chip waiting for the slowest one.
#pragma once
#include "ekf_kernels.h"
#include "icp_kernels.h"

// Barrier of the `n` workgroups of a team on a counter in device memory (zeroed by the host before the launch): the
// k-th barrier is passed when the counter reaches k n (`target`, kept by the caller; only thread 0's copy counts).
// The stages exchange their data through plain global memory, so the barrier is a release / acquire pair:
//   * `local` == false: agent-scope release before the arrival (on gfx950: `buffer_wbl2 sc1`, every dirty line of the XCD's L2
//     goes out to memory - the only way another XCD sees the data) and an agent-scope acquire (`buffer_inv sc1`) after;
//   * `local` == true - every workgroup of the team reported the same XCC_ID at the head of the launch (TeamEnv): they share
//     one L2.  Every wavefront waits until the L2 has acknowledged its own stores (s_waitcnt vmcnt(0)), the arrival and the
//     polls stay agent-scope atomics, the acquire stays (it empties the CU's vector L1): no write-back of the L2, which cost
//     35 MB of HBM writes per scan (round 2's PMC pass) at ~14 barriers per scan.  tools/hip/sync_probe.hip measures both.
// n == 1 (a team of one workgroup): a workgroup barrier.
// A wait that exceeds TEAM_BAR_POLLS, or sees `*abort_word` set, returns false: the caller leaves the kernel (and so does, at
// its next barrier, every other workgroup of the team).  With `st` given the expiry is an error of that sequence
// (ERR_GN_TIMEOUT + its abort word); without (the job barrier: nobody is working on a sequence) only `*abort_word` goes up.
#define TEAM_BAR_POLLS (1u << 23) /* polls with an s_sleep between them: several seconds of EXECUTED time (a wall-clock limit would fire
                                     falsely when the queue is time-sliced with another process's and the team sleeps in between) */
__device__ __forceinline__ bool team_sync(unsigned* word, unsigned n, unsigned& target, int* abort_word, DevState* st, const bool local) {
    __shared__ int s_bad;
    if (n <= 1u) {  // one workgroup: its waves share the CU's L1, but atomics execute in the L2 behind it - same release / acquire, no counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return true;
    }
    if (local) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    target += n;
    if (threadIdx.x == 0) {
        int bad = 0;
        if (!local) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned polls = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++polls & 255u) == 0u && (gn_abort_seen(abort_word) || polls > TEAM_BAR_POLLS)) { bad = 1; break; }
        }
        if (bad) {
            if (st) gn_raise_abort(st);
            __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_bad = bad;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return s_bad == 0;
}
// what a stage needs to synchronise its team: the barrier word, the team's size, the sequence's abort word, the mode
struct TeamEnv { unsigned* word; int* abort_word; DevState* st; bool local; };
__device__ __forceinline__ bool team_sync(const TeamEnv& e, unsigned n, unsigned& target) { return team_sync(e.word, n, target, e.abort_word, e.st, e.local); }
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

// The stages are separate (not inlined) functions: the Gauss-Newton loop and the filter step each need the whole register
// file of a wavefront, and inlined into one body they spill into each other's loops.  Each loads the scan's context from
// the sequence table itself (scalar loads), like the per-stage kernels.
#define SEQ_FAIL 0xFFFFFFFFu
#ifndef SEQ_U
#define SEQ_U 8  /* points per thread and pass in K1 and the map update (see Slice): 4 -> 8 took K0-K4 of a team of 2 from 1585 to 1406 us */
#endif
#ifndef SEQ_UM
#define SEQ_UM SEQ_U  /* ... in the map update */
#endif
#ifndef SEQ_UM_ALT
#define SEQ_UM_ALT 4  /* the map update's second instance (test hook: the result must not depend on the points per thread) */
#endif
#ifndef SEQ_U2
#define SEQ_U2 16  /* ... in K3 / K3b / K4 (a few registers per point: an index, a slot, two flags); they share their block size through bcnt1 / bcnt2 */
#endif
// make STAGES=1: workgroup 0 of every sequence adds the wall-clock ticks of every stage and of every barrier wait to
// st->dbg_sums[0..19] (tools/free_vs_lockstep.py prints them)
#ifdef SEQ_STAGE_CLOCKS
#define SQ_CLK_DECL long long sq_t = (long long)wall_clock64(); const bool sq_me = wg == 0 && threadIdx.x == 0
#define SQ_CLK(i) do { const long long n_ = (long long)wall_clock64(); if (sq_me) st->dbg_sums[i] += (double)(n_ - sq_t); sq_t = n_; } while (0)
#else
#define SQ_CLK_DECL do { } while (0)
#define SQ_CLK(i) do { } while (0)
#endif
// K0-K4 of scan k by the team's `nw` working workgroups (this one is number `wg`); returns the barrier target, SEQ_FAIL on abort
__device__ __noinline__ unsigned sq_prepare(const SeqCtx* a, int s, int k, int wg, int nw, unsigned target, unsigned* word, bool local) {
    const Ctx c = load_seq_ctx(a, s, k);
    DevState* st = c.st;
    const int BS = (int)blockDim.x * SEQ_U, nbs = (c.n_in + BS - 1) / BS;
    const int BS2 = (int)blockDim.x * SEQ_U2, nbs2 = (c.n_in + BS2 - 1) / BS2;
    Slice sl;
    sl.nb = nbs; sl.clk = wg == 0 ? 1 : 0;
    const TeamEnv te = {word, &st->gn_abort, st, local};
    SQ_CLK_DECL;
    if (wg == 0 && st->pro_next != k + 1) d_scan_prologue(c, true);  // (with a filter it has usually been run already: sq_filter)
    SQ_CLK(0);
    if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
    SQ_CLK(1);
    int wave_valid = 0;  // this wavefront's valid points over all its blocks (integer sum: any grouping gives the same total)
    if (c.in_f32 && !c.in_range) {  // (uniform) the next pass's raw points travel while this pass waits for its atomics
        RawF32<SEQ_U> raw;
        sl.b = wg;
        if (wg < nbs) raw = k1_load_raw<SEQ_U>(c, sl);
        for (sl.b = wg; sl.b < nbs; sl.b += nw) {
            const RawF32<SEQ_U> cur = raw;
            Slice nx = sl;
            nx.b = sl.b + nw;
            if (nx.b < nbs) raw = k1_load_raw<SEQ_U>(c, nx);
            d_deskew_vds1<SEQ_U>(c, sl, &cur, &wave_valid);
        }
    } else {
        for (sl.b = wg; sl.b < nbs; sl.b += nw) d_deskew_vds1<SEQ_U>(c, sl, nullptr, &wave_valid);
    }
    if ((threadIdx.x & 63) == 0 && wave_valid) atomicAdd(&st->n_valid, wave_valid);
    SQ_CLK(2);
    if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
    SQ_CLK(3);
    sl.nb = nbs2;
    const unsigned tag = lookback_tag(k);
    {   // K3: pass-1 winners -> frame_downsample, one pass with look-back (the next pass's first loads a pass ahead, as in K1)
        PreI32<SEQ_U2> pw;
        sl.b = wg;
        if (wg < nbs2) pw = stage_load_i32<SEQ_U2>(c.slot1, c.n_in, sl);
        for (sl.b = wg; sl.b < nbs2; sl.b += nw) {
            const PreI32<SEQ_U2> cur = pw;
            Slice nx = sl;
            nx.b = sl.b + nw;
            if (nx.b < nbs2) pw = stage_load_i32<SEQ_U2>(c.slot1, c.n_in, nx);
            d_compact_fd<SEQ_U2, true>(c, sl, &cur, tag);
        }
    }
    SQ_CLK(4);
    if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
    SQ_CLK(5);
    // K3b / K4 walk the COMPACT frame_downsample (N_d entries, a quarter of the raw indices)
    const int nbf = (st->n_down + BS2 - 1) / BS2;
    sl.nb = nbf;
    for (sl.b = wg; sl.b < nbf; sl.b += nw) d_vds2_fd<SEQ_U2>(c, sl);
    SQ_CLK(6);
    if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
    SQ_CLK(7);
    for (sl.b = wg; sl.b < nbf; sl.b += nw) d_compact_src<SEQ_U2, true>(c, sl, tag);
    SQ_CLK(8);
    return target;
}
// K7-K11 of scan k by all `nw` workgroups of the team.  The blocks of a phase are handed out through a counter (`ctr`, five
// words zeroed by the team's leader before the scan): the filter workgroup joins when its filter step is done and takes what
// is left - nobody waits for it, and a team of two does not leave the whole update to one workgroup.  Which workgroup
// inserts which block does not show in the result (voxel contents are ranked by point index, counters are integer sums).
__device__ __forceinline__ int sq_grab(unsigned* ctr) {
    __shared__ int s_blk;
    __syncthreads();
    if (threadIdx.x == 0) s_blk = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return s_blk;
}
// UM: points (blocks of the pool) per thread and pass.  Nothing in the result depends on it - block partitioning only - and the kernel carries
// two instances (SEQ_UM, the default, and SEQ_UM_ALT; SeqRun::um_alt picks, ptl_batch_debug_set_map_points_per_thread) so that a GPU test can
// hold that true: round 3-4's "map update fails at 4 points per thread" was a miscompiled branch chain in insert a (see there), found
// only because somebody built another U.
template <int UM>
__device__ __noinline__ unsigned sq_map_update(const SeqCtx* a, int s, int k, int wg, int nw, unsigned target, int rebuild, unsigned* word, unsigned* ctr, bool local) {
    const Ctx c = load_seq_ctx(a, s, k);
    DevState* st = c.st;
    const int BS = (int)blockDim.x, BU = BS * UM, nbd = (st->n_down_ins + BU - 1) / BU;
    Slice sl;
    sl.nb = nbd; sl.clk = 0;
    const TeamEnv te = {word, &st->gn_abort, st, local};
    SQ_CLK_DECL;
    for (sl.b = sq_grab(ctr); sl.b < nbd; sl.b = sq_grab(ctr)) d_map_insert_a<UM>(c, c.fd, &st->n_down_ins, 0, 1, sl);
    SQ_CLK(10);
    if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
    SQ_CLK(11);
    for (sl.b = sq_grab(ctr + 1); sl.b < nbd; sl.b = sq_grab(ctr + 1)) d_map_insert_b<UM, true>(c, &st->n_down_ins, 0, sl);
    SQ_CLK(12);
    if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
    SQ_CLK(13);
    SQ_CLK(14);  // (phase c is part of the prune pass here: d_map_insert_b / d_map_prune, FUSE)
    SQ_CLK(15);
    const int nvb = blk_virtual_count(c, st);  // small blocks below their high-water mark, then the full ones below theirs
    const int nbpu = (nvb + BU - 1) / BU;
    sl.nb = nbpu;
    for (sl.b = sq_grab(ctr + 3); sl.b < nbpu; sl.b = sq_grab(ctr + 3)) d_map_prune<UM, true>(c, nullptr, 1, sl);
    SQ_CLK(16);
    if (c.n_small > 0) {  // (uniform) two block classes: the voxels the batch took past a small block's capacity move to full blocks
        if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
        const int nm = st->mig_n < c.n_small ? st->mig_n : c.n_small, nbm = (nm + BS - 1) / BS;
        sl.nb = nbm;
        for (sl.b = sq_grab(ctr + 2); sl.b < nbm; sl.b = sq_grab(ctr + 2)) d_map_migrate(c, sl);
        if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;  // (the rebuild below and the next scan's list see the moved voxels; mig_n is reset behind it)
        if (wg == 0 && threadIdx.x == 0) st->mig_n = 0;
    }
    const int nbp = (blk_virtual_count(c, st) + BS - 1) / BS;
    sl.nb = nbp;
    if (rebuild) {  // drop the tombstones: empty table, re-enter the live voxels
        if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
        unsigned long long* tw = (unsigned long long*)c.tab;
        const size_t nwords = ((size_t)c.tmask + 1) * (sizeof(TabEnt) / 8);
        for (size_t i = (size_t)wg * BS + threadIdx.x; i < nwords; i += (size_t)nw * BS) tw[i] = ~0ull;
        if (wg == 0 && threadIdx.x == 0) st->tab_used = 0u;
        if (!team_sync(te, (unsigned)nw, target)) return SEQ_FAIL;
        for (sl.b = sq_grab(ctr + 4); sl.b < nbp; sl.b = sq_grab(ctr + 4)) d_map_rebuild(c, sl);
    }
    return target;
}
template <int PC, int GC>
__device__ __noinline__ void sq_gauss_newton(const SeqCtx* a, int s, int k, int G, int wg) {
    const Ctx c = load_seq_ctx(a, s, k);
    gn8_body<PC, GC>(c, 0, G, wg);
}
// the filter step that follows scan k: update with its pose, predict through the IMU samples before scan k + 1
// ... and, while the others are still updating the map, K0 of scan k + 1: it needs the new pose only (the lockstep driver
// runs it beside the map update too), and whichever team takes scan k + 1 finds it done
__device__ __noinline__ void sq_filter(const SeqCtx* a, int s, int k) {
    const SeqCtx* my = a + s;
    const int i0 = my->imu_end[k], i1 = (k + 1 < my->n_scans) ? my->imu_end[k + 1] : i0;
    d_ekf_step(my->ekf, my->imu, i0, i1, my->c.traj + 16 * (size_t)k, nullptr, my->res_poses + 16 * (size_t)k,
               my->res_t + k, my->rows + 8 * (size_t)k, 1);
    if (k + 1 < my->n_scans) {
        __syncthreads();
        const Ctx c = load_seq_ctx(a, s, k + 1);
        d_scan_prologue(c, true);
        __syncthreads();
        if (threadIdx.x == 0) c.st->pro_next = k + 2;
    }
}

struct SeqRun {
    int S, k0, k1, with_ekf, rebuild_every, G;  // G: workgroups per team (run-time value; the GC instances fix it at compile time)
    int um_alt;          // test hook: the map update runs its SEQ_UM_ALT instance (ptl_batch_debug_set_map_points_per_thread)
    int force_agent;     // PTL_TEAM_SYNC=agent: every team barrier keeps the agent-scope release (no XCD-local shortcut)
    int dbg_dead_block;  // test hook (ptl_batch_debug_stall_block): this workgroup of the grid leaves right before the job barrier of its
    int dbg_dead_round;  // dbg_dead_round-th job (0 = the first); -1 = none
    int balance_margin;  // >= 0: a team helps another XCD as soon as that XCD's least-advanced free sequence is this many scans behind its own
                         // XCD's (sched_most_behind); -1: only when its own XCD has nothing for it (rounds 3-5).  PTL_SCHED_MARGIN, default 1
};
// Why a team left the launch other than "all scans done".  Sticky bits in one word of device memory per batch (never cleared by a
// launch; ptl_batch_wait reports them): no exit of the free-running kernel is silent.
#define SEQ_EXIT_XCC_BARRIER 1u   /* a teammate never reached the head-of-launch barrier */
#define SEQ_EXIT_JOB_BARRIER 2u   /* a teammate never reached the job barrier */
#define SEQ_EXIT_IDLE 4u          /* TEAM_IDLE_ROUNDS rounds without work while sequences were still pending */
#define SEQ_EXIT_SEQUENCE 8u      /* a team gave up on a sequence (its err_flags say why) */
#define SEQ_EXIT_INCOMPLETE 16u   /* k_sched_check: a sequence did not reach the last scan of the launch */

// scheduler state of one XCD's sequences (slot q <-> sequence x + 8 q): the next scan of each and whether a team is on it
#define SEQ_SLOTS 32
struct SeqSched { int next_scan[SEQ_SLOTS]; int busy[SEQ_SLOTS]; };  // 256 B
__global__ void k_sched_init(SeqSched* sc, int S, int k0, int k1) {
    const int x = threadIdx.x / SEQ_SLOTS, q = threadIdx.x % SEQ_SLOTS;
    if (x >= 8) return;
    sc[x].next_scan[q] = (x + 8 * q < S) ? k0 : k1;
    sc[x].busy[q] = 0;
}
// After every launch of kx_seq_run: did every sequence get to scan k1?  A slot that did not (its team left while holding it,
// or nobody was left to run it) gets ERR_GN_TIMEOUT on its sequence and SEQ_EXIT_INCOMPLETE on the batch: the host must not
// report scans that never ran as done.  (A sequence taken off the schedule by sched_abandon stands at k1 and carries its
// own error flag already.)
__global__ void k_sched_check(const SeqCtx* a, const SeqSched* sc, int S, int k1, unsigned* status) {
    const int x = threadIdx.x / SEQ_SLOTS, q = threadIdx.x % SEQ_SLOTS;
    if (x >= 8 || x + 8 * q >= S) return;
    if (sc[x].next_scan[q] < k1 || sc[x].busy[q] != 0) {
        atomicOr(&a[x + 8 * q].c.st->err_flags, ERR_GN_TIMEOUT);
        atomicOr(status, SEQ_EXIT_INCOMPLETE);
    }
}
// The team leader's choice: the sequence of this XCD with the fewest scans done that nobody is working on.  Returns its slot,
// SCHED_DONE when every sequence of the XCD has reached k1, SCHED_RETRY when all that is left is in other teams' hands right
// now (after a short bounded wait: the caller goes round its team barrier and asks again, so no wait in the kernel is long
// enough to outlast a teammate's poll budget).
#define SCHED_DONE (-1)
#define SCHED_RETRY (-2)
#define SCHED_POLLS 256
// (the FIRST WAVEFRONT of the leader workgroup asks: lane q reads slot q - one memory round trip for the whole table instead of up
// to two per slot one after the other by a single lane, at agent scope each; the choice is the serial loop's: fewest scans done,
// lowest slot on a tie.  Lane 0 takes the slot; every lane returns the same answer.)
__device__ __forceinline__ int sched_pick(SeqSched* sc, int nslots, int k1, int* scan_out, unsigned max_polls = SCHED_POLLS) {
    const int lane = (int)(threadIdx.x & 63u);
    for (unsigned polls = 0; polls < max_polls; ++polls) {
        int k = k1, busy = 1;
        if (lane < nslots) {
            k = __hip_atomic_load(&sc->next_scan[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            busy = __hip_atomic_load(&sc->busy[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool pend = k < k1;
        if (__ballot(pend) == 0ull) return SCHED_DONE;
        unsigned key = (pend && busy == 0) ? (((unsigned)k << 6) | (unsigned)lane) : 0xFFFFFFFFu;
        for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
        if (key != 0xFFFFFFFFu) {
            const int best = (int)(key & 63u);
            int res = -3, kk = 0;  // -3: the slot went to somebody else / finished meanwhile - look again
            if (lane == 0 && atomicCAS(&sc->busy[best], 0, 1) == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // what the team that ran the previous scan wrote
                kk = __hip_atomic_load(&sc->next_scan[best], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (kk < k1) res = best;
                else __hip_atomic_store(&sc->busy[best], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (finished meanwhile)
            }
            res = __shfl(res, 0); kk = __shfl(kk, 0);
            if (res >= 0) { *scan_out = kk; return res; }
            continue;  // (lost the swap, or the slot was done: the table has changed - look again straight away)
        }
        __builtin_amdgcn_s_sleep(16);
    }
    return SCHED_RETRY;
}
// Where is the sequence that is furthest behind?  The leader's first wavefront reads all eight tables in one round trip (8 lanes per XCD, 4
// slots each) and returns, in every lane: *k_home = the fewest scans done among the FREE sequences of XCD x (INT_MAX: none), and the XCD
// other than x whose least-advanced free sequence has the fewest scans done, with that count in *k_other (-1 / INT_MAX: none).
// Round 6: the XCDs' sequences differ in cost by several per cent, and a team that only ever looks abroad when its own XCD has NOTHING left
// meets the slow XCD's sequences at the end of the run - a handful of chains with a few scans each that cannot be split, the other teams idle
// (the driver's 20-step form lost ~ 6 % there).  Helping as soon as somebody is a scan behind keeps all sequences of the device within a scan
// of each other: the run ends everywhere at once.  Nothing is lost in the caches - 30 sequences time-share one XCD's 4 MB L2, a sequence
// finds nothing of its previous scan there either way - and every hand-over is agent-scope already.
__device__ __forceinline__ int sched_most_behind(const SeqSched* sched, int S, int k1, int x, int* k_home, int* k_other) {
    const int lane = (int)(threadIdx.x & 63u), xo = lane >> 3;
    const int nso = xo < S ? (S - xo + 7) >> 3 : 0;
    int kk[4], bz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int slot = (lane & 7) * 4 + j;
        kk[j] = k1; bz[j] = 1;
        if (slot < nso) {
            kk[j] = __hip_atomic_load(&sched[xo].next_scan[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bz[j] = __hip_atomic_load(&sched[xo].busy[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    int kmin = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (kk[j] < k1 && bz[j] == 0) kmin = min(kmin, kk[j]);
    for (int o = 4; o > 0; o >>= 1) kmin = min(kmin, __shfl_xor(kmin, o));  // (the 8 lanes of an XCD)
    *k_home = __shfl(kmin, 8 * x);
    unsigned key = (xo != x && kmin != 0x7FFFFFFF) ? (((unsigned)kmin << 3) | (unsigned)xo) : 0xFFFFFFFFu;
    for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
    *k_other = key == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(key >> 3);
    return key == 0xFFFFFFFFu ? -1 : (int)(key & 7u);
}
__device__ __forceinline__ void sched_release(SeqSched* sc, int q, int next_scan) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (always the full release: the next scan may run on any team - any XCD as far as this protocol knows)
    __hip_atomic_store(&sc->next_scan[q], next_scan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sc->busy[q], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// a team that gives up on its sequence (abort word, poll budget) takes it off the schedule: nobody waits for its remaining scans
__device__ __forceinline__ void sched_abandon(SeqSched* sc, int q, int k1) {
    __hip_atomic_store(&sc->next_scan[q], k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ... and lets go of it: k_sched_check then finds it at k1 and free, and leaves the diagnosis to the flag the team raised (a slot that is
    // still busy after the launch belongs to a team that died holding it - only that gets the time-out flag)
    __hip_atomic_store(&sc->busy[q], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// bar: per team 64 words - [0] barrier of the whole team, [32] of the map update, [40], [41] the leader's job (sequence,
// scan), [42] the XCC ids its workgroups reported (one bit each), [43] the team's own abort word (the job barrier),
// [44..48] block counters of the map update's phases
#define SEQ_MAX_TEAMS 512
#define JOB_DONE 0xFFFFFFFFu
#define JOB_RETRY 0xFFFFFFFEu
#define TEAM_IDLE_ROUNDS (1u << 20)  /* rounds of SCHED_POLLS polls a team keeps asking for work before it leaves quietly */
template <int PC, int GC>
// (make T256=1: 256-thread workgroups, two per CU - the kernel then has to be held to 2 wavefronts per SIMD explicitly: a launch
// bound of 256 threads alone would let the register allocator plan for one)
__global__ __launch_bounds__(GN8_MAX_THREADS) SEQ_OCC void kx_seq_run(const SeqCtx* a, SeqRun r, SeqSched* sched, unsigned* bar, unsigned* status) {
    const int x = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3), J = (int)(gridDim.x >> 3);
    const int G = GC > 0 ? GC : r.G;
    const int t = j / G, wg = j % G;
    const int nslots = (r.S - x + 7) >> 3;  // sequences of this XCD: x, x + 8, ...
    // no sequence on this XCD / a workgroup beyond the last whole team / a team that can never be needed (more teams than sequences)
    if (x >= r.S || (t + 1) * G > J || t >= nslots) return;
    unsigned* tb = bar + (size_t)(x + 8 * t) * 64;
    SeqSched* sc = sched + x;
    // with a filter, the team's last workgroup is the filter workgroup: after the Gauss-Newton loop it steps the filter
    // while the others update the map (and is back, waiting, long before they are done); K0-K4 are everybody's.  A team of
    // one workgroup does both, one after the other.
    const bool solo = G == 1;
    const bool fwg = r.with_ekf && !solo && wg == G - 1;
    unsigned t_all = 0u, t_work = 0u;
    const bool lead = wg == 0 && threadIdx.x == 0, clkf = (fwg || (solo && r.with_ekf)) && threadIdx.x == 0;
    int* team_abort = (int*)&tb[43];
    // do the team's workgroups share one L2?  Everybody reports its XCC_ID, one full (agent-scope) barrier, everybody reads the
    // mask: one bit set = one XCD = the barriers from here on need no L2 write-back (team_sync).  Any other placement - the
    // dispatch order is not specified - keeps the agent-scope protocol.
    bool local = solo;
    if (!solo) {
        if (threadIdx.x == 0) __hip_atomic_fetch_or(&tb[42], 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!team_sync(tb, (unsigned)G, t_all, team_abort, nullptr, false)) {
            if (threadIdx.x == 0) atomicOr(status, SEQ_EXIT_XCC_BARRIER);
            return;
        }
        const unsigned m = __hip_atomic_load(&tb[42], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // One bit = every workgroup of the team reported the same XCC_ID.  Hardware assumptions behind the shortcut (gfx950): the
        // workgroups of one XCD share ONE L2, a store that has been counted out of vmcnt is visible to every later load that
        // goes to that L2, and a workgroup stays on the XCD it was dispatched to for the whole launch.  PTL_TEAM_SYNC=agent
        // (SeqRun::force_agent) keeps the agent-scope protocol regardless; tests compare the two bit for bit.
        local = !r.force_agent && (m & (m - 1u)) == 0u;
    }
    int q_mine = -1, k_mine = 0, x_mine = x;  // the leader's job: slot, scan, and the XCD whose scheduler holds the sequence
    unsigned idle = 0u;
    int round = 0;
    const unsigned my_xcc = xcc_id();
    for (;;) {
        if (wg == 0 && threadIdx.x < 64u) {  // the leader workgroup's first wavefront: hand the finished scan back (lane 0), take the next job
            if (lead && q_mine >= 0) sched_release(sched + x_mine, q_mine, k_mine + 1);
            int k = 0, xs = x;
            int q = -1;
            if (r.balance_margin >= 0) {  // is another XCD's least-advanced free sequence behind ours?  Then that one first.
                int kh, ko;
                const int xb = sched_most_behind(sched, r.S, r.k1, x, &kh, &ko);
                if (xb >= 0 && kh != 0x7FFFFFFF && ko + r.balance_margin <= kh) {
                    const int qo = sched_pick(sched + xb, (r.S - xb + 7) >> 3, r.k1, &k, 1u);
                    if (qo >= 0) { q = qo; xs = xb; }
                }
            }
            if (q < 0) q = sched_pick(sc, nslots, r.k1, &k, 16u);  // (a short wait at home, then a look at the neighbours, then round the team barrier and again)
            if (q < 0) {
                // Nothing to do at home: is a sequence of another XCD waiting for a team?  (24 sequences per XCD differ enough in cost
                // that the XCDs finish 5-10 ms apart in a 150 ms run, and the L2 they would share is far too small for any of this
                // to live there anyway.)  The hand-over of a sequence between teams is agent-scope either way (sched_release / pick).
                bool all_done = q == SCHED_DONE;
                for (int d = 1; d < 8 && q < 0; ++d) {
                    const int xo = (x + d) & 7;
                    if (xo >= r.S) continue;
                    const int qo = sched_pick(sched + xo, (r.S - xo + 7) >> 3, r.k1, &k, 1u);
                    if (qo >= 0) { q = qo; xs = xo; }
                    else if (qo == SCHED_RETRY) all_done = false;
                }
                if (q < 0) q = all_done ? SCHED_DONE : SCHED_RETRY;
            }
            q_mine = q; k_mine = k; x_mine = xs;
            if (lead) {
                __hip_atomic_store(&tb[40], q == SCHED_DONE ? JOB_DONE : q == SCHED_RETRY ? JOB_RETRY : (unsigned)(xs + 8 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&tb[41], (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int i = 0; i < 5; ++i) __hip_atomic_store(&tb[44 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the map update's block counters
            }
        }
        if ((int)blockIdx.x == r.dbg_dead_block && round == r.dbg_dead_round) return;  // test hook: a workgroup that dies between two scans
        ++round;
        // the job barrier.  An expiry (a teammate that never came) ends the team - visibly: the batch's status word says so, and
        // the sequence the leader has just taken (busy, scans left) gets ERR_GN_TIMEOUT and goes off the schedule, so that the
        // other teams neither wait for it nor find it stuck (this exit used to be silent)
        if (!team_sync(tb, (unsigned)G, t_all, team_abort, nullptr, local)) {
            if (threadIdx.x == 0) atomicOr(status, SEQ_EXIT_JOB_BARRIER);
            if (lead && q_mine >= 0) {
                gn_raise_abort(a[x_mine + 8 * q_mine].c.st);
                sched_abandon(sched + x_mine, q_mine, r.k1);
            }
            return;
        }
        const unsigned job = __hip_atomic_load(&tb[40], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (job == JOB_DONE) return;
        if (job == JOB_RETRY) {
            if (++idle > TEAM_IDLE_ROUNDS) {  // sequences are pending in other teams' hands and never come back
                if (threadIdx.x == 0) atomicOr(status, SEQ_EXIT_IDLE);
                return;
            }
            continue;
        }
        idle = 0u;
        const int s = (int)job, k = (int)__hip_atomic_load(&tb[41], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int q = s >> 3;
        const SeqCtx* my = a + s;
        DevState* st = my->c.st;
        const TeamEnv te = {tb, &st->gn_abort, st, local};
        // (every exit below takes the sequence off the schedule first: the other teams must not wait for scans nobody will run)
#define SEQ_LEAVE do { if (threadIdx.x == 0) { sched_abandon(sched + (s & 7), q, r.k1); atomicOr(status, SEQ_EXIT_SEQUENCE); } return; } while (0)
        if (lead) {
            // where the sequence's scans run: [0] scans taken by a team of another XCD than the sequence's home, [1] scans that run
            // on another XCD than the sequence's previous scan did (what the agent-scope hand-over of sched_release / sched_pick
            // has to get right: the previous scan's data sits in another L2), [2] the XCC this scan runs on + 1
            if ((s & 7) != x) st->sched_cnt[0] += 1ull;
            if (st->sched_cnt[2] != 0ull && st->sched_cnt[2] != (unsigned long long)my_xcc + 1ull) st->sched_cnt[1] += 1ull;
            st->sched_cnt[2] = (unsigned long long)my_xcc + 1ull;
        }
        const long long c0 = (long long)wall_clock64();
        t_all = sq_prepare(a, s, k, wg, G, t_all, tb, local);  // (all G workgroups: the filter workgroup has nothing else to do here)
        if (t_all == SEQ_FAIL) SEQ_LEAVE;
        const long long c1 = (long long)wall_clock64();
        if (!team_sync(te, (unsigned)G, t_all)) SEQ_LEAVE;  // source ready (map and filter: complete since the scan before)
        const long long c2 = (long long)wall_clock64();
        sq_gauss_newton<PC, GC>(a, s, k, G, wg);
        if (gn_abort_seen(&st->gn_abort)) SEQ_LEAVE;
        const long long c3 = (long long)wall_clock64();
        if (!team_sync(te, (unsigned)G, t_all)) SEQ_LEAVE;  // new pose, trajectory row
        const long long c4 = (long long)wall_clock64();
        long long c5f = c4, c5g = c4;
        if (fwg) { sq_filter(a, s, k); c5g = (long long)wall_clock64(); }  // ... and then it joins the map update
        const int rebuild = (r.rebuild_every > 0 && ((k + 1) % r.rebuild_every) == 0) ? 1 : 0;
        t_work = r.um_alt ? sq_map_update<SEQ_UM_ALT>(a, s, k, wg, G, t_work, rebuild, tb + 32, tb + 44, local)
                          : sq_map_update<SEQ_UM>(a, s, k, wg, G, t_work, rebuild, tb + 32, tb + 44, local);
        if (t_work == SEQ_FAIL) SEQ_LEAVE;
        if (solo && r.with_ekf) { c5f = (long long)wall_clock64(); sq_filter(a, s, k); c5g = (long long)wall_clock64(); }
        const long long c5 = (long long)wall_clock64();
        if (lead) {
            st->seq_clk[0] += c1 - c0; st->seq_clk[1] += c2 - c1; st->seq_clk[2] += c3 - c2; st->seq_clk[3] += c4 - c3; st->seq_clk[4] += c5 - c4;
            st->seq_clk[6] += 1;
        }
        if (clkf) st->seq_clk[5] += c5g - c5f;
        if (!team_sync(te, (unsigned)G, t_all)) SEQ_LEAVE;  // the scan is complete: the sequence may go to another team
#undef SEQ_LEAVE
    }
}
