#!/usr/bin/env python3
"""bench.py -- lidar scans/sec (ICP+EKF) on synthetic 128x1024 sweeps, MI355X.

A "step" is one pass of the hot path (reference cli/ekf_bench.py:493-563: IMU predicts -> scan
registration against the local map -> EKF pose update) over one batch of synthetic input = one sweep
of each sequence this rank owns.  Inputs (all sweeps + IMU) are resident in HBM before the timed
region.  Warm-up steps are the first W sweeps of the same sequence (they also build the local map, so
the timed region is steady state).  One rank per GPU; ranks own independent sequences (no data-path
collective); the only collective is the final trajectory all-gather over RCCL.

    python bench.py --gpus 1 --steps 200 --warmup 20
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before anything initialises HIP: see ptudes-lab_amd/_lib.py
os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # cpu_baseline's OpenMP pass: idle threads must not burn the cgroup quota

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GN_EVENT_EVERY = 8
GATHER_TIMEOUT_S = 120
CPU_MIN_SWEEPS = 300  # sweeps offered to the CPU baseline (it stops at its time budget)
HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


def icp_bytes(stats):
    """Algorithmic bytes of one Gauss-Newton launch (B_icp): per iteration the source is
    read once (12 B/pt), 27 hash slots are probed per point (16 B each) and every candidate map point is
    read (12 B each): sum_i [12 N_s + 27*16 N_s + 12 C_i]."""
    return stats["iterations"] * (12 + 27 * 16) * stats["n_src"] + 12 * stats["sum_cand"]


def scan_bytes(s, n_raw):
    """B_scan of SURVEY.md 8(d) for one sweep"""
    nv, nd, ns, mv = s["n_valid"], s["n_down"], s["n_src"], s["map_voxels"]
    b_pre = 12 * n_raw + 12 * nv
    b_ds = (12 + 16) * nv + 12 * nd + (12 + 16) * nd + 12 * ns
    b_map = (12 + 16 + 12) * nd + (16 + 12) * mv
    return b_pre + b_ds + icp_bytes(s) + b_map


def _cpu_pass(seq, n_total, use_imu_prediction, budget_s, with_ekf=True, oracle_over=None):
    from oracle import cpu as orc
    icp = orc.ICP(max_range=seq.max_range, min_range=seq.min_range, **(oracle_over or {}))
    ekf = orc.EKF()
    t01 = seq.column_times()
    kiss, res = [], []
    spent, done = 0.0, 0
    for k in range(n_total):
        x = seq.scan(k).astype(np.float64)  # rendering is not part of the timed work
        a, b = seq.imu_range_for_scan(k)
        t0 = time.perf_counter()
        if with_ekf:
            for i in range(a, b):
                ekf.process_imu(seq.imu[i, 1:4], seq.imu[i, 4:7], seq.imu[i, 0])
        guess = ekf.pose_mat() if use_imu_prediction else None
        pose = icp.register_frame(x, t01, guess)
        if with_ekf:
            ekf.process_pose(pose)
        spent += time.perf_counter() - t0
        kiss.append(pose)
        res.append(ekf.pose_mat() if with_ekf else pose)
        done += 1
        if spent >= budget_s:
            break
    return done, spent, np.array(kiss), np.array(res)


def cpu_baseline(seq, n_total, use_imu_prediction, budget_s=20.0, with_ekf=True, oracle_over=None):
    """The CPU oracle (kind "port": our C restatement of the reference path) timed on this host on the first
    sweeps of the same sequence: one sequential pass (~1/3 of the budget; also the parity check of the GPU
    trajectory) and one pass with the loops kiss-icp runs under TBB spread over every usable core (oracle.h
    orc_set_threads).  `value` is the faster of the two."""
    from oracle import cpu as orc
    from ptudes_lab_amd.synth import usable_cores
    cores = usable_cores()
    orc.set_threads(1)
    d1, s1, kiss, res = _cpu_pass(seq, n_total, use_imu_prediction, budget_s / 3.0, with_ekf, oracle_over)
    v1, vm, dm, sm = d1 / s1, 0.0, 0, 0.0
    if cores > 1:
        orc.set_threads(cores)
        dm, sm, kiss_m, res_m = _cpu_pass(seq, n_total, use_imu_prediction, budget_s * 2.0 / 3.0, with_ekf, oracle_over)
        orc.set_threads(1)
        vm = dm / sm
        if dm > d1:
            kiss, res = kiss_m, res_m
    multi = vm > v1
    return dict(value=max(v1, vm), unit="scans/s", cores=cores if multi else 1, kind="port",
                single_thread_value=v1, multi_thread_value=vm if cores > 1 else None,
                sample=f"first {dm if multi else d1} sweeps of sequence seed {seq.seed} (cold start), "
                       f"{sm if multi else s1:.1f} s wall on {cores if multi else 1} threads "
                       f"(plus {d1} sweeps / {s1:.1f} s single-thread), oracle/liboracle.so; host has "
                       f"{os.cpu_count()} logical cores, {cores} usable under the cgroup quota"), kiss, res


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, worker=None, timeout_s=3600.0, out=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's JSON
    line.  Runs before anything in this process has imported torch or touched HIP (a process that has initialised the GPU
    must not be replaced or forked); the children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT exactly as
    `python -m torch.distributed.run` would set them.  A rank that dies takes the others down (exact PIDs) instead of
    leaving them in a barrier.  Returns the worst exit code.  `worker` replaces [python, bench.py] (CPU test stub)."""
    import subprocess
    import threading
    out = out or sys.stdout
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, text=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    t_end = time.monotonic() + timeout_s
    rcs = [None] * n
    failed = False
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0):
                    failed = True
        if (failed or time.monotonic() > t_end) and any(rc is None for rc in rcs):
            time.sleep(5.0 if failed else 0.0)  # the others usually notice by themselves (gloo peer reset)
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()
                    rcs[r] = p.wait() or 9
                else:
                    rcs[r] = p.returncode
            if not failed:
                print(f"bench.py: ranks still running after {timeout_s:.0f} s, killed", file=sys.stderr)
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    for ln in lines:  # stdout carries the result line(s) only; anything else a library printed there goes to stderr
        (out if ln.lstrip().startswith("{") else sys.stderr).write(ln)
    out.flush()
    worst = 0
    for rc in rcs:
        if rc:
            worst = rc if rc > 0 else 128 - rc
    if worst:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
    return worst


def pmc_traffic_for(workload_key):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC pass (profiles/) whose recorded workload
    is THIS run's workload, or None: a counter value belongs to the run it was collected on."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload_key") == workload_key and d.get("traffic_bytes_per_launch") is not None:
            best = (d["traffic_bytes_per_launch"], os.path.basename(f))
    return best


def workload_key(args, S):
    """what a PMC pass has to have been collected on to speak for this run"""
    return (f"{args.rows}x{args.cols}_r{args.min_range:g}-{args.max_range:g}_v{(args.voxel_size or args.max_range / 100):g}_"
            f"seed{args.seed_base}_S{S}_W{args.warmup}_K{args.steps}_{'cv' if args.const_velocity else 'imu'}"
            f"{'' if not args.icp_only else '_icponly'}{'' if not args.gn_lanes else '_L%d' % args.gn_lanes}"
            f"{'_free' if free_running(args, S) else ''}")


def free_running(args, S):
    """does this run use the free-running batch driver (one persistent launch, every sequence at its own pace)?"""
    return S > 1 and not args.lockstep and args.gn_lanes in (0, 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seqs-per-gpu", type=int, default=48,
                    help="independent sequences per GPU (SURVEY.md 8(e), second level), batched runner.  The sequences s = x (mod 8) "
                         "live on XCD x; up to 8 / 16 sequences: one / two teams of workgroups per XCD, one sequence each; more "
                         "(default 48, up to 64): four teams per XCD that take the sequences' scans as they come free, so the "
                         "sequences advance evenly (6.0k / 9.9k / 13.0k / 14.8k scans/s for 8 / 16 / 32 / 48).  One Gauss-Newton "
                         "loop alone leaves an XCD latency-bound, more of them fill the gaps.  1 = the single-sequence latency "
                         "pipeline (one sequence over the whole chip)")
    ap.add_argument("--seed-base", type=int, default=1000, help="sequence s of SURVEY.md 8(d) uses seed seed_base + s")
    ap.add_argument("--equal-work", action="store_true",
                    help="every rank registers its own copy of sequences seed_base .. seed_base + S - 1 (equal work per GPU) "
                         "instead of the SURVEY.md 8(e) sharding: rank r owns the sequences s with s %% world == r")
    ap.add_argument("--distinct-seeds", action="store_true", help="(the default now; kept for old command lines)")
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--max-range", type=float, default=70.0)
    ap.add_argument("--min-range", type=float, default=1.0)
    ap.add_argument("--const-velocity", action="store_true",
                    help="use the constant-velocity guess instead of --use-imu-prediction (reference default)")
    ap.add_argument("--icp-only", action="store_true", help="no IMU / EKF at all (BASELINE config 2): constant-velocity guess")
    ap.add_argument("--gn-wgs", type=int, default=0, help="workgroups of the persistent GN kernel (0 = library default)")
    ap.add_argument("--gn-threads", type=int, default=0)
    ap.add_argument("--gn-lanes", type=int, default=0, help="lanes per source point of the GN kernel: 32 | 8 (0 = the runner's default)")
    ap.add_argument("--lockstep", action="store_true",
                    help="batched runs: one launch per stage for all sequences (a step waits for its slowest sequence) instead of "
                         "the free-running kernel")
    ap.add_argument("--scans-per-launch", type=int, default=0, help="free-running driver: scans per persistent launch (0 = library default)")
    ap.add_argument("--voxel-size", type=float, default=0.0, help="override the map voxel size (default max_range/100)")
    ap.add_argument("--map-blocks", type=int, default=0, help="voxel-block pool capacity")
    ap.add_argument("--map-table", type=int, default=0, help="map hash-table slots (power of two)")
    ap.add_argument("--workload-name", type=str, default="")
    ap.add_argument("--device", type=int, default=-1, help="GPU index for this rank (default LOCAL_RANK %% visible devices)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-sequence", action="store_true",
                    help="skip the extra figure of the default line: sequence 0 alone through the single-sequence latency pipeline")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--dump-traj", type=str, default="", help="rank 0 writes the gathered (T, 8) NC-GT rows of every sequence to this .npz")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's plain `python bench.py --gpus N`: this process becomes the launcher (it has not touched torch or HIP)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launched under a launcher with a different rank count)")
    rank = int(os.environ.get("RANK", "0"))
    # stdout is for the one JSON line: native libraries (gloo, RCCL, the HIP runtime) print their own notes to fd 1, so
    # from here on fd 1 is stderr and the line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if args.seqs_per_gpu < 1 or args.seqs_per_gpu > 64:
        sys.exit("bench.py: --seqs-per-gpu must be in [1, 64]")
    multi = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)  # a rank of a multi-process run
    dist = None
    torch = None
    ctl = None
    if multi:
        # torch FIRST: it bundles its own HIP runtime, and libptudes_mi.so (linked against the system's) has to find that
        # one already loaded - the other order puts two runtimes into the process and torch then sees no GPU
        import datetime
        import torch
        import torch.distributed as dist
        n_dev = torch.cuda.device_count()  # (does not initialise a device)
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import _lib, core, synth
    if not multi:
        n_dev = _lib.lib().ptl_device_count()  # (hipGetDeviceCount: does not initialise a device)
    if n_dev < 1:
        sys.exit("bench.py: no HIP device - the HIP path is the only path")
    local_rank = (int(os.environ.get("LOCAL_RANK", "0")) % n_dev) if args.device < 0 else args.device
    shared_device = world > n_dev  # several ranks on one GPU (1-GPU box): RCCL refuses duplicate devices, the gather goes over gloo
    if multi:
        # RCCL carries the one collective of the path, the trajectory gather after the run.  Its communicator is brought
        # up there and not before: a live RCCL communicator in the process stretches the cross-stream hand-overs of the
        # scan pipeline from 63 to 110 us per scan (2700 -> 2400 scans/s, measured with one rank).  Barriers and the
        # max-over-ranks clock go through a host-side gloo group.
        if shared_device:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
            ctl = dist.group.WORLD
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", timeout=datetime.timedelta(seconds=600))
            ctl = dist.new_group(backend="gloo")

    K, W, S = args.steps, args.warmup, args.seqs_per_gpu
    n_total = W + K
    with_ekf = not args.icp_only
    use_imu = with_ekf and not args.const_velocity
    pps = args.rows * args.cols
    icp_over = dict(scan_cols=args.cols)
    if args.gn_wgs: icp_over["gn_workgroups"] = args.gn_wgs
    if args.gn_threads: icp_over["gn_threads"] = args.gn_threads
    if args.gn_lanes: icp_over["gn_lanes_per_point"] = args.gn_lanes
    if args.voxel_size: icp_over["voxel_size"] = args.voxel_size
    if args.map_blocks: icp_over["map_block_capacity"] = args.map_blocks
    if args.map_table: icp_over["map_table_capacity"] = args.map_table
    # rank r owns the sequences s with s % world == r, sequence s has seed seed_base + s - independent
    # sequences, nothing shared, no data-path collective.  They differ by up to 40 % in GN iterations per scan, so the
    # max-over-ranks clock of such a run is the slowest sequence's; --equal-work gives every rank a private copy of the
    # same sequence(s) instead (the pure scaling measurement).
    if world > 1:  # the ranks of one host share its cores: no oversubscription while every rank renders its sweeps
        synth.set_threads(max(2, min(16, synth.usable_cores() // world)))
    seq_ids = [j if args.equal_work else rank + world * j for j in range(S)]
    # the CPU baseline wants a sample of its own size (>= CPU_MIN_SWEEPS sweeps within its time budget) however short the
    # timed GPU run is: sequence 0 of rank 0 is generated that long, the GPU registers its first n_total sweeps
    n_cpu = max(n_total, CPU_MIN_SWEEPS) if (world == 1 and not args.no_cpu_baseline) else n_total
    seqs = [synth.make_sequence(seed=args.seed_base + s, n_scans=(n_cpu if j == 0 else n_total), H=args.rows, W=args.cols,
                                min_range=args.min_range, max_range=args.max_range) for j, s in enumerate(seq_ids)]
    n_imu = seqs[0].imu_range_for_scan(n_total - 1)[1] if with_ekf else 0
    # S == 1: the single-sequence runner (its Gauss-Newton kernel caches hash probes across iterations);
    # S > 1: all sequences of this rank advance in lockstep in one batched runner (one launch per stage for all)
    class _One:
        def __init__(self):
            self.r = core.SeqRunner(n_total, pps, n_imu, max_range=args.max_range, min_range=args.min_range,
                                    use_imu_prediction=use_imu, with_ekf=with_ekf, device_id=local_rank, **icp_over)
        def upload_scan(self, j, k, x): self.r.upload_scan(k, x)
        def upload_imu(self, j, rows, ends): self.r.upload_imu(rows, ends)
        def run(self, n): self.r.run(n)
        def enqueue(self, n): self.r.enqueue(n)
        def wait(self): self.r.wait()
        def results(self, j): return self.r.results()
        def profile(self, **kw): return self.r.profile(**kw)
        def copy_traj(self, j, ptr, n): return self.r.copy_traj(ptr, n)
    free = free_running(args, S)
    runner = _One() if S == 1 else core.BatchRunner(S, n_total, pps, n_imu, max_range=args.max_range,
                                                    min_range=args.min_range, use_imu_prediction=use_imu,
                                                    with_ekf=with_ekf, device_id=local_rank, free_running=free,
                                                    scans_per_launch=args.scans_per_launch, **icp_over)
    for j, sq in enumerate(seqs):
        for k in range(n_total):
            runner.upload_scan(j, k, sq.scan(k))
        runner.upload_imu(j, sq.imu[:n_imu] if with_ekf else np.zeros((0, 7)),
                          [sq.imu_range_for_scan(k)[1] if with_ekf else 0 for k in range(n_total)])

    def barrier():
        if dist is not None:
            dist.barrier(group=ctl)

    def sync():
        if torch is not None and not shared_device:
            torch.cuda.synchronize()
        core.device_sync(local_rank)

    # warm-up: cold start + the first W sweeps (untimed)
    runner.run(W)
    # HIP events around every 8th launch of the dominant kernel: two event records per scan cost ~18 us (4 %) of
    # command-processor time on the critical path
    ev_every = GN_EVENT_EVERY if S == 1 else 1  # (the batched runner times every launch)
    runner.profile(enable=ev_every, reset=True)
    barrier(); sync()
    t0 = time.perf_counter()
    runner.enqueue(K)
    runner.wait()
    sync()
    dt_own = time.perf_counter() - t0  # this rank's own K steps
    barrier()
    dt = time.perf_counter() - t0
    per_rank = [K * S / dt_own]
    if dist is not None:
        from ptudes_lab_amd import parallel
        dt = parallel.max_over_ranks(dt, dist, device="cpu", group=ctl)
        box = [None] * world
        dist.all_gather_object(box, K * S / dt_own, group=ctl)
        per_rank = [float(v) for v in box]

    # per-rank accounting
    outs = [runner.results(j) for j in range(S)]
    gn_ms, gn_n = runner.profile(enable=False)
    gn_bytes, gn_bytes_all, b_scan = 0.0, 0.0, 0.0
    iters = []
    for o in outs:
        for k, s in enumerate(o["stats"][W:]):
            if free:  # the launch carries the whole pipeline of every scan
                gn_bytes += scan_bytes(s, pps)
            elif (W + k) % ev_every == 0:  # the launches the HIP events bracketed
                gn_bytes += icp_bytes(s)
            gn_bytes_all += icp_bytes(s)
            b_scan += scan_bytes(s, pps)
            iters.append(s["iterations"])
    seq_clk = [runner.seq_clocks(j) for j in range(S)] if free else None
    n_timed = sum(len(o["stats"]) - W for o in outs)
    assert n_timed == K * S, (n_timed, K, S)

    # final trajectory gather: the only collective (T x 8 NC-GT rows per sequence, RCCL all-gather; gloo when several
    # ranks share one GPU, which RCCL refuses).  It runs after the timed region and brings the RCCL communicator up; a
    # failure or a stall there is reported in the line AND in the exit code.
    gathered, gather_err = None, None
    if dist is not None and with_ekf:
        import threading
        from ptudes_lab_amd import parallel
        box = {}

        def _gather():
            try:
                if shared_device:
                    rows = torch.zeros((S, n_total, 8), dtype=torch.float64)
                    counts = []
                    for j in range(S):
                        o = outs[j]
                        counts.append(len(o["res_t"]))
                        rows[j, : counts[-1]] = torch.from_numpy(parallel.poses_to_rows(o["res_t"], o["res_poses"]))
                    box["out"] = parallel.gather_trajectories(rows, counts, dist)
                else:
                    torch.cuda.set_device(local_rank)
                    rows = torch.zeros((S, n_total, 8), dtype=torch.float64, device="cuda")
                    counts = [runner.copy_traj(j, rows[j].data_ptr(), n_total) for j in range(S)]
                    box["out"] = parallel.gather_trajectories(rows, counts, dist)
            except Exception as e:  # noqa: BLE001
                box["err"] = repr(e)

        th = threading.Thread(target=_gather, daemon=True)
        th.start()
        th.join(timeout=GATHER_TIMEOUT_S)
        gathered = box.get("out")
        gather_err = "no answer after %d s" % GATHER_TIMEOUT_S if th.is_alive() else box.get("err")
        if parallel.max_over_ranks(0.0 if gather_err is None else 1.0, dist, device="cpu", group=ctl) > 0 and gather_err is None:
            gather_err = "failed on another rank"  # every rank takes the same exit below

    if rank == 0:
        from ptudes_lab_amd.ins.data import calc_ate
        o, sq = outs[0], seqs[0]
        est = o["res_poses"] if with_ekf else o["kiss_poses"]
        gt = sq.gt_poses(0.5)
        g0i = np.linalg.inv(gt[0])
        gt_rel = np.array([g0i @ g for g in gt])
        ate_r, ate_t = calc_ate(list(est), list(gt_rel[: len(est)]))
        rmse_gt = float(np.sqrt(np.mean(np.sum((est[:, :3, 3] - gt_rel[: len(est), :3, 3]) ** 2, 1))))
        avg_gn_s = (gn_ms / 1e3) / max(gn_n, 1)
        avg_gn_bytes = gn_bytes / max(gn_n, 1)  # one launch carries the GN loops (free-running: the whole scans) of all S sequences
        achieved = avg_gn_bytes / avg_gn_s if avg_gn_s > 0 else 0.0
        if args.equal_work:
            seeds_txt = f"{args.seed_base}..{args.seed_base + S - 1}" + (", a private copy on every rank (equal work per GPU)" if world > 1 else "")
        else:
            seeds_txt = f"{args.seed_base}..{args.seed_base + world * S - 1}" + (f" (rank r owns s % {world} == r)" if world > 1 else "")
        mode_txt = "ICP only, constant-velocity guess" if not with_ekf else \
            f"ICP + IMU-EKF ({'--use-imu-prediction' if use_imu else 'constant-velocity guess'})"
        wkey = workload_key(args, S)
        pmc = pmc_traffic_for(wkey) if world == 1 else None
        line = {
            "metric": f"lidar scans/sec (ICP+EKF) on {args.rows}x{args.cols} sweeps",
            "value": K * S * world / dt, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic {args.rows}x{args.cols} sweeps, random-walk SE(3) GT, {mode_txt}, "
                                   f"min/max range {args.min_range}/{args.max_range} m, voxel {(args.voxel_size or args.max_range / 100):.2f} m"
                                   + (f" [{args.workload_name}]" if args.workload_name else ""),
                       "workload_key": wkey,
                       "sequences_per_gpu": S, "sequence_seeds": seeds_txt,
                       "driver": "single sequence" if S == 1 else
                                 "free-running (one persistent launch, every sequence at its own pace)" if free else "lockstep (one launch per stage)",
                       "scans_per_sequence": n_total, "parallelism": f"{world} independent sequence shard(s), no data-path collective"},
            "per_rank_scans_per_s": {"values": per_rank, "min": min(per_rank), "mean": float(np.mean(per_rank)), "max": max(per_rank),
                                     "note": "each rank's own K steps / its own wall time; `value` uses the max-over-ranks clock"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": pmc[0] if pmc else None,
                         "traffic_source": pmc[1] if pmc else None,
                         "measured_frac": (pmc[0] / avg_gn_s / HBM_PEAK) if (pmc and avg_gn_s > 0) else None,
                         "kernel": ("k_gn_loop8" if args.gn_lanes == 8 else "k_gn_loop") if S == 1 else
                                   "kx_seq_run" if free else ("kx_gn_loop" if args.gn_lanes == 32 else "kx_gn_loop8"),
                         "avg_launch_us": 1e6 * avg_gn_s, "algorithmic_bytes_per_launch": avg_gn_bytes,
                         "launches": gn_n,
                         "timed_launches": (f"all {gn_n} persistent launches of the {K} steps (HIP events)" if free else
                                            f"every {ev_every}th of {K} (HIP events)"),
                         "note": ("the free-running kernel carries the WHOLE per-scan pipeline of every sequence: frac = B_scan of SURVEY "
                                  "8(d) (pre-processing + down-sampling + the Gauss-Newton iterations + map update) of all scans of a "
                                  "launch / launch time / peak - it can exceed 1: the answer cache settles most point-iterations "
                                  "without the 27 probes and the candidate reads the formula charges, and most of the rest hits the "
                                  "L2; what reaches HBM is `traffic`; " if free else
                                  "frac = ALGORITHMIC bytes (SURVEY 8(d): 27 probes x 16 B + every candidate x 12 B + the source, per "
                                  "iteration) / launch time / peak; ") +
                                 "measured_frac = PMC HBM bytes of the same workload / launch time / peak"},
            "whole_scan": {"algorithmic_bytes_per_scan": b_scan / max(n_timed, 1),
                           "achieved_GBps": (b_scan * world / dt) / 1e9 if world == 1 else None,
                           "gn_share_of_wall": ((gn_ms / 1e3) / dt) if free else (avg_gn_s * K) / dt,
                           "mean_gn_iterations": float(np.mean(iters))},
            "map": {"voxels_end": o["stats"][-1]["map_voxels"], "points_end": o["stats"][-1]["map_points"],
                    "n_src_mean": float(np.mean([s["n_src"] for s in o["stats"][W:]])),
                    "n_down_mean": float(np.mean([s["n_down"] for s in o["stats"][W:]]))},
            "sequence_phases_us_per_scan": None if seq_clk is None else {
                "columns": ["K0-K4", "wait", "gauss_newton", "wait", "map_update", "filter_workgroup"],
                "mean": [float(np.mean([c[i] for c in seq_clk])) for i in range(6)],
                "slowest_sequence_total": float(max(sum(c[:5]) for c in seq_clk)),
                "mean_sequence_total": float(np.mean([sum(c[:5]) for c in seq_clk])),
                "note": "100 MHz device clock of workgroup 0 of every sequence (since the cold start); the run lasts as long as its slowest sequence"},
            "accuracy": {"ate_vs_gt_ref_style_rot": ate_r, "ate_vs_gt_ref_style_trans_m2": ate_t,
                         "rmse_vs_gt_m": rmse_gt},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, kiss_cpu, res_cpu = cpu_baseline(sq, n_cpu, use_imu, args.cpu_budget, with_ekf=with_ekf,
                                                  oracle_over={"voxel_size": args.voxel_size} if args.voxel_size else None)
            m = min(len(res_cpu), len(est))
            line["cpu_baseline"] = cb
            d = np.linalg.norm(est[:m, :3, 3] - res_cpu[:m, :3, 3], axis=1)
            line["parity_vs_oracle"] = {"scans": m, "max_dpos_m": float(d.max()), "rmse_dpos_m": float(np.sqrt(np.mean(d ** 2)))}
            line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"]
        else:
            line["cpu_baseline"] = None
        if world == 1 and S > 1 and not args.no_single_sequence:
            # SURVEY.md 8(e) wants both figures: k sequences per GPU (`value`) and one.  Sequence 0 alone, same sweeps, through
            # the single-sequence latency pipeline (one sequence over the whole chip, 32-lane kernel), after the timed region.
            runner.close()  # (its 2 x 16 handles' streams would share hardware queues with the pipeline measured next)
            one = core.SeqRunner(n_total, pps, n_imu, max_range=args.max_range, min_range=args.min_range,
                                 use_imu_prediction=use_imu, with_ekf=with_ekf, device_id=local_rank,
                                 **{k: v for k, v in icp_over.items() if k not in ("gn_lanes_per_point", "gn_threads", "gn_workgroups")})
            for k in range(n_total):
                one.upload_scan(k, sq.scan(k))
            one.upload_imu(sq.imu[:n_imu] if with_ekf else np.zeros((0, 7)),
                           [sq.imu_range_for_scan(k)[1] if with_ekf else 0 for k in range(n_total)])
            one.run(W)
            one.profile(enable=GN_EVENT_EVERY, reset=True)
            core.device_sync(local_rank)
            t1 = time.perf_counter()
            one.enqueue(K)
            one.wait()
            core.device_sync(local_rank)
            dt1 = time.perf_counter() - t1
            ms1, n1 = one.profile(enable=False)
            o1 = one.results()
            est1 = o1["res_poses"] if with_ekf else o1["kiss_poses"]
            line["single_sequence"] = {"value": K / dt1, "unit": "scans/s", "ms_per_step": 1e3 * dt1 / K, "kernel": "k_gn_loop",
                                       "avg_launch_us": 1e3 * ms1 / max(n1, 1),
                                       "max_dpos_vs_batched_m": float(np.linalg.norm(est1[:, :3, 3] - est[:len(est1), :3, 3], axis=1).max()),
                                       "note": "sequence 0 alone (--seqs-per-gpu 1): one sequence over the whole chip"}
            one.close()
        if gathered is not None:
            line["gathered_trajectories"] = {"sequences": len(gathered), "rows_each": sorted({len(v) for v in gathered.values()}),
                                             "backend": "gloo (ranks share a GPU; RCCL refuses duplicate devices)" if shared_device else "nccl (RCCL)"}
            if args.dump_traj:
                np.savez(args.dump_traj, **{f"rank{r}_seq{j}": v for (r, j), v in gathered.items()},
                         seeds=np.array([[r, j, args.seed_base + (j if args.equal_work else r + world * j)] for (r, j) in gathered]))
        elif gather_err is not None:
            line["gathered_trajectories"] = {"error": gather_err}
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        if gather_err is not None:  # a communicator in an unknown state: leave without the collective shutdown, and say so
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(3)
        dist.barrier(group=ctl)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
