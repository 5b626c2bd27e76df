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
HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


def pmc_traffic():
    """HBM bytes per k_gn_loop launch from the committed rocprofv3 PMC passes (profiles/), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1])).get("k_gn_loop_traffic_bytes_per_launch")
    except Exception:
        return None


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


def _cpu_pass(seq, n_total, use_imu_prediction, budget_s):
    from oracle import cpu as orc
    icp = orc.ICP(max_range=seq.max_range, min_range=seq.min_range)
    ekf = orc.EKF()
    t01 = seq.column_times()
    kiss, res = [], []
    spent, done = 0.0, 0
    for k in range(n_total):
        x = seq.scan(k).astype(np.float64)  # rendering is not part of the timed work
        a, b = seq.imu_range_for_scan(k)
        t0 = time.perf_counter()
        for i in range(a, b):
            ekf.process_imu(seq.imu[i, 1:4], seq.imu[i, 4:7], seq.imu[i, 0])
        guess = ekf.pose_mat() if use_imu_prediction else None
        pose = icp.register_frame(x, t01, guess)
        ekf.process_pose(pose)
        spent += time.perf_counter() - t0
        kiss.append(pose)
        res.append(ekf.pose_mat())
        done += 1
        if spent >= budget_s:
            break
    return done, spent, np.array(kiss), np.array(res)


def cpu_baseline(seq, n_total, use_imu_prediction, budget_s=20.0):
    """The CPU oracle (kind "port": our C restatement of the reference path) timed on this host on the first
    sweeps of the same sequence: one sequential pass (~1/3 of the budget; also the parity check of the GPU
    trajectory) and one pass with the loops kiss-icp runs under TBB spread over every usable core (oracle.h
    orc_set_threads).  `value` is the faster of the two."""
    from oracle import cpu as orc
    from ptudes_lab_amd.synth import usable_cores
    cores = usable_cores()
    orc.set_threads(1)
    d1, s1, kiss, res = _cpu_pass(seq, n_total, use_imu_prediction, budget_s / 3.0)
    v1, vm, dm, sm = d1 / s1, 0.0, 0, 0.0
    if cores > 1:
        orc.set_threads(cores)
        dm, sm, kiss_m, res_m = _cpu_pass(seq, n_total, use_imu_prediction, budget_s * 2.0 / 3.0)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seqs-per-gpu", type=int, default=1)
    ap.add_argument("--seed-base", type=int, default=1000, help="sequence s of SURVEY.md 8(d) uses seed seed_base + s")
    ap.add_argument("--distinct-seeds", action="store_true",
                    help="rank r takes sequences seed_base + r, + world + r, ... (SURVEY.md 8(e)) instead of every rank "
                         "registering its own copy of sequences seed_base .. seed_base + S - 1")
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--max-range", type=float, default=70.0)
    ap.add_argument("--min-range", type=float, default=1.0)
    ap.add_argument("--const-velocity", action="store_true",
                    help="use the constant-velocity guess instead of --use-imu-prediction (reference default)")
    ap.add_argument("--gn-wgs", type=int, default=0, help="workgroups of the persistent GN kernel (0 = library default)")
    ap.add_argument("--gn-threads", type=int, default=0)
    ap.add_argument("--voxel-size", type=float, default=0.0, help="override the map voxel size (default max_range/100)")
    ap.add_argument("--map-blocks", type=int, default=0, help="voxel-block pool capacity")
    ap.add_argument("--map-table", type=int, default=0, help="map hash-table slots (power of two)")
    ap.add_argument("--workload-name", type=str, default="")
    ap.add_argument("--device", type=int, default=-1, help="GPU index for this rank (default LOCAL_RANK)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # one process per GPU, started by the launcher (the driver's command line): this script does not spawn ranks itself
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N ranks with\n  python -m torch.distributed.run "
                 f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus {args.gpus} ...")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.device < 0 else args.device
    dist = None
    torch = None
    ctl = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):  # launched by torch.distributed.run
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # RCCL carries the one collective of the path, the trajectory gather after the run.  Its communicator is brought
        # up there and not before: a live RCCL communicator in the process stretches the cross-stream hand-overs of the
        # scan pipeline from 63 to 110 us per scan (2700 -> 2400 scans/s, measured with one rank).  Barriers and the
        # max-over-ranks clock go through a host-side gloo group.
        dist.init_process_group(backend="nccl")
        ctl = dist.new_group(backend="gloo")

    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import core, synth

    K, W, S = args.steps, args.warmup, args.seqs_per_gpu
    n_total = W + K
    use_imu = not args.const_velocity
    pps = args.rows * args.cols
    icp_over = dict(scan_cols=args.cols)
    if args.gn_wgs: icp_over["gn_workgroups"] = args.gn_wgs
    if args.gn_threads: icp_over["gn_threads"] = args.gn_threads
    if args.voxel_size: icp_over["voxel_size"] = args.voxel_size
    if args.map_blocks: icp_over["map_block_capacity"] = args.map_blocks
    if args.map_table: icp_over["map_table_capacity"] = args.map_table
    seqs = []
    for j in range(S):
        # Weak scaling = the same work on every GPU: by default each rank registers its own copy of the same sequence(s)
        # (own map, own poses, own filter, nothing shared).  The sequences of SURVEY.md 8(d) differ by +-20 % in GN
        # iterations per scan, so with --distinct-seeds (ids s with s % world == rank, 8(e)) the max-over-ranks clock
        # measures the slowest sequence, not the scaling.
        seed = args.seed_base + (rank + world * j if args.distinct_seeds else j)
        seqs.append(synth.make_sequence(seed=seed, n_scans=n_total, H=args.rows, W=args.cols, min_range=args.min_range,
                                        max_range=args.max_range))
    n_imu = seqs[0].imu_range_for_scan(n_total - 1)[1]
    # S == 1: the single-sequence runner (its Gauss-Newton kernel caches hash probes across iterations);
    # S > 1: all sequences of this rank advance in lockstep in one batched runner (one launch per stage for all)
    class _One:
        def __init__(self):
            self.r = core.SeqRunner(n_total, pps, n_imu, max_range=args.max_range, min_range=args.min_range,
                                    use_imu_prediction=use_imu, with_ekf=True, device_id=local_rank, **icp_over)
        def upload_scan(self, j, k, x): self.r.upload_scan(k, x)
        def upload_imu(self, j, rows, ends): self.r.upload_imu(rows, ends)
        def run(self, n): self.r.run(n)
        def enqueue(self, n): self.r.enqueue(n)
        def wait(self): self.r.wait()
        def results(self, j): return self.r.results()
        def profile(self, **kw): return self.r.profile(**kw)
        def copy_traj(self, j, ptr, n): return self.r.copy_traj(ptr, n)
    runner = _One() if S == 1 else core.BatchRunner(S, n_total, pps, n_imu, max_range=args.max_range,
                                                    min_range=args.min_range, use_imu_prediction=use_imu,
                                                    with_ekf=True, device_id=local_rank, **icp_over)
    for j, sq in enumerate(seqs):
        for k in range(n_total):
            runner.upload_scan(j, k, sq.scan(k))
        runner.upload_imu(j, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n_total)])

    def barrier():
        if dist is not None:
            dist.barrier(group=ctl)

    def sync():
        if torch is not None:
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
    sync(); barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        from ptudes_lab_amd import parallel
        dt = parallel.max_over_ranks(dt, dist, device="cpu", group=ctl)

    # per-rank accounting
    outs = [runner.results(j) for j in range(S)]
    gn_ms, gn_n = runner.profile(enable=False)
    gn_bytes, gn_bytes_all, b_scan = 0.0, 0.0, 0.0
    iters = []
    for o in outs:
        for k, s in enumerate(o["stats"][W:]):
            if (W + k) % ev_every == 0:  # the launches the HIP events bracketed
                gn_bytes += icp_bytes(s)
            gn_bytes_all += icp_bytes(s)
            b_scan += scan_bytes(s, pps)
            iters.append(s["iterations"])
    n_timed = sum(len(o["stats"]) - W for o in outs)
    assert n_timed == K * S, (n_timed, K, S)

    # final trajectory gather: the only collective (T x 8 NC-GT rows per sequence, RCCL all-gather).  It runs after the
    # timed region and brings the RCCL communicator up; a failure or a stall there is reported in the line, not fatal.
    gathered, gather_err = None, None
    if dist is not None:
        import threading
        from ptudes_lab_amd import parallel
        box = {}

        def _gather():
            try:
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
        gt = sq.gt_poses(0.5)
        g0i = np.linalg.inv(gt[0])
        gt_rel = np.array([g0i @ g for g in gt])
        ate_r, ate_t = calc_ate(list(o["res_poses"]), list(gt_rel[: len(o["res_poses"])]))
        rmse_gt = float(np.sqrt(np.mean(np.sum((o["res_poses"][:, :3, 3] - gt_rel[: len(o["res_poses"]), :3, 3]) ** 2, 1))))
        avg_gn_s = (gn_ms / 1e3) / max(gn_n, 1)
        avg_gn_bytes = gn_bytes / max(gn_n, 1)  # one launch carries the GN loops of all S sequences
        achieved = avg_gn_bytes / avg_gn_s if avg_gn_s > 0 else 0.0
        if args.distinct_seeds:
            seeds_txt = f"{args.seed_base}..{args.seed_base + world * S - 1} (rank r: s % {world} == r)"
        else:
            seeds_txt = f"{args.seed_base}..{args.seed_base + S - 1}" + (", a private copy on every rank (equal work per GPU)" if world > 1 else "")
        line = {
            "metric": f"lidar scans/sec (ICP+EKF) on {args.rows}x{args.cols} sweeps",
            "value": K * S * world / dt, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic {args.rows}x{args.cols} sweeps, random-walk SE(3) GT, "
                                   f"ICP + IMU-EKF ({'--use-imu-prediction' if use_imu else 'constant-velocity guess'}), "
                                   f"min/max range {args.min_range}/{args.max_range} m, voxel {(args.voxel_size or args.max_range / 100):.2f} m"
                                   + (f" [{args.workload_name}]" if args.workload_name else ""),
                       "sequences_per_gpu": S, "sequence_seeds": seeds_txt,
                       "scans_per_sequence": n_total, "parallelism": f"{world} independent sequence shard(s), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": pmc_traffic(), "kernel": "k_gn_loop" if S == 1 else "kb_gn_loop",
                         "avg_launch_us": 1e6 * avg_gn_s, "algorithmic_bytes_per_launch": avg_gn_bytes,
                         "launches": gn_n, "timed_launches": f"every {ev_every}th of {K} (HIP events)"},
            "whole_scan": {"algorithmic_bytes_per_scan": b_scan / max(n_timed, 1),
                           "achieved_GBps": (b_scan * world / dt) / 1e9 if world == 1 else None,
                           "gn_share_of_wall": (avg_gn_s * K) / dt,
                           "mean_gn_iterations": float(np.mean(iters))},
            "map": {"voxels_end": o["stats"][-1]["map_voxels"], "points_end": o["stats"][-1]["map_points"],
                    "n_src_mean": float(np.mean([s["n_src"] for s in o["stats"][W:]])),
                    "n_down_mean": float(np.mean([s["n_down"] for s in o["stats"][W:]]))},
            "accuracy": {"ate_vs_gt_ref_style_rot": ate_r, "ate_vs_gt_ref_style_trans_m2": ate_t,
                         "rmse_vs_gt_m": rmse_gt},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, kiss_cpu, res_cpu = cpu_baseline(sq, n_total, use_imu, args.cpu_budget)
            m = len(res_cpu)
            line["cpu_baseline"] = cb
            d = np.linalg.norm(o["res_poses"][:m, :3, 3] - res_cpu[:, :3, 3], axis=1)
            line["parity_vs_oracle"] = {"scans": m, "max_dpos_m": float(d.max()), "rmse_dpos_m": float(np.sqrt(np.mean(d ** 2)))}
            line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"]
        else:
            line["cpu_baseline"] = None
        if gathered is not None:
            line["gathered_trajectories"] = {"sequences": len(gathered), "rows_each": sorted({len(v) for v in gathered.values()})}
        elif gather_err is not None:
            line["gathered_trajectories"] = {"error": gather_err}
        print(json.dumps(line), flush=True)
    if dist is not None:
        if gather_err is not None:  # a communicator in an unknown state: leave without the collective shutdown
            sys.stdout.flush()
            os._exit(0)
        dist.barrier(group=ctl)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
