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
DEFAULT_SEQS = 240  # 30 per XCD on its 16 teams of 2: in the driver's 20-step form 33.9 k scans/s against 32.1 k for 192 (the end of the run is filled better), the same over 100 steps (profiles/r04_p_sequences_per_gpu.txt)
PARITY_EXTRA_SEQS = 8     # sequences beside sequence 0 whose trajectories are checked against the oracle (single rank; all its threads)
PARITY_EXTRA_SWEEPS = 40  # ... over their first sweeps
DEFAULT_REPEATS = 3       # timed repeats of the K steps (>= 3, the median is reported)
GATHER_TIMEOUT_S = 120
CPU_MIN_SWEEPS = 300  # sweeps offered to the CPU baseline (it stops at its time budget)
HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_RATE = 6.29e12  # B/s a streaming copy reaches on MI355X (same guide)


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


# ---- as-executed byte model of the free-running kernel ("Executed bytes"): every global load / store the
# kernel issues, at the width the lane requests (no cache-line rounding, no cache assumed), driven by the counters the
# kernel keeps (ptl_batch_exec_counters) and the per-scan statistics.  Unit costs in bytes:
EXEC_COST = {
    # Gauss-Newton loop (gn8_body): the first iteration of a scan reads every source point (24); the positions then live in the
    # workgroup's LDS (GN8_LDS_PTS = 3072 per workgroup), only the points beyond that are written (24) and read back (24) through
    # src_cur every iteration; after the first iteration phase A also reads the voxel key (8) and the 144-byte answer row
    # (position of the last search, its four nearest candidates, the bound on everybody else, ids: GN8_KCAND = 4)
    "source_read": 24, "point_iteration_in_memory": 48, "point_iteration_later": 144, "lds_points_per_workgroup": 3072,
    # a full search reads the probe row (128) + key (8), writes the answer row (144) + the winner's voxel (4); the searches of a scan's
    # first iteration (every source point once) read neither: their rows are rebuilt whatever they hold
    "search": 284, "search_first_iteration_not_read": 136,
    # a rebuilt probe row: 27 hash-table entries of 16 B, the row (128) and the key (8) written
    "row_rebuilt": 27 * 16 + 136,
    "map_point_read": 24,
    # per iteration and workgroup of a team of G: its row stored (36 words of 8 B), G rows read once by its first wavefront
    "exchange_word": 8, "exchange_words_per_row": 36,
    # stages, per raw point: K1 f32 xyz (12) + slot1 write (4) | K3 slot1 (4)   (round 4: K3b / K4 walk the compact frame_downsample, not the raw indices)
    "raw_point": 12 + 4 + 4,
    # per valid point: K1 writes the deskewed point (24)   (round 6: only a run's head keeps its slot, so K3 reads the slot's winner index
    # per CLAIM, not per valid point)
    "valid_point": 24,
    # per voxel claim (run head) of either down-sampling pass: key read (8) + compare-and-swap (8) + index read (4) + atomicMin (4), and
    # the winner index K3 / K4 read back through the head's slot (4)
    "vds_claim": 24 + 4,
    # per frame_down point: K3 reads the point (24), writes fd (24), releases the pass-1 slot (12); K3b reads fd (24), writes its pass-2 slot (4); K4 reads slot (4) (the winner index: per claim, below);
    # map insert a: fd read (24), world point written (24), table key read (8), list push (4), slot + link written (8);
    # b: slot (4) + table entry (16: the stored count comes with it) + list walk (~8) + world point (24) read, block written (24);
    # c: none in the free-running kernel - the prune pass counts the batch in (round 4; before: rank + length written and read back, block header read)
    "down_point": (24 + 24 + 12) + (24 + 4) + 4 + (24 + 24 + 8 + 4 + 8) + (4 + 16 + 8 + 24 + 24),
    # per source point: K4 reads + writes it (48) and releases its pass-2 slot (12)
    "source_point": 48 + 12,
    # prune: header (12) + first point (24) of every block below the pool's high-water mark (~ live voxels); the publish of a voxel the
    # batch touched (header 8 + table entry 8) is not counted
    "map_voxel": 36,
    # the per-column deskew table (12 doubles per column), written once per scan and read through the caches: counted once
    "scan_column": 96,
}


def exec_cost_for_build(info):
    """EXEC_COST with the terms that follow compile-time constants of the LOADED library taken from ptl_build_info (the
    table must not silently describe another build): answer-row size, positions kept in LDS, table entry sizes.  Returns the
    table and a list of what differed from the default build."""
    c = dict(EXEC_COST)
    notes = []
    row = 8 * info["ans_row_doubles"]
    if row != 144:
        notes.append(f"answer row {row} B (KCAND {info['kcand']})")
    c["point_iteration_later"] = row  # (the voxel key travels in the row since round 5: no separate 8-byte read)
    c["search"] = 128 + 8 + row + 4
    c["lds_points_per_workgroup"] = info["lds_points"]
    if info["lds_points"] != 3072:
        notes.append(f"{info['lds_points']} positions in LDS")
    c["row_rebuilt"] = 27 * info["tab_entry_bytes"] + 136
    if (info["seq_u"], info["seq_u2"]) != (8, 16):
        notes.append(f"SEQ_U {info['seq_u']}/{info['seq_u2']} (stage terms are per point: unchanged)")
    if info["diagnostics"]:
        notes.append("diagnostic clocks compiled in (their atomics are not in the model)")
    return c, notes


def executed_bytes(cnt, stats, n_raw, G, cols=1024, cost=None):
    """bytes the free-running kernel requested from memory for the scans in `stats` (ptl_icp_stats rows of the timed scans of
    one sequence), given the delta of its executed-work counters over those scans"""
    c = cost or EXEC_COST
    pi = cnt["point_iterations"]
    first = sum(s["n_src"] for s in stats if s["iterations"] > 0)
    in_mem = sum(max(s["n_src"] - max(G, 1) * c["lds_points_per_workgroup"], 0) * s["iterations"] for s in stats)  # point-iterations through src_cur
    gn = (c["source_read"] * first + c["point_iteration_in_memory"] * in_mem + c["point_iteration_later"] * max(pi - first, 0) + c["search"] * cnt["searches"]
          - c["search_first_iteration_not_read"] * min(first, cnt["searches"])
          + c["row_rebuilt"] * cnt["rows_rebuilt"] + c["map_point_read"] * cnt["map_points_read"]
          + cnt["gn_iterations"] * G * (1 + G) * c["exchange_words_per_row"] * c["exchange_word"])
    st = 0
    for s in stats:
        st += (c["raw_point"] * n_raw + c["valid_point"] * s["n_valid"] + c["down_point"] * s["n_down"]
               + c["source_point"] * s["n_src"] + c["map_voxel"] * s["map_voxels"] + c["scan_column"] * cols)
    st += c["vds_claim"] * (cnt["vds1_claims"] + cnt["vds2_claims"])
    return gn + st, gn, st


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
    upstream = upstream_kiss_leg(seq, min(n_total, max(d1, dm)), budget_s / 3.0, kiss)
    return dict(value=max(v1, vm), unit="scans/s", cores=cores if multi else 1, kind="port", upstream_kiss_icp=upstream,
                single_thread_value=v1, multi_thread_value=vm if cores > 1 else None,
                sample=f"first {dm if multi else d1} sweeps of sequence seed {seq.seed} (cold start), "
                       f"{sm if multi else s1:.1f} s wall on {cores if multi else 1} threads "
                       f"(plus {d1} sweeps / {s1:.1f} s single-thread), oracle/liboracle.so; host has "
                       f"{os.cpu_count()} logical cores, {cores} usable under the cgroup quota"), kiss, res


def upstream_kiss_leg(seq, n, budget_s, kiss_oracle):
    """if and only if `import kiss_icp` (0.2.9 / 0.2.10) succeeds on this box, the same array feed through the
    upstream package, driven by the calls the reference makes (oracle/upstream_kiss.py; ICP only, ground-truth-free constant-velocity
    guess), timed, and compared with the oracle's ICP-only trajectory when that is what the oracle ran.  Never required: returns
    {"available": False, "why": ...} otherwise.  No reference file is involved."""
    from oracle import upstream_kiss as up
    mod, ver = up.available()
    if mod is None:
        return {"available": False, "why": ver}
    icp = up.Upstream(seq.max_range, seq.min_range)
    t01 = seq.column_times()
    spent, done = 0.0, 0
    for k in range(n):
        xyz = np.asarray(seq.scan(k), np.float64)
        sel = np.linalg.norm(xyz, axis=1) > 0
        t0 = time.perf_counter()
        icp.register_frame(xyz[sel], t01[sel])
        spent += time.perf_counter() - t0
        done += 1
        if spent >= budget_s:
            break
    return {"available": True, "kind": f"upstream kiss-icp {ver}", "value": done / spent, "unit": "scans/s", "sweeps": done,
            "mode": "ICP only, constant-velocity guess (reference kiss.py:102-105)",
            "poses_last": np.asarray(icp.poses[-1]).tolist()}


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


def source_code_id(flags=""):
    """csrc/Makefile's CODE_ID recomputed from the source tree: sha256 over the kernel sources, the C-ABI header, the Makefile and the
    experiment flags (none for the product build), 12 hex digits"""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ptudes-lab_amd", "csrc")
    files = sorted(glob.glob(os.path.join(d, "*.hip"))) + sorted(glob.glob(os.path.join(d, "*.h"))) + \
        [os.path.join(d, "..", "..", "include", "ptudes_mi.h"), os.path.join(d, "Makefile")]
    for f in files:
        h.update(open(f, "rb").read())
    h.update((flags + "\n").encode())
    return h.hexdigest()[:12]


def code_id():
    """identity of the LOADED library (ptl_code_id: baked in by csrc/Makefile from its sources and flags): recorded in the line and in
    every PMC summary, so that a counter pass collected on another build - an older kernel, an experiment build behind PTL_LIB_PATH,
    a stale .so - is not taken for this build's traffic.  Falls back to the source tree's id for a library that
    predates the symbol."""
    try:
        from ptudes_lab_amd import _lib
        return _lib.lib().ptl_code_id().decode()
    except Exception:  # noqa: BLE001
        return source_code_id()


def pmc_traffic_for(workload_key, scans_per_launch=None, code=None):
    """The committed rocprofv3 PMC pass (profiles/) whose recorded workload is THIS run's workload, or None: a counter value
    belongs to the workload it was collected on.  Returns a dict: bytes (HBM bytes per launch of the dominant kernel: the
    free-running kernel's summaries record bytes per SCAN, scaled here by this run's scans per launch; per-launch kernels record
    bytes per launch), file, stale (collected on another build than the loaded one), and what the pass covered - warmup, steps,
    executed_bytes_per_scan (None for summaries that predate those fields).  A pass of this very build beats a later file of
    another build; among equals the latest file wins."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload_key") != workload_key:
            continue
        stale = code is not None and d.get("code_id") != code  # collected on another build than the loaded library
        if d.get("traffic_bytes_per_scan") is not None and scans_per_launch:
            nbytes = d["traffic_bytes_per_scan"] * scans_per_launch
        elif d.get("traffic_bytes_per_launch") is not None and not scans_per_launch:
            nbytes = d["traffic_bytes_per_launch"]
        else:
            continue
        cand = {"bytes": nbytes, "file": os.path.basename(f), "stale": stale, "warmup": d.get("warmup"), "steps": d.get("steps"),
                "executed_bytes_per_scan": d.get("executed_bytes_per_scan"), "per_scan": d.get("traffic_bytes_per_scan") is not None}
        if best is None or not stale or best["stale"]:
            best = cand
    return best


def pmc_for_this_run(pmc, W, K, exec_bytes_per_scan):
    """How a counter pass of this workload and build may speak for THIS run.  A free-running summary is bytes per scan
    over the scans the pass covered; per-scan traffic changes with the scans (the map grows, the iteration count falls), so:
    same warm-up and step counts -> ("same scans", bytes); other counts, both runs' executed bytes per scan known -> ("scaled", bytes x
    the ratio of executed bytes per scan, this run : the pass); else (None, None) and the line keeps the executed-byte model."""
    if not pmc or pmc["stale"]:
        return None, None
    if not pmc["per_scan"] or (pmc["warmup"], pmc["steps"]) == (W, K):
        return "same scans", pmc["bytes"]
    if pmc["executed_bytes_per_scan"] and exec_bytes_per_scan:
        return "scaled", pmc["bytes"] * exec_bytes_per_scan / pmc["executed_bytes_per_scan"]
    return None, None


def workload_key(args, S):
    """what a PMC pass has to have been collected on to speak for this run.  Per-launch kernels (one Gauss-Newton launch per
    step): including the warm-up and step counts, which decide what the map holds during the timed launches.  The free-running
    kernel (one launch for the whole run, summaries per scan): without them, with the team geometry instead."""
    free = free_running(args, S)
    return (f"{args.rows}x{args.cols}_r{args.min_range:g}-{args.max_range:g}_v{(args.voxel_size or args.max_range / 100):g}_"
            f"seed{args.seed_base}_S{S}_" + ("" if free else f"W{args.warmup}_K{args.steps}_") +
            f"{'cv' if args.const_velocity else 'imu'}"
            f"{'' if not args.icp_only else '_icponly'}{'' if not args.gn_lanes else '_L%d' % args.gn_lanes}"
            f"{'' if not getattr(args, 'map_small_blocks', 0) else '_sb%d' % args.map_small_blocks}"
            f"{'' if not getattr(args, 'range_input', False) else '_range'}"
            f"{'' if not getattr(args, 'ray_jitter_deg', 0.0) else '_jit%g' % args.ray_jitter_deg}"
            + (f"_free_g{getattr(args, 'gn_wgs', 0) or 256}x{getattr(args, 'gn_threads', 0) or 512}_t{getattr(args, 'team_wgs', 0)}" if free else ""))


def default_team_wgs(S):
    """bench default: the library's geometry"""
    return 0


def free_running(args, S):
    """does this run use the free-running batch driver (one persistent launch, every sequence at its own pace)?"""
    return S > 1 and not args.lockstep and args.gn_lanes in (0, 8)


class RangeFeed:
    """A synth.Sequence seen through its range images (--range-input): `range_image(k)` = the sweep as the sensor reports it (u32
    millimetres per pixel, 0 = no return: reference kiss.py:59-61 works on such a scan), `scan(k)` = the points the device LUT makes of
    it (ptl_lut_apply: exactly what K1 computes) - what the oracle and the CPU baseline are fed, so both sides see the same points.
    The synthetic sensor counts its columns counter-clockwise (column j looks along 2 pi j / W and fires at j / W of the sweep); an
    Ouster counts clockwise (XYZLut: 2 pi (1 - v / W)), so the LUT gets a lidar_to_sensor that mirrors y - pixel column v is then
    synthetic column v, and the per-column times of the deskew (kiss.py:34-35) are the firing times."""

    def __init__(self, sq, lut):
        self._sq, self._lut = sq, lut

    def __getattr__(self, name):
        return getattr(self._sq, name)

    def range_image(self, k):
        x = np.asarray(self._sq.scan(k), dtype=np.float64)
        return np.round(np.sqrt((x * x).sum(axis=1)) * 1000.0).astype(np.uint32)

    def scan(self, k):
        return self._lut(self.range_image(k))


def synthetic_lut(core, rows, cols, device_id=0):
    l2s = np.diag([1.0, -1.0, 1.0, 1.0])
    return core.Lut(rows, cols, np.linspace(45.0, -45.0, rows), np.zeros(rows), 0.0, l2s, None, device_id=device_id)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=DEFAULT_REPEATS,
                    help="timed repeats: each one is a cold start, the W warm-up sweeps (untimed) and EXACTLY K timed steps; the line "
                         "reports the median repeat (SURVEY.md 8(d)) and lists them all")
    ap.add_argument("--verify-all", action="store_true",
                    help="after the run: every sequence of the batch once more ALONE (single-sequence runner with a team's workgroup "
                         "count) and compared bit for bit - minutes for 240 sequences")
    ap.add_argument("--seqs-per-gpu", type=int, default=DEFAULT_SEQS,
                    help="independent sequences per GPU (SURVEY.md 8(e), second level), batched runner.  The sequences s = x (mod 8) "
                         "live on XCD x, whose teams of workgroups take their scans as they come free (--team-wgs).  1 = the "
                         "single-sequence latency pipeline (one sequence over the whole chip)")
    ap.add_argument("--team-wgs", type=int, default=-1,
                    help="workgroups per team of the free-running kernel (-1: bench default for the number of sequences, 0: the "
                         "library's).  Smaller teams spread the per-iteration fixed costs over more points and want more sequences")
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
    ap.add_argument("--map-small-blocks", type=int, default=0, help="small (128-byte, 5-point) voxel blocks beside the --map-blocks full ones: sparse maps (config 5)")
    ap.add_argument("--rebuild-every", type=int, default=0, help="scans between two rebuilds of the map hash table (tombstones dropped; 0 = library default)")
    ap.add_argument("--range-input", action="store_true",
                    help="sweeps enter as raw range images (SURVEY.md 8(f) rank 1: 512 KB of u32 millimetres per 128x1024 sweep instead of 1.57 MB of "
                         "float32 xyz; reference kiss.py:28-29, 59-61) and stay range images in HBM: 240 sequences x 1 000 sweeps fit (126 GB). "
                         "The LUT is the synthetic sensor's; oracle and CPU baseline get the same points through the numpy restatement of the LUT")
    ap.add_argument("--ray-jitter-deg", type=float, default=0.0,
                    help="per-ray angular jitter of the synthetic sensor (takes the sampling lattice away: 21-28 Gauss-Newton iterations per scan "
                         "instead of 35-40, DESIGN.md 6) - the stage-bound regime; part of the workload key")
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

    seqs_explicit = any(a == "--seqs-per-gpu" or a.startswith("--seqs-per-gpu=") for a in sys.argv[1:])
    if args.seqs_per_gpu < 1 or args.seqs_per_gpu > 256:
        sys.exit("bench.py: --seqs-per-gpu must be in [1, 256]")
    if args.team_wgs < 0:
        args.team_wgs = default_team_wgs(args.seqs_per_gpu)
    multi = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)  # a rank of a multi-process run
    dist = None
    ctl = None
    # the product library FIRST, on the system's HIP runtime - the runtime every single-rank run uses.  torch comes in afterwards and for
    # the CONTROL PLANE only (a gloo group: barriers, the max-over-ranks clock, the 128 bytes of the RCCL id); it never touches a GPU
    # here, so whichever HIP runtime it finds mapped is of no consequence.  The one collective of the path - the trajectory gather - is
    # the library's own (ptl_comm_* / ptl_batch_gather_trajectories: ncclAllGather from librccl).
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import _lib, core, synth
    n_dev = _lib.lib().ptl_device_count()  # (hipGetDeviceCount: does not initialise a device)
    if n_dev < 1:
        sys.exit("bench.py: no HIP device - the HIP path is the only path")
    local_rank = (int(os.environ.get("LOCAL_RANK", "0")) % n_dev) if args.device < 0 else args.device
    shared_device = world > n_dev  # several ranks on one GPU (1-GPU box): RCCL refuses duplicate devices, the gather goes through host rows over gloo
    if multi:
        import datetime
        import torch.distributed as dist
        # The RCCL communicator is brought up AFTER the timed region (a live communicator in the process stretched the cross-stream
        # hand-overs of the single-sequence pipeline from 63 to 110 us per scan, measured with one rank in round 1).
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
        ctl = dist.group.WORLD
        # the RCCL id travels over a group of its own: a rank stuck in that broadcast (its thread is abandoned after GATHER_TIMEOUT_S)
        # must not interleave with the control plane's collectives on `ctl`
        id_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=600))

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
    if args.map_small_blocks: icp_over["map_small_blocks"] = args.map_small_blocks
    if args.rebuild_every: icp_over["rebuild_every"] = args.rebuild_every
    # rank r owns the sequences s with s % world == r, sequence s has seed seed_base + s - independent
    # sequences, nothing shared, no data-path collective.  They differ by up to 40 % in GN iterations per scan, so the
    # max-over-ranks clock of such a run is the slowest sequence's; --equal-work gives every rank a private copy of the
    # same sequence(s) instead (the pure scaling measurement).
    if world > 1:  # the ranks of one host share its cores: no oversubscription while every rank renders its sweeps
        synth.set_threads(max(2, min(16, synth.usable_cores() // world)))
    # the CPU baseline wants a sample of its own size (>= CPU_MIN_SWEEPS sweeps within its time budget) however short the
    # timed GPU run is: sequence 0 of rank 0 is generated that long, the GPU registers its first n_total sweeps
    n_cpu = max(n_total, CPU_MIN_SWEEPS) if (world == 1 and not args.no_cpu_baseline) else n_total

    def make_seq(j, s):
        return synth.make_sequence(seed=args.seed_base + s, n_scans=(n_cpu if j == 0 else n_total), H=args.rows, W=args.cols,
                                   min_range=args.min_range, max_range=args.max_range, ray_jitter_deg=args.ray_jitter_deg)
    lut = synthetic_lut(core, args.rows, args.cols, local_rank) if args.range_input else None
    if lut is not None:
        _make = make_seq

        def make_seq(j, s):  # noqa: F811
            return RangeFeed(_make(j, s), lut)
    seq0 = make_seq(0, 0 if args.equal_work else rank)
    n_imu = seq0.imu_range_for_scan(n_total - 1)[1] if with_ekf else 0

    def feed_sweep(r, k, sq, j=None):
        """sweep k of sq into runner r (a batch takes the sequence slot j)"""
        if lut is not None:
            r.upload_range(*(() if j is None else (j,)), k, sq.range_image(k))
        else:
            r.upload_scan(*(() if j is None else (j,)), k, sq.scan(k))
    # S == 1: the single-sequence runner (its Gauss-Newton kernel caches hash probes across iterations);
    # S > 1: all sequences of this rank advance in lockstep in one batched runner (one launch per stage for all)
    class _One:
        def __init__(self):
            self.r = core.SeqRunner(n_total, pps, n_imu, max_range=args.max_range, min_range=args.min_range,
                                    use_imu_prediction=use_imu, with_ekf=with_ekf, device_id=local_rank, **icp_over)
            if lut is not None: self.r.set_lut(lut)
        def upload_scan(self, j, k, x): self.r.upload_scan(k, x)
        def upload_range(self, j, k, x): self.r.upload_range(k, x)
        def upload_imu(self, j, rows, ends): self.r.upload_imu(rows, ends)
        def run(self, n): self.r.run(n)
        def enqueue(self, n): self.r.enqueue(n)
        def wait(self): self.r.wait()
        def results(self, j): return self.r.results()
        def profile(self, **kw): return self.r.profile(**kw)
        def copy_traj(self, j, ptr, n): return self.r.copy_traj(ptr, n)
    t_setup = time.perf_counter()
    s_asked = S
    while True:
        free = free_running(args, S)
        try:
            runner = _One() if S == 1 else core.BatchRunner(S, n_total, pps, n_imu, max_range=args.max_range,
                                                            min_range=args.min_range, use_imu_prediction=use_imu,
                                                            with_ekf=with_ekf, device_id=local_rank, free_running=free,
                                                            scans_per_launch=args.scans_per_launch,
                                                            team_workgroups=args.team_wgs if free else 0,
                                                            range_input=lut is not None, **icp_over)
            if S > 1 and lut is not None:
                runner.set_lut(lut)
            break
        except RuntimeError as e:
            # the DEFAULT sequence count is sized for the 288 GB of an MI355X (about 0.8 GB per sequence): on a part with less memory -
            # ptl_batch_create says what it needs and what is free - the default shrinks until it fits; an explicit
            # --seqs-per-gpu that does not fit stays an error.  Every rank of a run takes the same decision from the same numbers.
            if args.seqs_per_gpu == DEFAULT_SEQS and "of device memory" in str(e) and S > 16 and not seqs_explicit:
                S -= 16
                continue
            raise
    if S != s_asked:
        print(f"bench.py rank {rank}: {s_asked} sequences do not fit this device, running {S}", file=sys.stderr, flush=True)
    seq_ids = [j if args.equal_work else rank + world * j for j in range(S)]
    seqs = [seq0] + [make_seq(j, s) for j, s in enumerate(seq_ids) if j > 0]
    team_g, teams = runner.team_geometry() if free else (0, 0)
    t_render, t_upload = 0.0, 0.0
    for j, sq in enumerate(seqs):
        for k in range(n_total):
            ta = time.perf_counter()
            x = sq.range_image(k) if lut is not None else sq.scan(k)
            tb = time.perf_counter()
            if lut is not None:
                runner.upload_range(j, k, x)
            else:
                runner.upload_scan(j, k, x)
            t_render += tb - ta
            t_upload += time.perf_counter() - tb
        runner.upload_imu(j, sq.imu[:n_imu] if with_ekf else np.zeros((0, 7)),
                          [sq.imu_range_for_scan(k)[1] if with_ekf else 0 for k in range(n_total)])
    # where a slow start-up of a many-rank run goes: per rank, to stderr (fd 1 is stderr here)
    print(f"bench.py rank {rank}: {S} sequences x {n_total} sweeps - handles {time.perf_counter() - t_setup - t_render - t_upload:.1f} s, "
          f"rendering {t_render:.1f} s ({synth.usable_cores()} usable cores), upload {t_upload:.1f} s", file=sys.stderr, flush=True)

    def barrier():
        if dist is not None:
            dist.barrier(group=ctl)

    def sync():
        core.device_sync(local_rank)  # hipDeviceSynchronize on this rank's device: every stream of the runner

    # HIP events around every 8th launch of the dominant kernel: two event records per scan cost ~18 us (4 %) of
    # command-processor time on the critical path
    ev_every = GN_EVENT_EVERY if S == 1 else 1  # (the batched runner times every launch)
    # R timed repeats (>= 3, median).  Each: cold start + the first W sweeps (untimed, they build the local map),
    # then EXACTLY K steps between barrier + synchronize on both sides, max over ranks.  The kernel events cover the timed
    # launches of all repeats (warm-up launches excluded), the byte counters and results are the last repeat's; the runs are
    # deterministic, so every repeat must reproduce the first one's trajectories bit for bit (checked, reported).
    R = max(1, args.repeats)
    rep_dt, rep_dt_own = [], []
    first_traj, repeats_identical = None, True
    cnt0 = None
    for rep in range(R):
        runner.profile(enable=False, reset=(rep == 0))
        t_warm = time.perf_counter()
        runner.run(W)
        if rep == 0:
            print(f"bench.py rank {rank}: warm-up ({W} sweeps of every sequence) {time.perf_counter() - t_warm:.2f} s", file=sys.stderr, flush=True)
        cnt0 = [runner.exec_counters(j) for j in range(S)] if free else None
        clk0 = [runner.seq_clocks_raw(j) for j in range(S)] if free else None  # (the warm-up's share of the phase clocks)
        runner.profile(enable=ev_every, reset=False)
        barrier(); sync()
        t0 = time.perf_counter()
        runner.enqueue(K)
        runner.wait()
        sync()
        dt_own = time.perf_counter() - t0  # this rank's own K steps
        barrier()
        dt = time.perf_counter() - t0
        print(f"bench.py rank {rank}: repeat {rep}: timed region {dt_own:.3f} s own, {dt:.3f} s to the barrier", file=sys.stderr, flush=True)
        if dist is not None:
            from ptudes_lab_amd import parallel
            dt = parallel.max_over_ranks(dt, dist, device="cpu", group=ctl)
        rep_dt.append(dt); rep_dt_own.append(dt_own)
        if R > 1:
            probe = [runner.results(j)["kiss_poses"] for j in sorted({0, S // 2, S - 1})]
            if first_traj is None:
                first_traj = probe
            else:
                repeats_identical = repeats_identical and all(np.array_equal(a, b) for a, b in zip(first_traj, probe))
    order = sorted(range(R), key=lambda i: rep_dt[i])
    med = order[(R - 1) // 2]  # the median repeat (the lower one of an even count)
    dt, dt_own = rep_dt[med], rep_dt_own[med]
    per_rank = [K * S / dt_own]
    if dist is not None:
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
    seq_clk_timed = None
    if free:  # the same clocks over the TIMED scans alone (sums now minus sums after the warm-up)
        seq_clk_timed = []
        for j in range(S):
            (t1, n1), (t0_, n0) = runner.seq_clocks_raw(j), clk0[j]
            seq_clk_timed.append(tuple((a - b) / max(n1 - n0, 1) / 100.0 for a, b in zip(t1, t0_)))
    exec_bytes, exec_gn, exec_stages, cnt_tot = 0.0, 0.0, 0.0, None
    exec_cost, exec_notes, migr = None, [], None
    if free:  # the as-executed byte model over the timed scans: counters now minus counters after the warm-up
        exec_cost, exec_notes = exec_cost_for_build(core.build_info())
        migr = [runner.sched_counters(j) for j in range(S)]
        cnt_tot = {k: 0 for k in core.BatchRunner.EXEC_COUNTERS}
        for j, o in enumerate(outs):
            c1 = runner.exec_counters(j)
            d = {k: c1[k] - cnt0[j][k] for k in c1}
            tot, gnb, stb = executed_bytes(d, o["stats"][W:], pps, team_g, args.cols, exec_cost)
            exec_bytes += tot; exec_gn += gnb; exec_stages += stb
            for k in d:
                cnt_tot[k] += d[k]
    n_timed = sum(len(o["stats"]) - W for o in outs)
    assert n_timed == K * S, (n_timed, K, S)

    # final trajectory gather: the only collective (T x 8 NC-GT rows per sequence: the library's ncclAllGather, device rows in, host
    # rows out; host rows over gloo when several ranks share one GPU, which RCCL refuses).  It runs after the timed region and brings
    # the RCCL communicator up; a failure or a stall there is reported in the line AND in the exit code.
    gathered, gather_err, rccl_stalled = None, None, False
    if dist is not None:
        import threading
        from ptudes_lab_amd import parallel

        def _host_rows():
            import torch
            rows = torch.zeros((S, n_total, 8), dtype=torch.float64)
            counts = []
            for j in range(S):
                o = outs[j]
                if with_ekf:
                    t_rows, p_rows = o["res_t"], o["res_poses"]
                else:  # ICP only: the registration's own poses, stamped with the scan index (what ptl_batch_gather_trajectories sends then)
                    t_rows, p_rows = np.arange(len(o["kiss_poses"]), dtype=np.float64), o["kiss_poses"]
                counts.append(len(t_rows))
                rows[j, : counts[-1]] = torch.from_numpy(parallel.poses_to_rows(t_rows, p_rows))
            return parallel.gather_trajectories(rows, counts, dist)

        def _gather_rccl(box):
            try:
                comm = parallel.Comm.over(dist, id_group, device_id=local_rank)  # rank 0 makes the id, gloo carries it, everybody joins
                if S == 1:
                    ptr, n_rows = runner.r.traj_device()
                    box["out"] = comm.gather_rows(ptr, 1, n_total, [min(n_rows, n_total)])
                else:
                    box["out"] = comm.gather_batch(runner)
                comm.close()
            except Exception as e:  # noqa: BLE001
                box["err"] = repr(e)

        def _gather_gloo(box):
            try:
                box["out"] = _host_rows()
            except Exception as e:  # noqa: BLE001
                box["err"] = repr(e)

        def _attempt(fn):
            box = {}  # (its own: a thread that is stuck in an earlier attempt must not write into a later one's)
            th = threading.Thread(target=fn, args=(box,), daemon=True)
            th.start()
            th.join(timeout=GATHER_TIMEOUT_S)
            err = "no answer after %d s" % GATHER_TIMEOUT_S if th.is_alive() else box.get("err")
            if err is not None:
                print(f"bench.py rank {rank}: trajectory gather ({fn.__name__[8:]}) failed: {err}", file=sys.stderr, flush=True)
            any_failed = parallel.max_over_ranks(0.0 if err is None else 1.0, dist, device="cpu", group=ctl) > 0
            return box.get("out"), err, any_failed, th.is_alive()

        host_only = shared_device or (S == 1 and not with_ekf)  # (a single ICP-only sequence keeps no rows on the device)
        gather_backend = "gloo, host rows (ranks share a GPU; RCCL refuses duplicate devices)" if shared_device else \
            "gloo, host rows (single ICP-only sequence)" if host_only else \
            "RCCL ncclAllGather through libptudes_mi.so (ptl_batch_gather_trajectories), id over gloo"
        gathered, gather_err, any_failed, stalled = _attempt(_gather_gloo if host_only else _gather_rccl)
        if any_failed and not host_only:
            # the library's RCCL path raised or did not answer somewhere (a first contact with a node's RCCL / xGMI set-up): the measurement
            # above does not depend on it, so every rank falls back to host rows over the control plane - and the line says which way the
            # rows came.  (A thread stuck inside RCCL stays behind as a daemon; such a rank leaves without the collective shutdown below.)
            rccl_stalled = parallel.max_over_ranks(1.0 if stalled else 0.0, dist, device="cpu", group=ctl) > 0
            rccl_err = gather_err or "failed on another rank"
            gathered, gather_err, any_failed, stalled = _attempt(_gather_gloo)
            gather_backend = f"gloo, host rows - FALLBACK after the RCCL gather failed: {rccl_err}"
        if any_failed and gather_err is None:
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
        avg_gn_bytes = R * gn_bytes / max(gn_n, 1)  # one launch carries the GN loops (free-running: the whole scans) of all S sequences; gn_bytes: one repeat's
        achieved = avg_gn_bytes / avg_gn_s if avg_gn_s > 0 else 0.0
        if args.equal_work:
            seeds_txt = f"{args.seed_base}..{args.seed_base + S - 1}" + (", a private copy on every rank (equal work per GPU)" if world > 1 else "")
        else:
            seeds_txt = f"{args.seed_base}..{args.seed_base + world * S - 1}" + (f" (rank r owns s % {world} == r)" if world > 1 else "")
        mode_txt = "ICP only, constant-velocity guess" if not with_ekf else \
            f"ICP + IMU-EKF ({'--use-imu-prediction' if use_imu else 'constant-velocity guess'})"
        wkey = workload_key(args, S)
        launches = max(gn_n, 1)  # (the timed launches of all R repeats)
        scans_per_launch_mean = (R * K * S / launches) if free else None
        cid = code_id()
        pmc = pmc_traffic_for(wkey, scans_per_launch_mean, cid) if world == 1 else None
        alg_frac = achieved / HBM_PEAK  # SURVEY 8(d)'s brute-force bytes / launch time / peak
        if free:
            # what the kernel itself requested from memory (EXEC_COST x its own counters): the roofline figure of the line
            avg_exec = R * exec_bytes / launches
            achieved = avg_exec / avg_gn_s if avg_gn_s > 0 else 0.0
        exec_achieved = achieved  # executed-byte model (free-running) / algorithmic bytes (the per-launch kernels)
        exec_per_scan = (exec_bytes / max(n_timed, 1)) if free else None
        pmc_how, pmc_bytes = pmc_for_this_run(pmc, W, K, exec_per_scan) if avg_gn_s > 0 else (None, None)
        fresh_pmc = pmc_how is not None
        if fresh_pmc:  # a counter pass of THIS workload on THIS build: what the memory system moved is the roofline figure
            achieved = pmc_bytes / avg_gn_s
        roof = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "frac_of_copy_rate": achieved / HBM_COPY_RATE,
                "frac_kind": ("HBM traffic of the dominant kernel (rocprofv3 PMC pass of this workload on this build, profiles/" + pmc["file"] +
                              ": 2 x FETCH_SIZE + WRITE_SIZE, corrected as the guide prescribes"
                              + (f"; the pass covered the same scans as this run: warm-up {W}, {K} steps" if pmc_how == "same scans" else
                                 f"; the pass covered warm-up {pmc['warmup']}, {pmc['steps']} steps, this run warm-up {W}, {K} steps: its bytes per scan are "
                                 f"SCALED by the ratio of executed bytes per scan, this run : the pass = {exec_per_scan / pmc['executed_bytes_per_scan']:.4f}")
                              + ") / launch time (HIP events, this run) / peak; "
                              "frac_of_copy_rate: / the 6.29 TB/s a streaming copy reaches on this part" if fresh_pmc else
                              "executed bytes: every load / store the kernel issued, at the width requested (bench.py EXEC_COST x the "
                              "kernel's own counters, DESIGN.md 3) / launch time / peak - no counter pass of this build and workload is committed" if free else
                              "algorithmic bytes of SURVEY 8(d) (27 probes x 16 B + every candidate x 12 B + the source, per iteration) / "
                              "launch time / peak; the kernel prunes the search exactly and skips ~3/4 of the candidate bytes"),
                "executed_frac": (exec_achieved / HBM_PEAK) if free else None,
                "algorithmic_frac": alg_frac, "algorithmic_bytes_per_launch": avg_gn_bytes,
                "executed_bytes_per_launch": (R * exec_bytes / launches) if free else None,
                "executed_model_build": ({"info": core.build_info(), "differs_from_default_build": exec_notes} if free else None),
                "executed_bytes_per_scan": (exec_bytes / max(n_timed, 1)) if free else None,
                "executed_split_per_scan": ({"gauss_newton": exec_gn / max(n_timed, 1), "stages": exec_stages / max(n_timed, 1)} if free else None),
                "executed_counters_per_scan": ({k: v / max(n_timed, 1) for k, v in cnt_tot.items()} if free else None),
                "traffic": (pmc_bytes if fresh_pmc else pmc["bytes"]) if pmc else None, "traffic_source": pmc["file"] if pmc else None,
                "traffic_stale": (pmc["stale"] if pmc else None),  # true: the PMC pass was collected on another build than the loaded library (config.code_id)
                "traffic_scans": (None if not pmc else {"how": pmc_how or ("other build" if pmc["stale"] else "other scans, not scalable: frac keeps the executed model"),
                                                        "pass_warmup": pmc["warmup"], "pass_steps": pmc["steps"], "run_warmup": W, "run_steps": K}),
                "measured_frac": ((pmc_bytes if fresh_pmc else pmc["bytes"]) / avg_gn_s / HBM_PEAK) if (pmc and avg_gn_s > 0) else None,
                "kernel": ("k_gn_loop8" if args.gn_lanes == 8 else "k_gn_loop") if S == 1 else
                          "kx_seq_run" if free else ("kx_gn_loop" if args.gn_lanes == 32 else "kx_gn_loop8"),
                "avg_launch_us": 1e6 * avg_gn_s, "launches": gn_n,
                "scans_per_launch": scans_per_launch_mean,
                "timed_launches": (f"all {gn_n} persistent launches of the {R} x {K} timed steps (HIP events)" if free else
                                   f"every {ev_every}th of {R} x {K} (HIP events)"),
                "note": ("the free-running kernel carries the WHOLE per-scan pipeline of every sequence.  frac = measured HBM traffic / launch time "
                         "/ peak when a counter pass of this build exists (else = executed_frac); executed_frac = executed bytes / launch time / peak; algorithmic_frac = B_scan of SURVEY 8(d) (brute-force 27-voxel search) the same way - above 1 because the "
                         "answer cache settles most point-iterations without the probes and candidate reads that formula charges; "
                         "traffic = HBM bytes (PMC: 2 x FETCH_SIZE + WRITE_SIZE) of a committed pass of this workload, per scan x this run's "
                         "scans per launch; measured_frac = traffic / launch time / peak.  FETCH_SIZE counts what the L2s request from the "
                         "fabric: reads served by the 256 MiB Infinity Cache are in it, so `of the HBM peak` is an upper bound on what HBM itself served" if free else
                         "measured_frac = PMC HBM bytes of the same workload / launch time / peak")}
        line = {
            "metric": f"lidar scans/sec (ICP+EKF) on {args.rows}x{args.cols} sweeps",
            "value": K * S * world / dt, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "repeats": {"n": R, "reported": "median", "ms_per_step": [1e3 * v / K for v in rep_dt],
                        "values": [K * S * world / v for v in rep_dt], "spread_rel": (max(rep_dt) - min(rep_dt)) / dt,
                        "bit_identical_trajectories": (repeats_identical if R > 1 else None),
                        "note": "each repeat = cold start + W untimed warm-up sweeps + exactly K timed steps (barrier + synchronize on both sides)"},
            "config": {"workload": f"synthetic {args.rows}x{args.cols} sweeps, random-walk SE(3) GT, {mode_txt}, "
                                   f"min/max range {args.min_range}/{args.max_range} m, voxel {(args.voxel_size or args.max_range / 100):.2f} m"
                                   + (", sweeps as raw range images (u32 mm per pixel, 4 resident bytes per pixel; LUT on device)" if args.range_input else "")
                                   + (f", ray pattern jittered by +-{args.ray_jitter_deg:g} deg per ray and sweep" if args.ray_jitter_deg else "")
                                   + (f" [{args.workload_name}]" if args.workload_name else ""),
                       "workload_key": wkey, "code_id": cid,
                       "library_built_from_this_source_tree": cid == source_code_id(),  # false: an experiment build (flags) or a stale .so
                       "sequences_per_gpu": S, "sequence_seeds": seeds_txt,
                       "driver": "single sequence" if S == 1 else
                                 "free-running (one persistent launch, every sequence at its own pace)" if free else "lockstep (one launch per stage)",
                       "team_workgroups": team_g if free else None, "teams": teams if free else None,
                       "scheduling": (None if migr is None else {
                           "scans_run_by_a_team_of_another_xcd": sum(m["stolen"] for m in migr),
                           "cross_xcd_handovers": sum(m["cross_xcd_handovers"] for m in migr),
                           "note": "last repeat, warm-up included; a sequence whose previous scan ran on another XCD is handed over at agent scope"}),
                       "scans_per_sequence": n_total, "parallelism": f"{world} independent sequence shard(s), no data-path collective"},
            "per_rank_scans_per_s": {"values": per_rank, "min": min(per_rank), "mean": float(np.mean(per_rank)), "max": max(per_rank),
                                     "note": "each rank's own K steps / its own wall time; `value` uses the max-over-ranks clock"},
            "roofline": roof,
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
                # the timed scans alone: the warm-up's sweeps are cheaper (an empty, then a small map), so the means since the cold start
                # understate what a timed scan costs - and a "never idle" rate formed from them overstates what the teams could do
                "timed_scans_mean": [float(np.mean([c[i] for c in seq_clk_timed])) for i in range(6)],
                "timed_scans_mean_total": float(np.mean([sum(c[:5]) for c in seq_clk_timed])),
                "teams_busy_fraction": ((K * S / dt) * float(np.mean([sum(c[:5]) for c in seq_clk_timed])) * 1e-6 / max(teams, 1)) if (world == 1 and teams) else None,
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
            if S > 1 and PARITY_EXTRA_SEQS > 0:
                # two more sequences of the batch, picked by a fixed generator, against the oracle over their first sweeps
                from oracle import cpu as orc
                orc.set_threads(synth.usable_cores())
                pick = sorted(int(v) for v in np.random.default_rng(args.seed_base + S).choice(np.arange(1, S), size=min(PARITY_EXTRA_SEQS, S - 1), replace=False))
                m2 = min(n_total, PARITY_EXTRA_SWEEPS)
                worst = 0.0
                for j in pick:
                    _, _, kiss_j, res_j = _cpu_pass(seqs[j], m2, use_imu, 1e9, with_ekf,
                                                    {"voxel_size": args.voxel_size} if args.voxel_size else None)
                    est_j = outs[j]["res_poses"] if with_ekf else outs[j]["kiss_poses"]
                    worst = max(worst, float(np.linalg.norm(est_j[:m2, :3, 3] - res_j[:m2, :3, 3], axis=1).max()))
                orc.set_threads(1)
                line["parity_vs_oracle"]["extra_sequences"] = {"sequences": pick, "seeds": [args.seed_base + seq_ids[j] for j in pick],
                                                               "sweeps_each": m2, "max_dpos_m": worst}
        else:
            line["cpu_baseline"] = None
        if world == 1 and S > 1 and free and args.verify_all:
            # every sequence of the batch once more alone, with a team's workgroup count: bit for bit
            bad = []
            for j in range(S):
                one = core.SeqRunner(n_total, pps, n_imu, max_range=args.max_range, min_range=args.min_range, use_imu_prediction=use_imu,
                                     with_ekf=with_ekf, device_id=local_rank,
                                     **dict({k: v for k, v in icp_over.items() if k != "map_small_blocks"}, gn_workgroups=team_g, gn_lanes_per_point=8, gn_threads=args.gn_threads or 512,
                                            map_block_capacity=icp_over.get("map_block_capacity", 1 << 19) + icp_over.get("map_small_blocks", 0)))
                if lut is not None:
                    one.set_lut(lut)
                for k in range(n_total):
                    feed_sweep(one, k, seqs[j])
                one.upload_imu(seqs[j].imu[:n_imu] if with_ekf else np.zeros((0, 7)),
                               [seqs[j].imu_range_for_scan(k)[1] if with_ekf else 0 for k in range(n_total)])
                one.run(n_total)
                o1 = one.results()
                if not (np.array_equal(o1["kiss_poses"], outs[j]["kiss_poses"]) and o1["stats"] == outs[j]["stats"]):
                    bad.append(j)
                one.close()
            line["verify_all"] = {"sequences": S, "differ_from_their_single_runs": bad}
        if world == 1 and S > 1 and not args.no_single_sequence:
            # SURVEY.md 8(e) wants both figures: k sequences per GPU (`value`) and one.  Sequence 0 alone, same sweeps, through
            # the single-sequence latency pipeline (one sequence over the whole chip, 32-lane kernel), after the timed region.
            runner.close()  # (its 2 x 16 handles' streams would share hardware queues with the pipeline measured next)
            one = core.SeqRunner(n_total, pps, n_imu, max_range=args.max_range, min_range=args.min_range,
                                 use_imu_prediction=use_imu, with_ekf=with_ekf, device_id=local_rank,
                                 **{k: v for k, v in icp_over.items() if k not in ("gn_lanes_per_point", "gn_threads", "gn_workgroups", "map_small_blocks")})
            if lut is not None:
                one.set_lut(lut)
            for k in range(n_total):
                feed_sweep(one, k, sq)
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
                                             "backend": gather_backend}
            if args.dump_traj:
                np.savez(args.dump_traj, **{f"rank{r}_seq{j}": v for (r, j), v in gathered.items()},
                         seeds=np.array([[r, j, args.seed_base + (j if args.equal_work else r + world * j)] for (r, j) in gathered]))
        elif gather_err is not None:
            line["gathered_trajectories"] = {"error": gather_err}
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        if gather_err is not None:  # no gather at all (the fallback failed too): leave without the collective shutdown, and say so in the exit code
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(3)
        dist.barrier(group=ctl)
        if rccl_stalled:  # the rows came the other way; a thread is still inside RCCL: no collective shutdown, no atexit
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
