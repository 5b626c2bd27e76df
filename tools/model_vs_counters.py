"""bench.py's executed-byte model against the counters, KERNEL BY KERNEL (the model was only ever checked in total).

The free-running kernel is one dispatch - its counters cannot be split - but the lockstep driver runs the same stage bodies and the same
Gauss-Newton body (kx_gn_loop8 = gn8_body) as separate launches, so rocprofv3 gives FETCH_SIZE / WRITE_SIZE / L2 requests per kernel:

    python3 tools/model_vs_counters.py run OUT.json                  (the profiled program: 32 sequences, lockstep, 40 sweeps)
    rocprofv3 --pmc FETCH_SIZE -d DIR_F -o f --output-format csv -- python3 tools/model_vs_counters.py run /dev/null      (and WRITE_SIZE, and the
                                                                      TCP_TCC_READ_REQ_sum / TCP_TCC_WRITE_REQ_sum pair, one pass each)
    python3 tools/model_vs_counters.py table OUT.json DIR_F DIR_W DIR_R > profiles/…_model_vs_counters.txt

Model side = EXEC_COST's unit costs (bytes a lane REQUESTS) x the run's counters, split by the kernel that issues them; counter side = HBM bytes
(2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction) and L2 read requests (128-byte lines) per scan of a sequence.  A ratio above 1 is
lines against requested bytes (a scattered 8-byte read moves a 128-byte line) less what the caches absorb."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S, N = 32, 40


def run(out):
    import numpy as np
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import core, synth
    seqs = [synth.make_sequence(seed=1000 + s, n_scans=N) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(N - 1)[1]
    b = core.BatchRunner(S, N, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, free_running=False)
    for s, sq in enumerate(seqs):
        for k in range(N):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(N)])
    b.run()
    tot = collections.Counter()
    for s in range(S):
        for st in b.results(s)["stats"]:
            for k in ("n_in", "n_valid", "n_down", "n_src", "map_voxels", "iterations"):
                tot[k] += st[k]
            tot["first_iteration_points"] += st["n_src"] if st["iterations"] > 0 else 0
        for k, v in b.exec_counters(s).items():
            tot[k] += v
    tot["scans"] = S * N
    json.dump(dict(tot), open(out, "w"), indent=1)
    print("scans", S * N, {k: round(v / (S * N), 1) for k, v in tot.items()}, file=sys.stderr)


def per_kernel(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]] += float(r["Counter_Value"])
    return acc


def table(stats, dir_f, dir_w, dir_r):
    import bench
    c = bench.EXEC_COST
    t = json.load(open(stats))
    n = t["scans"]
    p = {k: v / n for k, v in t.items()}  # per scan of a sequence
    G = 8  # workgroups of a sequence in the lockstep run (32 sequences: 64 per XCD / 4... kx_assign: J / 4)
    later = p["point_iterations"] - p["first_iteration_points"]
    gn = (c["source_read"] * p["first_iteration_points"] + c["point_iteration_later"] * later + c["search"] * p["searches"]
          - c["search_first_iteration_not_read"] * min(p["first_iteration_points"], p["searches"]) + c["row_rebuilt"] * p["rows_rebuilt"]
          + c["map_point_read"] * p["map_points_read"] + p["gn_iterations"] * G * (1 + G) * c["exchange_words_per_row"] * c["exchange_word"])
    model = collections.OrderedDict([
        ("kb_deskew_vds1", ("K1: f32 point 12 + slot word 4 per raw point, deskewed point 24 per valid point, 24 per voxel claim",
                            16 * p["n_in"] + 24 * p["n_valid"] + c["vds_claim"] * p["vds1_claims"])),
        ("kb_count_w1", ("K2 (lockstep only: the free-running kernel looks back instead): not in the model", None)),
        ("kb_compact_fd", ("K3: slot word 4 per raw point, winner index 4 per valid point, point read 24 + written 24 + slot released 12 per frame_down point",
                           4 * p["n_in"] + 4 * p["n_valid"] + 60 * p["n_down"])),
        ("kb_vds2_fd", ("K3b: point 24 + slot 4 per frame_down point, 24 per voxel claim", 28 * p["n_down"] + c["vds_claim"] * p["vds2_claims"])),
        ("kb_count_w2", ("(lockstep only): not in the model", None)),
        ("kb_compact_src", ("K4: slot 4 + index 4 per frame_down point, 48 + 12 per source point", 8 * p["n_down"] + c["source_point"] * p["n_src"])),
        ("kx_gn_loop8", ("Gauss-Newton loop (gn8_body, the free-running kernel's loop): rows, searches, rebuilt rows, stored points, exchange", gn)),
        ("kb_map_insert_a", ("insert a: 68 per frame_down point", 68 * p["n_down"])),
        ("kb_map_insert_b", ("insert b: 76 per frame_down point", 76 * p["n_down"])),
        ("kb_map_insert_c", ("insert c (lockstep / per-call only: the free-running kernel's prune pass publishes): not in the model", None)),
        ("kb_map_prune", ("prune: 36 per live voxel (the lockstep launch also reads the empty directory entries below the pool's capacity)", c["map_voxel"] * p["map_voxels"])),
        ("kb_scan_prologue", ("K0: the per-column deskew table, 96 per column", c["scan_column"] * 1024)),
        ("kb_ekf_step", ("filter step: < 3 KB of state", None)),
    ])
    F, W = per_kernel(dir_f, "FETCH_SIZE"), per_kernel(dir_w, "WRITE_SIZE")
    R, WR = per_kernel(dir_r, "TCP_TCC_READ_REQ_sum"), per_kernel(dir_r, "TCP_TCC_WRITE_REQ_sum")
    print(f"bench.py EXEC_COST against the counters, kernel by kernel: lockstep driver, {S} sequences x {N} sweeps (seeds 1000..{1000 + S - 1}, ICP + IMU-EKF), all launches;")
    print("per scan of a sequence.  model = bytes the lanes request; HBM = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024); L2 reads = TCP_TCC_READ_REQ x 128 B.")
    print(f"per scan: {p['n_in']:.0f} raw, {p['n_valid']:.0f} valid, {p['n_down']:.0f} frame_down, {p['n_src']:.0f} source points, {p['map_voxels']:.0f} live voxels, "
          f"{p['iterations']:.1f} iterations, {p['point_iterations']:.0f} point-iterations, {p['searches']:.0f} searches, {p['rows_rebuilt']:.0f} rows rebuilt, {p['map_points_read']:.0f} stored points read")
    print()
    print(f"{'kernel':<18} {'model MB':>9} {'HBM MB':>8} {'HBM/model':>9} {'L2 read MB':>10} {'L2 write req':>12}   terms")
    tm = th = 0.0
    for k, (what, m) in model.items():
        hbm = (2 * F.get(k, 0.0) + W.get(k, 0.0)) * 1024 / n
        l2r = R.get(k, 0.0) * 128 / n
        if m is not None:
            tm += m
            th += hbm
        print(f"{k:<18} {('%.2f' % (m / 1e6)) if m is not None else '-':>9} {hbm / 1e6:8.2f} {('%.2f' % (hbm / m)) if m else '-':>9} {l2r / 1e6:10.2f} {WR.get(k, 0.0) / n:12.0f}   {what}")
    others = sorted(set(F) | set(W) - set(model))
    rest = sum((2 * F.get(k, 0.0) + W.get(k, 0.0)) * 1024 / n for k in others if k not in model)
    print(f"{'modelled kernels':<18} {tm / 1e6:9.2f} {th / 1e6:8.2f} {th / tm:9.2f}")
    print(f"(kernels outside the table - resets, uploads' helpers, finish: {rest / 1e6:.2f} MB of HBM traffic per scan)")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        table(*sys.argv[2:6])
