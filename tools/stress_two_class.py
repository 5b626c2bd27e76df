"""Long run of the two block classes against the one-class layout on the DENSE default workload (nearly every voxel outgrows its small block: the
migration pass, both free stacks and the in-kernel table rebuild are exercised every scan): 16 sequences x 220 sweeps, launches of 50, a rebuild every
64 scans, pools sized close to the need - every pose and every per-scan statistic must be bit-equal.  profiles/r05_f_two_class_stress.txt"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd
from ptudes_lab_amd import core, synth
S, n = 16, 220
seqs = [synth.make_sequence(seed=1170 + s, n_scans=n) for s in range(S)]
n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
res = {}
for name, kw in (("one", dict(map_block_capacity=65536)), ("two", dict(map_block_capacity=45056, map_small_blocks=12288))):
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, rebuild_every=64, scans_per_launch=50, **kw)
    for s, sq in enumerate(seqs):
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
    b.run()
    res[name] = [b.results(s) for s in range(S)]
    print(name, "voxels at the end", [r["stats"][-1]["map_voxels"] for r in res[name]][:6], "max voxels", max(st["map_voxels"] for r in res[name] for st in r["stats"]))
    b.close()
bad = [s for s in range(S) if not (np.array_equal(res["one"][s]["kiss_poses"], res["two"][s]["kiss_poses"]) and res["one"][s]["stats"] == res["two"][s]["stats"])]
print("sequences that differ between one and two block classes over", n, "sweeps:", bad)
