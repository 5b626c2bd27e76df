import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd
from ptudes_lab_amd import core, synth, _lib as L
# PTL_TOOL_WORKLOAD=config5: BASELINE config 5 (64 x 2048 sweeps, 0.1 m voxels, 100 m range, two block classes) instead of the default workload
if os.environ.get("PTL_TOOL_WORKLOAD") == "config5":
    SEQ_KW = dict(H=64, W=2048, max_range=100.0)
    RUN_KW = dict(max_range=100.0, voxel_size=0.1, scan_cols=2048, map_block_capacity=600000, map_small_blocks=2200000, map_table_capacity=1 << 25)
    RUN_KW.update({k[4:].lower(): int(v) for k, v in os.environ.items() if k.startswith("PTL_KW_")})  # e.g. PTL_KW_MAP_TABLE_CAPACITY=...
else:
    SEQ_KW, RUN_KW = {}, {}
S=int(sys.argv[1]); n=int(sys.argv[3]) if len(sys.argv) > 3 else 40; TG=int(sys.argv[2]) if len(sys.argv) > 2 else 0
seqs=[synth.make_sequence(seed=1000+s, n_scans=n, **SEQ_KW) for s in range(S)]
n_imu=seqs[0].imu_range_for_scan(n-1)[1]
b=core.BatchRunner(S,n,seqs[0].H*seqs[0].W,n_imu,use_imu_prediction=True,with_ekf=True,team_workgroups=TG,**RUN_KW)
for s,sq in enumerate(seqs):
    for k in range(n): b.upload_scan(s,k,sq.scan(k))
    b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
b.run()
nscan_all = n
out=(C.c_int64*8)(); L.check(L.lib().ptl_batch_gn_phases(b._h,out))
o=np.array(list(out),dtype=float); it=o[5]
print("S",S,"grid iters",it,"ticks/iter: nn %.0f wgred %.0f barrier %.0f gridred %.0f solve %.0f"%tuple(o[:5]/it),"total",o[:5].sum()/it, "sum seq iters", sum(sum(st["iterations"] for st in b.results(s)["stats"]) for s in range(S)))

print("wg0 of seq 0: misses/iter %.1f, phase A until compaction %.0f ticks/iter" % (o[6]/it, o[7]/it))
icp = C.c_void_p()
G = 32
try:
    # per-workgroup point-loop ticks and misses of sequence 0 (needs ptl_batch_icp)
    L.check(L.lib().ptl_batch_icp(b._h, 0, C.byref(icp)))
    wc = (C.c_int64 * 512)(); L.check(L.lib().ptl_icp_gn_wg_clocks(icp, wc, 256))
    w = np.array(list(wc), dtype=float)
    print("point loop per workgroup, ticks/iter:", np.round(w[:G] / it).astype(int).tolist())
    print("misses per workgroup per iter:", np.round(w[G:2 * G] / it, 1).tolist())
    if w[40] > 0:
        print("searches (all workgroups of sequence 0): %.0f per scan, survivor rounds per search %.2f, row rebuilt (voxel changed / first iteration) %.1f %%" % (w[40] / n, w[41] / w[40], 100 * w[42] / w[40]))
    if w[52] > 0:
        print("point loop of workgroup 0: first iteration %.0f ticks per scan, the others %.0f per iteration" % (w[52] / nscan_all, w[53] / max(it - nscan_all, 1)))
    if w[59] > 0:
        print("workgroup 0 of sequence 0, iterations > 0: per chunk of phase A: wait for the loads + evaluation %.0f ticks, compaction %.0f; chunks per iteration %.1f;  per search pass %.0f ticks (%.1f passes per iteration)"
              % (w[56] / w[59], w[57] / w[59], w[59] / max(it - nscan_all, 1), w[58] / max(w[60], 1), w[60] / max(it - nscan_all, 1)))
    if w[70] > 0:
        print("serial tail of an iteration (workgroup 0 of sequence 0), ticks: sums from the moments %.0f | 6x6 solve %.0f | Exp + flags %.0f" % tuple(w[67:70] / w[70]))
    if w[66] > 0:
        print("one search (thread 0 of the first teams' workgroup 0, iterations > 0, every step waited out), ticks: row + key %.0f | rebuild %.0f | boxes + first round %.0f | survivors %.0f | reductions + answer row %.0f   (%d searches)"
              % tuple(list(w[61:66] / w[66]) + [int(w[66])]))
    if w[44] + w[45] > 0:
        print("bound of the others after a search: third-nearest candidate %.1f %%, box of a dropped voxel %.1f %%;  bound / winner's distance in [1,1.2) [1.2,1.5) [1.5,2) [2,3) [3,..): %s %%"
              % (100 * w[44] / (w[44] + w[45]), 100 * w[45] / (w[44] + w[45]), np.round(100 * w[46:51] / max(w[46:51].sum(), 1), 1).tolist()))
except Exception as e:
    print("no per-wg clocks:", e)

try:
    import ctypes
    class _DS(ctypes.Structure): pass
    # dbg_sums of sequence 0's DevState: phase-B sub-steps of workgroup 0 / group 0 (ticks summed over its passes)
    st = np.zeros(2048, dtype=np.uint8)
    from ptudes_lab_amd import _lib
    print("phase B sub-steps: see ptl_icp_debug_sums")
    ds = (C.c_double * 32)(); L.check(L.lib().ptl_icp_debug_sums(icp, ds))
    d = np.array(list(ds)); n = max(d[4], 1)
    nscan = max(sum(1 for st in b.results(0)["stats"] if st["iterations"] > 0), 1)
    print("misses of workgroup 0 by iteration index (mean per scan):", np.round(d[8:32] / nscan, 1).tolist())
    tot = max(d[4] + d[5] + d[6] + d[7], 1)
    print("searches of workgroup 0 after iteration 0: %.0f per scan; voxel changed %.1f %%, no answer stored %.1f %%, SAME neighbour again %.1f %%, another neighbour %.1f %%"
          % (tot / nscan, 100 * d[5] / tot, 100 * d[6] / tot, 100 * d[7] / tot, 100 * d[4] / tot))
except Exception as e:
    print("no sub-steps:", e)
