"""Diagnostic for the free-running map update at another points-per-thread value (make EXTRA="-DSEQ_UM=4 -DMAP_DIAG"): runs a batch
scan by scan (one launch per scan), and after every scan records per sequence the error flags, the ERR_TABLE site counters of a
MAP_DIAG build (dbg_sums[24..31]), the map's size and a digest of its sorted points.  Two runs (PTL_LIB_PATH = the variant / the
default build) are compared by `um_diag.py cmp A.json B.json`: first scan and sequence where the maps differ.

    PTL_LIB_PATH=... python tools/um_diag.py run OUT.json S N [TEAM_WGS]
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(out, S, n, team):
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import _lib as L, core, synth
    seqs = [synth.make_sequence(seed=1010 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, scans_per_launch=1, team_workgroups=team)
    for s, sq in enumerate(seqs):
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
    rec = dict(build=core.build_info(), geometry=b.team_geometry(), scans=[])
    hs = []
    for s in range(S):
        h = C.c_void_p()
        L.check(L.lib().ptl_batch_icp(b._h, s, C.byref(h)))
        hs.append(h)
    for k in range(n):
        err = None
        try:
            if k == 0:
                b.run(1)
            else:
                b.enqueue(1)
                b.wait()
        except Exception as e:  # noqa: BLE001
            err = str(e)
        row = dict(k=k, err=err, seqs=[])
        for s in range(S):
            dbg = (C.c_double * 32)()
            L.lib().ptl_icp_debug_sums(hs[s], dbg)
            nv, npnt = C.c_int64(), C.c_int64()
            L.lib().ptl_icp_map_size(hs[s], C.byref(nv), C.byref(npnt))
            pts = np.empty((max(npnt.value, 1) + 64, 3))
            w = C.c_int64()
            L.lib().ptl_icp_map_points(hs[s], L.dptr(pts), len(pts), C.byref(w))
            p = pts[:w.value]
            p = p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
            row["seqs"].append(dict(voxels=nv.value, points=npnt.value, exported=w.value, digest=hashlib.sha1(p.tobytes()).hexdigest()[:16],
                                    diag=[float(dbg[i]) for i in range(24, 32)]))
        rec["scans"].append(row)
        if err:
            print("scan", k, "error:", err)
            for s in range(S):
                d = row["seqs"][s]["diag"]
                if any(d[:4]):
                    print("  sequence", s, "sites (a: no slot, a: table 3/4 full, b: list too long, rebuild):", d[:4], "first:", d[4:7])
            break
    json.dump(rec, open(out, "w"))
    print("wrote", out, "scans", len(rec["scans"]), "build", rec["build"]["seq_u"], rec["geometry"])


def cmp(a, b):
    A, B = json.load(open(a)), json.load(open(b))
    for ra, rb in zip(A["scans"], B["scans"]):
        for s, (x, y) in enumerate(zip(ra["seqs"], rb["seqs"])):
            if (x["voxels"], x["points"], x["digest"]) != (y["voxels"], y["points"], y["digest"]):
                print(f"first difference: scan {ra['k']} sequence {s}: {x} vs {y}")
                return 1
    print("maps equal over", min(len(A["scans"]), len(B["scans"])), "scans x", len(A["scans"][0]["seqs"]), "sequences;", "errors:", [r["err"] for r in A["scans"] if r["err"]], [r["err"] for r in B["scans"] if r["err"]])
    return 0


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    else:
        sys.exit(cmp(sys.argv[2], sys.argv[3]))
