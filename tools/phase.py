"""in-kernel phase clocks of the Gauss-Newton kernel (workgroup 0) and per-workgroup search-phase clocks.

Needs a library built with the clocks compiled in:  make -C ptudes-lab_amd/csrc clean && make -C ptudes-lab_amd/csrc PHASES=1
(the default build leaves them out: they cost registers and serialise on s_memtime; all values then read 0).
    python tools/phase.py [gn_workgroups gn_threads]
"""
import sys, ctypes as C, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd
from ptudes_lab_amd import core, synth, _lib as L
n=60
seq = synth.make_sequence(seed=1000, n_scans=n)
kw = dict(gn_workgroups=int(sys.argv[1]), gn_threads=int(sys.argv[2])) if len(sys.argv)>2 else {}
r = core.SeqRunner(n, seq.H*seq.W, seq.imu_range_for_scan(n-1)[1], max_range=70., min_range=1., use_imu_prediction=True, with_ekf=True, **kw)
for k in range(n): r.upload_scan(k, seq.scan(k))
r.upload_imu(seq.imu[:seq.imu_range_for_scan(n-1)[1]], [seq.imu_range_for_scan(k)[1] for k in range(n)])
r.run()
icp = C.c_void_p(); L.check(L.lib().ptl_seq_icp(r._h, C.byref(icp)))
out = (C.c_int64*8)(); L.check(L.lib().ptl_icp_gn_phases(icp, out))
o = np.array(list(out), dtype=float); it=o[5]
print("iters", it, "ticks/iter: nn %.0f wgred %.0f barrier %.0f gridred %.0f solve %.0f" % tuple(o[:5]/it), "total/iter", o[:5].sum()/it, "| leader: members in at +%.0f, group sum out at +%.0f (ticks after own publish)" % (o[6]/it, o[7]/it))

G = kw.get("gn_workgroups", 256)
wc = (C.c_int64 * (2 * G))(); L.check(L.lib().ptl_icp_gn_wg_clocks(icp, wc, G))
w = np.array(list(wc), dtype=float).reshape(2, G) / it
print("search phase per workgroup, ticks/iter (all waves): min %.0f median %.0f max %.0f; first wave: min %.0f median %.0f max %.0f" % (w[0].min(), np.median(w[0]), w[0].max(), w[1].min(), np.median(w[1]), w[1].max()))
print("by XCD slot (wg & 7), all-waves mean:", np.round([w[0][x::8].mean() for x in range(8)]))
print("slowest 8 workgroups:", np.argsort(w[0])[-8:], np.round(np.sort(w[0])[-8:]))
