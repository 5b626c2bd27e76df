import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd
from ptudes_lab_amd import core, synth, _lib as L
S=int(sys.argv[1]); n=40
seqs=[synth.make_sequence(seed=1000+s, n_scans=n) for s in range(S)]
n_imu=seqs[0].imu_range_for_scan(n-1)[1]
b=core.BatchRunner(S,n,seqs[0].H*seqs[0].W,n_imu,use_imu_prediction=True,with_ekf=True)
for s,sq in enumerate(seqs):
    for k in range(n): b.upload_scan(s,k,sq.scan(k))
    b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
b.run()
out=(C.c_int64*8)(); L.check(L.lib().ptl_batch_gn_phases(b._h,out))
o=np.array(list(out),dtype=float); it=o[5]
print("S",S,"grid iters",it,"ticks/iter: nn %.0f wgred %.0f barrier %.0f gridred %.0f solve %.0f"%tuple(o[:5]/it),"total",o[:5].sum()/it, "sum seq iters", sum(sum(st["iterations"] for st in b.results(s)["stats"]) for s in range(S)))
