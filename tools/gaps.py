"""Mean hand-over gaps of the scan pipeline from a rocprofv3 --kernel-trace csv dir:
GN end -> K0 start, K4 end -> GN start, GN end -> EKF start, EKF end -> GN start, map end -> GN start."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
def name(k): return k[2].split('(')[0].replace('void ', '')
gn = [k for k in rows if name(k).startswith('k_gn_loop')]
out = {k: [] for k in ('gn_end->k0', 'k4_end->gn', 'gn_end->ekf', 'ekf_end->gn', 'map_end->gn', 'k0..k4', 'gn')}
for a, b in zip(gn[20:-1], gn[21:]):
    mid = [k for k in rows if a[1] <= k[0] < b[0]]
    g = lambda n: [k for k in mid if name(k) == n]
    k0, k4, ekf, pr = g('k_scan_prologue'), g('k_compact_src'), g('k_ekf_step'), g('k_map_prune')
    if not (k0 and k4): continue
    out['gn_end->k0'].append(k0[0][0] - a[1]); out['k4_end->gn'].append(b[0] - k4[0][1]); out['k0..k4'].append(k4[0][1] - k0[0][0])
    out['gn'].append(a[1] - a[0])
    if ekf: out['gn_end->ekf'].append(ekf[0][0] - a[1]); out['ekf_end->gn'].append(b[0] - ekf[0][1])
    if pr: out['map_end->gn'].append(b[0] - pr[0][1])
for k, v in out.items():
    if v: print('%-12s mean %7.1f us  median %7.1f  (n=%d)' % (k, np.mean(v) / 1e3, np.median(v) / 1e3, len(v)))
