import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
gn = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id'), r.get('Stream_Id')) for r in rows if 'k_gn_loop' in r['Kernel_Name']]
gn.sort()
print(len(gn), 'gn kernels; queues', sorted({g[2] for g in gn}))
ov = 0; tot = 0
for i in range(1, len(gn)):
    a, b = gn[i-1], gn[i]
    tot += b[1]-b[0]
    ov += max(0, min(a[1], b[1]) - b[0])
print('total gn time %.1f ms, overlapped with predecessor %.1f ms' % (tot/1e6, ov/1e6))
for g in gn[200:208]: print(g[0]-gn[200][0], g[1]-gn[200][0], g[2], g[3])
import collections
byq = collections.defaultdict(list)
for g in gn: byq[g[2]].append(g)
t0 = gn[0][0]
for q, v in byq.items():
    print('queue', q, 'n', len(v), 'first start %.2f ms last end %.2f ms' % ((v[0][0]-t0)/1e6, (v[-1][1]-t0)/1e6))
allk = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'][:30]) for r in rows)
i0 = len(allk)//2
for k in allk[i0:i0+30]: print('%.1f %.1f q%s %s' % ((k[0]-t0)/1e3, (k[1]-t0)/1e3, k[2], k[3]))
