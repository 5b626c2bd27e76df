"""print a window of the kernel timeline (start, end in us, queue, name) from a rocprofv3 --kernel-trace csv dir"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
allk = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'][:28]) for r in rows)
t0 = allk[0][0]
i0 = int(len(allk) * 0.6)
for k in allk[i0:i0 + int(sys.argv[2]) if len(sys.argv) > 2 else i0 + 40]:
    print('%.1f %.1f (%.1f) q%s %s' % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[2], k[3]))
