#!/bin/bash
# any counter set of the dominant kernel:  tools/pmc_any.sh KERNEL_PREFIX "CTR1 CTR2 ..." [bench args...]   (counters only)
set -euo pipefail
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
KP="${1:?usage: pmc_any.sh KERNEL_PREFIX \"COUNTERS\" [bench args]}"; CT="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/pmc_any"
# shellcheck disable=SC2086
rocprofv3 --pmc $CT --output-format csv -d "$R/gpurun_out/pmc_any" -o s -- python3 "$R/bench.py" --no-cpu-baseline --no-single-sequence --steps 40 --warmup 10 "$@" > /dev/null 2> "$R/gpurun_out/pmc_any.err" || tail -5 "$R/gpurun_out/pmc_any.err"
KP="$KP" D="$R/gpurun_out/pmc_any" python3 - <<'PY'
import collections, csv, glob, os
f = glob.glob(os.environ["D"] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no csv"); raise SystemExit
acc = collections.defaultdict(lambda: [set(), 0.0])
for r in csv.DictReader(open(f[0])):
    if r["Kernel_Name"].replace("void ", "").startswith(os.environ["KP"]):
        acc[r["Counter_Name"]][0].add(r["Dispatch_Id"]); acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
for k, v in acc.items():
    print("%-28s launches %3d  mean per launch %.4g" % (k, len(v[0]), v[1] / max(len(v[0]), 1)))
PY
rm -rf "$R/gpurun_out/pmc_any"
