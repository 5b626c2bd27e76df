#!/bin/bash
# SQ instruction / cycle counters of the dominant kernel:  tools/pmc_sq.sh KERNEL_PREFIX [bench args...]  (counters only, one set per run)
set -euo pipefail
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
KP="${1:?usage: pmc_sq.sh KERNEL_PREFIX [bench args]}"
shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  rm -rf "$R/gpurun_out/pmc_sq"
  # shellcheck disable=SC2086
  rocprofv3 --pmc $set --output-format csv -d "$R/gpurun_out/pmc_sq" -o s -- python3 "$R/bench.py" --no-cpu-baseline --no-single-sequence --repeats 1 --steps 40 --warmup 10 "$@" > /dev/null 2> "$R/gpurun_out/pmc_sq.err"
  KP="$KP" D="$R/gpurun_out/pmc_sq" python3 - <<'PY'
import collections, csv, glob, os
f = glob.glob(os.environ["D"] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no csv"); raise SystemExit
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if r["Kernel_Name"].replace("void ", "").startswith(os.environ["KP"]):
        acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
for k, v in acc.items():
    print("%-24s launches %3d  mean per launch %.4g" % (k, v[0], v[1] / v[0]))
PY
done
rm -rf "$R/gpurun_out/pmc_sq"
