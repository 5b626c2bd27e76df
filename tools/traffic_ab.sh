#!/bin/bash
# A/B of library builds on the GPU box: scans/s (plain run) and HBM bytes per scan of kx_seq_run (FETCH_SIZE / WRITE_SIZE passes)
#   tools/traffic_ab.sh "bench args" name1=libpath1 [name2=libpath2 ...]      -> gpurun_out/traffic_ab.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
ARGS="$1"; shift
OUT="$R/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for nv in "$@"; do
  name="${nv%%=*}"; lib="${nv#*=}"
  export PTL_LIB_PATH="$lib"
  # shellcheck disable=SC2086
  python3 "$R/bench.py" --no-cpu-baseline --no-single-sequence $ARGS > "$OUT/ab_${name}.json" 2> "$OUT/ab_${name}.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/ab_pmc_$c"
    # shellcheck disable=SC2086
    rocprofv3 --pmc $c --output-format csv -d "$OUT/ab_pmc_$c" -o p -- python3 "$R/bench.py" --no-cpu-baseline --no-single-sequence $ARGS > "$OUT/ab_${name}_pmc.json" 2>> "$OUT/ab_${name}.err"
  done
  python3 "$R/tools/pmc_summary.py" "$OUT/ab_${name}_traffic.json" "$OUT/ab_${name}_pmc.json" "FETCH_SIZE=$OUT/ab_pmc_FETCH_SIZE" "WRITE_SIZE=$OUT/ab_pmc_WRITE_SIZE" > /dev/null
  rm -rf "$OUT/ab_pmc_FETCH_SIZE" "$OUT/ab_pmc_WRITE_SIZE"
  python3 - "$name" "$OUT/ab_${name}.json" "$OUT/ab_${name}_traffic.json" <<'PY' | tee -a "$OUT/traffic_ab.txt"
import json, sys
b = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); t = json.load(open(sys.argv[3]))
r = b["roofline"]; ph = b["sequence_phases_us_per_scan"]["mean"]
print("%-10s %7.0f scans/s  exec %.1f MB/scan (GN %.1f stages %.1f)  HBM %.1f MB/scan (fetch x2 %.1f, write %.1f) = %.2f TB/s at the plain run's rate  K0-4 %.0f GN %.0f map %.0f us" % (
    sys.argv[1], b["value"], r["executed_bytes_per_scan"] / 1e6, r["executed_split_per_scan"]["gauss_newton"] / 1e6, r["executed_split_per_scan"]["stages"] / 1e6,
    t["traffic_bytes_per_scan"] / 1e6, t["fetch_x2_bytes_per_scan"] / 1e6, t["write_bytes_per_scan"] / 1e6, t["traffic_bytes_per_scan"] * b["value"] / 1e12, ph[0], ph[2], ph[4]))
PY
done
