#!/bin/bash
# One round's measurements on the GPU box (run through gpurun):
#   tools/profile_round.sh TAG [bench args...]        e.g.  tools/profile_round.sh r02_a --seqs-per-gpu 8
# -> gpurun_out/TAG_bench.json            the bench line (with cpu_baseline)
#    gpurun_out/TAG_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (no cpu_baseline)
#    gpurun_out/TAG_pmc_hbm_traffic.json  FETCH_SIZE / WRITE_SIZE passes (separate runs, counters only) + the workload key
# Copy what should be judged into profiles/.
set -euo pipefail
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
TAG="${1:?usage: profile_round.sh TAG [bench args]}"
shift
OUT="$R/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
rm -rf "$OUT/prof_$TAG" "$OUT/pmc_${TAG}_fetch" "$OUT/pmc_${TAG}_write"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o kt -- python3 "$R/bench.py" --no-cpu-baseline --repeats 1 "$@" > "$OUT/${TAG}_prof_bench.json" 2> "$OUT/${TAG}_prof.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_${TAG}_fetch" -o f -- python3 "$R/bench.py" --no-cpu-baseline --repeats 1 "$@" > "$OUT/${TAG}_pmc_bench.json" 2> "$OUT/${TAG}_pmc_f.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_${TAG}_write" -o w -- python3 "$R/bench.py" --no-cpu-baseline --repeats 1 "$@" > /dev/null 2> "$OUT/${TAG}_pmc_w.err"
cd "$R"
cp "$(find "$OUT/prof_$TAG" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
# every dispatch of the dominant kernel (the free-running kernel has a short warm-up launch and the timed one: the stats' average mixes them)
python3 - "$(find "$OUT/prof_$TAG" -name '*kernel_trace.csv' | head -1)" "$OUT/${TAG}_prof_bench.json" "$OUT/${TAG}_dominant_dispatches.json" <<'PY'
import csv, json, sys
line = json.loads([ln for ln in open(sys.argv[2]).read().splitlines() if ln.strip().startswith("{")][-1])
dom = line["roofline"]["kernel"]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].replace("void ", "").startswith(dom)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
timed = d[-line["roofline"]["launches"]:] if dom == "kx_seq_run" else d[-line["steps"]:]
json.dump({"kernel": dom, "dispatches": len(d), "durations_us": d if len(d) <= 16 else None, "timed_dispatches": len(timed),
           "timed_mean_us": sum(timed) / max(len(timed), 1), "bench_avg_launch_us_same_run": line["roofline"]["avg_launch_us"],
           "bench_value_under_profiler": line["value"]}, open(sys.argv[3], "w"), indent=1)
PY
python3 tools/pmc_summary.py "$OUT/${TAG}_pmc_hbm_traffic.json" "$OUT/${TAG}_pmc_bench.json" "FETCH_SIZE=$OUT/pmc_${TAG}_fetch" "WRITE_SIZE=$OUT/pmc_${TAG}_write"
# keep only the small summaries
rm -rf "$OUT/prof_$TAG" "$OUT/pmc_${TAG}_fetch" "$OUT/pmc_${TAG}_write"
head -c 700 "$OUT/${TAG}_bench.json"; echo; head -6 "$OUT/${TAG}_kernel_stats.csv"
