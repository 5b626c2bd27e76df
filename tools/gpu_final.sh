#!/bin/bash
# end-of-round check on the GPU box: the whole GPU suite, smoke, and the round's profile sets (bench line + rocprofv3 stats + PMC)
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
O="$R/gpurun_out"; mkdir -p "$O"
TAG="${1:-final}"
cd "$R"
timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=5 > "$O/${TAG}_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/${TAG}_pytest.txt"; tail -9 "$O/${TAG}_pytest.txt"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh ${TAG}_default > "$O/${TAG}_prof_default.txt" 2>&1
bash tools/profile_round.sh ${TAG}_s8 --seqs-per-gpu 8 > "$O/${TAG}_prof_s8.txt" 2>&1
bash tools/profile_round.sh ${TAG}_s16 --seqs-per-gpu 16 --no-cpu-baseline > "$O/${TAG}_prof_s16.txt" 2>&1
bash tools/profile_round.sh ${TAG}_s32_lockstep --seqs-per-gpu 32 --lockstep --no-cpu-baseline --no-single-sequence > "$O/${TAG}_prof_s32_lockstep.txt" 2>&1
for t in default s8 s16 s32_lockstep; do python3 - <<PY
import json
d=json.loads(open("$O/${TAG}_${t}_bench.json").read())
r=d["roofline"]; print("$t", round(d["value"],1), "frac %.3f launch %.1f us traffic %s"%(r["frac"],r["avg_launch_us"],r["traffic"]), d.get("parity_vs_oracle"), d["cpu_baseline"] and d["cpu_baseline"]["value"], d.get("single_sequence",{}).get("value"))
PY
done
