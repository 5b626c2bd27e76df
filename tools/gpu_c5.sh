#!/bin/bash
# config 5 (dense map) with a build variant:  tools/gpu_c5.sh "MAKEVARS" [extra bench args]
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
VARS="$1"; shift
make -C ptudes-lab_amd/csrc -B libptudes_mi.so $VARS > "$O/c5_make.txt" 2>&1 || { tail -5 "$O/c5_make.txt"; exit 1; }
timeout 900 python3 bench.py --seqs-per-gpu 1 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 60 --warmup 20 --map-blocks 2097152 --map-table 8388608 --gn-lanes 8 --gn-threads 512 --cpu-budget 3 "$@" > "$O/c5_bench.json" 2> "$O/c5_bench.err" || tail -3 "$O/c5_bench.err"
python3 - <<PY
import json
d=json.loads(open("$O/c5_bench.json").read())
print("[$VARS $@]", round(d["value"],1), "GN us", round(d["roofline"]["avg_launch_us"],1), "iters", d["whole_scan"]["mean_gn_iterations"], d.get("parity_vs_oracle"))
PY
make -C ptudes-lab_amd/csrc -B libptudes_mi.so > /dev/null 2>&1
