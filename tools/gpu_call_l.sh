#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "whole_chip or lanes" > "$O/r02_l_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_l_pytest.txt"
tail -5 "$O/r02_l_pytest.txt"
C5="--seqs-per-gpu 1 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 60 --warmup 20 --map-blocks 2097152 --map-table 8388608"
for a in "$C5 --gn-lanes 8 --gn-threads 512 --cpu-budget 6" "--seqs-per-gpu 1 --gn-lanes 8 --gn-threads 512 --no-cpu-baseline" "--seqs-per-gpu 1 --gn-lanes 8 --gn-threads 512 --gn-wgs 64 --no-cpu-baseline"; do
timeout 900 python3 bench.py $a > "$O/r02_l_bench.json" 2> "$O/r02_l_bench.err"; echo "rc $?"
python3 - <<PY
import json
d=json.loads(open("$O/r02_l_bench.json").read())
print("$a"[-60:], d["value"], d["roofline"]["avg_launch_us"], d["whole_scan"]["gn_share_of_wall"], d["whole_scan"]["mean_gn_iterations"], d.get("parity_vs_oracle"))
PY
done
