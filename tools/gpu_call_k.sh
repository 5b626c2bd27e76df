#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_dewarp.py tests/test_gpu_dropin.py tests/test_tracking_diagnosis.py -m gpu -x -q > "$O/r02_k_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_k_pytest.txt"
tail -25 "$O/r02_k_pytest.txt"
bash tools/profile_round.sh r02_k_config5 --seqs-per-gpu 1 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 60 --warmup 20 --map-blocks 2097152 --map-table 8388608 --workload-name "config 5: dense map" > "$O/r02_k_prof_config5.txt" 2>&1
python3 - <<PY
import json
d=json.loads(open("$O/r02_k_config5_bench.json").read())
print(d["value"], d["roofline"], d.get("parity_vs_oracle"), d["cpu_baseline"])
PY
