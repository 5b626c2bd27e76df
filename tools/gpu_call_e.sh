#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q --durations=5 -k "batch or lanes" > "$O/r02_e_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_e_pytest.txt"
tail -6 "$O/r02_e_pytest.txt"
bash tools/profile_round.sh r02_e_s8_L8 --seqs-per-gpu 8 --steps 40 --warmup 20 --no-cpu-baseline > "$O/r02_e_prof.txt" 2>&1
head -8 "$O/r02_e_s8_L8_kernel_stats.csv"
python3 - <<PY
import json
d=json.loads(open("$O/r02_e_s8_L8_bench.json").read())
print(d["value"], d["roofline"]["avg_launch_us"], d["whole_scan"])
p=json.load(open("$O/r02_e_s8_L8_pmc_hbm_traffic.json")); print(p.get("hbm_bytes_per_launch"))
PY
