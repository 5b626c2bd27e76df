#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q --durations=5 -k "batch or lanes or icp_only or config5" > "$O/r02_c_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_c_pytest.txt"
timeout 600 python3 bench.py --seqs-per-gpu 8 --no-cpu-baseline > "$O/r02_c_bench_s8.json" 2> "$O/r02_c_bench_s8.err"; echo "rc $?" >> "$O/r02_c_bench_s8.err"
tail -12 "$O/r02_c_pytest.txt"; head -c 300 "$O/r02_c_bench_s8.json"; echo; tail -3 "$O/r02_c_bench_s8.err"
python3 - <<PY
import json
d=json.loads(open("$O/r02_c_bench_s8.json").read())
print(d["value"], d["roofline"]["avg_launch_us"], d["whole_scan"])
PY
