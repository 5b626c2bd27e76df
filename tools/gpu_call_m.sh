#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_gpu_batch.py -m gpu -x -q > "$O/r02_m_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_m_pytest.txt"
tail -5 "$O/r02_m_pytest.txt"
for a in "--seqs-per-gpu 8" "--seqs-per-gpu 16" "--seqs-per-gpu 32" "--seqs-per-gpu 24"; do
timeout 900 python3 bench.py $a --no-cpu-baseline > "$O/r02_m_bench.json" 2> "$O/r02_m_bench.err"; echo "rc $?"
python3 - <<PY
import json
d=json.loads(open("$O/r02_m_bench.json").read())
print("$a", d["value"], d["roofline"]["avg_launch_us"], d["whole_scan"]["gn_share_of_wall"], d["roofline"]["frac"])
PY
done
