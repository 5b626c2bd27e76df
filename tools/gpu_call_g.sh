#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q -k "batch or lanes" > "$O/r02_g_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_g_pytest.txt"
tail -4 "$O/r02_g_pytest.txt"
timeout 600 python3 bench.py --seqs-per-gpu 8 --no-cpu-baseline > "$O/r02_g_bench_s8.json" 2> "$O/r02_g_bench_s8.err"; echo "rc $?"
python3 - <<PY
import json
d=json.loads(open("$O/r02_g_bench_s8.json").read())
print(d["value"], d["roofline"]["avg_launch_us"], d["whole_scan"])
PY
make -C ptudes-lab_amd/csrc -B libptudes_mi.so PHASES=1 > "$O/r02_g_make.txt" 2>&1
{ python3 tools/phase_batch.py 8;} > "$O/r02_g_phases.txt" 2>&1
make -C ptudes-lab_amd/csrc -B libptudes_mi.so > /dev/null 2>&1
cat "$O/r02_g_phases.txt"
