#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q -k "batch or lanes or whole_chip" > "$O/r02_p_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_p_pytest.txt"
tail -4 "$O/r02_p_pytest.txt"
for a in "--seqs-per-gpu 8" "--seqs-per-gpu 16" "--seqs-per-gpu 32"; do
timeout 900 python3 bench.py $a --no-cpu-baseline > "$O/r02_p_bench.json" 2> "$O/r02_p_bench.err"; echo "rc $?"
python3 - <<PY
import json
d=json.loads(open("$O/r02_p_bench.json").read())
print("$a", d["value"], d["roofline"]["avg_launch_us"], d["whole_scan"]["gn_share_of_wall"], d["roofline"]["frac"])
PY
done
make -C ptudes-lab_amd/csrc -B libptudes_mi.so PHASES=1 > "$O/r02_p_make.txt" 2>&1
{ python3 tools/phase_batch.py 8;} > "$O/r02_p_phases.txt" 2>&1
make -C ptudes-lab_amd/csrc -B libptudes_mi.so > /dev/null 2>&1
cat "$O/r02_p_phases.txt"
