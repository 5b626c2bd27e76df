#!/bin/bash
# A/B of a build variant on the GPU box:  tools/gpu_ab.sh "MAKEVAR=VALUE ..." [S values...]
# builds the library with the given make variables, runs the batch / lanes parity tests and bench for each S, restores the default build
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
VARS="$1"; shift
make -C ptudes-lab_amd/csrc -B libptudes_mi.so $VARS > "$O/ab_make.txt" 2>&1 || { tail -5 "$O/ab_make.txt"; exit 1; }
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q -k "batch or lanes or whole_chip" > "$O/ab_pytest.txt" 2>&1; echo "[$VARS] pytest rc $?"; tail -2 "$O/ab_pytest.txt"
for S in "$@"; do
timeout 900 python3 bench.py --seqs-per-gpu $S --no-cpu-baseline > "$O/ab_bench.json" 2> "$O/ab_bench.err" || tail -3 "$O/ab_bench.err"
python3 - <<PY
import json
d=json.loads(open("$O/ab_bench.json").read())
print("[$VARS] S=$S", round(d["value"],1), "GN us", round(d["roofline"]["avg_launch_us"],1), "share", round(d["whole_scan"]["gn_share_of_wall"],3))
PY
done
make -C ptudes-lab_amd/csrc -B libptudes_mi.so > /dev/null 2>&1
