#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
V=$PWD/tools/variants
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for u in 2 16; do
  PTL_LIB_PATH=$V/lib_um$u.so timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -k "equals_independent or long_run or handle_both or icp_only" > $O/pytest_um$u.log 2>&1; tail -3 $O/pytest_um$u.log
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05b/bench_default.json"))
print(d["value"], d["ms_per_step"], d.get("repeats"), d["roofline"].get("kernel_ms_per_launch"), d["single_sequence"]["value"])
PY
