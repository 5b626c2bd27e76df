#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
V=$PWD/tools/variants
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -k "rccl or default_is" > $O/pytest_contract.log 2>&1; tail -4 $O/pytest_contract.log; grep -n "gather failed" $O/pytest_contract.log | head -3
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_handover.py -x -q > $O/pytest_batch.log 2>&1; tail -3 $O/pytest_batch.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench_default.json"))
print("default", d["value"], d["sequence_phases_us_per_scan"]["mean"], d["roofline"]["executed_bytes_per_scan"])
PY
C5="--seqs-per-gpu 96 --team-wgs 4 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 40 --warmup 20 --map-blocks 3145728 --map-table 33554432 --workload-name config5 --no-single-sequence --repeats 1 --no-cpu-baseline"
for v in default keep3 keep5; do
  if [ $v = default ]; then unset PTL_LIB_PATH; else export PTL_LIB_PATH=$V/lib_$v.so; fi
  timeout 900 python bench.py $C5 > $O/config5_$v.json 2> $O/config5_$v.err
  python - $v <<'PY'
import json,sys
d=json.load(open("gpurun_out/r05d/config5_%s.json"%sys.argv[1]))
c=d["roofline"]["executed_counters_per_scan"]
print("config5", sys.argv[1], round(d["value"],1), [round(x) for x in d["sequence_phases_us_per_scan"]["mean"]], "searches", round(c["searches"]), "rebuilt", round(c["rows_rebuilt"]), "mappts", round(c["map_points_read"]), "exec MB", round(d["roofline"]["executed_bytes_per_scan"]/1e6))
PY
done
