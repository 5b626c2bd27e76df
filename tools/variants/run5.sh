#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05e/bench_default.json"))
print("default", d["value"], d["sequence_phases_us_per_scan"]["mean"], d["roofline"]["executed_bytes_per_scan"])
PY
C5="--team-wgs 4 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 40 --warmup 20 --map-table 33554432 --workload-name config5 --no-single-sequence --repeats 1 --no-cpu-baseline"
run() { # name, extra args
  n=$1; shift
  timeout 1200 python bench.py $C5 "$@" > $O/config5_$n.json 2> $O/config5_$n.err
  python - $n <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/r05e/config5_%s.json"%sys.argv[1]))
    c=d["roofline"]["executed_counters_per_scan"]
    print("config5", sys.argv[1], round(d["value"],1), [round(x) for x in d["sequence_phases_us_per_scan"]["mean"]], "slowest", round(d["sequence_phases_us_per_scan"]["slowest_sequence_total"]), "searches", round(c["searches"]), "map", d["map"]["voxels_end"], d["map"]["points_end"])
except Exception as e:
    print("config5", sys.argv[1], "failed", e); print(open("gpurun_out/r05e/config5_%s.err"%sys.argv[1]).read()[-600:])
PY
}
run one96 --seqs-per-gpu 96 --map-blocks 3145728
run two96 --seqs-per-gpu 96 --map-blocks 600000 --map-small-blocks 2200000
run two128 --seqs-per-gpu 128 --map-blocks 600000 --map-small-blocks 2200000
run two160 --seqs-per-gpu 160 --map-blocks 600000 --map-small-blocks 2200000
