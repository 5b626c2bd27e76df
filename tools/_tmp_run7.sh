cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_handover.py tests/test_gpu_batch.py -m gpu -x -q > $O/r06_g_pytest_subset.txt 2>&1; tail -3 $O/r06_g_pytest_subset.txt
C5="--seqs-per-gpu 160 --team-wgs 4 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 40 --warmup 20 --map-blocks 600000 --map-small-blocks 2200000 --map-table 33554432 --workload-name config5 --no-single-sequence --no-cpu-baseline --repeats 1"
for m in -1 1 2; do
  PTL_SCHED_MARGIN=$m python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-single-sequence --no-cpu-baseline > $O/r06_g_margin${m}_bench.json 2> $O/r06_g_margin${m}_bench.err
done
for m in -1 1; do
  PTL_SCHED_MARGIN=$m python3 bench.py --steps 200 --warmup 20 --no-single-sequence --no-cpu-baseline --repeats 1 > $O/r06_g_margin${m}_bench_200.json 2> $O/r06_g_margin${m}_bench_200.err
  PTL_SCHED_MARGIN=$m python3 bench.py $C5 > $O/r06_g_margin${m}_config5_bench.json 2> $O/r06_g_margin${m}_config5_bench.err
done
python3 bench.py --steps 60 --warmup 10 --no-single-sequence --no-cpu-baseline --repeats 1 > $O/r06_g_product_bench_60.json 2> $O/r06_g_product_bench_60.err
PTL_LIB_PATH=$PWD/tools/variants/lib_blocksum_probe.so python3 bench.py --steps 60 --warmup 10 --no-single-sequence --no-cpu-baseline --repeats 1 > $O/r06_g_blocksum_bench_60.json 2> $O/r06_g_blocksum_bench_60.err
python3 bench.py --steps 60 --warmup 10 --no-single-sequence --no-cpu-baseline --repeats 1 > $O/r06_g_product2_bench_60.json 2> $O/r06_g_product2_bench_60.err
PTL_LIB_PATH=$PWD/tools/variants/lib_blocksum_probe.so python3 bench.py --steps 60 --warmup 10 --no-single-sequence --no-cpu-baseline --repeats 1 > $O/r06_g_blocksum2_bench_60.json 2> $O/r06_g_blocksum2_bench_60.err
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r06_g_*bench*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]; ph = d["sequence_phases_us_per_scan"]
        print(f.split("/")[-1], round(d["value"], 1), [round(v) for v in d["repeats"]["values"]], [round(x) for x in ph["mean"]], "slowest/mean seq", round(ph["slowest_sequence_total"]), round(ph["mean_sequence_total"]), d["config"]["scheduling"])
    except Exception as e:
        print(f, "failed", e); print(open(f.replace(".json", ".err")).read()[-1000:])
PY
