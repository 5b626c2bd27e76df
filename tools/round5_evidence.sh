#!/bin/bash
# Round 5's measurements of the final state on the GPU box (run through gpurun, part A or B):
#   tools/round5_evidence.sh A|B TAG      -> gpurun_out/TAG_*   (copy what should be judged into profiles/)
# Before part A, in the build container (the diagnostic builds travel with the snapshot; *.so is git-ignored):
#   make -C ptudes-lab_amd/csrc OUT=$PWD/tools/variants/lib_phases.so PHASES=1 && make -C ptudes-lab_amd/csrc OUT=$PWD/tools/variants/lib_stages.so STAGES=1
# A: GPU suite + smoke, the driver's command (bench line, kernel stats, dispatches, HBM counters), the same command again with that
#    counter pass in place (roofline.frac = measured), 200 steps, SQ / L2 counters, phase and stage clocks (diagnostic builds in
#    tools/variants/), the --icp-only line (BASELINE config 2 as written)
# B: BASELINE config 5 (160 dense sequences, two block classes), the 1 000-sweep run, --verify-all, the per-call loop
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
PART="${1:?A or B}"; TAG="${2:-r05_z}"
O="$R/gpurun_out"; mkdir -p "$O"; V="$R/tools/variants"
cd "$R"
C5="--seqs-per-gpu 160 --team-wgs 4 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 40 --warmup 20 --map-blocks 600000 --map-small-blocks 2200000 --map-table 33554432 --workload-name config5 --no-single-sequence"
if [ "$PART" = A ]; then
  timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=5 > "$O/${TAG}_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/${TAG}_pytest.txt"; tail -4 "$O/${TAG}_pytest.txt"
  python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a "$O/${TAG}_pytest.txt"
  bash tools/profile_round.sh "$TAG" --gpus 1 --steps 20 --warmup 5 > "$O/${TAG}_prof.txt" 2>&1; tail -2 "$O/${TAG}_prof.txt" | cut -c1-300
  cp "$O/${TAG}_pmc_hbm_traffic.json" "$R/profiles/"   # (on the box: the lines below find this build's counter pass)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/${TAG}_bench_with_counters.json" 2> "$O/${TAG}_bench_with_counters.err"
  python3 bench.py --steps 200 --warmup 20 > "$O/${TAG}_bench_200steps.json" 2> "$O/${TAG}_bench_200steps.err"
  bash tools/pmc_sq.sh kx_seq_run > "$O/${TAG}_sq_counters_kx_seq_run.txt" 2>&1
  { bash tools/pmc_any.sh kx_seq_run "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" --repeats 1; bash tools/pmc_any.sh kx_seq_run "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" --repeats 1; } > "$O/${TAG}_l2_counters_kx_seq_run.txt" 2>&1
  PTL_LIB_PATH="$V/lib_phases.so" python3 tools/phase_batch.py 240 2 > "$O/${TAG}_gn_phase_clocks.txt" 2>&1
  PTL_LIB_PATH="$V/lib_stages.so" python3 tools/stage_clocks.py 240 2 30 > "$O/${TAG}_stage_clocks_teams_of_2.txt" 2>&1
  python3 bench.py --icp-only --steps 100 --warmup 10 > "$O/${TAG}_bench_icp_only.json" 2> "$O/${TAG}_bench_icp_only.err"
  python3 - "$O" "$TAG" <<'PY'
import json, sys
O, T = sys.argv[1:3]
for f in ("bench", "bench_with_counters", "bench_200steps", "bench_icp_only"):
    try:
        d = json.load(open(f"{O}/{T}_{f}.json")); r = d["roofline"]
        print(f, round(d["value"]), "frac", round(r["frac"], 3), "stale", r["traffic_stale"], "exec", round(r["executed_frac"], 3), "MB/scan", r["traffic"] and round(r["traffic"] / r["scans_per_launch"] / 1e6, 1), [round(x) for x in d["sequence_phases_us_per_scan"]["mean"]])
    except Exception as e:
        print(f, "failed", e)
PY
else
  bash tools/profile_round.sh "${TAG}_config5" $C5 > "$O/${TAG}_config5_prof.txt" 2>&1; tail -2 "$O/${TAG}_config5_prof.txt" | cut -c1-300
  cp "$O/${TAG}_config5_pmc_hbm_traffic.json" "$R/profiles/"
  python3 bench.py $C5 --cpu-budget 20 > "$O/${TAG}_config5_bench_with_counters.json" 2> "$O/${TAG}_config5_bench_with_counters.err"
  python3 bench.py --seqs-per-gpu 128 --steps 990 --warmup 10 --repeats 1 --no-single-sequence --cpu-budget 10 > "$O/${TAG}_bench_1000scans_s128.json" 2> "$O/${TAG}_bench_1000scans_s128.err"
  python3 bench.py --steps 20 --warmup 5 --verify-all --repeats 1 --no-cpu-baseline --no-single-sequence > "$O/${TAG}_bench_verify_all.json" 2> "$O/${TAG}_bench_verify_all.err"
  python3 tools/percall.py > "$O/${TAG}_percall.txt" 2>&1
  python3 - "$O" "$TAG" <<'PY'
import json, sys
O, T = sys.argv[1:3]
for f in ("config5_bench", "config5_bench_with_counters", "bench_1000scans_s128", "bench_verify_all"):
    try:
        d = json.load(open(f"{O}/{T}_{f}.json")); r = d["roofline"]
        print(f, round(d["value"], 1), "frac", round(r["frac"], 3), "stale", r["traffic_stale"], "MB/scan", r["traffic"] and round(r["traffic"] / r["scans_per_launch"] / 1e6, 1), d.get("parity_vs_oracle"), d.get("verify_all"))
    except Exception as e:
        print(f, "failed", e)
PY
  tail -4 "$O/${TAG}_percall.txt" | cut -c1-200
fi
