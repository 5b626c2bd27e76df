#!/bin/bash
# Round 6's measurements on the GPU box (run through gpurun):   tools/round6_evidence.sh PART TAG   -> gpurun_out/TAG_*
# Before, in the build container (the diagnostic builds travel with the snapshot; *.so is git-ignored):
#   make -C ptudes-lab_amd/csrc OUT=$PWD/tools/variants/lib_phases.so PHASES=1 && make -C ptudes-lab_amd/csrc OUT=$PWD/tools/variants/lib_stages.so STAGES=1
# T:  the GPU suite + smoke
# C5: BASELINE config 5 measured like the default workload (none of this existed): phase clocks of the Gauss-Newton
#     loop, stage clocks, SQ and L2 counters - 160 dense sequences on teams of 4
# D:  the default workload's line (driver's command), kernel stats, dispatches, HBM counters, the same command with that pass in place,
#     200 steps, SQ / L2 counters, phase and stage clocks
# B5: config 5's line, kernel stats, dispatches, HBM counters, the same command with that pass in place, stage clocks
# J:  the jitter workload (the stage-bound regime): line, kernel stats, HBM counters
# X:  --icp-only, --range-input (driver's form), --verify-all, the per-call loop
# S:  tools/stream_feed.py: the sweep ring, resident against streamed
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
PART="${1:?T, C5, D, B5, J, X or S}"; TAG="${2:-r06_a}"
O="$R/gpurun_out"; mkdir -p "$O"; V="$R/tools/variants"
cd "$R"
C5="--seqs-per-gpu 160 --team-wgs 4 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 40 --warmup 20 --map-blocks 600000 --map-small-blocks 2200000 --map-table 33554432 --workload-name config5 --no-single-sequence"
case "$PART" in
T)
  timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=5 > "$O/${TAG}_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/${TAG}_pytest.txt"; tail -4 "$O/${TAG}_pytest.txt"
  python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a "$O/${TAG}_pytest.txt"
  ;;
C5)
  PTL_TOOL_WORKLOAD=config5 PTL_LIB_PATH="$V/lib_phases.so" timeout 900 python3 tools/phase_batch.py 160 4 60 > "$O/${TAG}_config5_gn_phase_clocks.txt" 2>&1; tail -3 "$O/${TAG}_config5_gn_phase_clocks.txt" | cut -c1-400
  PTL_TOOL_WORKLOAD=config5 PTL_LIB_PATH="$V/lib_stages.so" timeout 900 python3 tools/stage_clocks.py 160 4 60 > "$O/${TAG}_config5_stage_clocks_teams_of_4.txt" 2>&1; tail -3 "$O/${TAG}_config5_stage_clocks_teams_of_4.txt" | cut -c1-400
  timeout 1500 bash tools/pmc_sq.sh kx_seq_run $C5 > "$O/${TAG}_config5_sq_counters_kx_seq_run.txt" 2>&1; cat "$O/${TAG}_config5_sq_counters_kx_seq_run.txt"
  { timeout 600 bash tools/pmc_any.sh kx_seq_run "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" --repeats 1 $C5; timeout 600 bash tools/pmc_any.sh kx_seq_run "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" --repeats 1 $C5; } > "$O/${TAG}_config5_l2_counters_kx_seq_run.txt" 2>&1; cat "$O/${TAG}_config5_l2_counters_kx_seq_run.txt"
  ;;
D)
  bash tools/profile_round.sh "$TAG" --gpus 1 --steps 20 --warmup 5 > "$O/${TAG}_prof.txt" 2>&1; tail -2 "$O/${TAG}_prof.txt" | cut -c1-300
  cp "$O/${TAG}_pmc_hbm_traffic.json" "$R/profiles/"   # (on the box: the line below finds this build's counter pass)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/${TAG}_bench_with_counters.json" 2> "$O/${TAG}_bench_with_counters.err"
  python3 bench.py --steps 200 --warmup 20 > "$O/${TAG}_bench_200steps.json" 2> "$O/${TAG}_bench_200steps.err"
  bash tools/pmc_sq.sh kx_seq_run > "$O/${TAG}_sq_counters_kx_seq_run.txt" 2>&1
  { bash tools/pmc_any.sh kx_seq_run "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" --repeats 1; bash tools/pmc_any.sh kx_seq_run "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" --repeats 1; } > "$O/${TAG}_l2_counters_kx_seq_run.txt" 2>&1
  PTL_LIB_PATH="$V/lib_phases.so" python3 tools/phase_batch.py 240 2 > "$O/${TAG}_gn_phase_clocks.txt" 2>&1
  PTL_LIB_PATH="$V/lib_stages.so" python3 tools/stage_clocks.py 240 2 30 > "$O/${TAG}_stage_clocks_teams_of_2.txt" 2>&1
  ;;
J)
  JIT="--steps 20 --warmup 5 --ray-jitter-deg 0.3 --workload-name jitter"
  bash tools/profile_round.sh "${TAG}_jitter" $JIT > "$O/${TAG}_jitter_prof.txt" 2>&1; tail -2 "$O/${TAG}_jitter_prof.txt" | cut -c1-300
  cp "$O/${TAG}_jitter_pmc_hbm_traffic.json" "$R/profiles/"
  python3 bench.py $JIT --no-cpu-baseline > "$O/${TAG}_jitter_bench_with_counters.json" 2> "$O/${TAG}_jitter_bench_with_counters.err"
  ;;
X)
  python3 bench.py --icp-only --steps 100 --warmup 10 > "$O/${TAG}_bench_icp_only.json" 2> "$O/${TAG}_bench_icp_only.err"
  python3 bench.py --steps 20 --warmup 5 --range-input --no-single-sequence --cpu-budget 10 > "$O/${TAG}_bench_range_input.json" 2> "$O/${TAG}_bench_range_input.err"
  python3 bench.py --steps 20 --warmup 5 --verify-all --repeats 1 --no-cpu-baseline --no-single-sequence > "$O/${TAG}_bench_verify_all.json" 2> "$O/${TAG}_bench_verify_all.err"
  python3 tools/percall.py > "$O/${TAG}_percall.txt" 2>&1; tail -4 "$O/${TAG}_percall.txt" | cut -c1-160
  ;;
S)
  # a recording of any length at the full batch width: the sweep ring fed while the kernel runs (pageable, then page-locked host buffer)
  PTL_STREAM_PIN=0 timeout 1500 python3 tools/stream_feed.py 240 120 32 16 > "$O/${TAG}_stream_feed_pageable.txt" 2>&1; cut -c1-300 "$O/${TAG}_stream_feed_pageable.txt"
  PTL_STREAM_PIN=1 timeout 1500 python3 tools/stream_feed.py 240 120 32 16 > "$O/${TAG}_stream_feed_pinned.txt" 2>&1; cut -c1-300 "$O/${TAG}_stream_feed_pinned.txt"
  ;;
B5)
  bash tools/profile_round.sh "${TAG}_config5" $C5 > "$O/${TAG}_config5_prof.txt" 2>&1; tail -2 "$O/${TAG}_config5_prof.txt" | cut -c1-300
  cp "$O/${TAG}_config5_pmc_hbm_traffic.json" "$R/profiles/"
  python3 bench.py $C5 --cpu-budget 20 > "$O/${TAG}_config5_bench_with_counters.json" 2> "$O/${TAG}_config5_bench_with_counters.err"
  PTL_TOOL_WORKLOAD=config5 PTL_LIB_PATH="$V/lib_stages.so" timeout 900 python3 tools/stage_clocks.py 160 4 60 > "$O/${TAG}_config5_stage_clocks_teams_of_4.txt" 2>&1
  PTL_TOOL_WORKLOAD=config5 PTL_LIB_PATH="$V/lib_phases.so" timeout 900 python3 tools/phase_batch.py 160 4 60 > "$O/${TAG}_config5_gn_phase_clocks.txt" 2>&1
  ;;
esac
python3 - "$O" "$TAG" <<'PY'
import glob, json, sys
O, T = sys.argv[1:3]
for f in sorted(glob.glob(f"{O}/{T}_*bench*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        ph = d.get("sequence_phases_us_per_scan") or {}
        print(f.split("/")[-1], round(d["value"], 1), "frac", round(r["frac"], 3), "stale", r["traffic_stale"], (r.get("traffic_scans") or {}).get("how"), "exec", r["executed_frac"] and round(r["executed_frac"], 3),
              "MB/scan", r["traffic"] and r.get("scans_per_launch") and round(r["traffic"] / r["scans_per_launch"] / 1e6, 1), (d.get("parity_vs_oracle") or {}).get("max_dpos_m"), [round(x) for x in ph.get("mean", [])],
              "timed", [round(x) for x in ph.get("timed_scans_mean", [])], "busy", ph.get("teams_busy_fraction"), d.get("verify_all"), (d.get("single_sequence") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
PY
