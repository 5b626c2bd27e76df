#!/bin/bash
# end-of-round check on the GPU box: the whole GPU suite, smoke, and the round's profile set of the driver's command
#   tools/gpu_final.sh TAG     -> gpurun_out/TAG_pytest.txt, TAG_bench.json, TAG_kernel_stats.csv, TAG_pmc_hbm_traffic.json, TAG_sq_counters_kx_seq_run.txt
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
O="$R/gpurun_out"; mkdir -p "$O"
TAG="${1:-final}"
cd "$R"
timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=5 > "$O/${TAG}_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/${TAG}_pytest.txt"; tail -9 "$O/${TAG}_pytest.txt"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh "$TAG" --gpus 1 --steps 20 --warmup 5 > "$O/${TAG}_prof.txt" 2>&1; tail -2 "$O/${TAG}_prof.txt"
bash tools/pmc_sq.sh kx_seq_run > "$O/${TAG}_sq_counters_kx_seq_run.txt" 2>&1
