#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=6 > "$O/r02_j_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_j_pytest.txt"
tail -12 "$O/r02_j_pytest.txt"
bash tools/profile_round.sh r02_j_default > "$O/r02_j_prof_default.txt" 2>&1
bash tools/profile_round.sh r02_j_single --seqs-per-gpu 1 > "$O/r02_j_prof_single.txt" 2>&1
bash tools/profile_round.sh r02_j_config5 --seqs-per-gpu 1 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 60 --warmup 20 --map-blocks 2097152 --map-table 8388608 --workload-name "config 5: dense map" > "$O/r02_j_prof_config5.txt" 2>&1
for t in default single config5; do head -c 250 "$O/r02_j_${t}_bench.json"; echo; tail -2 "$O/r02_j_${t}_bench.err"; done
