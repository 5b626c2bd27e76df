#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
bash tools/profile_round.sh r02_d_s8_L8 --seqs-per-gpu 8 --steps 40 --warmup 20 --no-cpu-baseline
bash tools/profile_round.sh r02_d_s8_L32 --seqs-per-gpu 8 --steps 40 --warmup 20 --no-cpu-baseline --gn-lanes 32
cat gpurun_out/r02_d_s8_L8_pmc_hbm_traffic.json | head -60
