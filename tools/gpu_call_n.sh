#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline > "$O/r02_n_torchrun1.json" 2> "$O/r02_n_torchrun1.err"; echo "torchrun rc $?"
tail -3 "$O/r02_n_torchrun1.err"; python3 - <<PY
import json
d=json.loads([l for l in open("$O/r02_n_torchrun1.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d.get("gathered_trajectories"), d["per_rank_scans_per_s"])
PY
timeout 900 python3 -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q > "$O/r02_n_pytest.txt" 2>&1; tail -3 "$O/r02_n_pytest.txt"
bash tools/profile_round.sh r02_n_default > "$O/r02_n_prof_default.txt" 2>&1
bash tools/profile_round.sh r02_n_s8 --seqs-per-gpu 8 > "$O/r02_n_prof_s8.txt" 2>&1
bash tools/profile_round.sh r02_n_config5 --seqs-per-gpu 1 --rows 64 --cols 2048 --max-range 100 --voxel-size 0.1 --steps 60 --warmup 20 --map-blocks 2097152 --map-table 8388608 --gn-lanes 8 --gn-threads 512 --workload-name "config 5: dense map" > "$O/r02_n_prof_config5.txt" 2>&1
for t in default s8 config5; do python3 - <<PY
import json
d=json.loads(open("$O/r02_n_${t}_bench.json").read())
r=d["roofline"]; print("$t", round(d["value"],1), "frac %.3f launch %.1f us traffic %s"%(r["frac"],r["avg_launch_us"],r["traffic"]), d.get("parity_vs_oracle"), d["cpu_baseline"] and d["cpu_baseline"]["value"])
PY
done
