#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1200 python3 -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q > "$O/r02_o_pytest.txt" 2>&1; tail -15 "$O/r02_o_pytest.txt"
