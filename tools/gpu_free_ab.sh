#!/bin/bash
# A/B of a build variant of the free-running kernel on the GPU box:  tools/gpu_free_ab.sh "MAKEVAR=VALUE ..." S [n] [key=value ...]
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
VARS="$1"; shift
make -C ptudes-lab_amd/csrc -B libptudes_mi.so $VARS > "$O/ab_make.txt" 2>&1 || { tail -5 "$O/ab_make.txt"; exit 1; }
timeout 900 python3 tools/free_vs_lockstep.py "$@"
make -C ptudes-lab_amd/csrc -B libptudes_mi.so > /dev/null 2>&1
