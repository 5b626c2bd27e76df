#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
make -C ptudes-lab_amd/csrc -B libptudes_mi.so PHASES=1 > "$O/r02_f_make.txt" 2>&1
{ python3 tools/phase_batch.py 8; python3 tools/phase_batch.py 1;} > "$O/r02_f_phases.txt" 2>&1
make -C ptudes-lab_amd/csrc -B libptudes_mi.so > /dev/null 2>&1
cat "$O/r02_f_phases.txt"
