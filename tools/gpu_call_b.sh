#!/bin/bash
# GPU call b: full GPU test suite, then phase clocks of the batched kernel (PHASES build, rebuilt on the box)
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 2400 python3 -m pytest tests -m gpu -x -q --durations=8 > "$O/r02_b_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_b_pytest.txt"
make -C ptudes-lab_amd/csrc -B libptudes_mi.so PHASES=1 > "$O/r02_b_make.txt" 2>&1
{ python3 tools/phase_batch.py 8; python3 tools/phase_batch.py 4; python3 tools/phase.py 32 1024; python3 tools/phase.py; } > "$O/r02_b_phases.txt" 2>&1
make -C ptudes-lab_amd/csrc -B libptudes_mi.so > /dev/null 2>&1
tail -15 "$O/r02_b_pytest.txt"; cat "$O/r02_b_phases.txt"
