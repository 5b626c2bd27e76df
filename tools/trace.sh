#!/bin/bash
# kernel timeline of a bench run -> per-scan gaps:  tools/trace.sh [bench args...]
set -euo pipefail
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/trace_x"
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_x" -o kt -- python3 "$R/bench.py" --no-cpu-baseline "$@" > /dev/null 2> "$R/gpurun_out/trace_x.err"
cd "$R" && python3 tools/gaps.py gpurun_out/trace_x
find gpurun_out/trace_x -type f -size +30M -delete
