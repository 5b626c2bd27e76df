#!/bin/bash
# first GPU call of round 2: third-party probe, GPU tests, single-sequence / batched / self-launched bench lines
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
{ echo "== import kiss_icp"; python3 -c "import kiss_icp; print('kiss_icp', kiss_icp.__version__)" 2>&1 | tail -2
  echo "== pip download"; timeout 40 pip download kiss-icp==0.2.10 --no-deps -d /tmp/kd 2>&1 | tail -3
  echo "== pip list | grep -i kiss / ouster / rosbags"; pip list 2>/dev/null | grep -i -E "kiss|ouster|rosbags|open3d" || echo none
  nproc; rocminfo | grep -c gfx950; } > "$O/r02_a_probe.txt" 2>&1
timeout 2000 python3 -m pytest tests -m gpu -x -q > "$O/r02_a_pytest.txt" 2>&1; echo "pytest rc $?" >> "$O/r02_a_pytest.txt"
timeout 600 python3 bench.py > "$O/r02_a_bench_default.json" 2> "$O/r02_a_bench_default.err"; echo "rc $?" >> "$O/r02_a_bench_default.err"
timeout 600 python3 bench.py --seqs-per-gpu 8 --no-cpu-baseline > "$O/r02_a_bench_s8.json" 2> "$O/r02_a_bench_s8.err"; echo "rc $?" >> "$O/r02_a_bench_s8.err"
timeout 600 python3 bench.py --seqs-per-gpu 4 --no-cpu-baseline > "$O/r02_a_bench_s4.json" 2> "$O/r02_a_bench_s4.err"; echo "rc $?" >> "$O/r02_a_bench_s4.err"
timeout 600 python3 bench.py --gpus 2 --gn-wgs 64 --steps 100 > "$O/r02_a_bench_g2.json" 2> "$O/r02_a_bench_g2.err"; echo "rc $?" >> "$O/r02_a_bench_g2.err"
tail -3 "$O/r02_a_pytest.txt"; for f in default s8 s4 g2; do head -c 400 "$O/r02_a_bench_$f.json"; echo; tail -2 "$O/r02_a_bench_$f.err"; done; cat "$O/r02_a_probe.txt"
