"""Team-geometry sweep of the free-running kernel on the GPU box: bench.py per configuration, one compact row each.

    python tools/geom_sweep.py OUT.txt STEPS WARMUP  S:team_wgs[:gn_wgs:gn_threads] ...

e.g.  python tools/geom_sweep.py gpurun_out/geom.txt 60 10 48:0 96:4 96:8:512:256 192:2 256:1
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path, steps, warm = sys.argv[1], sys.argv[2], sys.argv[3]
extra = [a for a in sys.argv[4:] if a.startswith("--")]
rows = []
with open(out_path, "a") as out:
    for spec in [a for a in sys.argv[4:] if not a.startswith("--")]:
        f = spec.split(":")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", warm, "--seqs-per-gpu", f[0],
               "--team-wgs", f[1], "--no-cpu-baseline", "--no-single-sequence"] + extra
        if len(f) > 2:
            cmd += ["--gn-wgs", f[2], "--gn-threads", f[3]]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT)
        wall = time.time() - t0
        line = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        if r.returncode or not line:
            msg = f"{spec}: rc {r.returncode} after {wall:.0f} s: {r.stderr[-600:]}"
        else:
            d = json.loads(line[-1])
            ro, ph = d["roofline"], d.get("sequence_phases_us_per_scan") or {}
            m = ph.get("mean", [0] * 6)
            msg = (f"{spec:>16}  {d['value']:8.0f} scans/s  team {d['config'].get('team_workgroups')} x{d['config'].get('teams')}  "
                   f"launch {ro['avg_launch_us'] / 1e3:8.1f} ms  frac {ro['frac']:.3f} alg {ro['algorithmic_frac']:.2f}  exec/scan {(ro.get('executed_bytes_per_scan') or 0) / 1e6:6.1f} MB  "
                   f"phases us K0-4 {m[0]:.0f} w {m[1]:.0f} GN {m[2]:.0f} w {m[3]:.0f} map {m[4]:.0f} filt {m[5]:.0f}  "
                   f"scan tot mean {ph.get('mean_sequence_total', 0):.0f} slowest {ph.get('slowest_sequence_total', 0):.0f}  it {d['whole_scan']['mean_gn_iterations']:.1f}  wall {wall:.0f} s")
        print(msg, flush=True)
        out.write(msg + "\n")
        out.flush()
