"""tools/layout_sim.py (host arithmetic behind DESIGN.md section 8's layout estimates) keeps running: a three-sweep map of the default
workload, its fill and the table rows."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_sim_reports_fill_and_table_occupancy():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "layout_sim.py"), "default", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    out = r.stdout
    m = re.search(r"-> (\d+) voxels, (\d+) points, ([\d.]+) points per voxel", out)
    assert m and int(m.group(1)) > 1000 and int(m.group(2)) > int(m.group(1)) and 1.0 < float(m.group(3)) <= 20.0
    rows = re.findall(r"table 2\^(\d+) slots .*?: ([\d.]+)% of the lines taken, ([\d.]+)% of the bricks off their home line", out)
    assert [int(a) for a, _, _ in rows] == list(range(21, 29))
    taken = [float(b) for _, b, _ in rows]
    off = [float(c) for _, _, c in rows]
    assert all(x >= y for x, y in zip(taken, taken[1:])) and all(x >= y - 1e-9 for x, y in zip(off, off[1:]))  # a bigger table is never worse
