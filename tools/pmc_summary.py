"""Summarise rocprofv3 --pmc passes (counter_collection csv) per kernel -> JSON for profiles/.

    python tools/pmc_summary.py OUT.json BENCH_LINE.json FETCH_SIZE=<dir> WRITE_SIZE=<dir>

BENCH_LINE.json is the line bench.py printed in the FETCH_SIZE pass: its config.workload_key is recorded, and bench.py
reports `roofline.traffic` only from a summary whose key equals the run's own.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B for
wide coalesced reads (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): the fetch side is doubled (an upper
bound for this kernel's 8-byte gathers, an uncalibrated width) and WRITE_SIZE is taken as reported."""
import collections
import csv
import glob
import json
import sys

out = {}
try:
    line = json.loads([ln for ln in open(sys.argv[2]).read().splitlines() if ln.strip().startswith("{")][-1])
    out["workload_key"] = line["config"]["workload_key"]
    out["dominant_kernel"] = line["roofline"]["kernel"]
    out["bench_value_under_profiler"] = line["value"]
except Exception as e:  # noqa: BLE001
    out["workload_key"] = None
    out["error"] = f"no bench line: {e!r}"
for arg in sys.argv[3:]:
    name, d = arg.split("=")
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    out[name] = {k: {"launches": v[0], "mean_per_launch_KB": v[1] / v[0]} for k, v in acc.items()}
dom = out.get("dominant_kernel") or "k_gn_loop"
gn = [k for k in out.get("FETCH_SIZE", {}) if k.startswith(dom)]
if gn:
    fk = sum(out["FETCH_SIZE"][k]["mean_per_launch_KB"] * out["FETCH_SIZE"][k]["launches"] for k in gn) / sum(out["FETCH_SIZE"][k]["launches"] for k in gn) * 1024
    wk_n = sum(out.get("WRITE_SIZE", {}).get(k, {}).get("launches", 0) for k in gn)
    wk = (sum(out["WRITE_SIZE"][k]["mean_per_launch_KB"] * out["WRITE_SIZE"][k]["launches"] for k in gn if k in out.get("WRITE_SIZE", {})) / wk_n * 1024) if wk_n else 0.0
    out["hbm_bytes_per_launch"] = {"kernel": dom, "fetch_reported": fk, "fetch_x2_gfx950": 2 * fk, "write_reported": wk,
                                   "total_corrected": 2 * fk + wk}
    out["traffic_bytes_per_launch"] = 2 * fk + wk  # what bench.py reports as roofline.traffic for this workload
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("workload_key", "hbm_bytes_per_launch")}))
