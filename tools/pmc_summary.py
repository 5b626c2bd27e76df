"""Summarise rocprofv3 --pmc passes (counter_collection csv) per kernel -> JSON for profiles/.

    python tools/pmc_summary.py OUT.json BENCH_LINE.json FETCH_SIZE=<dir> WRITE_SIZE=<dir>

BENCH_LINE.json is the line bench.py printed in the FETCH_SIZE pass: its config.workload_key is recorded, and bench.py
reports `roofline.traffic` only from a summary whose key equals the run's own.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B for
wide coalesced reads (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): the fetch side is doubled (an upper
bound for this kernel's 8-byte gathers, an uncalibrated width) and WRITE_SIZE is taken as reported.
The per-launch figure of the dominant kernel is the mean over its launches of the TIMED region - the last `steps`
launches (the free-running kernel: the last `roofline.launches`, a launch carries many scans) - not the warm-up's.
For the free-running kernel the summary also records bytes per scan (per launch / the line's roofline.scans_per_launch)."""
import collections
import csv
import glob
import json
import sys

out = {}
try:
    line = json.loads([ln for ln in open(sys.argv[2]).read().splitlines() if ln.strip().startswith("{")][-1])
    out["workload_key"] = line["config"]["workload_key"]
    out["code_id"] = line["config"].get("code_id")  # the kernel sources the pass ran on (bench.py code_id)
    out["dominant_kernel"] = line["roofline"]["kernel"]
    out["bench_value_under_profiler"] = line["value"]
    # the scans the pass covers: a free-running summary is per scan, and a scan's traffic depends on WHICH scans (the map grows, the
    # iteration count falls) - bench.py takes the figure as it is only for the same warm-up and step counts and otherwise scales it by
    # the ratio of executed bytes per scan between the two runs
    out["warmup"], out["steps"] = line.get("warmup"), line.get("steps")
    out["executed_bytes_per_scan"] = line["roofline"].get("executed_bytes_per_scan")
    n_timed = line["roofline"]["launches"] if line["roofline"]["kernel"] == "kx_seq_run" else line["steps"]
except Exception as e:  # noqa: BLE001
    out["workload_key"] = None
    out["error"] = f"no bench line: {e!r}"
    n_timed = 0
dom = out.get("dominant_kernel") or "k_gn_loop"
for arg in sys.argv[3:]:
    name, d = arg.split("=")
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    per_dispatch = collections.defaultdict(float)  # the dominant kernel: one value per dispatch (a counter row per XCD / dimension is summed)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
        if k.startswith(dom):
            per_dispatch[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    out[name] = {k: {"launches": v[0], "mean_per_launch_KB": v[1] / v[0]} for k, v in acc.items()}
    vals = [per_dispatch[d] for d in sorted(per_dispatch)]
    if n_timed and len(vals) >= n_timed:
        vals = vals[-n_timed:]
    out[name + "_timed"] = {"launches": len(vals), "mean_per_launch_KB": (sum(vals) / len(vals)) if vals else None}
ft, wt = out.get("FETCH_SIZE_timed", {}), out.get("WRITE_SIZE_timed", {})
if ft.get("mean_per_launch_KB") is not None:
    fk = ft["mean_per_launch_KB"] * 1024
    wk = (wt.get("mean_per_launch_KB") or 0.0) * 1024
    out["hbm_bytes_per_launch"] = {"kernel": dom, "fetch_reported": fk, "fetch_x2_gfx950": 2 * fk, "write_reported": wk,
                                   "total_corrected": 2 * fk + wk}
    out["traffic_bytes_per_launch"] = 2 * fk + wk  # what bench.py reports as roofline.traffic for this workload
    # the free-running kernel: one launch carries the whole run, so the figure that carries over to another step count of the same
    # workload is bytes per SCAN (bench.py scales it by its own scans per launch)
    try:
        spl = line["roofline"].get("scans_per_launch")
        if dom == "kx_seq_run" and spl:
            out["scans_per_launch"] = spl
            out["traffic_bytes_per_scan"] = (2 * fk + wk) / spl
            out["fetch_x2_bytes_per_scan"] = 2 * fk / spl
            out["write_bytes_per_scan"] = wk / spl
    except Exception:  # noqa: BLE001
        pass
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("workload_key", "hbm_bytes_per_launch")}))
