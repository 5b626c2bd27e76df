"""Summarise rocprofv3 --pmc passes (counter_collection csv) per kernel -> JSON for profiles/.

    python tools_pmc_summary.py OUT.json FETCH_SIZE=<dir> WRITE_SIZE=<dir>

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B for
wide coalesced reads (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): `hbm_bytes_per_launch_k_gn_loop` applies
that x2 to the fetch side (an upper bound for this kernel's 8-byte gathers) and takes WRITE_SIZE as reported."""
import csv, glob, json, sys, collections

out = {}
for arg in sys.argv[2:]:
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
gn = [k for k in out.get("FETCH_SIZE", {}) if k.startswith("k_gn_loop")]
if gn:
    fk = out["FETCH_SIZE"][gn[0]]["mean_per_launch_KB"] * 1024
    wk = out.get("WRITE_SIZE", {}).get(gn[0], {}).get("mean_per_launch_KB", 0.0) * 1024
    out["hbm_bytes_per_launch_k_gn_loop"] = {"fetch_reported": fk, "fetch_x2_gfx950": 2 * fk, "write_reported": wk,
                                             "total_corrected": 2 * fk + wk}
    out["k_gn_loop_traffic_bytes_per_launch"] = 2 * fk + wk  # what bench.py reports as roofline.traffic
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out.get("hbm_bytes_per_launch_k_gn_loop")))
