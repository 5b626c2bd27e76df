cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_sq
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq -o s -- python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 > /dev/null 2> $R/gpurun_out/pmc_sq.err
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)
if not f: print("no csv for $set"); raise SystemExit
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if r["Kernel_Name"].startswith("void k_gn_loop"):
        acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
for k, v in acc.items(): print("%-24s launches %3d  mean per launch %.4g" % (k, v[0], v[1] / v[0]))
PY
done
rm -rf $R/gpurun_out/pmc_sq
