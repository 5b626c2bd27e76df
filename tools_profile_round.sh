cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r01_n_bench_default.json 2> gpurun_out/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_n -o kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_n_bench.json 2> $R/gpurun_out/prof_n.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_n_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 60 --warmup 20 > /dev/null 2> $R/gpurun_out/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_n_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 60 --warmup 20 > /dev/null 2> $R/gpurun_out/pmc_w.err
cd $R
find gpurun_out/prof_n -name "*kernel_stats.csv" | head -2
python tools_pmc_summary.py gpurun_out/r01_n_pmc_hbm_traffic.json FETCH_SIZE=gpurun_out/pmc_n_fetch WRITE_SIZE=gpurun_out/pmc_n_write
# keep only the small summaries
find gpurun_out/pmc_n_fetch gpurun_out/pmc_n_write -name "*.csv" -size +8M -delete
cp $(find gpurun_out/prof_n -name "*kernel_stats.csv" | head -1) gpurun_out/r01_n_kernel_stats_default_bench.csv
rm -rf gpurun_out/prof_n/*/*.db gpurun_out/pmc_n_fetch gpurun_out/pmc_n_write
find gpurun_out/prof_n -type f -size +4M -delete
head -c 600 gpurun_out/r01_n_bench_default.json; echo; head -5 gpurun_out/r01_n_kernel_stats_default_bench.csv
