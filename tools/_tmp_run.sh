cd $GRAFT_REPO_ROOT
make -C ptudes-lab_amd/csrc -B libptudes_mi.so PHASES=1 > gpurun_out/ab_make.txt 2>&1 || tail -5 gpurun_out/ab_make.txt
timeout 600 python3 tools/phase_batch.py 16 2>&1 | grep -v "^point loop\|^misses per" | cut -c1-400
