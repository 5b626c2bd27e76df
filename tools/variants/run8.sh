#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
V=$PWD/tools/variants
PTL_LIB_PATH=$V/lib_opt4.so timeout 2400 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_handover.py -x -q > $O/pytest_opt.log 2>&1; tail -4 $O/pytest_opt.log
for v in noopt opt4 opt1 opt16 noopt opt4; do
  PTL_LIB_PATH=$V/lib_$v.so python tools/geom_sweep.py $O/ab_$v.txt 40 10 240:0 --repeats=1 2>&1 | cut -c1-330
done
for v in noopt opt4; do
  PTL_LIB_PATH=$V/lib_$v.so python tools/geom_sweep.py $O/ab20_$v.txt 20 5 240:0 --repeats=3 2>&1 | cut -c1-330
done
