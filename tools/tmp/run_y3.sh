L=$PWD/ptudes-lab_amd/csrc
python -m pytest tests/test_gpu_handover.py tests/test_gpu_batch.py -x -q -m gpu > gpurun_out/r04_y3_pytest_subset.txt 2>&1
tail -3 gpurun_out/r04_y3_pytest_subset.txt
for v in "" x_spec3 x_spec1 x_surv3 x_kc5 x_keep2 x_u2_8 ""; do
  if [ -n "$v" ]; then export PTL_LIB_PATH=$L/lib$v.so; else unset PTL_LIB_PATH; fi
  echo "== ${v:-base}" >> gpurun_out/r04_y3_variants.txt
  python tools/geom_sweep.py gpurun_out/r04_y3_variants.txt 100 10 240:0 --repeats=1
done
