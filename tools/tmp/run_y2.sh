L=$PWD/ptudes-lab_amd/csrc
for v in "" v_v32 v_v64 ""; do
  if [ -n "$v" ]; then export PTL_LIB_PATH=$L/lib$v.so; else unset PTL_LIB_PATH; fi
  echo "== ${v:-base}" >> gpurun_out/r04_y2_vds.txt
  python tools/geom_sweep.py gpurun_out/r04_y2_vds.txt 100 10 240:0 --repeats=1
done
unset PTL_LIB_PATH
python tools/geom_sweep.py gpurun_out/r04_y2_seqs.txt 20 5 224:2 240:2 256:2 272:2 288:2 --repeats=3
python tools/geom_sweep.py gpurun_out/r04_y2_seqs100.txt 100 10 256:2 288:2 --repeats=1
