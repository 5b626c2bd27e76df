"""MI355X-native lidar-odometry core: drop-in for the ptudes-lab `ekf-bench ouster` pose path
(KISS-ICP scan-to-local-map registration + ptudes ES-EKF), backed by hand-written HIP kernels for
gfx950 behind the C-ABI declared in include/ptudes_mi.h.

Module layout mirrors the reference so `ptudes.X` becomes `ptudes_lab_amd.X`:
  kiss.py          KissICPWrapper            (reference src/ptudes/kiss.py)
  ins/es_ekf.py    ESEKF                     (reference src/ptudes/ins/es_ekf.py)
  ins/data.py      IMU, NavState, calc_ate   (reference src/ptudes/ins/data.py)
  utils.py         pose-file writers/reader  (reference src/ptudes/utils.py)
  cli/ekf_bench.py `ekf-bench` commands      (reference src/ptudes/cli/ekf_bench.py)
  sequence.py      whole-sequence device-resident runner (the reference's driver loop, ekf_bench.py:493-563)
  synth.py         synthetic sweeps + IMU (test/bench data; the reference ships none)
"""
__version__ = "0.1.0"
