"""Which HIP runtime the product library ends up on when torch is in the process, and what that costs.
usage: tools/torch_order.py MODE [bench args]   MODE: plain | torch_first | torch_cuda_first | lib_first"""
import os, sys
mode = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if mode == "torch_first":
    import torch
elif mode == "torch_cuda_first":
    import torch
    torch.cuda.set_device(0); torch.zeros(4, device="cuda"); torch.cuda.synchronize()
elif mode == "lib_first":
    import ptudes_lab_amd
    from ptudes_lab_amd import _lib
    _lib.lib()
    import torch
    torch.cuda.set_device(0); torch.zeros(4, device="cuda"); torch.cuda.synchronize()
elif mode in ("dist_nccl", "dist_gloo", "dist_nccl_lazy"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    if mode == "dist_nccl":
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    elif mode == "dist_nccl_lazy":
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    else:
        dist.init_process_group(backend="gloo", rank=0, world_size=1)
    del os.environ["MASTER_ADDR"]
elif mode == "omp1":
    os.environ["OMP_NUM_THREADS"] = "1"
maps = open("/proc/self/maps").read()
print(mode, "runtimes:", sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l}), file=sys.stderr)
import bench
bench.main()
