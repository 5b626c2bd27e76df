"""ctypes binding of libptudes_mi.so (the C-ABI declared in include/ptudes_mi.h).

The library is the only compute backend: loading fails loudly when it has not been built, and every
handle constructor raises when no HIP device is present.  Nothing here imports torch.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTL_LIB_PATH: a diagnostic / experiment build of the same library (csrc/Makefile: make OUT=... PHASES=1 ...), A/B runs on the GPU box
LIB_PATH = os.environ.get("PTL_LIB_PATH") or os.path.join(_HERE, "csrc", "libptudes_mi.so")

PTL_F32, PTL_F64 = 0, 1
c_d_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)


ABI_VERSION = 6  # PTL_ABI_VERSION of the include/ptudes_mi.h this binding was written against


class _Cfg(C.Structure):
    """A configuration struct of the header: starts with {struct_size, abi_version}, which the CALLER fills in before the library
    sees the struct (PTL_CFG_INIT) - ptl_*_default_cfg / ptl_*_create refuse a layout that is not theirs instead of overrunning it."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(type(self))
        self.abi_version = ABI_VERSION


class IcpCfg(_Cfg):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("max_range", C.c_double), ("min_range", C.c_double), ("voxel_size", C.c_double),
                ("max_points_per_voxel", C.c_int32), ("initial_threshold", C.c_double),
                ("min_motion_th", C.c_double), ("deskew", C.c_int32), ("max_iterations", C.c_int32),
                ("convergence", C.c_double), ("device_id", C.c_int32), ("scan_cols", C.c_int32),
                ("max_points_per_scan", C.c_int64), ("map_block_capacity", C.c_int64),
                ("map_table_capacity", C.c_int64), ("gn_workgroups", C.c_int32), ("rebuild_every", C.c_int32),
                ("gn_threads", C.c_int32), ("gn_lanes_per_point", C.c_int32), ("map_small_blocks", C.c_int64)]


class IcpStats(C.Structure):
    _fields_ = [("sigma", C.c_double), ("err_dt", C.c_double), ("err_drot", C.c_double),
                ("iterations", C.c_int32), ("n_corr_last", C.c_int32), ("n_in", C.c_int64),
                ("n_valid", C.c_int64), ("n_down", C.c_int64), ("n_src", C.c_int64),
                ("sum_cand", C.c_int64), ("map_voxels", C.c_int64), ("map_points", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class EkfCfg(_Cfg):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("init_grav", C.c_double * 3), ("init_bacc", C.c_double * 3), ("init_bgyr", C.c_double * 3),
                ("device_id", C.c_int32)]


class SeqCfg(_Cfg):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("icp", IcpCfg), ("ekf", EkfCfg), ("n_scans", C.c_int64), ("points_per_scan", C.c_int64),
                ("n_imu", C.c_int64), ("use_imu_prediction", C.c_int32), ("with_ekf", C.c_int32), ("range_input", C.c_int32),
                ("resident_scans", C.c_int32)]


_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
# every exported symbol of include/ptudes_mi.h with its prototype
PROTOTYPES = {
    "ptl_abi_version": (C.c_int, []),
    "ptl_sizeof_cfg": (C.c_int64, [C.c_int]),
    "ptl_last_error": (C.c_char_p, []),
    "ptl_backend": (C.c_int, []),
    "ptl_device_count": (C.c_int, []),
    "ptl_icp_default_cfg": (C.c_int, [C.POINTER(IcpCfg), C.c_double, C.c_double]),
    "ptl_icp_create": (C.c_int, [C.POINTER(IcpCfg), _vpp]),
    "ptl_icp_destroy": (C.c_int, [_vp]),
    "ptl_icp_register_frame": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, c_d_p, C.c_double, c_d_p, c_d_p,
                                         C.POINTER(IcpStats)]),
    "ptl_icp_set_lazy_map_stats": (C.c_int, [_vp, C.c_int32]),
    "ptl_icp_num_poses": (C.c_int, [_vp, c_i64_p]),
    "ptl_icp_get_poses": (C.c_int, [_vp, c_d_p, C.c_int64, c_i64_p]),
    "ptl_icp_get_prediction": (C.c_int, [_vp, c_d_p]),
    "ptl_icp_map_size": (C.c_int, [_vp, c_i64_p, c_i64_p]),
    "ptl_icp_map_points": (C.c_int, [_vp, c_d_p, C.c_int64, c_i64_p]),
    "ptl_icp_last_frame_down": (C.c_int, [_vp, c_d_p, C.c_int64, c_i64_p]),
    "ptl_icp_last_source": (C.c_int, [_vp, c_d_p, C.c_int64, c_i64_p]),
    "ptl_icp_gn_wg_clocks": (C.c_int, [_vp, C.POINTER(C.c_int64), C.c_int32]),
    "ptl_range_stats": (C.c_int, [C.c_int, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, c_d_p]),
    "ptl_icp_debug_set_epoch": (C.c_int, [_vp, C.c_uint32]),
    "ptl_icp_debug_sums": (C.c_int, [_vp, c_d_p]),
    "ptl_icp_debug_stall_workgroup": (C.c_int, [_vp, C.c_int32]),
    "ptl_icp_debug_limit_capacity": (C.c_int, [_vp, C.c_int32, C.c_int64]),
    "ptl_icp_deskew": (C.c_int, [_vp, c_d_p, c_d_p, C.c_int64, c_d_p]),
    "ptl_icp_map_add": (C.c_int, [_vp, c_d_p, C.c_int64, c_d_p, C.c_int]),
    "ptl_icp_linear_system": (C.c_int, [_vp, c_d_p, C.c_int64, C.c_double, C.c_double, c_d_p, c_i64_p, c_i64_p]),
    "ptl_icp_align": (C.c_int, [_vp, c_d_p, C.c_int64, c_d_p, C.c_double, C.c_double, c_d_p,
                                C.POINTER(C.c_int32)]),
    "ptl_lut_create": (C.c_int, [C.c_int, C.c_int32, C.c_int32, c_d_p, c_d_p, C.c_double, c_d_p, c_d_p, _vpp]),
    "ptl_lut_destroy": (C.c_int, [_vp]),
    "ptl_lut_apply": (C.c_int, [_vp, C.POINTER(C.c_uint32), c_d_p]),
    "ptl_traj_poses_at": (C.c_int, [C.c_int, c_d_p, c_d_p, C.c_int64, C.c_double, C.c_double, c_d_p, C.c_int64, c_d_p, c_i64_p]),
    "ptl_lut_dewarp": (C.c_int, [_vp, C.POINTER(C.c_uint32), c_d_p, c_d_p, c_i64_p]),
    "ptl_icp_set_active_beams": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "ptl_icp_register_range": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint32), C.c_double, c_d_p, c_d_p,
                                         C.POINTER(IcpStats)]),
    "ptl_icp_profile": (C.c_int, [_vp, C.c_int, c_d_p, c_i64_p, C.c_int]),
    "ptl_icp_gn_phases": (C.c_int, [_vp, c_i64_p]),
    "ptl_ekf_default_cfg": (C.c_int, [C.POINTER(EkfCfg)]),
    "ptl_ekf_create": (C.c_int, [C.POINTER(EkfCfg), _vpp]),
    "ptl_ekf_destroy": (C.c_int, [_vp]),
    "ptl_ekf_process_imu": (C.c_int, [_vp, c_d_p, c_d_p, C.c_double]),
    "ptl_ekf_process_imu_batch": (C.c_int, [_vp, c_d_p, C.c_int64]),
    "ptl_ekf_process_pose": (C.c_int, [_vp, c_d_p, c_d_p]),
    "ptl_ekf_get_state": (C.c_int, [_vp, c_d_p, c_d_p]),
    "ptl_ekf_pose_mat": (C.c_int, [_vp, c_d_p]),
    "ptl_ekf_ts": (C.c_int, [_vp, c_d_p]),
    "ptl_icp_ekf_step": (C.c_int, [_vp, _vp, c_d_p, C.c_int64, _vp, C.c_int, C.c_int64, c_d_p, c_d_p, C.c_int32, c_d_p, c_d_p,
                                  c_d_p, C.POINTER(IcpStats)]),
    "ptl_seq_create": (C.c_int, [C.POINTER(SeqCfg), _vpp]),
    "ptl_seq_destroy": (C.c_int, [_vp]),
    "ptl_seq_upload_scan": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_float)]),
    "ptl_seq_upload_range": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_uint32)]),
    "ptl_seq_set_lut": (C.c_int, [_vp, _vp, C.c_int32]),
    "ptl_seq_upload_imu": (C.c_int, [_vp, c_d_p, c_i64_p]),
    "ptl_seq_run": (C.c_int, [_vp, C.c_int64]),
    "ptl_seq_advance": (C.c_int, [_vp, C.c_int64]),
    "ptl_seq_enqueue": (C.c_int, [_vp, C.c_int64]),
    "ptl_seq_wait": (C.c_int, [_vp]),
    "ptl_seq_copy_traj": (C.c_int, [_vp, _vp, C.c_int64, c_i64_p]),
    "ptl_device_sync": (C.c_int, [C.c_int]),
    "ptl_host_pin": (C.c_int, [C.c_int, _vp, C.c_uint64]),
    "ptl_host_unpin": (C.c_int, [_vp]),
    "ptl_seq_results": (C.c_int, [_vp, c_d_p, c_d_p, c_d_p, C.POINTER(IcpStats), C.c_int64, c_i64_p]),
    "ptl_seq_traj_device": (C.c_int, [_vp, _vpp, c_i64_p]),
    "ptl_seq_icp": (C.c_int, [_vp, _vpp]),
    "ptl_seq_profile": (C.c_int, [_vp, C.c_int, c_d_p, c_i64_p, C.c_int]),
    "ptl_batch_create": (C.c_int, [C.POINTER(SeqCfg), C.c_int32, _vpp]),
    "ptl_batch_destroy": (C.c_int, [_vp]),
    "ptl_batch_upload_scan": (C.c_int, [_vp, C.c_int32, C.c_int64, C.POINTER(C.c_float)]),
    "ptl_batch_set_lut": (C.c_int, [_vp, _vp, C.c_int32]),
    "ptl_batch_upload_range": (C.c_int, [_vp, C.c_int32, C.c_int64, C.POINTER(C.c_uint32)]),
    "ptl_batch_upload_imu": (C.c_int, [_vp, C.c_int32, c_d_p, c_i64_p]),
    "ptl_batch_run": (C.c_int, [_vp, C.c_int64]),
    "ptl_batch_enqueue": (C.c_int, [_vp, C.c_int64]),
    "ptl_batch_wait": (C.c_int, [_vp]),
    "ptl_batch_results": (C.c_int, [_vp, C.c_int32, c_d_p, c_d_p, c_d_p, C.POINTER(IcpStats), C.c_int64, c_i64_p]),
    "ptl_batch_copy_traj": (C.c_int, [_vp, C.c_int32, _vp, C.c_int64, c_i64_p]),
    "ptl_batch_gn_phases": (C.c_int, [_vp, c_i64_p]),
    "ptl_batch_icp": (C.c_int, [_vp, C.c_int32, _vpp]),
    "ptl_batch_profile": (C.c_int, [_vp, C.c_int, c_d_p, c_i64_p, C.c_int]),
    "ptl_batch_set_driver": (C.c_int, [_vp, C.c_int32, C.c_int64]),
    "ptl_batch_reset": (C.c_int, [_vp]),
    "ptl_batch_seq_clocks": (C.c_int, [_vp, C.c_int32, c_i64_p]),
    "ptl_batch_set_team_workgroups": (C.c_int, [_vp, C.c_int32]),
    "ptl_batch_team_workgroups": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ptl_batch_exec_counters": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_uint64)]),
    "ptl_batch_sched_counters": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_uint64)]),
    "ptl_batch_status": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "ptl_batch_debug_stall_block": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "ptl_batch_debug_set_map_points_per_thread": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int32)]),
    "ptl_build_info": (C.c_int, [C.POINTER(C.c_int32)]),
    "ptl_code_id": (C.c_char_p, []),
    "ptl_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "ptl_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, _vpp]),
    "ptl_comm_destroy": (C.c_int, [_vp]),
    "ptl_gather_trajectories": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, c_i64_p, c_d_p, c_i64_p]),
    "ptl_batch_gather_trajectories": (C.c_int, [_vp, _vp, c_d_p, c_i64_p]),
}

_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when libptudes_mi.so is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C ptudes-lab_amd/csrc).  There is no CPU fallback.")
        # A handle drives three HIP streams (registration, map update, filter) that hand over to each other once per
        # scan.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): as
        # soon as other streams exist in the process (torch's, an RCCL communicator's) ours share a queue and every
        # hand-over serialises behind the neighbour's commands: 63 -> 110 us per scan, measured.  The variable is read
        # when the runtime initialises, so this only helps if nothing touched the GPU yet.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        L = C.CDLL(LIB_PATH)
        older = bool(os.environ.get("PTL_LIB_PATH")) and os.environ.get("PTL_LIB_ALLOW_OLDER") == "1"  # A/B runs against a library built from an older commit
        for name, (res, args) in PROTOTYPES.items():
            if older and not hasattr(L, name):
                continue
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        if hasattr(L, "ptl_abi_version"):
            # the binding's layouts against the loaded library's, before any struct changes hands
            have = (L.ptl_abi_version(), [L.ptl_sizeof_cfg(i) for i in range(4)])
            want = (ABI_VERSION, [C.sizeof(t) for t in (IcpCfg, EkfCfg, SeqCfg, IcpStats)])
            if have != want and not older:
                raise RuntimeError(f"{LIB_PATH}: ABI version / sizeof(icp cfg, ekf cfg, seq cfg, icp stats) = {have}, this binding "
                                   f"(ptudes-lab_amd/_lib.py) was written for {want}: rebuild the library or update the binding")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().ptl_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise RuntimeError(f"libptudes_mi error {rc}: {msg}")


def dptr(a):
    return a.ctypes.data_as(c_d_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
