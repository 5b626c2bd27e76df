"""The built library's gfx950 code, checked for what cost round 3 a tenth of its throughput before it was found:
per-lane arrays that the compiler keeps in SCRATCH memory because of a run-time index (a `switch` with `M[10 + e]`, a chain
of selects turned into a load through selected addresses).  Such an access shows in the ISA as a scratch load / store whose
address is a VGPR; register spills (`off, s32 offset:N`) are a different matter and are counted only.

No GPU needed: the device code object is cut out of the library's fat binary and disassembled with the ROCm llvm-objdump."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ptudes-lab_amd", "csrc", "libptudes_mi.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# the hot device functions of the default build: the free-running kernel's stages (teams of 2 = the bench default, and the
# generic instance) and the single-sequence Gauss-Newton kernels
HOT = ["_Z15sq_gauss_newtonILi20ELi2EEvPK6SeqCtxiiii", "_Z15sq_gauss_newtonILi20ELi0EEvPK6SeqCtxiiii", "_Z15sq_gauss_newtonILi20ELi4EEvPK6SeqCtxiiii",
       "_Z10sq_preparePK6SeqCtxiiiijPjb", "_Z13sq_map_updateILi8EEjPK6SeqCtxiiiijiPjS3_b", "_Z13sq_map_updateILi4EEjPK6SeqCtxiiiijiPjS3_b", "_Z10k_gn_loop8ILi20ELi0EEv3Ctxi", "_Z9k_gn_loopILi20ELb0EEv3Ctxi"]


def device_code_object(tmp_path):
    data = open(LIB, "rb").read()
    at = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert at >= 0, "no uncompressed offload bundle in the library"
    (n,) = struct.unpack_from("<Q", data, at + 24)
    p = at + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        p += 24
        triple = data[p:p + tl].decode()
        p += tl
        if "gfx950" in triple:
            out = tmp_path / "device.o"
            out.write_bytes(data[at + off:at + off + size])
            return str(out)
    raise AssertionError("no gfx950 code object in the library")


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and the ROCm llvm-objdump")
def test_no_run_time_indexed_private_arrays_in_the_hot_functions(tmp_path):
    obj = device_code_object(tmp_path)
    # load: destination, ADDRESS VGPR, ...;  store: ADDRESS VGPR, data, ...   (a spill has `off` where the address VGPR would be)
    vgpr_addressed = re.compile(r"scratch_load_\w+\s+v\[?[\d:]+\]?,\s*v\d+,|scratch_store_\w+\s+v\d+,\s*v\[?[\d:]+\]?,")
    for sym in HOT:
        r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--disassemble-symbols=" + sym, obj], capture_output=True, text=True, timeout=120)
        body = [ln for ln in r.stdout.splitlines() if ln.startswith("\t")]
        assert len(body) > 500, f"{sym}: not found in the code object (renamed?)"
        scratch = [ln.split("//")[0].strip() for ln in body if "scratch_" in ln]
        dynamic = [ln for ln in scratch if vgpr_addressed.search(ln)]
        assert not dynamic, f"{sym}: {len(dynamic)} scratch accesses through a VGPR address (a per-lane array indexed at run time), e.g. {dynamic[:3]}"
        # spills: the callee-saved registers of a stage function (about 110 each way) plus what the register allocator could not keep;
        # a jump here means a loop has started to live in memory
        assert len(scratch) <= 420, f"{sym}: {len(scratch)} scratch instructions"
