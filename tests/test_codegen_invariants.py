"""What the hand-counted waits and the inline-asm MFMAs of the HIP kernels ASSUME about the code hipcc emits, checked on the built
library (no GPU): the device code objects are taken out of libsatools_hip.so with llvm-objdump, their kernel metadata and
disassembly read.

* `pair32s_kernel` (csrc/pair32s.hip) waits for the LDS-DMA pieces of the next tile with `s_waitcnt vmcnt(SPS)` / `vmcnt(2 SPS)`,
  SPS = the VM stores a wave issues per output subtile (4 plane stores, 8 f32 stores): a toolchain that merged or split those
  stores would let the wait return early (or late) without any error — the count is asserted here.
* Kernels that keep accumulators in the operands of inline-asm MFMAs must not spill (the compiler inserts no wait states between an
  MFMA it cannot see and the scratch store of its result: DESIGN.md toolchain note 14): the upsampler instantiations of
  conv1d_f16x3_ring16_kernel and the persistent ring GEMM are spill-free, the ring conv's other instantiations keep the few
  (non-accumulator) spills they were validated with."""
import os
import re
import shutil
import subprocess

import pytest

from satools_amd import _lib

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf of ROCm not found")
    _lib.lib()                                             # (raises with the build command if the library is missing)
    d = tmp_path_factory.mktemp("co")
    so = shutil.copy(_lib.library_path(), d)
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=d)
    cos = sorted(str(p) for p in d.iterdir() if "amdgcn" in p.name)
    assert cos, "no gfx950 code object inside the library"
    meta = {}
    for co in cos:
        notes = subprocess.run([readelf, "--notes", co], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:|\n\s+- \.args:", notes):
            m = re.search(r"\.name:\s+(\S+)", blk)
            sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
            if m and sp:
                meta[m.group(1)] = {"file": co, "vgpr_spill_count": int(sp.group(1)),
                                    "scratch": int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))}
    return {"meta": meta, "objdump": objdump}


def _disassemble(code_objects, symbol):
    info = code_objects["meta"][symbol]
    out = subprocess.run([code_objects["objdump"], "-d", f"--disassemble-symbols={symbol}", info["file"]], check=True,
                         capture_output=True, text=True).stdout
    assert symbol in out
    return out


def test_pair32s_store_count_matches_its_counted_waits(code_objects):
    meta = code_objects["meta"]
    seen = 0
    for y16, yf, sps in ((1, 0, 4), (0, 1, 8), (1, 1, 12)):
        for nw in (4, 8):
            sym = f"_ZN3sat14pair32s_kernelILb{y16}ELb{yf}ELi{nw}EEEvNS_7P32ArgsE"
            assert sym in meta, sym
            dis = _disassemble(code_objects, sym)
            stores = re.findall(r"\bbuffer_store_\w+", dis)
            assert len(stores) == sps, (sym, stores)
            assert not re.search(r"\b(global|flat|scratch)_store", dis), sym
            # the waits the source asks for, as immediates
            waits = set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", dis))
            assert {str(sps), str(2 * sps)} <= waits | {"0"} and (str(sps) in waits), (sym, sorted(waits, key=int))
            assert len(re.findall(r"buffer_load_dwordx4 .*\blds\b", dis)) == 8      # 4 pieces of the first image + 4 of the next
            seen += 1
    assert seen == 6


def test_kernels_with_asm_accumulators_do_not_spill_them(code_objects):
    meta = code_objects["meta"]
    spill = {k: v["vgpr_spill_count"] for k, v in meta.items()}
    ups = [k for k in spill if "conv1d_f16x3_ring16_kernelILi4ELb0ELi780E" in k or "conv1d_f16x3_ring16_kernelILi4ELb0ELi0E" in k]
    assert len(ups) == 2 and all(spill[k] == 0 and meta[k]["scratch"] == 0 for k in ups), {k: spill[k] for k in ups}
    walk = [k for k in spill if "gemm_f16x3_walk16_kernel" in k]
    assert walk and all(spill[k] == 0 for k in walk), {k: spill[k] for k in walk}
    # the resblock instantiations: a handful of spilled loop invariants (DMA offsets, toolchain note 13), none inside the K loop's
    # MFMA columns — validated bit for bit against the lean tile by tests/test_hip_parity.py; a jump in the count wants a new look
    ring = {k: spill[k] for k in spill if "conv1d_f16x3_ring16_kernelILi4ELb0ELin1E" in k or "conv1d_f16x3_ring16_kernelILi2ELb0ELin1E" in k}
    assert len(ring) == 2 and all(v <= 24 for v in ring.values()), ring
