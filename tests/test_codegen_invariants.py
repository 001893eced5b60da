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


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(r) for r in re.findall(r"\bv(\d+)\b", text))
    return out


def _mfma_result_hazards(dis, window=16):
    """instructions that touch the destination registers of an inline-asm MFMA within `window` instructions behind it without wait
    states (`s_nop 15`) in between — in listing order inside straight-line runs (the MFMA columns of the K loops are such runs).
    Another MFMA accumulating into the same registers is what the hardware interlocks; everything else (a vector copy, a scratch or
    memory store, an LDS write) would read a result the matrix pipe has not written back yet: the compiler does not see an MFMA in
    an asm statement and pads nothing."""
    ins = [l.split("//")[0].strip() for l in dis.splitlines() if re.match(r"^\s+[a-z_0-9]+ ", l)]
    recent, bad = [], []                        # [(index, dest registers)]
    for i, l in enumerate(ins):
        if l.startswith(("s_nop 15", "s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            recent = []                         # wait states; or the end of a straight-line run (listing order is no longer execution order)
            continue
        if l.startswith("v_mfma"):
            m = re.match(r"v_mfma\S*\s+v\[(\d+):(\d+)\]", l)
            assert m, l
            recent.append((i, set(range(int(m.group(1)), int(m.group(2)) + 1))))
            continue
        if l.startswith("s_") or not re.match(r"^(v_|scratch_|buffer_|global_|flat_|ds_)", l):
            continue
        recent = [(j, d) for j, d in recent if i - j <= window]
        touched = _regs(l)
        for j, d in recent:
            if touched & d:
                bad.append((ins[j], l))
    return bad, sum(1 for l in ins if l.startswith("v_mfma"))


def test_kernels_with_asm_accumulators_do_not_spill_them(code_objects):
    meta = code_objects["meta"]
    spill = {k: v["vgpr_spill_count"] for k, v in meta.items()}
    ups = [k for k in spill if "conv1d_f16x3_ring16_kernelILi4ELb0ELi780E" in k or "conv1d_f16x3_ring16_kernelILi4ELb0ELi0E" in k]
    assert len(ups) == 2 and all(spill[k] == 0 and meta[k]["scratch"] == 0 for k in ups), {k: spill[k] for k in ups}
    walk = [k for k in spill if "gemm_f16x3_walk16_kernel" in k]
    assert walk and all(spill[k] == 0 for k in walk), {k: spill[k] for k in walk}
    # round 5: the production instantiations with e4m3 cross terms (SAT_CONV_F16F8R) are spill-free
    f8 = [k for k in spill if "conv1d_f16x3_ring16_kernelILi4ELb0ELin1ELb1E" in k or "conv1d_f16x3_ring16_kernelILi2ELb0ELin1ELb1E" in k]
    assert len(f8) == 2 and all(spill[k] == 0 and meta[k]["scratch"] == 0 for k in f8), {k: spill[k] for k in f8}
    # the f16x3 resblock instantiations: spill-free since round 5 (round 4 shipped them with 5 / 17 spilled registers, and the WR = 2
    # form stored an accumulator to scratch right behind the MFMA that writes it — found by the hazard check below; with ONE loop tail
    # instead of three, conv_ring16.hip, the allocator keeps everything in registers)
    ring = {k: spill[k] for k in spill if "conv1d_f16x3_ring16_kernelILi4ELb0ELin1ELb0E" in k or "conv1d_f16x3_ring16_kernelILi2ELb0ELin1ELb0E" in k}
    assert len(ring) == 2 and all(v == 0 and meta[k]["scratch"] == 0 for k, v in ring.items()), ring
    # ... and WHERE they are (round-4 advisor item): nothing but another MFMA touches the registers an MFMA has just written — no scratch
    # store, no vector copy, no memory store — unless the tied wait states of mfma16_drain(acc) stand in between
    checked = 0
    for sym in list(ring) + f8 + ups + walk:
        dis = _disassemble(code_objects, sym)
        assert not re.search(r"\bv_accvgpr", dis), sym                  # (no accumulator lives in the AGPR half either)
        bad, n_mfma = _mfma_result_hazards(dis)
        assert n_mfma >= 100, (sym, n_mfma)
        assert not bad, (sym, bad[:4])
        checked += 1
    assert checked >= 6, checked
