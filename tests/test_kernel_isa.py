"""Guards on the generated code of the dominant kernel (CPU-only: hipcc cross-compiles gfx950 here).

The K loop of `zgemm_kernel` is bound by the FP64 matrix pipe, and on this chip every vector instruction a wave issues
between its MFMAs takes issue time from that pipe (134 -> 116 ms per C4 apply came from removing
them).  A source change that looks harmless can bring them back -- a struct copy that becomes a `memcpy` through
scratch, a select on a pointer, a mask on an LDS store -- so the steady-state loop of the shipped instantiations is
checked instruction class by instruction class, and no GEMM instantiation may spill or use scratch."""
import collections
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytdscf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def zgemm_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "zgemm.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
           "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, "zgemm.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text(), r.stderr


def _blocks(asm, mangled):
    i = asm.index(mangled + ":")
    body = asm[i:asm.index(".Lfunc_end", i)].splitlines()
    blocks, cur = [], []
    for line in body:
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append(cur)
            cur = []
        else:
            t = line.strip()
            if t and not t.startswith((".", ";", "//")):
                cur.append(t.split()[0])
    blocks.append(cur)
    return blocks


HOT = {  # 64x64 tile, BK = 16, 3M product: the four operand forms, the block-sparse list form, the two forms with the
    # reducing epilogue (the K loop is the same code: the epilogue must not leak instructions or registers into it)
    "NN": "_ZN6mitdvp12zgemm_kernelILi2ELi2ELi16ELb0ELb0ELb1ELb0ELb0EEEvNS_9ZgemmDescEiii",
    "NT": "_ZN6mitdvp12zgemm_kernelILi2ELi2ELi16ELb0ELb1ELb1ELb0ELb0EEEvNS_9ZgemmDescEiii",
    "TN": "_ZN6mitdvp12zgemm_kernelILi2ELi2ELi16ELb1ELb0ELb1ELb0ELb0EEEvNS_9ZgemmDescEiii",
    "TT": "_ZN6mitdvp12zgemm_kernelILi2ELi2ELi16ELb1ELb1ELb1ELb0ELb0EEEvNS_9ZgemmDescEiii",
    "SP": "_ZN6mitdvp12zgemm_kernelILi2ELi2ELi16ELb0ELb0ELb1ELb1ELb0EEEvNS_9ZgemmDescEiii",
    "NN-reduce": "_ZN6mitdvp12zgemm_kernelILi2ELi2ELi16ELb0ELb0ELb1ELb0ELb1EEEvNS_9ZgemmDescEiii",
    "NT-reduce": "_ZN6mitdvp12zgemm_kernelILi2ELi2ELi16ELb0ELb1ELb1ELb0ELb1EEEvNS_9ZgemmDescEiii",
}


@pytest.mark.parametrize("form", sorted(HOT))
def test_steady_state_loop_has_no_vector_instruction_but_the_3m_sums(zgemm_asm, form):
    asm, _ = zgemm_asm
    loops = [collections.Counter(b) for b in _blocks(asm, HOT[form]) if b.count("v_mfma_f64_16x16x4_f64") >= 96]
    assert loops, "no basic block holds two K tiles (96 MFMAs): the tile loop is no longer unrolled by the register-set parity"
    c = min(loops, key=lambda k: sum(k.values()))  # the FULL form (the general form for the last tiles has selects)
    assert c["v_mfma_f64_16x16x4_f64"] == 96
    assert c["buffer_load_dwordx4"] == 16 and c["ds_write_b128"] == 16 and c["ds_read_b128"] == 32 and c["s_barrier"] == 2
    valu = {k: v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma")}
    assert set(valu) <= {"v_fma_f64"} and valu.get("v_fma_f64", 0) == 32, valu
    assert not any(k.startswith(("scratch_", "global_load", "flat_load")) for k in c), c


def test_no_gemm_instantiation_spills_or_uses_scratch(zgemm_asm):
    _, remarks = zgemm_asm
    cur, seen = None, 0
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        if cur and "zgemm_kernel" in cur:
            m = re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
            if m:
                # the reducing-epilogue forms may keep a few scalars in vector lanes OUTSIDE the K loop (the loop itself is
                # checked instruction by instruction above: no v_readlane / v_writelane there); never memory
                if not (m.group(1) == "SGPRs Spill" and cur.endswith("ELb1EEEvNS_9ZgemmDescEiii")):
                    assert int(m.group(2)) == 0, (cur, line)
                seen += 1
            m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
            if m and "ILi2ELi2ELi16E" in cur:  # the 64x64 tile: two workgroups per CU
                assert int(m.group(1)) >= 2, (cur, line)
    assert seen >= 3 * 20
