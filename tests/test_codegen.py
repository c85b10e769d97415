"""Static check of the compiled gfx950 code of the register-ring scan kernels (no GPU needed: hipcc cross-compiles).

The ring kernels load doc granules with `global_load_dwordx4` in inline asm and let the compiler believe the destination
register holds the value at once; the counted `s_waitcnt vmcnt(N)` that really delivers it is a later asm statement with
the register as in/out operand (csrc/mfar_stage1.h).  That is only sound while hipcc keeps the register where the load
left it: a `v_mov` that reads a ring register between the load and the wait copies a value that has not arrived.  A
variant of the wide kernel whose wait statement sat behind a branch did exactly that (lost list entries, found on the
GPU); this test pins the property for every ring kernel of the build."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multifield-adaptive-retrieval_amd", "csrc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("asm") / "mfar.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", f"-I{ROOT}/include", f"-I{CSRC}", "--cuda-device-only", "-S",
           os.path.join(CSRC, "mfar_hip.hip"), "-o", str(out)]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernels(asm):
    for name in re.findall(r"^(_Z\d+mfar_stage1_\w+?_kernel8S1Params):", asm, re.M):
        i = asm.index(name + ":")
        yield name, asm[i:asm.index(".Lfunc_end", i)].split("\n")


def test_ring_registers_are_never_copied_before_their_wait(asm):
    checked = 0
    for name, body in _kernels(asm):
        ring = set()
        in_asm = False
        for line in body:
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            if not in_asm:
                continue                              # (hipcc's own loads are tracked by its waitcnt insertion)
            # the inline-asm ring loads: a per-lane 64-bit address, or a per-lane 32-bit offset on an SGPR base
            m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\], (?:v\[\d+:\d+\], off|v\d+, s\[\d+:\d+\])", line)
            if m:
                ring.update(range(int(m.group(1)), int(m.group(2)) + 1))
            m = re.search(r"global_atomic_add v(\d+), v\[\d+:\d+\], v\d+, off sc0", line)   # the asynchronous unit claim (s1_unit_claim_async)
            if m:
                ring.add(int(m.group(1)))
        if not ring:
            continue                                  # an LDS-ring kernel: docs never sit in registers while in flight
        checked += 1
        mf = [k for k, line in enumerate(body) if "v_mfma" in line]
        assert mf, name
        for line in body[max(0, mf[0] - 120):mf[-1] + 10]:      # prologue issue + the k-loop
            m = re.search(r"v_mov_b(?:32|64)_e32 v\[?(\d+)(?::(\d+))?\]?, v\[?(\d+)(?::(\d+))?\]?", line)
            if m:
                src = range(int(m.group(3)), int(m.group(4) or m.group(3)) + 1)
                assert not any(r in ring for r in src), f"{name}: '{line.strip()}' reads a doc-ring register"
    assert checked >= 28, checked                     # f32r / f16r / bf16r / f16w / bf16s / bf16w / bf16c and their 4-slot twins, full + sample pass


def test_scan_loops_do_not_touch_scratch(asm):
    """Spills are tolerated in the selection epilogue of the wide pass (256 VGPRs), never inside the MFMA loop."""
    for name, body in _kernels(asm):
        mf = [k for k, line in enumerate(body) if "v_mfma" in line]
        if not mf:
            continue
        inside = [line for line in body[mf[0]:mf[-1] + 1] if "scratch_" in line]
        assert not inside, (name, inside[:3])
