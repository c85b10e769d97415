#!/usr/bin/env python3
"""VGPR / spill / scratch figures of the stage-1 kernels from the gfx950 assembly (hipcc cross-compiles; no GPU needed).
    python tools/kernel_regs.py [pattern]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multifield-adaptive-retrieval_amd", "csrc")


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "stage1"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "mfar.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                               "-fhip-fp32-correctly-rounded-divide-sqrt", f"-I{ROOT}/include", f"-I{CSRC}", "--cuda-device-only", "-S",
                               os.path.join(CSRC, "mfar_hip.hip"), "-o", out] + sys.argv[2:], stderr=subprocess.DEVNULL)
        s = open(out).read()
    for blk in re.split(r"\n  - \.agpr_count:", s)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or pat not in name.group(1):
            continue
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        print(name.group(1)[:64].ljust(64), "vgpr", g("vgpr_count"), "spill", g("vgpr_spill_count"), "sgpr", g("sgpr_count"), "scratch",
              g("private_segment_fixed_size"))


if __name__ == "__main__":
    main()
