#!/usr/bin/env python3
"""What the MFMA loop of a stage-1 kernel looks like in the gfx950 assembly: vmcnt waits, scratch traffic, instruction mix.
    python tools/kernel_loop.py <kernel-name-substring> [--dump]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multifield-adaptive-retrieval_amd", "csrc")


def main():
    pat = sys.argv[1]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "mfar.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                               "-fhip-fp32-correctly-rounded-divide-sqrt", f"-I{ROOT}/include", f"-I{CSRC}", "--cuda-device-only", "-S",
                               os.path.join(CSRC, "mfar_hip.hip"), "-o", out] + [a for a in sys.argv[2:] if a.startswith("-D")], stderr=subprocess.DEVNULL)
        s = open(out).read()
    for name in re.findall(r"^(_Z\d+mfar_stage1_\w+?_kernel8S1Params):", s, re.M):
        if pat not in name:
            continue
        i = s.index(name + ":")
        body = s[i:s.index(".Lfunc_end", i)].split("\n")
        mf = [k for k, l in enumerate(body) if "v_mfma" in l]
        loop = body[mf[0] - 12:mf[-1] + 1]
        mix = {}
        for l in loop:
            t = l.strip().split(" ")[0]
            if t and not t.startswith((";", ".", "s_nop")):
                mix[t] = mix.get(t, 0) + 1
        print(name, "mfma", len(mf))
        print("  vmcnt waits:", [l.strip() for l in loop if "s_waitcnt" in l and "vmcnt" in l])
        print("  scratch in loop:", [l.strip() for l in loop if "scratch_" in l][:4])
        print("  mix:", dict(sorted(mix.items(), key=lambda kv: -kv[1])[:24]))
        if "--dump" in sys.argv:
            print("\n".join(loop))


if __name__ == "__main__":
    main()
