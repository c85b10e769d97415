"""The adaptive policy of the certified screen (AUTO-OFF per field with probes, reversible inline repair) is a host-only state machine in
csrc/mfar_policy.h, driven by libmfar_hip.so from the certificate flags of finished launches.  Here the same header is compiled on the CPU
with AddressSanitizer + UndefinedBehaviorSanitizer and run through scenario simulations (tests/host/policy_sim.cpp): clean data, two of
eight fields failing, every field failing (a launch = the exact pass + one probe in 64), a 30 % field (stays on, repaired inline), the
strict bf16 rule, mode 0, reversibility.  The GPU side of it is tests/test_gpu_parity.py::test_auto_off_switches_clustered_fields_...."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_screen_policy_scenarios_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "policy_sim")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "multifield-adaptive-retrieval_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "policy_sim.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=200,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert out.stdout.startswith("OK policy scenarios")
