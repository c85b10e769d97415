"""Sanitizer build of the HOST half of libmfar_hip.so: the stage-1 chunk-table builder and the payload /
workspace layouts live in csrc/mfar_tables.h, plain C++ that mfar_hip.hip includes unchanged.  Here the same header is compiled
on the CPU with AddressSanitizer + UndefinedBehaviorSanitizer + checked std::vector indexing and fuzzed over
(rows per field, fields, dim, list depth k, compute units, workgroups per CU, waves, sample sizing); every table must satisfy the
invariants the kernels rely on (tests/host/tables_fuzz.cpp).  CPU only: GPU sanitizers are not available on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_table_builder_and_layouts_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "tables_fuzz")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D_GLIBCXX_ASSERTIONS",
                           "-I", os.path.join(ROOT, "multifield-adaptive-retrieval_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "tables_fuzz.cpp"), "-o", exe])
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=500,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert out.stdout.startswith("OK ") and int(out.stdout.split()[1]) > 20000
