"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and exports
every symbol include/mfar_hip.h declares; argument validation that needs no device; the host mirror keeps the
reference's names and signatures (pinned by tests/golden/cli_signatures.json, schema.json)."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mfar_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mfar_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mfar import _native
    L = _native.lib()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), f"libmfar_hip.so does not export {name}"
    assert sorted(_native.SIGNATURES) == declared, "python binding table and header disagree"
    # the version is pinned exactly: header, library and binding table must move together on every signature change
    hdr = open(os.path.join(ROOT, "include", "mfar_hip.h")).read()
    want = int(re.search(r"#define\s+MFAR_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert L.mfar_version() == want == _native.ABI_VERSION == 107


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """A library that reports another ABI version has other argument lists behind the same names: lib() must refuse it."""
    from mfar import _native
    _native.lib()
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "ABI_VERSION", _native.ABI_VERSION + 1)
    with pytest.raises(ImportError, match="ABI version"):
        _native.lib()


def test_no_device_fails_loudly_not_silently():
    """Without a GPU every compute entry point must return an error (and the python layer must raise):
    there is no CPU fallback."""
    from mfar import _native
    from mfar.data.index import MultiFieldIndex
    if _native.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_native.MfarError):
        MultiFieldIndex(10, 1, 32)
    L = _native.lib()
    h = ctypes.c_void_p()
    assert L.mfar_index_create(ctypes.byref(h), 0, 10, 0, 1, 32, 0) != 0
    assert b"hip" in L.mfar_last_error().lower()


def test_argument_validation_without_device():
    from mfar import _native
    L = _native.lib()
    h = ctypes.c_void_p()
    assert L.mfar_index_create(None, 0, 10, 0, 1, 32, 0) == -1
    assert L.mfar_index_create(ctypes.byref(h), 0, 10, 0, 1, 33, 0) == -1      # dim % 32
    assert L.mfar_index_create(ctypes.byref(h), 0, 10, 0, 99, 32, 0) == -1     # n_fields
    assert L.mfar_index_create(ctypes.byref(h), 0, -1, 0, 1, 32, 0) == -1
    assert L.mfar_index_create(ctypes.byref(h), 0, 2**32, 0, 1, 32, 0) == -1   # ids must fit 32 bits
    assert L.mfar_index_create(ctypes.byref(h), 0, 10, 0, 1, 32, 7) == -1      # unknown dtype
    assert L.mfar_payload_bytes(64, 8, 100) > 64 * 800 * 8 * 4
    assert L.mfar_payload_bytes(-1, 8, 100) == 0
    assert L.mfar_merge_workspace_bytes(64, 8, 100) > 64 * 800 * 8 * 4
    assert L.mfar_retrieve_fields(None, None, 1, 100, 1, None, None, 0, None) == -1
    assert L.mfar_set_wgs_per_cu(None, 2) == -1
    assert L.mfar_set_stage2_mode(None, 1) == -1
    assert L.mfar_stage2_stats(None, None, None, None, None) == -1


def test_python_mirror_keeps_reference_signatures():
    from mfar.data.index import DenseFlatIndex, Index, candidate_encoding_stream
    p = list(inspect.signature(DenseFlatIndex.__init__).parameters)
    assert p[:7] == ["self", "model", "vectors", "numeric_ids_to_keys", "keys_to_numeric_ids", "device", "vector_batch_size"]
    assert inspect.signature(DenseFlatIndex.__init__).parameters["vector_batch_size"].default == 1048576
    assert list(inspect.signature(DenseFlatIndex.retrieve_batch).parameters) == ["self", "queries", "top_k"]
    assert list(inspect.signature(DenseFlatIndex.score_batch).parameters) == ["self", "queries", "keys"]
    # the reference's five parameters in the reference's order; `as_tensor` is a trailing keyword extension (default off)
    ps = inspect.signature(candidate_encoding_stream).parameters
    assert list(ps)[:5] == ["encoder", "corpus", "batch_size", "multiprocess", "show_progress"]
    assert list(ps)[5:] == ["as_tensor"] and ps["as_tensor"].default is False
    assert inspect.isabstract(Index)


def test_bench_spawns_its_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher starts two child ranks (RANK / WORLD_SIZE / MASTER_* in their environment) before
    anything touches the GPU and relays rank 0's line.  In this container there is no GPU: both ranks fail, the parent must exit
    non-zero, print NO result line and leave no process behind (the GPU half is tests/test_gpu_multirank.py)."""
    import subprocess
    import sys
    from mfar import _native
    if _native.device_count() > 0:
        pytest.skip("GPU present: covered by tests/test_gpu_multirank.py")
    env = dict(os.environ, MFAR_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--docs", "20000", "--fields", "2", "--dim", "64",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "rank" in out.stderr and "exited with code" in out.stderr


def test_integration_stub_compiles_and_names_only_declared_symbols():
    """INTEGRATION.md section 2 (executed verbatim on the GPU box by tests/test_gpu_integration.py): here, without a GPU, the block must
    compile, pin the header's ABI number and call nothing the header does not declare."""
    import re
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", md[md.index("## 2. The binding a maintainer"):], re.S).group(1)
    compile(code, "INTEGRATION.md#2", "exec")
    header = open(os.path.join(ROOT, "include", "mfar_hip.h")).read()
    abi = int(re.search(r"#define MFAR_ABI_VERSION (\d+)", header).group(1))
    assert f"mfar_version() == {abi}" in code
    used = set(re.findall(r"_L\.(mfar_\w+)", code))
    assert used and all(re.search(r"\b%s\(" % name, header) for name in used), sorted(used)
    assert {"mfar_pipeline_create", "mfar_pipeline_submit", "mfar_pipeline_result", "mfar_search_two_stage"} <= used


def test_argument_validation_returns_error_codes_without_a_device():
    """Every entry point validates its arguments before it touches the device: bad shapes come back as MFAR_ERR_INVALID with a message --
    on this box too, where there is no GPU (the valid calls then fail with MFAR_ERR_HIP / INVALID 'no such device', never with a crash)."""
    import ctypes
    from mfar import _native
    L = _native.lib()
    h = ctypes.c_void_p()
    bad = [dict(n=-1), dict(F=0), dict(F=33), dict(E=0), dict(E=48), dict(dt=7), dict(off=-5), dict(n=2 ** 32)]
    for kw in bad:
        a = dict(dev=0, n=100, off=0, F=4, E=64, dt=0)
        a.update(kw)
        rc = L.mfar_index_create(ctypes.byref(h), a["dev"], a["n"], a["off"], a["F"], a["E"], a["dt"])
        assert rc == -1 and not h.value, (kw, rc)
        assert len(L.mfar_last_error()) > 0
    assert L.mfar_index_create(None, 0, 100, 0, 4, 64, 0) == -1
    # NULL handles / pointers
    assert L.mfar_index_write_rows(None, 0, 0, 1, None, 0, None) == -1
    assert L.mfar_retrieve_fields(None, None, 1, 100, 1, None, None, 0, None) == -1
    assert L.mfar_search_two_stage(None, None, 1, None, 1, None, 100, 100, 1, None, None, None, None, None, None, 0, None) == -1
    assert L.mfar_pipeline_create(None, None, None, 1, None, 100, 100, 1, 64, 0, 0, 0) == -1
    p = ctypes.c_void_p()
    assert L.mfar_pipeline_create(ctypes.byref(p), None, None, 1, None, 100, 100, 1, 64, 0, 0, 0) == -1 and not p.value
    t = ctypes.c_int64()
    assert L.mfar_pipeline_submit(None, None, 1, 0, None, ctypes.byref(t)) == -1
    assert L.mfar_pipeline_result(None, 0, None, None, None, 0, None) == -1
    assert L.mfar_pipeline_flush(None) == -1
    L.mfar_pipeline_destroy(None)                                   # no-ops on NULL
    L.mfar_index_destroy(None)
    assert L.mfar_set_auto_off(None, 1, 0, 0) == -1 and L.mfar_set_screen(None, 1, 1.0) == -1
    # pure size helpers need no device
    assert L.mfar_payload_bytes(64, 8, 100) > 0 and L.mfar_lists_bytes(64, 8, 100) > 0 and L.mfar_topk_bytes(64, 100) > 0
    assert L.mfar_payload_bytes(-1, 8, 100) == 0 and L.mfar_max_split_batch(None, 100) == 0
