"""Property tests of the CPU oracle (hypothesis) and an AddressSanitizer/UBSan pass over its C code (sanitizers are
CPU-only on this pool)."""
import os
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import mfar_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(seed, F, D, E, Q, dup):
    rng = np.random.default_rng(seed)
    slab = (rng.standard_normal((F, D, E)) * 0.5 + 0.05).astype(np.float32)
    if dup and D > 3:
        slab[:, rng.choice(D, size=min(dup, D), replace=False)] = slab[:, :1]      # identical rows -> exact ties
    q = rng.standard_normal((Q, E)).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.2).astype(np.float32)
    return slab, q, W


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10**6), D=st.integers(1, 400), k=st.integers(1, 128), Q=st.integers(1, 5), sentinel=st.booleans(),
       dup=st.integers(0, 20))
def test_retrieve_is_canonical_topk_of_scores(seed, D, k, Q, sentinel, dup):
    slab, q, _ = _data(seed, 1, D, 32, Q, dup)
    sc = O.c_scores(slab[0], q)
    ids, top = O.c_retrieve(slab[0], q, k, sentinel)
    for i in range(Q):
        keep = np.nonzero(sc[i] > 0)[0] if sentinel else np.arange(D)
        order = keep[np.lexsort((keep, -sc[i][keep].astype(np.float64)))][:k]
        n = order.size
        assert np.array_equal(ids[i, :n], order) and np.array_equal(top[i, :n].view(np.uint32), sc[i][order].view(np.uint32))
        assert (ids[i, n:] == (0 if sentinel else -1)).all()
        assert (top[i, n:] == 0).all() if sentinel else np.isneginf(top[i, n:]).all()


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10**6), D=st.integers(2, 500), S=st.integers(1, 8), sentinel=st.booleans(), dup=st.integers(0, 30))
def test_sharded_lists_merge_to_unsharded(seed, D, S, sentinel, dup):
    slab, q, _ = _data(seed, 1, D, 32, 3, dup)
    gi, gs = O.c_retrieve(slab[0], q, 100, sentinel)
    bounds = [D * r // S for r in range(S + 1)]
    parts = [O.c_retrieve(slab[0][bounds[r]:bounds[r + 1]], q, 100, sentinel, row_offset=bounds[r]) for r in range(S)]
    for i in range(3):
        mi, ms = O.c_merge_lists(np.stack([p[0][i] for p in parts]), np.stack([p[1][i] for p in parts]), sentinel)
        assert np.array_equal(mi, gi[i]) and np.array_equal(ms.view(np.uint32), gs[i].view(np.uint32))


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10**6), F=st.integers(1, 6), D=st.integers(1, 300), masked=st.integers(0, 63))
def test_two_stage_invariants(seed, F, D, masked):
    slab, q, W = _data(seed, F, D, 32, 3, 5)
    mask = np.array([0.0 if (masked >> f) & 1 else 1.0 for f in range(F)], np.float32)
    r = O.c_two_stage(slab, q, W, mask)
    assert (r["n_valid"] == np.minimum(r["n_cand"], 100)).all()
    for i in range(3):
        n = r["n_valid"][i]
        ids, sc = r["ids"][i, :n], r["scores"][i, :n]
        assert len(set(ids.tolist())) == n and (np.diff(sc) <= 0).all()
        # every result is a member of some per-field list, and its score is the mixer applied to its exact field scores
        assert set(ids.tolist()) <= set(r["field_ids"][i].ravel().tolist())
        w = O.c_gate(q[i], W)
        x = O.c_score_candidates(slab, q[i:i + 1], ids[None])[0]
        assert np.array_equal(O.c_mix(x, w, mask).view(np.uint32), sc.view(np.uint32))
    if masked == 0:       # mask of ones == no mask
        r2 = O.c_two_stage(slab, q, W, None)
        assert np.array_equal(r2["ids"], r["ids"]) and np.array_equal(r2["scores"].view(np.uint32), r["scores"].view(np.uint32))


def test_bf16_round_is_rne_and_idempotent():
    x = np.array([1.0, 1.00390625, 1.005859375, 1.0078125, -3.1415927, 1e-30, 65504.0, 0.0], np.float32)
    r = O.bf16_round(x)
    assert (r.view(np.uint32) & 0xFFFF == 0).all() and np.array_equal(O.bf16_round(r), r)
    assert r[1] == np.float32(1.0) and r[2] == np.float32(1.0078125)          # tie -> even, above tie -> up
    assert np.max(np.abs(r - x) / np.maximum(np.abs(x), 1e-38)) <= 2.0 ** -8


def test_oracle_c_code_under_asan_ubsan():
    """Build the oracle with -fsanitize=address,undefined and drive every entry point in a child process."""
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/libmfar_oracle_asan.so"])
    code = r'''
import ctypes, numpy as np, sys
sys.path.insert(0, %r)
from oracle import mfar_oracle as O
O._LIB = None
import oracle.mfar_oracle as M
M.build = lambda force=False: %r
rng = np.random.default_rng(0)
slab = rng.standard_normal((3, 333, 40)).astype(np.float32); q = rng.standard_normal((4, 40)).astype(np.float32)
W = rng.standard_normal((40, 3)).astype(np.float32)
for sentinel in (True, False):
    r = O.c_two_stage(slab, q, W, np.array([1, 0, 1], np.float32), k1=100, k2=100, sentinel=sentinel)
    O.c_retrieve(slab[0][:7], q, 128, sentinel, row_offset=5)
O.c_score_candidates(slab, q, np.array([[0, 332, -1, 999]] * 4, np.int64))
O.c_merge_lists(np.zeros((3, 100), np.int64), np.zeros((3, 100), np.float32), True)
O.bf16_round(slab); O.c_exp(-3.0)
with O.chain("natural"): O.c_scores(slab[0], q)
print("asan-ok")
''' % (ROOT, os.path.join(ROOT, "oracle", "_build", "libmfar_oracle_asan.so"))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "asan-ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
