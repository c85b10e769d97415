"""The oracle (oracle/) against the golden vectors captured from the real reference (tools/gen_golden.py).

Two restatements are pinned here:
  * the torch port (`ref_*`): same torch ops as the reference -> ids identical, scores ~1e-6;
  * the C oracle (`c_*`, the fmaf-chain contract the HIP kernels reproduce bit for bit): ids identical under the
    canonical tie-break, scores within 1e-4 (the tolerance BASELINE.json's north_star states).
"""
import json
import os

import numpy as np
import pytest

from oracle import mfar_oracle as O

TOL = 1e-4


def _cases(npz):
    return sorted({k.split("__")[0] for k in npz.files})


def test_c_oracle_builds_and_exports():
    L = O.lib()
    assert L.mfar_oracle_version() == 1


def test_retrieve_batch_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "retrieve_batch.npz"))
    for n in _cases(z):
        V, q, k, ch = z[n + "__V"], z[n + "__q"], int(z[n + "__k"]), int(z[n + "__chunk"])
        gi, gs = z[n + "__ids"], z[n + "__scores"]
        ri, rs = O.ref_retrieve_batch(V, q, k, ch)
        assert np.array_equal(ri, gi), n
        np.testing.assert_allclose(rs, gs, rtol=0, atol=1e-5, err_msg=n)
        ci, cs = O.c_retrieve(V, q, k, sentinel=True)
        assert np.array_equal(ci, gi), n
        np.testing.assert_allclose(cs, gs, rtol=0, atol=TOL, err_msg=n)


def test_retrieve_zero_sentinel_semantics(golden_dir):
    """index.py:192-193: lists start as k x (row 0, 0.0): negative scores never enter, short lists are padded."""
    z = np.load(os.path.join(golden_dir, "retrieve_batch.npz"))
    gi, gs = z["g1_negative__ids"], z["g1_negative__scores"]
    assert (gs >= 0).all() and (gs[:, -1] == 0).all() and (gi[gs == 0] == 0).all()
    gi, gs = z["g1_smallD__ids"], z["g1_smallD__scores"]
    assert (gs[:, 60:] == 0).all() and (gi[:, 60:] == 0).all()
    # clean mode of the oracle (no sentinel) pads with (-1, -inf) instead
    ci, cs = O.c_retrieve(z["g1_smallD__V"], z["g1_smallD__q"], 100, sentinel=False)
    assert (ci[:, 60:] == -1).all() and np.isneginf(cs[:, 60:]).all() and (ci[:, :60] >= 0).all()


def test_score_batch_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "score_batch.npz"))
    V, q, cand, gs = z["V"], z["q"], z["cand"], z["scores"]
    assert bool(z["unknown_key_raises"])
    for qi in range(q.shape[0]):
        rs = O.ref_score_batch(V, q[qi:qi + 1], cand)
        np.testing.assert_allclose(rs[0], gs[qi], rtol=0, atol=1e-5)
    cs = O.c_score_candidates(V[None], q, np.broadcast_to(cand, (q.shape[0], cand.size)).copy())
    np.testing.assert_allclose(cs[:, :, 0], gs, rtol=0, atol=TOL)


def test_linear_weights_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "linear_weights.npz"))
    W, x2, q1, x3, q4, w2 = z["W"], z["x2"], z["q1"], z["x3"], z["q4"], z["w2"]
    np.testing.assert_allclose(O.ref_linear_weights(x2, q1, W), z["eval_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.ref_linear_weights(x3, q4, W), z["train_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.ref_linear_weights(x3, None, w2, query_cond=False), z["nocond_out"], rtol=0, atol=1e-6)
    # C oracle: gate + mix, eval shape [C,F] x [1,E]
    w = O.c_gate(q1[0], W)
    np.testing.assert_allclose(O.c_mix(x2, w), z["eval_out"][0], rtol=0, atol=TOL)
    for b in range(x3.shape[0]):
        np.testing.assert_allclose(O.c_mix(x3[b], O.c_gate(q4[b], W)), z["train_out"][b], rtol=0, atol=TOL)
        np.testing.assert_allclose(O.c_mix(x3[b], O.c_gate(q4[b], w2[:, 0], query_cond=False)), z["nocond_out"][b], rtol=0, atol=TOL)


def test_deterministic_exp_accuracy():
    xs = np.concatenate([-np.logspace(-6, np.log10(79.0), 400), [0.0, -1e-8, -0.5, -20.0]]).astype(np.float32)
    got = np.array([O.c_exp(float(x)) for x in xs], dtype=np.float64)
    ref = np.exp(xs.astype(np.float64))
    rel = np.abs(got - ref) / ref
    assert rel.max() < 3e-7, rel.max()
    assert O.c_exp(-100.0) == 0.0 and O.c_exp(0.0) == 1.0


def test_trec_eval_step_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "trec_eval_step.npz"))
    for n in _cases(z):
        b = str(z[n + "__base"]) if n + "__base" in z.files else n
        slab, q, W, mask = z[b + "__slab"], z[b + "__q"], z[b + "__W"], z[n + "__mask"]
        gi, gs = z[n + "__ids"], z[n + "__scores"]
        pi, ps = O.ref_two_stage(slab, q, W, mask)
        assert np.array_equal(pi, gi), n
        np.testing.assert_allclose(ps, gs, rtol=2e-6, atol=1e-5, err_msg=n)  # set-order dependent gather matmul
        r = O.c_two_stage(slab, q, W, mask)
        assert (r["n_valid"] == 100).all()
        O.assert_topk_equivalent(r["ids"], r["scores"], gi, gs, tol=TOL, what=n)
        assert np.array_equal(r["ids"], gi), n   # the goldens are tie-free: ids match exactly


def test_first_qres_line_format(golden_dir):
    """trec.py:49-50 line format as printed by contrastive.py:699-704."""
    z = np.load(os.path.join(golden_dir, "trec_eval_step.npz"))
    line = str(z["t_f4__first_line"])
    parts = line.split("\t")
    assert len(parts) == 6 and parts[0] == "qid0" and parts[1] == "0" and parts[3] == "0" and parts[5] == "0"
    assert float(parts[4]) == float(np.float32(float(parts[4])))  # an fp32 value printed via python float repr


def test_two_stage_fewer_candidates_than_k():
    """contrastive.py:696: topk(k=100) over fewer than 100 candidates raises in the reference; the oracle reports
    n_valid < k2 so the host layer can raise the same way."""
    rng = np.random.default_rng(5)
    slab = rng.standard_normal((1, 40, 32)).astype(np.float32) - 2.0   # mostly negative: lists padded with doc 0
    q = rng.standard_normal((2, 32)).astype(np.float32)
    W = rng.standard_normal((32, 1)).astype(np.float32)
    r = O.c_two_stage(slab, q, W)
    assert (r["n_valid"] == r["n_cand"]).all() and (r["n_cand"] < 100).all()
    assert (r["ids"][0, r["n_valid"][0]:] == -1).all()
    with pytest.raises(RuntimeError):
        O.ref_two_stage(slab, q, W)


def test_sharded_merge_equals_unsharded():
    """per-shard lists merged == unsharded list, for 1/2/4/8 shards (oracle-level statement of the
    multi-GPU contract; the HIP merge kernel is checked against the same property on the GPU)."""
    rng = np.random.default_rng(11)
    D, E, Q, k = 1000, 32, 5, 100
    V = (rng.standard_normal((D, E)) * 0.5 + 0.1).astype(np.float32)
    V[100:140] = V[100]  # identical "empty field" rows -> exact ties across shard boundaries
    q = rng.standard_normal((Q, E)).astype(np.float32)
    for sentinel in (True, False):
        gi, gs = O.c_retrieve(V, q, k, sentinel)
        for S in (1, 2, 4, 8):
            bounds = [D * r // S for r in range(S + 1)]
            parts = [O.c_retrieve(V[bounds[r]:bounds[r + 1]], q, k, sentinel, row_offset=bounds[r]) for r in range(S)]
            for i in range(Q):
                mi, ms = O.c_merge_lists(np.stack([p[0][i] for p in parts]), np.stack([p[1][i] for p in parts]), sentinel)
                assert np.array_equal(mi, gi[i]) and np.array_equal(ms, gs[i]), (sentinel, S, i)


def test_schema_golden_shape(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "schema.json")))
    assert [len(d[f"{ds}|all_dense"]) for ds in ("mag", "prime", "amazon")] == [5, 22, 8]


# ------------------------------------------------------------------------------------------------ larger goldens (SURVEY 8(c) sizes)
def test_retrieve_batch_large_golden(golden_dir):
    """G1 at D = 5000 x E = 768 through the reference's chunk-merge path (vector_batch_size = 256)."""
    z = np.load(os.path.join(golden_dir, "retrieve_batch_large.npz"))
    n = "g1_5000x768_chunk256"
    V, q = z[n + "__V16"].astype(np.float32), z[n + "__q16"].astype(np.float32)
    gi, gs = z[n + "__ids"], z[n + "__scores"]
    ri, rs = O.ref_retrieve_batch(V, q, 100, 256)
    assert np.array_equal(ri, gi)
    np.testing.assert_allclose(rs, gs, rtol=0, atol=1e-5)
    ci, cs = O.c_retrieve(V, q, 100, sentinel=True)
    assert np.array_equal(ci, gi)
    np.testing.assert_allclose(cs, gs, rtol=0, atol=TOL)


def check_tie_grid_case(z, ids, scores):
    """The tie-heavy golden: scores must equal the reference's BIT FOR BIT (exact arithmetic on the value grid); ids must
    be canonical, and equal the reference's wherever the reference had no choice (scores strictly above its k-th)."""
    gi, gs = z["g1_ties_grid__ids"], z["g1_ties_grid__scores"]
    V, q = z["g1_ties_grid__Vg"].astype(np.float32) / 8.0, z["g1_ties_grid__qg"].astype(np.float32) / 8.0
    full = (q.astype(np.float64) @ V.astype(np.float64).T).astype(np.float32)     # exact: every partial sum is representable
    assert np.array_equal(np.asarray(scores).view(np.uint32), gs.view(np.uint32))
    for i in range(q.shape[0]):
        above = gs[i] > gs[i, -1]
        assert np.array_equal(np.asarray(ids)[i][above], gi[i][above]), i
        # canonical order: (score desc, id asc) over ALL rows, zero sentinel (only positive scores enter)
        order = np.lexsort((np.arange(V.shape[0]), -full[i].astype(np.float64)))
        order = order[full[i][order] > 0][:100]
        want = np.zeros(100, np.int64)
        want[:order.size] = order
        assert np.array_equal(np.asarray(ids)[i], want), i


def test_retrieve_tie_heavy_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "retrieve_batch_large.npz"))
    V, q = z["g1_ties_grid__Vg"].astype(np.float32) / 8.0, z["g1_ties_grid__qg"].astype(np.float32) / 8.0
    gs = z["g1_ties_grid__scores"]
    ri, rs = O.ref_retrieve_batch(V, q, 100, 1000)
    assert np.array_equal(np.sort(rs, axis=1)[:, ::-1].view(np.uint32), gs.view(np.uint32))     # same score multiset
    ci, cs = O.c_retrieve(V, q, 100, sentinel=True)
    check_tie_grid_case(z, ci, cs)
    assert (np.diff(gs[0]) == 0).sum() > 90            # query 0 hits the block of 500 identical rows: a list full of ties


@pytest.mark.parametrize("name", ["L_f1_e768", "L_f4_e768", "L_f8_e768"])
def test_trec_eval_step_large_golden(golden_dir, name):
    """G5 at E = 768, F in {1, 4, 8}: the unmodified reference trec_eval_step vs the port and the C oracle."""
    z = np.load(os.path.join(golden_dir, f"trec_eval_step_{name}.npz"))
    slab, q, W, mask = (z[k].astype(np.float32) for k in ("slab16", "q16", "W16", "mask"))
    gi, gs = z["ids"], z["scores"]
    pi, ps = O.ref_two_stage(slab, q, W, mask)
    assert np.array_equal(pi, gi)
    np.testing.assert_allclose(ps, gs, rtol=2e-6, atol=1e-5)
    r = O.c_two_stage(slab, q, W, mask)
    O.assert_topk_equivalent(r["ids"], r["scores"], gi, gs, tol=TOL, what=name)
    assert np.array_equal(r["ids"], gi)


# ------------------------------------------------------------------------------------------------ the mismatch classifier (bench.py's gate)
def _run(ids, scores, fid, fsc):
    return dict(ids=np.asarray(ids), scores=np.asarray(scores, np.float32), field_ids=np.asarray(fid), field_scores=np.asarray(fsc, np.float32))


def test_classify_topk_mismatch():
    fid = np.array([[1, 2, 3, 4]])
    fsc = np.array([[4.0, 3.0, 2.0, 1.0]])
    a = _run([1, 2, 3], [3.0, 2.0, 1.0], fid, fsc)
    assert O.classify_topk_mismatch(a, _run([1, 2, 3], [3.0, 2.0, 1.0], fid, fsc))[0] == "identical"
    # order swap inside a 1e-5 tie
    b = _run([2, 1, 3], [3.0, 3.0 - 1e-5, 1.0], fid, fsc)
    a2 = _run([1, 2, 3], [3.0, 3.0 - 1e-5, 1.0], fid, fsc)
    assert O.classify_topk_mismatch(a2, b)[0] == "order_in_tie"
    # order swap with a real gap
    assert O.classify_topk_mismatch(a, _run([2, 1, 3], [3.0, 2.0, 1.0], fid, fsc))[0] == "other"
    # the last id differs, both were candidates and sit at the cut-off
    c = _run([1, 2, 4], [3.0, 2.0, 1.0 + 1e-5], fid, fsc)
    assert O.classify_topk_mismatch(a, c)[0] == "final_cutoff_tie"
    # ... with a score far above the cut-off it is a real disagreement
    assert O.classify_topk_mismatch(a, _run([1, 4, 2], [3.0, 2.5, 2.0], fid, fsc))[0] == "other"
    # the other side never had the id as a candidate: explained only by a near-tie at the END of its stage-1 list
    fid_b, fsc_b = np.array([[1, 2, 3, 9]]), np.array([[4.0, 3.0, 2.0, 1.0 + 2e-5]])
    d = _run([1, 2, 9], [3.0, 2.0, 1.0], fid_b, fsc_b)
    a3 = _run([1, 2, 4], [3.0, 2.0, 1.0], fid, fsc)
    assert O.classify_topk_mismatch(a3, d)[0] == "stage1_cutoff_tie"
    fsc_far = np.array([[4.0, 3.0, 2.0, 1.5]])
    assert O.classify_topk_mismatch(a3, _run([1, 2, 9], [3.0, 2.0, 1.0], fid_b, fsc_far))[0] == "other"
    # scores of common ids further apart than the tolerance
    assert O.classify_topk_mismatch(a, _run([1, 2, 3], [3.0, 2.0, 1.001], fid, fsc))[0] == "other"


def test_tolerant_checker_is_absolute():
    ids = np.arange(5)[None]
    sc = np.array([[50.0, 40.0, 30.0, 20.0, 10.0]], np.float32)
    O.assert_topk_equivalent(ids, sc, ids, sc + 5e-5, tol=1e-4)
    with pytest.raises(AssertionError):
        O.assert_topk_equivalent(ids, sc, ids, sc + 2e-3, tol=1e-4)            # 1e-4 is absolute, not scaled by |score|
    O.assert_topk_equivalent(ids, sc, ids, sc + 2e-3, tol=1e-4, relative=True)
    swapped = np.array([[0, 1, 2, 4, 3]])
    with pytest.raises(AssertionError):                                        # the last position gets no free pass
        O.assert_topk_equivalent(ids, sc, swapped, sc, tol=1e-4)
