"""CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the mFAR dense multi-field scoring path.

Two restatements of the reference algorithm live here (citations relative to /root/reference):

1. `ref_*` -- a torch "port" that performs the SAME torch ops as the reference, in the same order
   (index.py:181-232 `matmul`/`cat`/`topk`, weighting.py:17-29 `softmax`/`sum`, contrastive.py:669-704).
   It is the code timed as `cpu_baseline` (kind "port") by bench.py and the bridge between the golden
   vectors (captured from the real reference by tools/gen_golden.py) and the C oracle.

2. `c_*` -- ctypes bindings of oracle/mfar_oracle.c, the plain-C restatement whose arithmetic contract
   (k-ordered fmaf chains, deterministic exp, canonical (score desc, id asc) tie-break) the HIP kernels
   reproduce bit for bit.

Both are pinned against tests/golden/ by tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess
from functools import reduce

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile oracle/mfar_oracle.c with gcc (seconds). Returns the .so path."""
    so = os.path.join(_HERE, "_build", "libmfar_oracle.so")
    src = os.path.join(_HERE, "mfar_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i64, f32p, i64p, i32p = ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)
        L.mfar_oracle_dot.restype = ctypes.c_float
        L.mfar_oracle_dot.argtypes = [f32p, f32p, ctypes.c_int]
        L.mfar_oracle_exp.restype = ctypes.c_float
        L.mfar_oracle_exp.argtypes = [ctypes.c_float]
        L.mfar_oracle_scores.restype = None
        L.mfar_oracle_scores.argtypes = [f32p, i64, ctypes.c_int, f32p, ctypes.c_int, f32p]
        L.mfar_oracle_retrieve.restype = ctypes.c_int
        L.mfar_oracle_retrieve.argtypes = [f32p, i64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64, i64p, f32p]
        L.mfar_oracle_gate.restype = None
        L.mfar_oracle_gate.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p]
        L.mfar_oracle_mix.restype = None
        L.mfar_oracle_mix.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, f32p, f32p]
        L.mfar_oracle_score_candidates.restype = ctypes.c_int
        L.mfar_oracle_score_candidates.argtypes = [f32p, ctypes.c_int, i64, ctypes.c_int, i64, f32p, ctypes.c_int, i64p, ctypes.c_int, f32p]
        L.mfar_oracle_two_stage.restype = ctypes.c_int
        L.mfar_oracle_two_stage.argtypes = [f32p, ctypes.c_int, i64, ctypes.c_int, f32p, ctypes.c_int, f32p, ctypes.c_int, f32p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, i64p, f32p, i32p, i64p, f32p, i32p]
        L.mfar_oracle_set_chain.restype = None
        L.mfar_oracle_set_chain.argtypes = [ctypes.c_int]
        L.mfar_oracle_bf16_round.restype = None
        L.mfar_oracle_bf16_round.argtypes = [f32p, f32p, i64]
        L.mfar_oracle_merge_lists.restype = ctypes.c_int
        L.mfar_oracle_merge_lists.argtypes = [i64p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64p, f32p]
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


# ----------------------------------------------------------------------------- C oracle bindings
class chain:
    """`with chain("natural"):` switches the dot-product order to the bf16-slab contract (natural dim order); the default
    is the fp32-slab contract (the MFMA order documented in mfar_oracle.c)."""

    def __init__(self, mode: str):
        self.natural = {"mfma_f32": 0, "natural": 1}[mode]

    def __enter__(self):
        lib().mfar_oracle_set_chain(self.natural)

    def __exit__(self, *a):
        lib().mfar_oracle_set_chain(0)


def bf16_round(a):
    """fp32 -> bf16 (round to nearest even) -> fp32: the values a bf16 slab holds."""
    a = _f32(a)
    out = np.empty_like(a)
    lib().mfar_oracle_bf16_round(_p(a, ctypes.c_float), _p(out, ctypes.c_float), a.size)
    return out


def c_exp(x: float) -> np.float32:
    return np.float32(lib().mfar_oracle_exp(ctypes.c_float(x)))


def c_scores(V, q):
    V, q = _f32(V), _f32(q)
    D, E = V.shape
    out = np.empty((q.shape[0], D), dtype=np.float32)
    lib().mfar_oracle_scores(_p(V, ctypes.c_float), D, E, _p(q, ctypes.c_float), q.shape[0], _p(out, ctypes.c_float))
    return out


def c_retrieve(V, q, k, sentinel=True, row_offset=0):
    V, q = _f32(V), _f32(q)
    D, E = V.shape
    Q = q.shape[0]
    ids = np.empty((Q, k), dtype=np.int64)
    sc = np.empty((Q, k), dtype=np.float32)
    rc = lib().mfar_oracle_retrieve(_p(V, ctypes.c_float), D, E, _p(q, ctypes.c_float), Q, k, int(bool(sentinel)), row_offset,
                                    _p(ids, ctypes.c_int64), _p(sc, ctypes.c_float))
    assert rc == 0, rc
    return ids, sc


def c_gate(q1, W, query_cond=True):
    q1, W = _f32(q1).reshape(-1), _f32(W)
    if query_cond:
        E, F = W.shape
    else:
        F, E = W.size, q1.size
    w = np.empty(F, dtype=np.float32)
    lib().mfar_oracle_gate(_p(q1, ctypes.c_float), _p(W, ctypes.c_float), E, F, int(bool(query_cond)), _p(w, ctypes.c_float))
    return w


def c_mix(x, w, mask=None):
    x, w = _f32(x), _f32(w)
    C, F = x.shape
    m = _f32(mask) if mask is not None else None
    out = np.empty(C, dtype=np.float32)
    lib().mfar_oracle_mix(_p(x, ctypes.c_float), C, F, _p(w, ctypes.c_float), _p(m, ctypes.c_float), _p(out, ctypes.c_float))
    return out


def c_score_candidates(slab, q, cand, row_offset=0):
    slab, q = _f32(slab), _f32(q)
    F, D, E = slab.shape
    cand = np.ascontiguousarray(cand, dtype=np.int64)
    Q, C = cand.shape
    out = np.empty((Q, C, F), dtype=np.float32)
    rc = lib().mfar_oracle_score_candidates(_p(slab, ctypes.c_float), F, D, E, row_offset, _p(q, ctypes.c_float), Q,
                                            _p(cand, ctypes.c_int64), C, _p(out, ctypes.c_float))
    assert rc == 0
    return out


def c_two_stage(slab, q, W, mask=None, k1=100, k2=100, sentinel=True, query_cond=True):
    slab, q, W = _f32(slab), _f32(q), _f32(W)
    F, D, E = slab.shape
    Q = q.shape[0]
    m = _f32(mask).reshape(-1) if mask is not None else None
    ids = np.empty((Q, k2), dtype=np.int64)
    sc = np.empty((Q, k2), dtype=np.float32)
    nv = np.empty(Q, dtype=np.int32)
    nc = np.empty(Q, dtype=np.int32)
    fid = np.empty((Q, F, k1), dtype=np.int64)
    fsc = np.empty((Q, F, k1), dtype=np.float32)
    rc = lib().mfar_oracle_two_stage(_p(slab, ctypes.c_float), F, D, E, _p(q, ctypes.c_float), Q, _p(W, ctypes.c_float),
                                     int(bool(query_cond)), _p(m, ctypes.c_float), k1, k2, int(bool(sentinel)),
                                     _p(ids, ctypes.c_int64), _p(sc, ctypes.c_float), _p(nv, ctypes.c_int32),
                                     _p(fid, ctypes.c_int64), _p(fsc, ctypes.c_float), _p(nc, ctypes.c_int32))
    assert rc == 0, rc
    return dict(ids=ids, scores=sc, n_valid=nv, n_cand=nc, field_ids=fid, field_scores=fsc)


def c_search_fused(slab, q, W, mask=None, k=100, query_cond=True):
    """Fused mode (include/mfar_hip.h mfar_search_fused): per query the gate (contract softmax), the folded query
    [g_f * mask_f * q]_f, and the exhaustive top-k of its inner product with the concatenated field vectors -- the contract's
    fma chain over F * E dims, no zero sentinel."""
    slab, q, W = _f32(slab), _f32(q), _f32(W)
    F, D, E = slab.shape
    cat = np.ascontiguousarray(slab.transpose(1, 0, 2).reshape(D, F * E))
    m = _f32(mask).reshape(-1) if mask is not None else np.ones(F, np.float32)
    qf = np.empty((q.shape[0], F * E), np.float32)
    for i in range(q.shape[0]):
        g = c_gate(q[i], W, query_cond) * m                       # fp32 products, same operation order as the kernel
        qf[i] = (g[:, None] * q[i][None, :]).reshape(-1)
    return c_retrieve(cat, qf, k, sentinel=False)


def c_merge_lists(ids, scores, sentinel=True):
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    scores = _f32(scores)
    S, k = ids.shape
    oi = np.empty(k, dtype=np.int64)
    os_ = np.empty(k, dtype=np.float32)
    lib().mfar_oracle_merge_lists(_p(ids, ctypes.c_int64), _p(scores, ctypes.c_float), S, k, int(bool(sentinel)),
                                  _p(oi, ctypes.c_int64), _p(os_, ctypes.c_float))
    return oi, os_


# ----------------------------------------------------------------------------- torch port of the reference ops
def canon(ids, scores):
    """Canonical (score desc, id asc) ordering of one result list."""
    ids = np.asarray(ids, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    order = np.lexsort((ids, -scores.astype(np.float64)))
    return ids[order], scores[order]


def ref_retrieve_batch(V, q, top_k, vector_batch_size=1048576):
    """Same torch ops as DenseFlatIndex.retrieve_batch (index.py:181-212). Returns (ids[Q,k] int64, scores[Q,k])."""
    import torch
    V = np.asarray(V, dtype=np.float32)
    qe = torch.from_numpy(_f32(q))
    Q = qe.size(0)
    top_scores = torch.zeros((Q, top_k), dtype=torch.float32)        # zero sentinel, index.py:192
    top_indices = torch.zeros((Q, top_k), dtype=torch.int64)         # index.py:193
    for lb in range(0, V.shape[0], vector_batch_size):
        ub = min(V.shape[0], lb + vector_batch_size)
        vb = torch.from_numpy(V[lb:ub])
        scores = torch.matmul(qe, vb.t())                            # index.py:197
        cs = torch.cat([top_scores, scores], dim=1)
        ci = torch.cat([top_indices, torch.arange(lb, ub).unsqueeze(0).expand(Q, -1)], dim=1)
        ts, ti = torch.topk(cs, top_k, dim=1, largest=True, sorted=True)
        top_indices = ci[torch.arange(Q).unsqueeze(1), ti]
        top_scores = ts[:, :top_k]
    return top_indices.numpy(), top_scores.numpy()


def ref_score_batch(V, q1, rows):
    """index.py:227-232: gather rows, matmul -> [n_queries, len(rows)]."""
    import torch
    sel = torch.from_numpy(np.asarray(V, dtype=np.float32)[np.asarray(rows, dtype=np.int64)])
    return torch.matmul(torch.from_numpy(_f32(q1)).reshape(-1, sel.shape[1]), sel.t()).numpy()


def ref_linear_weights(x, q, W, query_cond=True):
    """weighting.py:17-29."""
    import torch
    x, W = torch.from_numpy(_f32(x)), torch.from_numpy(_f32(W))
    if query_cond:
        weights = torch.from_numpy(_f32(q)) @ W
    else:
        weights = W.transpose(1, 0)
    wd = torch.softmax(weights, dim=1)
    return torch.sum(wd.unsqueeze(1) * x, dim=-1).numpy()


def ref_two_stage(slab, q, W, mask=None, k1=100, k2=100, vector_batch_size=1048576, return_fields=False):
    """Port of trec_eval_step (contrastive.py:669-704) for query embeddings q[Q,E] (one embedding per query).
    Returns canonical (ids[Q,k2], scores[Q,k2]) [+ (field_ids[Q,F,k1], field_scores[Q,F,k1]) as torch.topk left them].
    Raises RuntimeError like torch.topk when C < k2."""
    import torch
    slab = np.asarray(slab, dtype=np.float32)
    F = slab.shape[0]
    q = _f32(q)
    Q = q.shape[0]
    stage1 = [ref_retrieve_batch(slab[f], q, k1, vector_batch_size) for f in range(F)]     # :672-674
    hits = [h[0] for h in stage1]
    m = torch.ones(F, 1) if mask is None else torch.from_numpy(_f32(mask)).reshape(F, 1)
    out_ids = np.empty((Q, k2), dtype=np.int64)
    out_sc = np.empty((Q, k2), dtype=np.float32)
    for i in range(Q):
        all_ids = sorted(reduce(lambda a, b: a | b, [set(h[i].tolist()) for h in hits]))  # :678-679
        new_hits = [torch.from_numpy(ref_score_batch(slab[f], q[i:i + 1], all_ids)) for f in range(F)]  # :681-683
        all_tens = torch.stack(new_hits, dim=0).squeeze(1) * m                              # :685-686
        scores = torch.from_numpy(ref_linear_weights(all_tens.t().numpy(), q[i:i + 1], W))  # :694
        values, indices = torch.topk(scores, k=k2, dim=1)                                   # :696
        ids = np.asarray(all_ids, dtype=np.int64)[indices.flatten().numpy()]
        out_ids[i], out_sc[i] = canon(ids, values.flatten().numpy())
    if return_fields:
        return out_ids, out_sc, np.stack(hits, axis=1), np.stack([h[1] for h in stage1], axis=1)
    return out_ids, out_sc


def assert_topk_equivalent(ids_a, sc_a, ids_b, sc_b, tol=1e-4, what="", relative=False):
    """Two canonical result lists computed with different fp32 summation orders: scores must agree within `tol`
    (ABSOLUTE -- the north-star's "scores within 1e-4 fp32"; `relative=True` scales it by max(1, max|score|) for data
    whose scores are far above 1), and ids must agree everywhere except inside runs of near-tied scores (gap <= 2*tol),
    where positions may be permuted.  The last position gets no special treatment: an id that differs there must be
    in a near-tie with its neighbour like anywhere else, or the id SETS must differ only by elements whose scores are
    within 2*tol of the cut-off (checked by `classify_topk_mismatch`, not here)."""
    ids_a, ids_b = np.asarray(ids_a), np.asarray(ids_b)
    sc_a, sc_b = np.asarray(sc_a, dtype=np.float64), np.asarray(sc_b, dtype=np.float64)
    assert ids_a.shape == ids_b.shape, (what, ids_a.shape, ids_b.shape)
    k = ids_a.shape[-1]
    for r in range(ids_a.reshape(-1, k).shape[0]):
        ia, ib = ids_a.reshape(-1, k)[r], ids_b.reshape(-1, k)[r]
        sa, sb = sc_a.reshape(-1, k)[r], sc_b.reshape(-1, k)[r]
        scale = 1.0
        if relative and np.isfinite(sa).any():
            scale = max(1.0, float(np.max(np.abs(sa[np.isfinite(sa)]))))
        np.testing.assert_allclose(sa, sb, rtol=0, atol=tol * scale, err_msg=f"{what} row {r} scores")
        for j in np.nonzero(ia != ib)[0]:
            near = any(0 <= jj < k and abs(sa[j] - sa[jj]) <= 2 * tol * scale for jj in (j - 1, j + 1))
            assert near, f"{what} row {r} pos {j}: ids {ia[j]} vs {ib[j]} with scores {sa[j]} vs {sb[j]}"


def classify_topk_mismatch(a, b, tol=1e-4):
    """Explain every difference between two runs of the two-stage scorer on the same inputs (different fp32 summation
    orders).  a, b: dicts with ids[k2], scores[k2] (canonical order), field_ids[F,k1], field_scores[F,k1] for ONE query.
    Returns one of
      "identical"        same ids in the same order
      "order_in_tie"     same id set; every position that differs sits in a run of scores within 2*tol
      "final_cutoff_tie" id sets differ only by ids whose mixed score is within 2*tol of the k2-th score on the side
                         that has them, and which the other side had as a CANDIDATE (it ranked them just below the cut)
      "stage1_cutoff_tie" ... or which the other side never saw as a candidate because, in every field whose list
                         carries the id on this side, its per-field score is within 2*tol of the other side's k1-th
                         (last) list score in that field (a rank-k1 near-tie in stage 1)
      "other"            anything else: a real disagreement
    plus the largest score difference over the ids both sides returned."""
    ia, ib = np.asarray(a["ids"]), np.asarray(b["ids"])
    sa, sb = np.asarray(a["scores"], dtype=np.float64), np.asarray(b["scores"], dtype=np.float64)
    pos_b = {int(x): j for j, x in enumerate(ib)}
    common = [(j, pos_b[int(x)]) for j, x in enumerate(ia) if int(x) in pos_b]
    dmax = max((abs(sa[j] - sb[jb]) for j, jb in common), default=0.0)
    if dmax > tol:
        return "other", dmax
    if np.array_equal(ia, ib):
        return "identical", dmax
    k = len(ia)
    if set(ia.tolist()) == set(ib.tolist()):
        for j in np.nonzero(ia != ib)[0]:
            if not any(0 <= jj < k and abs(sa[j] - sa[jj]) <= 2 * tol for jj in (j - 1, j + 1)):
                return "other", dmax
        return "order_in_tie", dmax
    cls = "final_cutoff_tie"
    for x, y in ((a, b), (b, a)):
        xi, xs = np.asarray(x["ids"]), np.asarray(x["scores"], dtype=np.float64)
        y_set = set(np.asarray(y["ids"]).tolist())
        y_cand = set(np.asarray(y["field_ids"]).reshape(-1).tolist())
        for j, d in enumerate(xi.tolist()):
            if d in y_set:
                continue
            if abs(xs[j] - xs[-1]) <= 2 * tol and d in y_cand:
                continue                                   # y ranked it just below its cut-off
            if d not in y_cand:
                # stage-1 explanation: in every field where x lists d, x's score of d is a near-tie with y's last entry
                fx, fsx = np.asarray(x["field_ids"]), np.asarray(x["field_scores"], dtype=np.float64)
                fsy = np.asarray(y["field_scores"], dtype=np.float64)
                fields = [f for f in range(fx.shape[0]) if d in fx[f].tolist()]
                ok = bool(fields) and all(abs(fsx[f][fx[f].tolist().index(d)] - fsy[f][-1]) <= 2 * tol for f in fields)
                if ok:
                    cls = "stage1_cutoff_tie"
                    continue
            return "other", dmax
    # the ids both sides kept may be permuted only inside near-ties
    return cls, dmax
