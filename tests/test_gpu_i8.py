"""GPU parity tests of the int8 first level under the wide screened pass (csrc/mfar_i8.h: int8 scan -> certified superset per
(query, field) -> fp16 rows of those only -> the screen's own k' lists, exact re-scoring and certificate).  It must never change
a bit: every case compares the per-field lists and the final top-k with the level switched off (`set_i8(0)`), which the other
GPU tests pin to the oracle, and the small cases also against the oracle directly.

Reference semantics: DenseFlatIndex.retrieve_batch per field (reference mfar/data/index.py:181-222) + the mixer
(mfar/modeling/contrastive.py:681-696)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mfar_oracle as O


@pytest.fixture(scope="module")
def idxmod():
    from mfar.data import index
    return index


def _corpus(rng, F, D, E, Q, mean=0.3, sigma=0.5, dup=0, heavy=False, outliers=0):
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * sigma + mean * mu * 4.0).astype(np.float32)
    if heavy:                       # heavy-tailed row norms in field 0: the rows of large norm land in the coarse segment
        slab[0] *= np.exp(rng.standard_normal((D, 1)) * 0.8).astype(np.float32)
    if outliers:                    # a few elements far outside the bulk (what a single quantisation step would be sized by)
        for f in range(F):
            r, c = rng.integers(0, D, outliers), rng.integers(0, E, outliers)
            slab[f, r, c] *= 9.0
    if dup:
        for f in range(F):
            rows = rng.choice(D, size=min(dup, D), replace=False)
            slab[f, rows] = slab[f, rows[0]]
    q = (rng.standard_normal((Q, E)) * sigma + mu * 2.0).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    return slab, q, W


def _load(idxmod, slab):
    F, D, E = slab.shape
    ix = idxmod.MultiFieldIndex(D, F, E, device=0)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    ix.set_screen(2)
    return ix


def _bits(a):
    return np.asarray(a).view(np.uint32)


def test_i8_level_leaves_every_bit(idxmod):
    """Shapes of both ring variants (dim 768: 6-slot doc ring; 256 / 128: 4-slot), ragged wide blocks (65 .. 128 queries), both
    sentinel modes, duplicates, heavy-tailed norms, outlier elements: the level is in use (stats), hands the fp16 level fewer rows
    than the scan appended, and lists + final top-k equal the fp16 screen's bit for bit."""
    rng = np.random.default_rng(800)
    cases = ((3, 40000, 768, 128, dict(mean=0.3)), (2, 50000, 256, 100, dict(mean=-0.2, dup=9)), (4, 30000, 128, 128, dict(heavy=True, outliers=40)),
             (1, 70000, 768, 97, dict(mean=0.0, outliers=10)))
    for F, D, E, Q, kw in cases:
        slab, q, W = _corpus(rng, F, D, E, Q, **kw)
        ix = _load(idxmod, slab)
        for sentinel in (True, False):
            ix.set_i8(0)
            ids0, sc0 = ix.retrieve_fields(q, 100, sentinel)
            r0 = ix.search(q, W, None, sentinel=sentinel)
            ix.set_i8(1)
            s0 = ix.i8_stats()
            ids1, sc1 = ix.retrieve_fields(q, 100, sentinel)
            s1 = ix.i8_stats()
            assert s1["built"] and s1["slab_bytes"] >= F * D * E * 0.9, s1
            n_lists = s1["n_lists"] - s0["n_lists"]
            assert n_lists == Q * F, (n_lists, Q, F)
            app, surv = s1["n_appended"] - s0["n_appended"], s1["n_survivors"] - s0["n_survivors"]
            assert 0 < surv <= app, (app, surv)
            assert np.array_equal(ids0, ids1), (F, D, E, sentinel, "list ids")
            assert np.array_equal(_bits(sc0), _bits(sc1)), (F, D, E, sentinel, "list score bits")
            r1 = ix.search(q, W, None, sentinel=sentinel)
            assert np.array_equal(r0["ids"], r1["ids"]) and np.array_equal(_bits(r0["scores"]), _bits(r1["scores"]))
        st = ix.i8_stats()
        assert st["n_failed"] <= 0.02 * st["n_lists"], st        # the exact pass is the exception
        if kw.get("outliers") or kw.get("heavy"):
            assert st["seg1_rows"] > 0, st                        # the rows beyond the fine step went to the coarse segment
        ix.close()


def test_i8_level_against_the_oracle(idxmod):
    """A case the C oracle finishes in seconds: lists and two-stage result equal the oracle's bits with the level on."""
    rng = np.random.default_rng(801)
    F, D, E, Q = 2, 20000, 128, 80
    slab, q, W = _corpus(rng, F, D, E, Q, dup=5)
    ix = _load(idxmod, slab)
    ix.set_i8(1)
    r = ix.search(q, W, None, return_fields=True)
    assert ix.i8_stats()["n_lists"] > 0
    o = O.c_two_stage(slab, q, W, np.ones(F, np.float32), sentinel=True)
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(_bits(r["scores"]), _bits(o["scores"]))
    ix.close()


def test_i8_level_survives_rows_written_later(idxmod):
    """Rows written after the slab was built invalidate it with the screen: the next search rebuilds both."""
    rng = np.random.default_rng(802)
    F, D, E, Q = 2, 30000, 256, 128
    slab, q, W = _corpus(rng, F, D, E, Q)
    ix = _load(idxmod, slab)
    ix.set_i8(1)
    a = ix.retrieve_fields(q, 100, True)
    slab2 = slab.copy()
    slab2[1, 100:4000] = (rng.standard_normal((3900, E)) * 0.5).astype(np.float32) + q[5] * 0.5
    ix.write_rows(1, 100, slab2[1, 100:4000])
    b = ix.retrieve_fields(q, 100, True)
    ix.set_i8(0)
    c = ix.retrieve_fields(q, 100, True)
    assert np.array_equal(b[0], c[0]) and np.array_equal(_bits(b[1]), _bits(c[1]))
    assert not np.array_equal(a[0], b[0])
    ix.close()


def test_i8_level_failed_lists_go_through_the_exact_pass(idxmod, monkeypatch):
    """Half of the rows in the coarse segment and elements far outside the bulk: the coarse step is huge, its bound useless, the
    chunk lists of the scan pass their capacity and are closed -- those (query, field) lists must come from the exact repair pass
    and still carry the fp16 screen's bits."""
    monkeypatch.setenv("MFAR_I8_PCT", "0.5")          # read when the index is created
    rng = np.random.default_rng(803)
    F, D, E, Q = 2, 60000, 128, 128
    slab, q, W = _corpus(rng, F, D, E, Q, outliers=300)
    slab[0, rng.integers(0, D, 50), rng.integers(0, E, 50)] *= 40.0
    ix = _load(idxmod, slab)
    ix.set_i8(0)
    ids0, sc0 = ix.retrieve_fields(q, 100, True)
    ix.set_i8(1)
    ids1, sc1 = ix.retrieve_fields(q, 100, True)
    st = ix.i8_stats()
    assert st["built"] and st["seg1_rows"] > 0.4 * F * D and st["n_failed"] > 0, st
    assert np.array_equal(ids0, ids1) and np.array_equal(_bits(sc0), _bits(sc1))
    ix.close()
