"""GPU parity tests of the certified two-level stage 2 (csrc/mfar_select.h: fp16 gather slab -> interval bounds on the mixed
score -> fp32 rows of the survivors only).  It must never change a bit: every case compares against the full gather
(`set_stage2_mode(0)`) and, where the oracle finishes in seconds, against `O.c_two_stage`.

Reference semantics: DenseFlatIndex.score_batch x F + mask + LinearWeights + topk (reference mfar/data/index.py:227-232,
mfar/modeling/contrastive.py:681-696)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mfar_oracle as O


@pytest.fixture(scope="module")
def idxmod():
    from mfar.data import index
    return index


# Every test of this file runs with BOTH kernel families of the tail (include/mfar_hip.h mfar_set_stage2_kernels): round 6's gate / front /
# bounds / select kernels (default) and the one-workgroup-per-query kernels of rounds 3-5.  Same bits, whichever runs.
_FAMILY = [1]


@pytest.fixture(autouse=True, params=[1, 0], ids=["round6_kernels", "round3_5_kernels"])
def _kernel_family(request):
    _FAMILY[0] = request.param
    yield
    _FAMILY[0] = 1


def _mk(rng, F, D, E, Q, mean=0.3, sigma=0.5, dup=0):
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * sigma + mean * mu * 4.0).astype(np.float32)
    if dup:
        for f in range(F):
            rows = rng.choice(D, size=min(dup, D), replace=False)
            slab[f, rows] = slab[f, rows[0]]
    q = (rng.standard_normal((Q, E)) * sigma + mu * 2.0).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    return slab, q, W


def _load(idxmod, slab, row_offset=0, screen=2):
    F, D, E = slab.shape
    ix = idxmod.MultiFieldIndex(D, F, E, device=0, row_offset=row_offset)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    ix.set_screen(screen)
    ix.set_stage2_kernels(_FAMILY[0])
    return ix


def _same(a, b, what):
    assert np.array_equal(np.asarray(a["ids"]), np.asarray(b["ids"])), (what, "ids")
    assert np.array_equal(np.asarray(a["scores"]).view(np.uint32), np.asarray(b["scores"]).view(np.uint32)), (what, "score bits")
    assert np.array_equal(np.asarray(a["n_valid"]), np.asarray(b["n_valid"])), (what, "n_valid")


def test_two_level_equals_full_gather_and_oracle(idxmod):
    """Several shapes (dims that are / are not multiples of 64, 3 .. 22 fields, duplicates, both sentinel modes): the two-level
    stage 2 is in use (stats), prunes (survivors < candidates) and returns the full gather's bits = the oracle's bits."""
    rng = np.random.default_rng(300)
    for F, D, E, Q, mean in ((8, 9000, 768, 70, 0.3), (22, 6000, 128, 33, 0.2), (3, 17000, 96, 20, 0.3), (5, 4000, 64, 130, -0.2)):
        slab, q, W = _mk(rng, F, D, E, Q, mean=mean, dup=7)
        mask = np.ones(F, np.float32)
        mask[1] = 0
        ix = _load(idxmod, slab)
        for sentinel in (True, False):
            ix.set_stage2_mode(1)
            s0 = ix.stage2_stats()
            r1 = ix.search(q, W, mask, sentinel=sentinel, return_fields=True)
            s1 = ix.stage2_stats()
            assert s1["two_level"] and s1["gather_slab_bytes"] >= F * D * E * 2, s1
            seen, kept = s1["n_candidates"] - s0["n_candidates"], s1["n_survivors"] - s0["n_survivors"]
            assert seen == int(np.asarray(r1["n_cand"]).sum()), (seen, "every candidate went through the prune kernel")
            if sentinel and mean > 0:
                assert kept < 0.6 * seen, (F, D, E, kept, seen)      # it does prune on this data
            ix.set_stage2_mode(0)
            r0 = ix.search(q, W, mask, sentinel=sentinel, return_fields=True)
            assert ix.stage2_stats()["n_candidates"] == s1["n_candidates"]      # mode 0 does not touch the prune kernel
            _same(r1, r0, (F, D, E, sentinel))
            assert np.array_equal(r1["n_cand"], r0["n_cand"])
            o = O.c_two_stage(slab, q, W, mask, sentinel=sentinel)
            assert np.array_equal(r1["ids"], o["ids"]) and np.array_equal(r1["scores"].view(np.uint32), o["scores"].view(np.uint32))
        ix.close()


def test_two_level_masks_of_any_sign_and_sweeps(idxmod):
    """Zero, negative and fractional mask entries (a negative entry swaps the interval ends), no query conditioning, and the
    one-pass mask sweep (survivors = union over the masks): bits of the full gather."""
    import torch
    rng = np.random.default_rng(301)
    F, D, E, Q = 6, 8000, 128, 40
    slab, q, W = _mk(rng, F, D, E, Q)
    ix = _load(idxmod, slab)
    masks = np.array([[1, 1, 1, 1, 1, 1], [0, 1, 0, 1, 1, 0], [1, -1, 1, 1, -0.5, 1], [-1, -1, -1, -1, -1, -1], [0.25, 2, 1, 0, 1, 3],
                      [0, 0, 0, 0, 0, 0]], np.float32)
    for qc in (True, False):
        Wq = W if qc else W[0].copy()
        for m in masks:
            ix.set_stage2_mode(1)
            r1 = ix.search(q, Wq, m, query_cond=qc)
            ix.set_stage2_mode(0)
            r0 = ix.search(q, Wq, m, query_cond=qc)
            _same(r1, r0, (qc, m.tolist()))
    # the sweep entry point: one union of survivors serves every mask
    dev = torch.device("cuda:0")
    qd, Wd, md = torch.from_numpy(q).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(masks).to(dev)
    fid, _ = ix.retrieve_fields(qd, 100, True)
    ix.set_stage2_mode(1)                       # mode 1 leaves sweeps of more than two masks to the full gather ...
    s0 = ix.stage2_stats()
    sw1 = ix.search_stage2_masks(qd, Wd, fid, md)
    torch.cuda.synchronize()
    assert ix.stage2_stats()["n_candidates"] == s0["n_candidates"]
    sw2 = ix.search_stage2_masks(qd, Wd, fid, md[:2].contiguous())
    torch.cuda.synchronize()
    assert ix.stage2_stats()["n_candidates"] > s0["n_candidates"]          # ... and prunes for two
    ix.set_stage2_mode(2)                       # mode 2: any number of masks, survivors = union over the masks
    s0 = ix.stage2_stats()
    sw = ix.search_stage2_masks(qd, Wd, fid, md)
    s1 = ix.stage2_stats()
    assert s1["n_candidates"] > s0["n_candidates"]
    torch.cuda.synchronize()
    for k_ in ("ids", "scores", "n_valid"):
        assert torch.equal(sw[k_], sw1[k_]) and torch.equal(sw[k_][:2], sw2[k_])
    ix.set_stage2_mode(0)
    for i, m in enumerate(masks):
        r0 = ix.search(q, W, m)
        assert np.array_equal(sw["ids"][i].cpu().numpy(), r0["ids"]), i
        assert np.array_equal(sw["scores"][i].cpu().numpy().view(np.uint32), r0["scores"].view(np.uint32)), i
        assert np.array_equal(sw["n_valid"][i].cpu().numpy(), r0["n_valid"]), i
    ix.close()


def test_two_level_when_the_bound_is_useless_or_data_is_hostile(idxmod):
    """eps_mult = 1e9 (every candidate survives), non-finite rows (NaN / inf in one field of some candidates: their mixed score is
    NaN and the mixer drops them), huge values, fewer valid candidates than k2: still the full gather's bits and n_valid."""
    rng = np.random.default_rng(302)
    F, D, E, Q = 4, 6000, 64, 25
    slab, q, W = _mk(rng, F, D, E, Q)
    ix = _load(idxmod, slab)
    ix.set_screen(2, 1e9)
    ix.set_stage2_mode(1)
    s0 = ix.stage2_stats()
    r1 = ix.search(q, W, None)
    s1 = ix.stage2_stats()
    assert s1["n_survivors"] - s0["n_survivors"] == s1["n_candidates"] - s0["n_candidates"] > 0
    ix.set_stage2_mode(0)
    _same(r1, ix.search(q, W, None), "eps_mult 1e9")
    ix.close()
    # hostile rows: the field statistics become non-finite -> eps = inf / NaN -> everything survives -> exact path decides
    bad = slab.copy()
    hot = rng.choice(D, 300, replace=False)
    bad[1, hot[:100]] = np.nan
    bad[2, hot[100:200]] = np.inf
    bad[3, hot[200:]] *= 1e30
    ix = _load(idxmod, bad)
    for sentinel in (True, False):
        ix.set_stage2_mode(1)
        r1 = ix.search(q, W, None, sentinel=sentinel)
        ix.set_stage2_mode(0)
        _same(r1, ix.search(q, W, None, sentinel=sentinel), ("hostile", sentinel))
    ix.close()
    # k2 larger than the number of positive-scoring rows: n_valid < k2 in both modes
    neg = -np.abs(slab[:, :3000])
    neg[:, :40] = np.abs(slab[:, :40])
    ix = _load(idxmod, neg.astype(np.float32))
    qq = np.abs(q)
    ix.set_stage2_mode(1)
    r1 = ix.search(qq, W, None)
    ix.set_stage2_mode(0)
    r0 = ix.search(qq, W, None)
    _same(r1, r0, "few candidates")
    assert (np.asarray(r0["n_valid"]) <= 41).all()
    ix.close()


def test_two_level_in_the_lists_first_exchange(idxmod):
    """Row shards (mfar_search_owned prunes against the OWNED candidates' own k2-th lower bound) and the pipelined searcher
    (two slots of two-level scratch in flight): bits of the unsharded full gather."""
    import torch
    from mfar.data.pipeline import PipelinedSearcher
    rng = np.random.default_rng(303)
    F, D, E, Q = 5, 12000, 128, 64
    slab, q, W = _mk(rng, F, D, E, Q, dup=5)
    mask = np.array([1, 1, 0, 1, 1], np.float32)
    dev = torch.device("cuda:0")
    qd, Wd, md = (torch.from_numpy(a).to(dev) for a in (q, W, mask))
    full = _load(idxmod, slab)
    full.set_stage2_mode(0)
    ref = full.search(q, W, mask)
    # pipelined, two-level on
    full.set_stage2_mode(1)
    ps = PipelinedSearcher(full, Wd, md, max_batch=16)
    tickets = [ps.submit(qd[c:c + 16]) for c in range(0, Q, 16)]
    got = [{k: v.cpu().numpy().copy() for k, v in ps.result(t).items()} for t in tickets]
    assert np.array_equal(np.concatenate([g["ids"] for g in got]), ref["ids"])
    assert np.array_equal(np.concatenate([g["scores"] for g in got]).view(np.uint32), ref["scores"].view(np.uint32))
    full.close()
    for S in (2, 3):
        bounds = [D * g // S for g in range(S + 1)]
        shards = [_load(idxmod, slab[:, bounds[g]:bounds[g + 1]], row_offset=bounds[g]) for g in range(S)]
        nl, nt = shards[0].lists_bytes(Q), shards[0].topk_bytes(Q)
        lists_all = torch.empty(S * nl, dtype=torch.uint8, device=dev)
        for g, sh in enumerate(shards):
            sh.retrieve_lists(qd, lists_all[g * nl:(g + 1) * nl], 100, True)
        topk_all = torch.empty(S * nt, dtype=torch.uint8, device=dev)
        for g, sh in enumerate(shards):
            sh.search_owned(lists_all, S, qd, Wd, topk_all[g * nt:(g + 1) * nt], md)
        r = idxmod.merge_topk(topk_all, S, Q)
        torch.cuda.synchronize()
        assert sum(sh.stage2_stats()["n_candidates"] for sh in shards) > 0
        assert np.array_equal(r["ids"].cpu().numpy(), ref["ids"]), S
        assert np.array_equal(r["scores"].cpu().numpy().view(np.uint32), ref["scores"].view(np.uint32)), S
        assert np.array_equal(r["n_valid"].cpu().numpy(), ref["n_valid"]), S
        for sh in shards:
            sh.close()


def test_two_level_reuses_stage1_scores(idxmod):
    """Known pairs: a candidate's score in the field whose stage-1 list it came from is taken from the list (`field_scores`), not
    gathered again.  Same bits with and without the lists' scores and as the full gather, zero-sentinel and clean lists (negative
    scores, -1 padding), short lists with (0, 0.0) padding; and the path is live -- handing it WRONG list scores changes results."""
    import torch
    rng = np.random.default_rng(305)
    dev = torch.device("cuda:0")
    for F, D, E, Q, mean in ((6, 9000, 128, 50, 0.3), (4, 5000, 64, 30, -0.35)):
        slab, q, W = _mk(rng, F, D, E, Q, mean=mean, dup=5)
        ix = _load(idxmod, slab)
        qd, Wd = torch.from_numpy(q).to(dev), torch.from_numpy(W).to(dev)
        md = torch.tensor([1.0] * (F - 1) + [0.0], device=dev)
        for sentinel in (True, False):
            fid, fsc = ix.retrieve_fields(qd, 100, sentinel)
            ix.set_stage2_mode(0)
            full = ix.search_stage2(qd, Wd, fid, md)
            ix.set_stage2_mode(1)
            plain = ix.search_stage2(qd, Wd, fid, md)
            reuse = ix.search_stage2(qd, Wd, fid, md, field_scores=fsc, sentinel=sentinel)
            wrong = ix.search_stage2(qd, Wd, fid, md, field_scores=fsc + 1.0, sentinel=sentinel)
            torch.cuda.synchronize()
            for k_ in ("ids", "scores", "n_valid"):
                assert torch.equal(plain[k_], full[k_]) and torch.equal(reuse[k_], full[k_]), (F, sentinel, k_)
            assert not torch.equal(wrong["scores"], full["scores"]), "the lists' scores are not being used"
            o = O.c_two_stage(slab, q, W, md.cpu().numpy(), sentinel=sentinel)
            assert np.array_equal(reuse["ids"].cpu().numpy(), o["ids"])
            assert np.array_equal(reuse["scores"].cpu().numpy().view(np.uint32), o["scores"].view(np.uint32))
        ix.close()


def test_score_dump_level_equals_row_gathers_and_oracle(idxmod):
    """The approximate level read from the wide screened pass's SCORE DUMP (include/mfar_hip.h "SCORE DUMP"; forced with
    set_stage2_dump(2)) instead of 16-bit row gathers: blocks of 65 .. 128 queries use it (info counter), prune, and return the bits of
    the gather path, of the full fp32 gather and of the oracle -- duplicates (the dump is indexed by UNIQUE row), masks of any sign, both
    sentinel modes, a mask sweep, hostile eps (everything survives), the split-phase pipeline over three slots, two row shards."""
    import torch
    rng = np.random.default_rng(305)
    for F, D, E, Q, mean, dup in ((22, 20000, 128, 128, 0.2, 300), (8, 17000, 768, 100, 0.3, 7), (3, 30000, 96, 65, -0.2, 0)):
        slab, q, W = _mk(rng, F, D, E, Q, mean=mean, dup=dup)
        mask = rng.choice(np.array([0.0, 1.0, -1.0, 0.5, 1.0, 1.0], np.float32), F)
        ix = _load(idxmod, slab)
        for sentinel in (True, False):
            o = O.c_two_stage(slab, q, W, mask, sentinel=sentinel)
            ix.set_stage2_dump(0)
            r_g = ix.search(q, W, mask, sentinel=sentinel)
            assert ix.stage2_dump_info()["n_launches"] == 0 or sentinel is False
            n0 = ix.stage2_dump_info()["n_launches"]
            ix.set_stage2_dump(2)
            s0 = ix.stage2_stats()
            r_d = ix.search(q, W, mask, sentinel=sentinel)
            s1 = ix.stage2_stats()
            info = ix.stage2_dump_info()
            assert info["n_launches"] == n0 + 1 and info["bytes_per_launch"] >= F * D * 256 * 0.5, info      # 16-bit codes: 256 B per scanned row
            if sentinel and mean > 0:
                assert s1["n_survivors"] - s0["n_survivors"] < 0.7 * (s1["n_candidates"] - s0["n_candidates"])      # the dump's bound prunes too
            _same(r_d, r_g, (F, D, E, sentinel, "dump vs gathers"))
            assert np.array_equal(r_d["ids"], o["ids"]) and np.array_equal(r_d["scores"].view(np.uint32), o["scores"].view(np.uint32))
        # more than 128 queries in one call: the dump would only hold the last block -- the gather path serves the call, same bits
        q2 = np.concatenate([q, q, q])[:168]
        n0 = ix.stage2_dump_info()["n_launches"]
        r2 = ix.search(q2, W, mask)
        assert ix.stage2_dump_info()["n_launches"] == n0
        o2 = O.c_two_stage(slab, q2, W, mask)
        assert np.array_equal(r2["ids"], o2["ids"]) and np.array_equal(r2["scores"].view(np.uint32), o2["scores"].view(np.uint32))
        # hostile bound: every candidate survives, still the same bits
        ix.set_screen(2, 1e9)
        r3 = ix.search(q, W, mask)
        o3 = O.c_two_stage(slab, q, W, mask)
        assert np.array_equal(r3["ids"], o3["ids"]) and np.array_equal(r3["scores"].view(np.uint32), o3["scores"].view(np.uint32))
        ix.set_screen(2, 1.0)
        ix.close()
    # the pipeline (three slots, coalesced launches of 128) and two row shards through the single-process exchange kernels
    from mfar.data.pipeline import PipelinedSearcher
    F, D, E = 6, 40000, 128
    slab, _, W = _mk(rng, F, D, E, 1, dup=11)
    ix = _load(idxmod, slab)
    ix.set_stage2_dump(2)
    dev = torch.device("cuda:0")
    Wd = torch.from_numpy(W).to(dev)
    qs = [(rng.standard_normal((64, E)) * 0.5 + 0.3).astype(np.float32) for _ in range(10)]
    ps = PipelinedSearcher(ix, Wd, None, max_batch=64)
    tk, got = [], []
    for i, qq in enumerate(qs):
        tk.append(ps.submit(torch.from_numpy(qq).to(dev)))
        if i >= ps.lag:
            got.append({k: v.clone() for k, v in ps.result(tk[i - ps.lag]).items()})
    for t in tk[len(got):]:
        got.append({k: v.clone() for k, v in ps.result(t).items()})
    torch.cuda.synchronize()
    assert ix.stage2_dump_info()["n_launches"] >= 5 and ps.n_redone == 0
    for qq, g in zip(qs, got):
        o = O.c_two_stage(slab, qq, W, None)
        assert np.array_equal(g["ids"].cpu().numpy(), o["ids"]) and np.array_equal(g["scores"].cpu().numpy().view(np.uint32), o["scores"].view(np.uint32))
    ix.close()
