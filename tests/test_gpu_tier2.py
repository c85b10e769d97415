"""TIER 2 of the certified screen -- the threshold rescan (csrc/mfar_screen.h, include/mfar_hip.h mfar_set_tier2) -- and the certified
screen on ENCODER-PRODUCED vectors.  Everything through the C ABI, ids and score BITS against the C oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mfar_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(r, o, fields=True):
    g = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in r.items()}
    ok = np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["scores"].view(np.uint32), o["scores"].view(np.uint32))
    if fields and "field_ids" in g:
        ok = ok and np.array_equal(g["field_ids"], o["field_ids"]) and np.array_equal(g["field_scores"].view(np.uint32), o["field_scores"].view(np.uint32))
    return bool(ok)


def _clustered(D, F, E, nq, noise, kinds=None):
    from mfar import synth
    cp = synth.SyntheticCorpus(D, F, E, n_queries=nq, seed=0xdeadbeef, device="cuda:0", field_kinds=kinds or ["clustered"] * F, cluster_noise=noise)
    slab = np.stack([cp.rows(f, 0, D).cpu().numpy() for f in range(F)])
    return cp, slab


@pytest.mark.parametrize("noise", [1e-2, 1e-4])
def test_tier2_finishes_the_lists_the_first_certificate_could_not(noise):
    """Clusters of near-duplicate rows: nearly every first certificate fails.  With tier 2 the failed lists are finished by the rescan --
    the exact pass is not needed -- and the result is the oracle's, bit for bit; without it (mode 0) the same bits come from the exact pass."""
    import torch
    from mfar.data import index as idxmod
    D, F, E, Q = 60_000, 4, 128, 128
    cp, slab = _clustered(D, F, E, 4 * Q, noise)
    ix = cp.build_index(idxmod)
    ix.set_auto_off(0)                                       # (the policy's switch-off is tested elsewhere: here every field stays screened)
    W, Wn = cp.W, cp.W.cpu().numpy()
    mask = np.array([1, 1, 0, 1], np.float32)
    q = cp.queries(0, Q)
    o = O.c_two_stage(slab, q.cpu().numpy(), Wn, mask)
    ix.set_tier2(0)
    r0 = ix.search(q, W, torch.from_numpy(mask).cuda(), return_fields=True)
    st0 = ix.screen_stats()
    assert _same(r0, o)
    assert st0["n_failed"] > Q, st0                          # the corpus does defeat the first certificate
    ix.set_tier2(2)
    r2 = ix.search(q, W, torch.from_numpy(mask).cuda(), return_fields=True)
    st2, t2 = ix.screen_stats(), ix.tier2_stats()
    assert _same(r2, o)
    assert t2["lists"] > Q, (t2, st0)                        # such lists went to tier 2 (fewer than before where ROW MODE, now active, certifies) ...
    assert t2["passed_on_to_exact"] <= t2["lists"] // 20, t2          # ... which finished (nearly) all of them
    assert st2["n_failed"] - st0["n_failed"] == t2["passed_on_to_exact"], (st0, st2, t2)
    assert t2["candidates_from_the_launch_scan"] + t2["lists_rescanned"] == t2["lists"], t2      # every such list: the launch's own scan, or the rescan
    ix.set_tier2(2 + 4)                                      # every list through the rescan (the fallback on demand): the same bits
    r6 = ix.search(q, W, torch.from_numpy(mask).cuda(), return_fields=True)
    t6 = ix.tier2_stats()
    assert _same(r6, o)
    assert t6["lists_rescanned"] - t2["lists_rescanned"] == t6["lists"] - t2["lists"] > Q, (t2, t6)
    ix.set_tier2(2)
    # 64-column blocks (two fp16 query terms, the other kernel family) take the same path
    r64 = ix.search(q[:40], W, torch.from_numpy(mask).cuda(), return_fields=True)
    assert _same(r64, {k: v[:40] for k, v in o.items()})
    ix.close()


def test_tier2_with_every_certificate_forced_to_fail_and_overflow_falls_back():
    """eps_mult makes EVERY first certificate fail and widens tier 2's band with it: moderate widening -> tier 2 finishes the lists from
    larger candidate sets; absurd widening -> the sets overflow (2048 rows / a full chunk list) and the exact pass decides.  Same bits always.
    Also: the zero sentinel with mostly negative scores (short lists, threshold = the sentinel), and no sentinel at all."""
    import torch
    from mfar.data import index as idxmod
    rng = np.random.default_rng(11)
    F, D, E, Q = 3, 40_000, 96, 100
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * 0.5 + 0.3 * mu * 4.0).astype(np.float32)
    slab[1] -= 2.2 * mu                                      # field 1: most scores negative -> short zero-sentinel lists
    q = (rng.standard_normal((Q, E)) * 0.5 + mu * 2.0).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    ix = idxmod.MultiFieldIndex(D, F, E, device=0)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    ix.set_auto_off(0)
    ix.set_tier2(2)
    for sentinel in (True, False):
        o = O.c_two_stage(slab, q, W, None, sentinel=sentinel)
        for mult, expect_overflow in ((30.0, False), (3000.0, True)):
            ix.set_screen(2, mult)
            t0, s0 = ix.tier2_stats(), ix.screen_stats()
            r = ix.search(q, W, None, sentinel=sentinel, return_fields=True)
            t1, s1 = ix.tier2_stats(), ix.screen_stats()
            assert _same(r, o), (sentinel, mult)
            lists, passed = t1["lists"] - t0["lists"], t1["passed_on_to_exact"] - t0["passed_on_to_exact"]
            assert lists > 0, (sentinel, mult, t0, t1)
            if expect_overflow:
                assert passed > 0 and s1["n_failed"] - s0["n_failed"] >= passed
            else:
                assert passed < lists, (sentinel, mult, lists, passed)
    ix.close()


def test_tier2_is_armed_by_failures_in_the_pipeline_and_keeps_fields_on():
    """Default settings (tier 2 auto, AUTO-OFF on) under the C-ABI pipeline: the first launches of a clustered corpus fail and are redone /
    repaired, their flags arm tier 2, and from then on the clustered fields stay ON (tier 2 finishes their lists: nothing is switched
    off, nothing redone).  Oracle bits before and after."""
    import torch
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    D, F, E, Q = 60_000, 4, 128, 64
    cp, slab = _clustered(D, F, E, 48 * Q, 1e-3, kinds=["plain", "clustered", "plain", "clustered"])
    ix = cp.build_index(idxmod)
    pl = NativePipeline(ix, cp.W, None, max_batch=Q)
    assert not ix.tier2_stats()["armed"]
    outs, tickets = [], []
    for j in range(48):
        tickets.append(pl.submit(cp.queries(j * Q, Q)))
        if j >= pl.lag:
            outs.append({k: v.clone() for k, v in pl.result(tickets[j - pl.lag]).items()})
    for t in tickets[48 - pl.lag:]:
        outs.append({k: v.clone() for k, v in pl.result(t).items()})
    torch.cuda.synchronize()
    t2, info = ix.tier2_stats(), ix.auto_off_info()
    assert t2["armed"] and t2["lists"] > 0, t2
    assert info["off"] == [], info                           # tier 2 finishes what the first certificate cannot: no field is switched off
    assert pl.n_redone <= 4, pl.n_redone                     # only the launches before tier 2 was armed
    Wn = cp.W.cpu().numpy()
    for j in (0, 1, 20, 47):
        assert _same(outs[j], O.c_two_stage(slab, cp.queries(j * Q, Q).cpu().numpy(), Wn, None), fields=False), j
    # steady state: failures are tier 2's, none reaches the exact pass
    s0, t0 = ix.screen_stats(), ix.tier2_stats()
    r = pl.result(pl.submit(cp.queries(5 * Q, Q)))
    s1, t1 = ix.screen_stats(), ix.tier2_stats()
    assert t1["lists"] > t0["lists"] and s1["n_failed"] - s0["n_failed"] == t1["passed_on_to_exact"] - t0["passed_on_to_exact"]
    pl.close()
    ix.close()


def test_certified_screen_on_encoder_produced_vectors_amazon_shaped():
    """an index of >= 20 000 rows built from TEXT -- amazon-shaped product families whose variants differ in one token,
    30 - 70 % of the fields missing -- through the BERT-base-shaped encoder (mean-pooled transformer outputs: a narrow cone, near-duplicate
    rows from near-duplicate texts), searched with the screen ON by the default pipeline: the oracle's bits, and the statistics say which
    tier finished the lists."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import encode_bench
    got = {}

    def check(module, st):
        from mfar.data.pipeline import NativePipeline
        dm = st.data_module
        dm.setup("test")
        with torch.no_grad():
            x = torch.cat([module.encode_query_batch(b) for loader in dm.test_dataloader()[:1] for b in loader]).contiguous()
        ix, W = module.slab, module._weights()
        assert ix.n_rows >= 20_000 and ix.screen_setting[0] == 1
        F = ix.n_fields
        slab = np.stack([ix.read_rows(f) for f in range(F)])
        Wn = W.cpu().numpy()
        pl = NativePipeline(ix, W, None, max_batch=64)
        outs = []
        for rnd in range(6):                                 # the adaptive policy (tier 2, ROW MODE, AUTO-OFF) learns over the first launches
            tk = [pl.submit(x[b:b + 64].contiguous()) for b in range(0, x.shape[0], 64)]
            outs = [{k: v.clone() for k, v in pl.result(t).items()} for t in tk]
        pl.close()
        st_ = ix.screen_stats()
        assert st_["built"] and st_["n_checked"] > 0
        o = O.c_two_stage(slab, x[:24].cpu().numpy(), Wn, None)
        assert _same({k: v[:24] for k, v in outs[0].items()}, o, fields=False)
        r = ix.search(x[:24], W, None, return_fields=True)   # the synchronous entry point, lists included
        assert _same(r, o)
        got.update(screen=st_, tier2=ix.tier2_stats(), off=ix.auto_off_info()["off"], geometry=float(torch.nn.functional.normalize(x, dim=1).mean(0).norm()))
        return {}
    check.__name__ = "oracle_check"
    encode_bench.run(20_000, 128, sweep=False, dataset="amazon", modes=("bf16",), probe=False, hooks=(check,))
    assert got["geometry"] > 0.9                             # the queries do sit in a narrow cone
    assert got["tier2"]["lists"] > 0, got                    # ... the first certificate does fail on such vectors, and tier 2 takes the lists
    assert got["tier2"]["passed_on_to_exact"] <= got["tier2"]["lists"] // 10, got


def test_tier2_on_a_bf16_index_keeps_the_chain_bits():
    """A bf16 index whose certificates fail (clusters that survive the bf16 rounding as distinct rows): until round 6 every failed list cost
    the exhaustive VALU chain pass (~30x a screened scan of its field); tier 2 rescans the slab itself, re-scores its candidates from the
    row-major companion with the natural-order chain and finishes the lists with exactly the bits the contract demands."""
    import torch
    from mfar.data import index as idxmod
    D, F, E, Q = 50_000, 3, 192, 100
    cp, slab = _clustered(D, F, E, 2 * Q, 2e-2)
    ix = cp.build_index(idxmod, dtype="bf16")
    ix.set_auto_off(0)
    ref = O.bf16_round(slab)
    q = cp.queries(0, Q)
    Wn = cp.W.cpu().numpy()
    with O.chain("natural"):
        o = O.c_two_stage(ref, q.cpu().numpy(), Wn, None)
    ix.set_tier2(0)
    r0 = ix.search(q, cp.W, None, return_fields=True)
    st0 = ix.screen_stats()
    assert _same(r0, o)
    assert st0["n_failed"] > Q // 2, st0                      # the corpus defeats the first certificate (the chain pass repaired those lists)
    ix.set_tier2(2)
    r2 = ix.search(q, cp.W, None, return_fields=True)
    st2, t2 = ix.screen_stats(), ix.tier2_stats()
    assert _same(r2, o)
    assert t2["lists"] > Q // 2 and t2["passed_on_to_exact"] <= t2["lists"] // 10, t2
    assert st2["n_failed"] - st0["n_failed"] == t2["passed_on_to_exact"], (st0, st2, t2)
    assert t2["candidates_from_the_launch_scan"] + t2["lists_rescanned"] == t2["lists"], t2      # every such list: the launch's own scan, or the rescan
    ix.set_tier2(2 + 4)                                      # every list through the rescan (the fallback on demand): the same bits
    r6 = ix.search(q, cp.W, None, return_fields=True)
    t6 = ix.tier2_stats()
    assert _same(r6, o)
    assert t6["lists_rescanned"] - t2["lists_rescanned"] == t6["lists"] - t2["lists"] > Q // 2, (t2, t6)
    ix.set_tier2(2)
    r64 = ix.search(q[:50], cp.W, None, return_fields=True)   # the 64-column pass (two bf16 query terms)
    assert _same(r64, {k: v[:50] for k, v in o.items()})
    ix.close()


def test_deep_scan_every_field_plain_clustered_and_sentinel_cases():
    """DEEP SCAN forced for every field (mode 2): no first certificate at all -- the scan's chunk lists hold the complete candidate sets above
    the sample-derived thresholds, the collect kernel narrows them to the band around the k-th best approximate score, tier 2's back half
    writes the lists.  Plain Gaussian rows, clusters of near-duplicates, a field whose scores are mostly negative (zero sentinel: the set
    must hold every row that can be positive), no sentinel, 128- and 40-query blocks: the oracle's bits, (nearly) nothing passed on."""
    import torch
    from mfar.data import index as idxmod
    D, F, E, Q = 60_000, 4, 128, 128
    for kinds, shift in ((None, 0.0), (["clustered", "plain", "clustered", "plain"], 0.0), (None, 1.6)):
        from mfar import synth
        cp = synth.SyntheticCorpus(D, F, E, n_queries=2 * Q, seed=0xdeadbeef, device="cuda:0", field_kinds=kinds, cluster_noise=1e-3)
        slab = np.stack([cp.rows(f, 0, D).cpu().numpy() for f in range(F)])
        if shift:
            slab[1] -= shift * cp.mu.cpu().numpy()           # field 1: most scores negative
        ix = idxmod.MultiFieldIndex(D, F, E, device=0)
        for f in range(F):
            ix.write_rows(f, 0, slab[f])
        ix.set_auto_off(0)
        ix.set_tier2(2)
        ix.set_deep_scan(2)
        assert ix.deep_scan_info()["fields"] == list(range(F))
        q = cp.queries(0, Q)
        Wn = cp.W.cpu().numpy()
        for sentinel in (True, False):
            o = O.c_two_stage(slab, q.cpu().numpy(), Wn, None, sentinel=sentinel)
            t0, s0 = ix.tier2_stats(), ix.screen_stats()
            r = ix.search(q, cp.W, None, sentinel=sentinel, return_fields=True)
            t1, s1 = ix.tier2_stats(), ix.screen_stats()
            assert _same(r, o), (kinds, shift, sentinel)
            assert t1["lists"] - t0["lists"] == Q * F, (t0, t1)          # every list took the deep path
            passed = t1["passed_on_to_exact"] - t0["passed_on_to_exact"]
            if not shift:
                assert passed <= Q * F // 20, (kinds, sentinel, passed)
            assert s1["n_failed"] - s0["n_failed"] == passed
            r40 = ix.search(q[:40], cp.W, None, sentinel=sentinel, return_fields=True)      # a 64-column block (two fp16 query terms)
            assert _same(r40, {k: v[:40] for k, v in o.items()}), (kinds, shift, sentinel, "64 columns")
        ix.close()


def test_deep_scan_is_learned_per_field_in_the_pipeline():
    """Deep scan in auto mode: the two clustered fields fail their first certificates launch after launch, tier 2 finishes their lists, and after
    eight such launches the policy makes exactly those fields DEEP fields (one scan instead of scan + rescan); the plain fields keep their
    certificates.  Oracle bits before and after the switch."""
    import torch
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    D, F, E, Q = 60_000, 4, 128, 64
    cp, slab = _clustered(D, F, E, 64 * Q, 1e-3, kinds=["plain", "clustered", "plain", "clustered"])
    ix = cp.build_index(idxmod)
    ix.set_deep_scan(1)                                      # (off by default: include/mfar_hip.h says why)
    pl = NativePipeline(ix, cp.W, None, max_batch=Q)
    outs, tickets = [], []
    n = 64
    for j in range(n):
        tickets.append(pl.submit(cp.queries(j * Q, Q)))
        if j >= pl.lag:
            outs.append({k: v.clone() for k, v in pl.result(tickets[j - pl.lag]).items()})
    for t in tickets[n - pl.lag:]:
        outs.append({k: v.clone() for k, v in pl.result(t).items()})
    torch.cuda.synchronize()
    info = ix.deep_scan_info()
    assert info["fields"] == [1, 3] and info["n_switched"] == 2, info
    assert ix.auto_off_info()["off"] == []
    Wn = cp.W.cpu().numpy()
    for j in (0, 2, 30, 63):
        assert _same(outs[j], O.c_two_stage(slab, cp.queries(j * Q, Q).cpu().numpy(), Wn, None), fields=False), j
    # a deep launch: the two deep fields' lists all count as tier-2 lists, none of the plain fields' do
    t0 = ix.tier2_stats()
    r = pl.result(pl.submit(cp.queries(7 * Q, Q)))
    t1 = ix.tier2_stats()
    assert t1["lists"] - t0["lists"] == 2 * Q and t1["passed_on_to_exact"] == t0["passed_on_to_exact"], (t0, t1)
    ix.set_deep_scan(0)                                      # switching it off: first attempts + rescans again, same bits
    assert ix.deep_scan_info()["fields"] == []
    r2 = pl.result(pl.submit(cp.queries(7 * Q, Q)))
    assert torch.equal(r["ids"], r2["ids"]) and torch.equal(r["scores"], r2["scores"])
    pl.close()
    ix.close()


def test_tier2_at_the_headline_shape_through_the_timed_path():
    """1 000 000 x 8 x 768 with every field made of near-duplicate clusters (`bench.py clustered_corpus`): the path the bench line times --
    two 64-query batches coalesced into one 128-column launch, both faces of the pipeline, nothing forced -- with tier 2 ARMED by a first
    failing launch, against the synchronous search of each half and, for 8 probe queries, the C oracle bit for bit (stage-1 lists proven
    complete by an exhaustive torch scan, O.c_two_stage on the union rows).  No list reaches the exact pass, no field is switched off."""
    import torch
    if torch.cuda.mem_get_info(0)[0] < 90 << 30:
        pytest.skip("needs ~80 GB of free HBM")
    from mfar.data import index as idxmod
    from mfar import synth
    from test_gpu_parity import _timed_path_check
    D, F, E, Q = 1_000_000, 8, 768, 64
    cp = synth.SyntheticCorpus(D, F, E, n_queries=8 * Q, seed=0xdeadbeef, device="cuda:0", field_kinds=["clustered"] * F, cluster_noise=1e-3)
    ix = cp.build_index(idxmod)
    mask = torch.ones(F, device="cuda:0")
    ix.search(cp.queries(4 * Q, 2 * Q), cp.W, mask)          # certificates fail, the exact pass repairs -- and the feedback arms tier 2
    torch.cuda.synchronize()
    assert ix.tier2_stats()["armed"]
    s0, t0 = ix.screen_stats(), ix.tier2_stats()
    _timed_path_check(idxmod, ix, cp, cp.W, mask, [0, 9, 21, 34, 42, 55, 60, 63])
    s1, t1 = ix.screen_stats(), ix.tier2_stats()
    assert t1["lists"] - t0["lists"] > 4 * Q * F and t1["passed_on_to_exact"] == t0["passed_on_to_exact"], (t0, t1)
    assert s1["n_failed"] == s0["n_failed"] and ix.auto_off_info()["off"] == []
    # ... and hardly any of them needed a second scan: the launch's own chunk lists held their candidates (sample threshold <= tier 2's,
    # no chunk compacted -- checked per list on the device)
    lists = t1["lists"] - t0["lists"]
    from_scan = t1["candidates_from_the_launch_scan"] - t0["candidates_from_the_launch_scan"]
    rescanned = t1["lists_rescanned"] - t0["lists_rescanned"]
    assert from_scan + rescanned == lists and rescanned <= lists // 10, (lists, from_scan, rescanned)
    # the fallback on demand (mode + 4: every list takes the rescan): the same bits
    ix.set_tier2(1 + 4)
    _timed_path_check(idxmod, ix, cp, cp.W, mask, [3, 17, 40, 63])
    t2 = ix.tier2_stats()
    assert t2["lists_rescanned"] - t1["lists_rescanned"] == t2["lists"] - t1["lists"] > 0, (t1, t2)
    assert t2["passed_on_to_exact"] == t1["passed_on_to_exact"]
    ix.close()
