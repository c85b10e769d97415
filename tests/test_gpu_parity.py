"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the oracle.

Bar: BIT-EXACT ids and scores against the C oracle (oracle/mfar_oracle.c documents the arithmetic contract);
within 1e-4 against the golden vectors captured from the real reference (different fp32 summation order)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mfar_oracle as O

TOL = 1e-4


@pytest.fixture(scope="module")
def idxmod():
    from mfar.data import index
    return index


def _mk(rng, F, D, E, Q, mean=0.3, dup=0):
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * 0.5 + mean * mu * 4.0).astype(np.float32)
    if dup and D > 4:
        for f in range(F):  # shared "empty field" vector -> exact ties (format.py:58-59)
            rows = rng.choice(D, size=min(dup, D), replace=False)
            slab[f, rows] = slab[f, rows[0]]
    q = (rng.standard_normal((Q, E)) * 0.5 + mu * 2.0).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    return slab, q, W


def _load(idxmod, slab, row_offset=0):
    F, D, E = slab.shape
    ix = idxmod.MultiFieldIndex(D, F, E, device=0, row_offset=row_offset)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    return ix


def _gather_rows(ix, rows):
    """Rows `rows` (global ids) of every field, row-major fp32 [F, len(rows), E], read back through the C ABI."""
    import torch
    out = np.empty((ix.n_fields, len(rows), ix.dim), np.float32)
    buf = torch.empty(1, ix.dim, device=f"cuda:{ix.device}")
    for f in range(ix.n_fields):
        dev = torch.empty(len(rows), ix.dim, device=f"cuda:{ix.device}")
        for j, x in enumerate(rows):
            ix.read_rows(f, int(x) - ix.row_offset, 1, out=buf)
            dev[j] = buf[0]
        out[f] = dev.cpu().numpy()
    return out


def _exhaustive_stage1_check(ix, q, fid, fsc, queries, slack=2e-3):
    """Independent, EXHAUSTIVE check of the stage-1 lists at any corpus size (plain torch fp32 matmul over every row read
    back from the index): no row outside a list scores above the list's last entry, every row inside scores at least
    that, and the listed scores agree with torch's.  `slack` covers the different fp32 summation orders."""
    import torch
    dev = f"cuda:{ix.device}"
    qs = q[queries].to(dev)
    step = 65536
    for f in range(ix.n_fields):
        S = torch.empty(len(queries), ix.n_rows, device=dev)
        for r0 in range(0, ix.n_rows, step):
            n = min(step, ix.n_rows - r0)
            rows = torch.empty(n, ix.dim, device=dev)
            ix.read_rows(f, r0, n, out=rows)
            S[:, r0:r0 + n] = qs @ rows.t()
        for j, qi in enumerate(queries):
            ids = torch.as_tensor(fid[qi, f], device=dev) - ix.row_offset
            sc = torch.as_tensor(fsc[qi, f], device=dev)
            real = sc > 0                                   # zero-sentinel padding entries are (row 0, 0.0)
            kth = float(sc[-1])
            assert torch.allclose(S[j, ids[real]], sc[real], rtol=0, atol=slack), (f, qi)
            inside = torch.zeros(ix.n_rows, dtype=torch.bool, device=dev)
            inside[ids[real]] = True
            above = (S[j] > max(kth, 0.0) + slack) & ~inside
            assert not bool(above.any()), (f, qi, int(above.sum()))


def _oracle_check_on_subset(ix, q, W, mask, r, queries, n_random=3000, seed=0):
    """The FINAL mixed top-k (and the stage-1 lists) of `queries` against the C oracle, bit for bit, on a row subset that
    contains every stage-1 list member of those queries plus random rows (the exhaustive check above proves the lists
    complete, so the oracle's per-field top-k over the subset is the per-field top-k over the corpus)."""
    fid, fsc = r["field_ids"].cpu().numpy(), r["field_scores"].cpu().numpy()
    ids, sc = r["ids"].cpu().numpy(), r["scores"].cpu().numpy()
    D = ix.row_offset + ix.n_rows
    rng = np.random.default_rng(seed)
    rows = np.unique(np.concatenate([fid[queries].ravel(), rng.integers(ix.row_offset, D, n_random), [ix.row_offset]]))
    sub = _gather_rows(ix, rows)
    qq = q[queries].cpu().numpy()
    o = O.c_two_stage(sub, qq, W.cpu().numpy(), None if mask is None else mask.cpu().numpy())
    for j, qi in enumerate(queries):
        assert np.array_equal(rows[o["field_ids"][j]], fid[qi]), ("stage-1 ids", qi)
        assert np.array_equal(o["field_scores"][j].view(np.uint32), fsc[qi].view(np.uint32)), ("stage-1 score bits", qi)
        assert np.array_equal(rows[o["ids"][j]], ids[qi]), ("final ids", qi)
        assert np.array_equal(o["scores"][j].view(np.uint32), sc[qi].view(np.uint32)), ("final score bits", qi)


def _timed_path_check(idxmod, ix, corpus, W, mask, probe, first_batch=0, oracle_chain=None):
    """The path bench.py TIMES, at whatever shape `ix` has: 128 queries submitted as two 64-query batches to a default
    PipelinedSearcher (three launches in flight, two batches coalesced into ONE launch of the wide 128-column scan, the automatic
    score-dump policy -- nothing forced), against (1) the synchronous 64-query search of each half (the 64-column kernels): ids, score
    bits, n_valid and the per-field lists must be equal, and (2) the C oracle, bit for bit, for the `probe` queries of the first half
    (stage-1 lists proven complete by the exhaustive torch scan, then O.c_two_stage on the union rows).  Returns the searcher."""
    import contextlib
    import torch
    from mfar.data.pipeline import NativePipeline, PipelinedSearcher
    Q = 64
    halves = [corpus.queries((first_batch + j) * Q, Q) for j in range(2)]
    refs = [ix.search(h, W, mask, return_fields=True) for h in halves]
    torch.cuda.synchronize()
    # both faces of the pipeline: the C-ABI one (mfar_pipeline_*: what bench.py's headline and INTEGRATION.md's binding drive) and the
    # Python one (mfar/data/pipeline.py: the row-sharded exchange and the mask sweeps build on it)
    for cls in (NativePipeline, PipelinedSearcher):
        ps = cls(ix, W, mask, max_batch=Q)
        assert ps.depth == 3 and ps.coalesce == 2 and ps.Qmax == 128, (cls.__name__, ps.depth, ps.coalesce, ps.Qmax)      # the bench's configuration
        tickets = [ps.submit(h) for h in halves]
        got = []
        for t in tickets:
            r = {k: v.clone() for k, v in ps.result(t).items()}
            fid, fsc = ps.lists(t)
            r.update(field_ids=fid.clone(), field_scores=fsc.clone())
            got.append(r)
        torch.cuda.synchronize()
        assert ps.n_redone == 0
        for g, ref in zip(got, refs):
            for key in ("ids", "scores", "n_valid", "field_ids", "field_scores"):
                assert torch.equal(g[key], ref[key]), (cls.__name__, "pipelined 128-column launch vs synchronous 64-query search", key)
        if cls is NativePipeline:
            ps.close()
    fid, fsc = got[0]["field_ids"].cpu().numpy(), got[0]["field_scores"].cpu().numpy()
    _exhaustive_stage1_check(ix, halves[0], fid, fsc, probe)
    with (oracle_chain if oracle_chain is not None else contextlib.nullcontext()):
        _oracle_check_on_subset(ix, halves[0], W, mask, got[0], probe)
    return ps


def test_rows_roundtrip_tiled_layout(idxmod):
    rng = np.random.default_rng(0)
    for D, E in [(1, 32), (63, 32), (64, 64), (65, 96), (257, 768), (1000, 32)]:
        slab = rng.standard_normal((2, D, E)).astype(np.float32)
        ix = _load(idxmod, slab)
        for f in range(2):
            assert np.array_equal(ix.read_rows(f), slab[f])
        if D > 10:  # partial overwrite
            new = rng.standard_normal((5, E)).astype(np.float32)
            ix.write_rows(1, 3, new)
            slab[1, 3:8] = new
            assert np.array_equal(ix.read_rows(1), slab[1])
            assert np.array_equal(ix.read_rows(1, 4, 2), slab[1, 4:6])
        ix.close()


def test_stage1_bit_exact_vs_oracle(idxmod):
    """DenseFlatIndex.retrieve_batch semantics (index.py:181-222), all fields at once; also proves the
    v_mfma_f32_32x32x2_f32 accumulation order is the chain the oracle documents."""
    rng = np.random.default_rng(1)
    cases = [(1, 1, 32, 1, 5), (2, 63, 32, 3, 100), (1, 64, 32, 64, 100), (3, 257, 64, 5, 10), (2, 1000, 768, 7, 100),
             (4, 5000, 32, 65, 100), (1, 3000, 32, 130, 128), (8, 777, 96, 9, 1)]
    for F, D, E, Q, k in cases:
        for sentinel in (True, False):
            for mean in (0.3, -0.4):
                slab, q, _ = _mk(rng, F, D, E, Q, mean=mean, dup=7)
                ix = _load(idxmod, slab)
                ids, sc = ix.retrieve_fields(q, k, sentinel)
                for f in range(F):
                    oi, osc = O.c_retrieve(slab[f], q, k, sentinel)
                    assert np.array_equal(ids[:, f], oi), (F, D, E, Q, k, sentinel, mean, f)
                    assert np.array_equal(sc[:, f].view(np.uint32), osc.view(np.uint32)), (F, D, E, Q, k, sentinel, mean, f)
                ix.close()


def test_stage1_many_workgroups_and_compactions(idxmod):
    """Ascending scores force every element through the append path (worst case for the running threshold: the LDS
    staging area overflows into the direct path and the lists compact every tile); several chunks per field exercise the
    cross-workgroup merge.  Both 16-bit scan kernels (LDS ring: 2 k-steps, register ring: 6 k-steps), the exact fp32 pass,
    and the deepest list the ABI allows (k = 128 -> 192 screened entries)."""
    rng = np.random.default_rng(2)
    for E, k in ((32, 100), (96, 100), (96, 128)):
        F, D, Q = 2, 40000, 64
        slab, q, _ = _mk(rng, F, D, E, Q)
        # make field 1 scores increase with the row index for query 0
        ramp = np.linspace(0.0, 3.0, D, dtype=np.float32)[:, None] * (q[0] / np.dot(q[0], q[0]))[None, :]
        slab[1] = (slab[1] * 0.01 + ramp).astype(np.float32)
        ix = _load(idxmod, slab)
        want = [O.c_retrieve(slab[f], q, k, True) for f in range(F)]
        for screen in (0, 2):
            ix.set_screen(screen)
            for wgs in (1, 2, 4):
                ix.set_wgs_per_cu(wgs)
                ids, sc = ix.retrieve_fields(q, k, True)
                for f in range(F):
                    assert np.array_equal(ids[:, f], want[f][0]), (E, k, screen, wgs, f)
                    assert np.array_equal(sc[:, f].view(np.uint32), want[f][1].view(np.uint32)), (E, k, screen, wgs, f)
        ix.close()


def test_stage1_golden_reference(golden_dir, idxmod):
    z = np.load(os.path.join(golden_dir, "retrieve_batch.npz"))
    for n in sorted({k.split("__")[0] for k in z.files}):
        V, q, k = z[n + "__V"], z[n + "__q"], int(z[n + "__k"])
        ix = _load(idxmod, V[None])
        ids, sc = ix.retrieve_fields(q, k, True)
        O.assert_topk_equivalent(ids[:, 0], sc[:, 0], z[n + "__ids"], z[n + "__scores"], tol=TOL, what=n)
        assert np.array_equal(ids[:, 0], z[n + "__ids"]), n
        ix.close()


def test_stage2_bit_exact_and_golden(golden_dir, idxmod):
    rng = np.random.default_rng(3)
    slab, q, _ = _mk(rng, 3, 500, 64, 4)
    ix = _load(idxmod, slab, row_offset=1000)
    cand = rng.integers(1000, 1500, size=(4, 33)).astype(np.int64)
    cand[0, 0] = 5          # outside the shard -> NaN
    cand[1, 1] = -1
    x = ix.score_candidates(q, cand)
    ox = O.c_score_candidates(slab, q, cand, row_offset=1000)
    assert np.array_equal(x.view(np.uint32)[~np.isnan(ox)], ox.view(np.uint32)[~np.isnan(ox)])
    assert np.isnan(x[0, 0]).all() and np.isnan(x[1, 1]).all() and np.isnan(ox[0, 0]).all()
    ix.close()
    z = np.load(os.path.join(golden_dir, "score_batch.npz"))
    ix = _load(idxmod, z["V"][None])
    x = ix.score_candidates(z["q"], np.broadcast_to(z["cand"], (z["q"].shape[0], z["cand"].size)).copy())
    np.testing.assert_allclose(x[:, :, 0], z["scores"], rtol=0, atol=TOL)
    ix.close()


def test_mixer_bit_exact_and_golden(golden_dir, idxmod):
    rng = np.random.default_rng(4)
    Q, C, F, E, k = 5, 300, 6, 64, 100
    x = (rng.standard_normal((Q, C, F)) * 3).astype(np.float32)
    x[:, 10:20] = x[:, 10:11]                       # ties -> id tie-break
    ids = np.stack([rng.permutation(5000)[:C] for _ in range(Q)]).astype(np.int64)
    q = rng.standard_normal((Q, E)).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.2).astype(np.float32)
    mask = np.array([1, 0, 1, 1, 0, 1], dtype=np.float32)
    ncand = np.array([C, C - 7, 100, 40, 0], dtype=np.int32)
    r = idxmod.mix_topk(x, ids, q, W, mask, ncand, k=k)
    for i in range(Q):
        n = int(ncand[i])
        w = O.c_gate(q[i], W)
        mixed = O.c_mix(x[i, :n], w, mask) if n else np.zeros(0, np.float32)
        oi, osc = O.canon(ids[i, :n], mixed)
        m = min(n, k)
        assert r["n_valid"][i] == m
        assert np.array_equal(r["ids"][i, :m], oi[:m]) and np.array_equal(r["scores"][i, :m].view(np.uint32), osc[:m].view(np.uint32))
        assert (r["ids"][i, m:] == -1).all() and np.isneginf(r["scores"][i, m:]).all()
    # not query-conditioned: LinearWeights(num_fields, 1) (contrastive.py:286-287)
    w2 = rng.standard_normal(F).astype(np.float32)
    r2 = idxmod.mix_topk(x, ids, None, w2, None, None, k=10, query_cond=False)
    for i in range(Q):
        oi, osc = O.canon(ids[i], O.c_mix(x[i], O.c_gate(q[i], w2, query_cond=False)))
        assert np.array_equal(r2["ids"][i], oi[:10]) and np.array_equal(r2["scores"][i].view(np.uint32), osc[:10].view(np.uint32))
    # golden LinearWeights.forward (weighting.py:17-29), eval shape
    z = np.load(os.path.join(golden_dir, "linear_weights.npz"))
    x2, q1, Wg = z["x2"], z["q1"], z["W"]
    rg = idxmod.mix_topk(x2[None], np.arange(x2.shape[0], dtype=np.int64)[None], q1, Wg, None, None, k=x2.shape[0])
    got = np.empty(x2.shape[0], np.float32)
    got[rg["ids"][0]] = rg["scores"][0]
    np.testing.assert_allclose(got, z["eval_out"][0], rtol=0, atol=TOL)


def test_selection_edge_cases_through_the_mixer(idxmod):
    """The selection every top-k kernel shares (mfar_device.h block_topk_regs: common-prefix skip, 256-bin histogram levels, the k-th
    key's bin finished by one wave, id refinement on ties) on the key sets that stress each branch, driven through `mfar_mix_topk` with
    one field (its mixed score is weight x candidate score, so the keys are what the test writes): all scores equal; a block of more
    than 512 equal scores around the k-th rank; 4096 scores that differ only in their lowest mantissa bits; two tight clusters of
    opposite sign (the k-th key's bin stays above 512 keys for several levels); huge dynamic range with zeros and denormals; n = k + 1,
    n = k, n < k."""
    rng = np.random.default_rng(67)
    C = 4096
    rows = []
    rows.append(np.full(C, 1.25, np.float32))                                                        # all equal
    a = rng.standard_normal(C).astype(np.float32)
    a[rng.choice(C, 1000, replace=False)] = np.float32(np.sort(a)[-60])                              # 1000 copies of the 60th best
    rows.append(a)
    rows.append((np.float32(1.0) + np.arange(C, dtype=np.float32) * np.float32(2.0 ** -23))[rng.permutation(C)])   # low mantissa bits only
    b = np.concatenate([5.0 + rng.random(2000) * 2.0 ** -10, -3.0 + rng.random(2096) * 2.0 ** -10]).astype(np.float32)
    rows.append(b[rng.permutation(C)])                                                                # two tight clusters
    c = (rng.standard_normal(C) * np.exp(rng.uniform(-80, 80, C))).astype(np.float32)
    c[:50] = 0.0
    c[50:80] = np.float32(1e-42)                                                                      # denormals
    c[80:100] = np.float32(-1e-42)
    rows.append(c[rng.permutation(C)])
    rows.append(np.round(rng.standard_normal(C) * 3).astype(np.float32))                              # few distinct values: ties everywhere
    x = np.stack(rows)[:, :, None].astype(np.float32)
    Q = x.shape[0]
    ids = np.stack([rng.permutation(100000)[:C] for _ in range(Q)]).astype(np.int64)
    w1 = np.array([1.0], dtype=np.float32)
    for k in (1, 37, 100, 128):
        for n in (C, 2049, 513, k + 1, k, max(1, k - 3)):
            ncand = np.full(Q, n, dtype=np.int32)
            r = idxmod.mix_topk(x, ids, None, w1, None, ncand, k=k, query_cond=False)
            for i in range(Q):
                mixed = O.c_mix(x[i, :n], O.c_gate(np.zeros(1, np.float32), w1, query_cond=False))
                oi, osc = O.canon(ids[i, :n], mixed)
                m = min(n, k)
                assert r["n_valid"][i] == m, (k, n, i)
                assert np.array_equal(r["ids"][i, :m], oi[:m]), (k, n, i)
                assert np.array_equal(r["scores"][i, :m].view(np.uint32), osc[:m].view(np.uint32)), (k, n, i)


def test_two_stage_bit_exact_vs_oracle(idxmod):
    rng = np.random.default_rng(5)
    for F, D, E, Q, mean, masked in [(1, 900, 32, 4, 0.3, []), (4, 1200, 32, 6, 0.3, [1]), (8, 700, 64, 5, -0.4, [2, 3]),
                                     (22, 400, 32, 3, 0.3, [5]), (5, 3000, 768, 66, 0.05, [])]:
        slab, q, W = _mk(rng, F, D, E, Q, mean=mean, dup=9)
        mask = np.ones(F, np.float32)
        mask[masked] = 0
        ix = _load(idxmod, slab)
        for sentinel in (True, False):
            r = ix.search(q, W, mask, sentinel=sentinel, return_fields=True)
            o = O.c_two_stage(slab, q, W, mask, sentinel=sentinel)
            assert np.array_equal(r["field_ids"], o["field_ids"])
            assert np.array_equal(r["field_scores"].view(np.uint32), o["field_scores"].view(np.uint32))
            assert np.array_equal(r["n_cand"], o["n_cand"]) and np.array_equal(r["n_valid"], o["n_valid"])
            assert np.array_equal(r["ids"], o["ids"]), (F, D, sentinel)
            assert np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32)), (F, D, sentinel)
        ix.close()


def test_two_stage_golden_reference(golden_dir, idxmod):
    """The unmodified reference trec_eval_step (contrastive.py:669-704) captured by tools/gen_golden.py."""
    z = np.load(os.path.join(golden_dir, "trec_eval_step.npz"))
    for n in sorted({k.split("__")[0] for k in z.files}):
        b = str(z[n + "__base"]) if n + "__base" in z.files else n
        slab, q, W, mask = z[b + "__slab"], z[b + "__q"], z[b + "__W"], z[n + "__mask"]
        ix = _load(idxmod, slab)
        r = ix.search(q, W, mask)
        assert (r["n_valid"] == 100).all()
        O.assert_topk_equivalent(r["ids"], r["scores"], z[n + "__ids"], z[n + "__scores"], tol=TOL, what=n)
        assert np.array_equal(r["ids"], z[n + "__ids"]), n
        ix.close()


def test_fewer_candidates_than_k(idxmod):
    rng = np.random.default_rng(6)
    slab = rng.standard_normal((1, 40, 32)).astype(np.float32) - 2.0
    q = rng.standard_normal((2, 32)).astype(np.float32)
    W = rng.standard_normal((32, 1)).astype(np.float32)
    ix = _load(idxmod, slab)
    r = ix.search(q, W, return_fields=True)
    o = O.c_two_stage(slab, q, W)
    assert np.array_equal(r["n_valid"], o["n_valid"]) and (r["n_valid"] < 100).all()
    assert np.array_equal(r["ids"], o["ids"])
    ix.close()


def test_sharded_search_equals_unsharded(idxmod):
    """row shards [D*g/S, D*(g+1)/S) (contrastive.py:470), per-shard payloads, one merge."""
    rng = np.random.default_rng(7)
    F, D, E, Q = 4, 2500, 32, 9
    slab, q, W = _mk(rng, F, D, E, Q, mean=0.2, dup=11)
    mask = np.array([1, 1, 0, 1], np.float32)
    for sentinel in (True, False):
        full = _load(idxmod, slab)
        ref = full.search(q, W, mask, sentinel=sentinel)
        o = O.c_two_stage(slab, q, W, mask, sentinel=sentinel)
        assert np.array_equal(ref["ids"], o["ids"])
        full.close()
        for S in (1, 2, 4, 8):
            bounds = [D * g // S for g in range(S + 1)]
            shards = [_load(idxmod, slab[:, bounds[g]:bounds[g + 1]], row_offset=bounds[g]) for g in range(S)]
            payloads = np.concatenate([sh.search_local(q, sentinel=sentinel) for sh in shards])
            r = idxmod.merge_payloads(payloads, S, q, W, mask, sentinel=sentinel)
            assert np.array_equal(r["ids"], ref["ids"]), (sentinel, S)
            assert np.array_equal(r["scores"].view(np.uint32), ref["scores"].view(np.uint32)), (sentinel, S)
            assert np.array_equal(r["n_valid"], ref["n_valid"])
            for sh in shards:
                sh.close()


def test_device_path_equals_host_path(idxmod):
    import torch
    rng = np.random.default_rng(8)
    slab, q, W = _mk(rng, 3, 1500, 64, 20)
    ix = idxmod.MultiFieldIndex(1500, 3, 64, device=0)
    dev = torch.device("cuda:0")
    for f in range(3):
        ix.write_rows(f, 0, torch.from_numpy(slab[f]).to(dev))
    host = ix.search(q, W, None, return_fields=True)
    d = ix.search(torch.from_numpy(q).to(dev), torch.from_numpy(W).to(dev), None, return_fields=True)
    torch.cuda.synchronize()
    for key in ("ids", "scores", "n_valid", "field_ids", "field_scores", "n_cand"):
        assert np.array_equal(d[key].cpu().numpy(), host[key]), key
    back = torch.empty(1500, 64, device=dev)
    ix.read_rows(1, 0, 1500, out=back)
    assert np.array_equal(back.cpu().numpy(), slab[1])
    ix.close()


def test_dense_flat_index_drop_in(golden_dir, idxmod):
    """The reference-shaped object: constructor, retrieve_batch -> [(key, score)], score_batch -> tensor, KeyError."""
    import torch
    z = np.load(os.path.join(golden_dir, "retrieve_batch.npz"))
    V, q, gi, gs = z["g1_basic__V"], z["g1_basic__q"], z["g1_basic__ids"], z["g1_basic__scores"]
    keys = [f"doc{i}" for i in range(V.shape[0])]
    ix = idxmod.DenseFlatIndex(None, V, keys, {k: i for i, k in enumerate(keys)})
    res = ix.retrieve_batch(q, top_k=100)
    assert len(res) == q.shape[0] and len(res[0]) == 100 and isinstance(res[0][0][0], str) and isinstance(res[0][0][1], float)
    assert [k for k, _ in res[0]] == [keys[j] for j in gi[0]]
    np.testing.assert_allclose([s for _, s in res[0]], gs[0], rtol=0, atol=TOL)

    class Enc:
        def encode(self, texts, convert_to_tensor=True, **kw):
            return torch.from_numpy(np.stack([q[int(t[1:])] for t in texts]))

    ix.model = Enc()
    s = ix.score_batch(["q2"], [keys[5], keys[77]])
    assert isinstance(s, torch.Tensor) and tuple(s.shape) == (1, 2)
    np.testing.assert_allclose(s.numpy()[0], V[[5, 77]] @ q[2], rtol=0, atol=TOL)
    with pytest.raises(KeyError):
        ix.score_batch(["q0"], ["nope"])
    # .vectors re-assignment after a "re-encode" (contrastive.py:494)
    V2 = V[::-1].copy()
    ix.vectors = V2
    res2 = ix.retrieve_batch(q[:1], top_k=100)
    assert [k for k, _ in res2[0]] == [keys[V.shape[0] - 1 - j] for j in gi[0]]


def test_full_size_properties(idxmod):
    """BASELINE.json's headline shape (1M docs x 8 fields x 768) is checked through size-independent properties:
    planted documents are found, lists are sorted, results are reproducible and invariant under re-sharding."""
    import torch
    if torch.cuda.mem_get_info(0)[0] < 90 << 30:
        pytest.skip("needs ~80 GB of free HBM")
    from mfar import synth
    D, F, E, Q = 1_000_000, 8, 768, 64
    corpus = synth.SyntheticCorpus(D, F, E, n_queries=2 * Q, seed=0xdeadbeef, device="cuda:0")
    ix = corpus.build_index(idxmod)
    q, W = corpus.queries(0, Q), corpus.W
    r1 = ix.search(q, W, None, return_fields=True)
    r2 = ix.search(q, W, None)
    torch.cuda.synchronize()
    assert torch.equal(r1["ids"], r2["ids"]) and torch.equal(r1["scores"], r2["scores"])          # idempotent
    sc = r1["scores"].cpu().numpy()
    assert (np.diff(sc, axis=1) <= 0).all() and (r1["n_valid"].cpu().numpy() == 100).all()          # sorted
    fs = r1["field_scores"].cpu().numpy()
    assert (np.diff(fs, axis=2) <= 0).all()
    rel = corpus.qrels(0, Q)
    hit = np.mean([len(set(r1["ids"][i, :20].tolist()) & rel[i]) / len(rel[i]) for i in range(Q)])
    assert hit > 0.9, hit                                                                          # planted docs found
    # the fp16 screen is on by default at this size: every list certified, and switching it off changes no bit
    st = ix.screen_stats()
    assert st["built"] and st["n_checked"] == 2 * Q * F and st["n_failed"] == 0, st
    ix.set_screen(0)
    r0 = ix.search(q, W, None, return_fields=True)
    torch.cuda.synchronize()
    for key in ("ids", "scores", "field_ids", "field_scores", "n_cand"):
        assert torch.equal(r0[key], r1[key]), key
    ix.set_screen(1)
    # re-sharding invariance at full size: two half-shards + merge == one shard
    half = D // 2
    shards = [corpus.build_index(idxmod, row0=0, n=half), corpus.build_index(idxmod, row0=half, n=D - half)]
    ix.close()
    payloads = torch.cat([s.search_local(q) for s in shards])
    rm = idxmod.merge_payloads(payloads, 2, q, W, None, n_fields=F)
    torch.cuda.synchronize()
    assert torch.equal(rm["ids"], r1["ids"]) and torch.equal(rm["scores"], r1["scores"])
    for s in shards:
        s.close()
    # oracle comparison AT the headline shape, 8 queries: (1) exhaustive torch check that the stage-1 lists are complete,
    # (2) the C oracle on the union rows: stage-1 lists, final ids and final score BITS
    ix = corpus.build_index(idxmod)
    queries = [0, 7, 13, 21, 34, 42, 55, 63]
    fid, fsc = r1["field_ids"].cpu().numpy(), r1["field_scores"].cpu().numpy()
    _exhaustive_stage1_check(ix, q, fid, fsc, queries)
    _oracle_check_on_subset(ix, q, W, None, r1, queries)
    # ... and the path bench.py times at this shape: the wide 128-column scan (mfar_stage1_f16w4_kernel) through the pipeline
    ps = _timed_path_check(idxmod, ix, corpus, W, None, queries)
    assert not ix.stage2_dump_info()["wanted"]                 # (1 M x 8: the automatic policy keeps stage 2 on the gather slab)
    del ps
    ix.close()


def test_pipelined_searcher_equals_plain_search(idxmod):
    """Two batches in flight on two streams (stage 1 of batch i+1 beside the tail of batch i) must not change a bit."""
    import torch
    from mfar.data.pipeline import PipelinedSearcher
    rng = np.random.default_rng(9)
    F, D, E, Q = 4, 6000, 64, 32
    slab, _, W = _mk(rng, F, D, E, 1, mean=0.2, dup=5)
    ix = _load(idxmod, slab)
    dev = torch.device("cuda:0")
    Wd = torch.from_numpy(W).to(dev)
    mask = torch.tensor([1, 1, 0, 1], dtype=torch.float32, device=dev)
    qs = [(rng.standard_normal((Q, E)) * 0.5 + 0.3).astype(np.float32) for _ in range(7)]
    ps = PipelinedSearcher(ix, Wd, mask, max_batch=Q)
    tickets, got = [], []
    for i, q in enumerate(qs):
        tickets.append(ps.submit(torch.from_numpy(q).to(dev)))
        if i >= 1:
            r = ps.result(tickets[i - 1])
            got.append({k: v.clone() for k, v in r.items()})
    r = ps.result(tickets[-1])
    got.append({k: v.clone() for k, v in r.items()})
    torch.cuda.synchronize()
    for q, g in zip(qs, got):
        ref = ix.search(q, W, mask.cpu().numpy())
        assert np.array_equal(g["ids"].cpu().numpy(), ref["ids"])
        assert np.array_equal(g["scores"].cpu().numpy().view(np.uint32), ref["scores"].view(np.uint32))
        assert np.array_equal(g["n_valid"].cpu().numpy(), ref["n_valid"])
    with pytest.raises(ValueError):
        ps.result(tickets[0])
    ix.close()


@pytest.mark.parametrize("name,D,F,n_shards", [("stark-prime", 129_375, 22, 1), ("stark-mag", 700_244, 5, 1),
                                               ("stark-amazon-8shards", 957_192, 8, 8)])
def test_baseline_config_shapes(idxmod, name, D, F, n_shards):
    """BASELINE.json configs[1..3] at their full shapes (synthetic vectors): size-independent properties, an oracle spot
    check on a row subset that provably contains every stage-1 winner, and (config 3) the 8-way row-sharded path."""
    import torch
    from mfar import synth
    E, Q = 768, 64
    need = D * F * E * 4 * (2 if n_shards > 1 else 1) + (8 << 30)
    if torch.cuda.mem_get_info(0)[0] < need:
        pytest.skip("not enough free HBM")
    corpus = synth.SyntheticCorpus(D, F, E, n_queries=2 * Q, seed=0xdeadbeef, device="cuda:0")
    q, W = corpus.queries(0, Q), corpus.W
    mask = torch.ones(F, device="cuda:0")
    mask[F // 2] = 0
    ix = corpus.build_index(idxmod)
    r = ix.search(q, W, mask, return_fields=True)
    torch.cuda.synchronize()
    ids, sc, fid, fsc = (r[k].cpu().numpy() for k in ("ids", "scores", "field_ids", "field_scores"))
    assert (r["n_valid"].cpu().numpy() == 100).all()
    assert (np.diff(sc, axis=1) <= 0).all() and (np.diff(fsc, axis=2) <= 0).all()
    assert all(len(set(row.tolist())) == 100 for row in ids)
    rel = corpus.qrels(0, Q)
    assert np.mean([len(set(ids[i, :20].tolist()) & rel[i]) / len(rel[i]) for i in range(Q)]) > 0.9
    # oracle comparison at the full shape for 8 queries: the stage-1 lists are proven complete by an exhaustive torch scan,
    # then the C oracle on the union rows must give the same stage-1 lists, final ids and final score BITS (mask included)
    queries = [0, 9, 18, 27, 36, 45, 54, 63]
    _exhaustive_stage1_check(ix, q, fid, fsc, queries)
    _oracle_check_on_subset(ix, q, W, mask, r, queries)
    # the path bench.py times at this shape (its `baseline_configs` legs): default pipeline, wide scan, automatic dump policy
    d0 = ix.stage2_dump_info()
    ps = _timed_path_check(idxmod, ix, corpus, W, mask, queries)
    d1 = ix.stage2_dump_info()
    if name == "stark-prime":      # 22 fields x 129 375 rows: stage 2's approximate level reads the scan's score dump (mfar_s2_lookup_kernel)
        assert d1["wanted"] and d1["n_launches"] > d0["n_launches"], (d0, d1)
    del ps
    if n_shards > 1:
        ix.close()
        bounds = [D * g // n_shards for g in range(n_shards + 1)]
        shards = [corpus.build_index(idxmod, row0=bounds[g], n=bounds[g + 1] - bounds[g]) for g in range(n_shards)]
        payloads = torch.cat([s.search_local(q) for s in shards])
        rm = idxmod.merge_payloads(payloads, n_shards, q, W, mask, n_fields=F)
        torch.cuda.synchronize()
        assert np.array_equal(rm["ids"].cpu().numpy(), ids) and np.array_equal(rm["scores"].cpu().numpy().view(np.uint32), sc.view(np.uint32))
        # the exchange path of the row-sharded run, as PipelinedSearcher issues it per launch: 128 queries through every shard's WIDE scan
        # (mfar_retrieve_lists), all lists "gathered", every shard scores the candidates it owns (mfar_search_owned), top-k payloads merged
        q2 = corpus.queries(0, 2 * Q)
        nl, nt = shards[0].lists_bytes(2 * Q), shards[0].topk_bytes(2 * Q)
        lists_all = torch.empty(n_shards * nl, dtype=torch.uint8, device="cuda:0")
        for g, sh in enumerate(shards):
            assert sh.max_split_batch(100) == 128
            sh.retrieve_lists(q2, lists_all[g * nl:(g + 1) * nl], 100, True)
        topk_all = torch.empty(n_shards * nt, dtype=torch.uint8, device="cuda:0")
        for g, sh in enumerate(shards):
            sh.search_owned(lists_all, n_shards, q2, W, topk_all[g * nt:(g + 1) * nt], mask)
        rx = idxmod.merge_topk(topk_all, n_shards, 2 * Q)
        torch.cuda.synchronize()
        assert np.array_equal(rx["ids"][:Q].cpu().numpy(), ids), "lists-first exchange over 8 row shards (first half == the unsharded search)"
        assert np.array_equal(rx["scores"][:Q].cpu().numpy().view(np.uint32), sc.view(np.uint32))
        for s in shards:
            s.close()
    else:
        ix.close()


# ------------------------------------------------------------------------------------------------ bf16 slab
def _load_bf16(idxmod, slab, row_offset=0):
    F, D, E = slab.shape
    ix = idxmod.MultiFieldIndex(D, F, E, device=0, row_offset=row_offset, dtype="bf16")
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    return ix


def test_bf16_slab_rows_are_rne_rounded(idxmod):
    rng = np.random.default_rng(20)
    slab = (rng.standard_normal((2, 300, 64)) * 3).astype(np.float32)
    ix = _load_bf16(idxmod, slab)
    assert ix.slab_bytes < 2 * 320 * 64 * 4      # half of an fp32 slab (rows padded to 256)
    for f in range(2):
        assert np.array_equal(ix.read_rows(f).view(np.uint32), O.bf16_round(slab[f]).view(np.uint32))
    ix.close()


def test_bf16_two_stage_vs_oracle(idxmod):
    """bf16 slab (BASELINE.json configs[4]): docs are stored RNE-rounded; stage 1 runs v_mfma_f32_32x32x16_bf16 against
    the query split exactly into three bf16 terms, so its scores equal the natural-order fp32 chain over the rounded docs
    up to fp32 summation order (<= 1e-4, tolerant id check); stage 2 and the mixer walk that chain exactly -> final
    scores are BIT-IDENTICAL to the oracle whenever the candidate sets agree."""
    rng = np.random.default_rng(21)
    exact_final = 0
    cases = [(1, 700, 32, 5, 0.3), (4, 3000, 64, 9, 0.3), (3, 2500, 768, 66, 0.05), (8, 900, 96, 7, -0.4)]
    for F, D, E, Q, mean in cases:
        slab, q, W = _mk(rng, F, D, E, Q, mean=mean, dup=5)
        mask = np.ones(F, np.float32)
        if F > 2:
            mask[1] = 0
        ix = _load_bf16(idxmod, slab)
        rs = O.bf16_round(slab)
        for sentinel in (True, False):
            r = ix.search(q, W, mask, sentinel=sentinel, return_fields=True)
            with O.chain("natural"):
                o = O.c_two_stage(rs, q, W, mask, sentinel=sentinel)
            for f in range(F):
                O.assert_topk_equivalent(r["field_ids"][:, f], r["field_scores"][:, f], o["field_ids"][:, f], o["field_scores"][:, f],
                                         tol=TOL, what=f"bf16 stage1 F{F} D{D} f{f}")
            O.assert_topk_equivalent(r["ids"], r["scores"], o["ids"], o["scores"], tol=TOL, what=f"bf16 final F{F} D{D}")
            if np.array_equal(r["field_ids"], o["field_ids"]):
                assert np.array_equal(r["ids"], o["ids"])
                assert np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
                exact_final += 1
        # stage 2 alone: bit-exact
        cand = rng.integers(0, D, size=(Q, 40)).astype(np.int64)
        with O.chain("natural"):
            ox = O.c_score_candidates(rs, q, cand)
        assert np.array_equal(ix.score_candidates(q, cand).view(np.uint32), ox.view(np.uint32))
        ix.close()
    assert exact_final >= 6      # near-ties at a list boundary are rare on this data


def test_bf16_sharded_equals_unsharded(idxmod):
    rng = np.random.default_rng(22)
    F, D, E, Q = 4, 2600, 64, 9
    slab, q, W = _mk(rng, F, D, E, Q, mean=0.2, dup=7)
    full = _load_bf16(idxmod, slab)
    ref = full.search(q, W, None)
    full.close()
    for S in (2, 8):
        bounds = [D * g // S for g in range(S + 1)]
        shards = [_load_bf16(idxmod, slab[:, bounds[g]:bounds[g + 1]], row_offset=bounds[g]) for g in range(S)]
        payloads = np.concatenate([sh.search_local(q) for sh in shards])
        r = idxmod.merge_payloads(payloads, S, q, W, None)
        assert np.array_equal(r["ids"], ref["ids"]) and np.array_equal(r["scores"].view(np.uint32), ref["scores"].view(np.uint32))
        for sh in shards:
            sh.close()


def test_bf16_stress_shape_per_gpu(idxmod):
    """BASELINE.json configs[4] (10M docs x 16 fields x 768d bf16 over 8 GPUs) at its PER-GPU shape: 1.25M x 16 x 768 bf16
    = 30.7 GB (+ the row-major companion for gathers), on the DEFAULT path: the certified pass over the bf16 slab itself (two bf16
    query terms; 64 columns here, 128 for larger blocks).  Size-independent properties, an EXHAUSTIVE torch check of the stage-1
    lists of three queries, and the oracle (natural-order chain over the bf16 rows, the bf16 contract) BIT FOR BIT -- stage-1 lists
    and the final top-100 -- on a row subset that holds every list member; a block of 128 queries through the wide pass returns the
    same bits; the plain bf16 MFMA pass (screen off) agrees to its 1e-4; re-sharding invariance."""
    import torch
    from mfar import synth
    D, F, E, Q = 1_250_000, 16, 768, 64
    if torch.cuda.mem_get_info(0)[0] < 130 << 30:
        pytest.skip("needs ~110 GB of free HBM")
    corpus = synth.SyntheticCorpus(D, F, E, n_queries=2 * Q, seed=0xdeadbeef, device="cuda:0")
    ix = corpus.build_index(idxmod, dtype="bf16")
    q, W = corpus.queries(0, Q), corpus.W
    r1 = ix.search(q, W, None, return_fields=True)
    r2 = ix.search(q, W, None)
    torch.cuda.synchronize()
    st = ix.screen_stats()
    assert st["built"] and st["n_checked"] >= 2 * Q * F and st["n_failed"] == 0 and st["screen_bytes"] == 0, st
    assert ix.stage2_stats()["gather_slab_bytes"] <= ix.slab_bytes * 1.01      # resident rows <= 2.0 x the slab (companion only)
    assert torch.equal(r1["ids"], r2["ids"]) and torch.equal(r1["scores"], r2["scores"])
    sc = r1["scores"].cpu().numpy()
    assert (np.diff(sc, axis=1) <= 0).all() and (r1["n_valid"].cpu().numpy() == 100).all()
    rel = corpus.qrels(0, Q)
    assert np.mean([len(set(r1["ids"][i, :20].tolist()) & rel[i]) / len(rel[i]) for i in range(Q)]) > 0.9
    probe = [0, 17, 63]
    _exhaustive_stage1_check(ix, q, r1["field_ids"].cpu().numpy(), r1["field_scores"].cpu().numpy(), probe)
    with O.chain("natural"):
        _oracle_check_on_subset(ix, q, W, None, r1, probe)
    # 128 queries in one call: the wide pass (128 columns per scan); the first 64 are the queries above -- same bits
    rw = ix.search(corpus.queries(0, 2 * Q), W, None, return_fields=True)
    torch.cuda.synchronize()
    for key in ("ids", "scores", "field_ids", "field_scores"):
        assert torch.equal(rw[key][:Q], r1[key]), key
    assert ix.screen_stats()["n_failed"] == 0
    # the path bench.py times at this shape (mfar_stage1_bf16c_kernel through the default pipeline): halves == the 64-column search, oracle bits
    ps = _timed_path_check(idxmod, ix, corpus, W, None, probe, oracle_chain=O.chain("natural"))
    del ps
    # plain pass vs certified pass: the plain MFMA pass's stage-1 scores agree with the chain to 1e-4, so a list may swap a member
    # at a near-tied cut-off and with it one final candidate; every document BOTH runs return carries the same exact score bits
    ix.set_screen(0)
    r3 = ix.search(q, W, None, return_fields=True)
    torch.cuda.synchronize()
    _exhaustive_stage1_check(ix, q, r3["field_ids"].cpu().numpy(), r3["field_scores"].cpu().numpy(), probe)
    i1, s1, i3, s3 = (r[k].cpu().numpy() for r in (r1, r3) for k in ("ids", "scores"))
    for qi in range(Q):
        common, a, b = np.intersect1d(i1[qi], i3[qi], return_indices=True)
        assert common.size >= 98, (qi, common.size)
        assert np.array_equal(s1[qi][a].view(np.uint32), s3[qi][b].view(np.uint32)), qi
    ix.close()
    half = D // 2
    shards = [corpus.build_index(idxmod, row0=0, n=half, dtype="bf16"), corpus.build_index(idxmod, row0=half, n=D - half, dtype="bf16")]
    rm = idxmod.merge_payloads(torch.cat([s.search_local(q) for s in shards]), 2, q, W, None, n_fields=F)
    torch.cuda.synchronize()
    assert torch.equal(rm["ids"], r1["ids"]) and torch.equal(rm["scores"], r1["scores"])
    for s in shards:
        s.close()


def test_empty_and_degenerate_inputs(idxmod):
    """Empty shard, empty query batch, a single row, k = 1: the edge cases of index.py:181-222 / contrastive.py:669-704."""
    rng = np.random.default_rng(30)
    E = 32
    q = rng.standard_normal((3, E)).astype(np.float32)
    W = rng.standard_normal((E, 2)).astype(np.float32)
    ix = idxmod.MultiFieldIndex(0, 2, E)                      # no rows at all
    ids, sc = ix.retrieve_fields(q, 100, sentinel=True)
    assert (ids == 0).all() and (sc == 0).all()               # lists are pure sentinel padding
    ids, sc = ix.retrieve_fields(q, 100, sentinel=False)
    assert (ids == -1).all() and np.isneginf(sc).all()
    r = ix.search(q, W, None, sentinel=False)
    assert (r["n_valid"] == 0).all() and (r["ids"] == -1).all()
    r = ix.search(q[:0], W, None)                             # empty batch
    assert r["ids"].shape == (0, 100)
    ix.close()
    one = np.abs(rng.standard_normal((2, 1, E))).astype(np.float32)
    ix = _load(idxmod, one)
    qq = np.abs(q)
    r = ix.search(qq, W, None, k1=1, k2=1, return_fields=True)
    o = O.c_two_stage(one, qq, W, None, k1=1, k2=1)
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
    assert (r["n_valid"] == 1).all() and (r["field_ids"] == 0).all()
    ix.close()


def test_lists_first_exchange_equals_unsharded(idxmod):
    """The two-collective multi-GPU path (mfar_retrieve_lists -> gather -> mfar_search_owned -> gather -> mfar_merge_topk)
    with 1/2/4/8 in-process shards on one GPU: bitwise the unsharded search, fp32 and bf16 slabs, both sentinel modes."""
    import torch
    rng = np.random.default_rng(40)
    F, D, E, Q = 4, 2700, 64, 11
    slab, q, W = _mk(rng, F, D, E, Q, mean=0.2, dup=9)
    mask = np.array([1, 0, 1, 1], np.float32)
    dev = torch.device("cuda:0")
    qd, Wd, md = (torch.from_numpy(a).to(dev) for a in (q, W, mask))
    for dtype, loader in (("f32", _load), ("bf16", _load_bf16)):
        for sentinel in (True, False):
            full = loader(idxmod, slab)
            ref = full.search(q, W, mask, sentinel=sentinel)
            full.close()
            for S in (1, 2, 4, 8):
                bounds = [D * g // S for g in range(S + 1)]
                shards = [loader(idxmod, slab[:, bounds[g]:bounds[g + 1]], row_offset=bounds[g]) for g in range(S)]
                nl, nt = shards[0].lists_bytes(Q), shards[0].topk_bytes(Q)
                lists_all = torch.empty(S * nl, dtype=torch.uint8, device=dev)
                for g, sh in enumerate(shards):
                    sh.retrieve_lists(qd, lists_all[g * nl:(g + 1) * nl], 100, sentinel)
                topk_all = torch.empty(S * nt, dtype=torch.uint8, device=dev)
                for g, sh in enumerate(shards):
                    sh.search_owned(lists_all, S, qd, Wd, topk_all[g * nt:(g + 1) * nt], md, sentinel=sentinel)
                r = idxmod.merge_topk(topk_all, S, Q)
                torch.cuda.synchronize()
                assert np.array_equal(r["ids"].cpu().numpy(), ref["ids"]), (dtype, sentinel, S)
                assert np.array_equal(r["scores"].cpu().numpy().view(np.uint32), ref["scores"].view(np.uint32)), (dtype, sentinel, S)
                assert np.array_equal(r["n_valid"].cpu().numpy(), ref["n_valid"]), (dtype, sentinel, S)
                for sh in shards:
                    sh.close()


# ------------------------------------------------------------------------------------------------ certified fp16 screen
def _check_stage1(ix, slab, q, k, sentinel, tag):
    ids, sc = ix.retrieve_fields(q, k, sentinel)
    for f in range(slab.shape[0]):
        oi, osc = O.c_retrieve(slab[f], q, k, sentinel)
        assert np.array_equal(ids[:, f], oi), (tag, f)
        assert np.array_equal(sc[:, f].view(np.uint32), osc.view(np.uint32)), (tag, f)


def test_screen_bit_exact_vs_oracle(idxmod):
    """Stage 1 through the fp16 screen (csrc/mfar_screen.h) is bit-identical to the oracle's exhaustive fp32 result:
    ragged shapes, both sentinel modes, mostly-negative scores (the zero sentinel cuts the lists short), exact ties."""
    rng = np.random.default_rng(11)
    cases = [(1, 300, 32, 3, 100), (2, 5000, 64, 65, 100), (3, 20000, 96, 7, 100), (8, 4097, 768, 9, 100), (2, 9000, 32, 130, 128),
             (4, 2500, 256, 64, 1), (1, 150, 32, 5, 100)]
    for F, D, E, Q, k in cases:
        for sentinel in (True, False):
            for mean, dup in ((0.3, 0), (-0.4, 7), (-0.05, 0)):
                slab, q, _ = _mk(rng, F, D, E, Q, mean=mean, dup=dup)
                ix = _load(idxmod, slab)
                ix.set_screen(2)
                _check_stage1(ix, slab, q, k, sentinel, (F, D, E, Q, k, sentinel, mean, dup))
                st = ix.screen_stats()
                assert st["built"] and st["n_checked"] == Q * F and st["screen_bytes"] > 0
                if dup == 0 and D >= 2000:
                    assert st["n_failed"] == 0, st        # random data: every list is certified by the screen alone
                ix.close()


def test_screen_fallback_and_slack(idxmod):
    """eps_mult = 1e9 makes every certificate fail: all fields go through the exact fall-back pass and the result must
    not change.  eps_mult = 1/16 still certifies everything on random data: the real error is far below the bound."""
    rng = np.random.default_rng(12)
    F, D, E, Q, k = 3, 30000, 128, 70, 100
    slab, q, W = _mk(rng, F, D, E, Q)
    ix = _load(idxmod, slab)
    ix.set_screen(2, 1e9)
    _check_stage1(ix, slab, q, k, True, "forced fall-back")
    st = ix.screen_stats()
    assert st["n_failed"] == st["n_checked"] == Q * F, st
    ix.set_screen(2, 1.0 / 16)
    _check_stage1(ix, slab, q, k, True, "eps/16")
    st2 = ix.screen_stats()
    assert st2["n_failed"] == st["n_failed"], (st, st2)
    # the whole scorer on top of the screened stage 1
    ix.set_screen(2, 1.0)
    r = ix.search(q, W, None, return_fields=True)
    o = O.c_two_stage(slab, q, W, None)
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
    ix.set_screen(0)
    r0 = ix.search(q, W, None, return_fields=True)
    for key in ("ids", "scores", "field_ids", "field_scores", "n_cand"):
        assert np.array_equal(np.asarray(r[key]).view(np.uint8), np.asarray(r0[key]).view(np.uint8)), key
    ix.close()


def test_screen_massive_ties_and_huge_values(idxmod):
    """Thousands of NEAR-identical rows at the top of every list (the same vector up to the last bits, as batched encoder
    forwards of one text produce) are no duplicate group, and their approximate scores cannot separate the k-th from the
    k'-th entry: the certificate must fail, the exact pass must take over, and the canonical tie-break must hold.
    Bit-identical rows, in contrast, are scanned once and expanded (test_screen_scans_unique_rows_and_expands_groups).
    A row with huge finite values overflows the norm statistics."""
    rng = np.random.default_rng(13)
    F, D, E, Q, k = 2, 20000, 64, 9, 100
    slab, q, _ = _mk(rng, F, D, E, Q)
    top = q.mean(0) * 3.0
    rows = rng.choice(D, size=3000, replace=False)
    # 3000 DISTINCT vectors (bit-identical rows would simply be scanned once), equal up to the last bits
    slab[0, rows] = top * (1.0 + np.arange(3000, dtype=np.float32)[:, None] * np.float32(2.0 ** -22))
    assert len(np.unique(slab[0, rows], axis=0)) == 3000
    ix = _load(idxmod, slab)
    ix.set_screen(2)
    _check_stage1(ix, slab, q, k, True, "near ties")
    st = ix.screen_stats()
    assert st["n_failed"] >= Q, st                         # field 0 fell back for every query
    slab[1, 17] *= np.float32(1e25)                        # field 1: row norm^2 overflows fp32
    ix.write_rows(1, 17, slab[1, 17:18])
    _check_stage1(ix, slab, q, k, False, "huge")
    ix.close()


def test_screen_follows_row_updates(idxmod):
    rng = np.random.default_rng(14)
    F, D, E, Q, k = 2, 18000, 96, 5, 100
    slab, q, _ = _mk(rng, F, D, E, Q)
    ix = _load(idxmod, slab)                               # auto mode: >= 16384 rows
    _check_stage1(ix, slab, q, k, True, "before")
    assert ix.screen_stats()["built"]
    new = (q[:3] * 5.0).astype(np.float32)                 # three rows that become everybody's best match
    slab[1, 100:103] = new
    ix.write_rows(1, 100, new)
    assert not ix.screen_stats()["built"]                  # stale until the next search rebuilds it
    _check_stage1(ix, slab, q, k, True, "after")
    assert ix.screen_stats()["built"]
    ix.close()


def test_stage2_gathers_group_representatives(idxmod):
    """Stage 2 reads a row's group representative instead of the row (bit-identical by construction, csrc/mfar_screen.h
    mfar_rep_of_kernel): same bits as the oracle on a corpus full of duplicate groups; the map is ignored as soon as a row
    is rewritten (a former duplicate that now differs must be scored from its own bytes) and rebuilt by the next search."""
    rng = np.random.default_rng(16)
    F, D, E, Q = 3, 20000, 96, 40
    slab, q, W = _mk(rng, F, D, E, Q)
    slab[0, 1000:] = slab[0, rng.integers(0, 1000, D - 1000)]          # field 0: 1000 distinct vectors over 20000 rows
    slab[1, ::2] = slab[1, 0]                                          # field 1: every other row is one vector
    ix = _load(idxmod, slab)
    o = O.c_two_stage(slab, q, W, None)
    r = ix.search(q, W, None, return_fields=True)
    assert ix.screen_stats()["built"]
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
    cand = np.tile(np.arange(0, 64, dtype=np.int64), (Q, 1))           # rows 0, 2, 4 .. of field 1 are duplicates of row 0
    x0 = np.asarray(ix.score_candidates(q, cand))
    ref = O.c_score_candidates(slab, q, cand)
    assert np.array_equal(x0.view(np.uint32), ref.view(np.uint32))
    slab[1, 4] = q[0] * 3.0                                            # a former duplicate changes
    ix.write_rows(1, 4, slab[1, 4:5])
    x1 = np.asarray(ix.score_candidates(q, cand))                      # tables are stale now: every row from its own bytes
    ref = O.c_score_candidates(slab, q, cand)
    assert np.array_equal(x1.view(np.uint32), ref.view(np.uint32)) and not np.array_equal(x1, x0)
    o = O.c_two_stage(slab, q, W, None)
    r = ix.search(q, W, None)                                          # rebuilds the tables
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
    x2 = np.asarray(ix.score_candidates(q, cand))
    assert np.array_equal(x2.view(np.uint32), ref.view(np.uint32))
    ix.close()


def test_mask_sweep_through_the_pipeline(idxmod):
    """PipelinedSearcher(masks=[M, F]) / mfar_search_stage2_masks: stage 1, the candidate union and stage 2 once per launch, the
    mixer once per mask -- every mask's results equal a separate search with that mask, bit for bit (full and short launches,
    coalesced and single batches)."""
    import torch
    from mfar.data.pipeline import PipelinedSearcher
    rng = np.random.default_rng(19)
    dev = torch.device("cuda:0")
    for F, D, E, Q, qb in ((4, 30000, 96, 150, 64), (3, 5000, 64, 70, 16), (6, 20000, 128, 64, 64)):
        slab, q, W = _mk(rng, F, D, E, Q)
        ix = _load(idxmod, slab)
        masks = (rng.random((5, F)) < 0.6).astype(np.float32)
        masks[0] = 1.0
        masks[1] = 0.0
        want = [ix.search(q, W, masks[m]) for m in range(len(masks))]
        ps = PipelinedSearcher(ix, torch.from_numpy(W).to(dev), None, max_batch=qb, masks=torch.from_numpy(masks).to(dev))
        tickets = [(c, ps.submit(torch.from_numpy(q[c:c + qb]).to(dev))) for c in range(0, Q, qb)]
        got = {}
        for i, (c, t) in enumerate(tickets):
            if i >= ps.lag:
                c0, t0 = tickets[i - ps.lag]
                got[c0] = {k: v.cpu().numpy().copy() for k, v in ps.result(t0).items()}
        for c, t in tickets[max(0, len(tickets) - ps.lag):]:
            got[c] = {k: v.cpu().numpy().copy() for k, v in ps.result(t).items()}
        for c, g in got.items():
            n = min(qb, Q - c)
            assert g["ids"].shape == (len(masks), n, 100)
            for m in range(len(masks)):
                assert np.array_equal(g["ids"][m], np.asarray(want[m]["ids"])[c:c + n]), (F, D, E, Q, qb, c, m)
                assert np.array_equal(g["scores"][m].view(np.uint32), np.asarray(want[m]["scores"])[c:c + n].view(np.uint32)), (c, m)
        ix.close()


def test_fine_repair_mode(idxmod):
    """Failed certificates are repaired over the finely cut table (every field a whole wave of chunks, walked by one wave of
    workgroups: S1_CHUNK_LOOP in csrc/mfar_stage1.h); mfar_set_repair_mode(1) adds the repair's own sample pass -- same bits
    as the oracle when every list fails, when one field fails (near-ties in field 0 only) and when nothing fails, in both
    modes; a bf16 index repairs to the same lists in both modes."""
    rng = np.random.default_rng(18)
    for F, D, E, Q in ((3, 40000, 96, 70), (5, 20000, 64, 33), (2, 70000, 768, 128)):
        slab, q, W = _mk(rng, F, D, E, Q)
        rows = rng.choice(D, size=1500, replace=False)         # field 0: 1500 distinct vectors, equal up to the last bits
        slab[0, rows] = (q[0] * 4.0) * (1.0 + np.arange(1500, dtype=np.float32)[:, None] * np.float32(2.0 ** -22))
        ix = _load(idxmod, slab)
        o = O.c_two_stage(slab, q, W, None)
        for thorough, eps_mult in ((False, 1.0), (False, 1e9), (True, 1.0), (True, 1e9)):
            ix.set_repair_mode(thorough)
            ix.set_screen(2, eps_mult)
            r = ix.search(q, W, None, return_fields=True)
            assert np.array_equal(r["field_ids"], o["field_ids"]), (F, D, E, Q, thorough, eps_mult)
            assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
        st = ix.screen_stats()
        assert st["n_failed"] >= Q * F, st
        ix.close()
    slab, q, W = _mk(rng, 3, 30000, 128, 100)
    ix = _load_bf16(idxmod, slab)
    ix.set_screen(2, 1e9)
    r0 = ix.search(q, W, None, return_fields=True)
    ix.set_repair_mode(True)
    r1 = ix.search(q, W, None, return_fields=True)
    for key in ("ids", "scores", "field_ids", "field_scores"):
        assert np.array_equal(np.asarray(r0[key]).view(np.uint8), np.asarray(r1[key]).view(np.uint8)), key
    ix.close()


def test_pipelined_searcher_with_screen_and_redo(idxmod):
    """The split-phase pipeline over a screened index: certified batches flow through; with an impossible proof
    (eps_mult = 1e9) every batch reports a failed certificate and result() redoes it exactly -- same bits either way."""
    import torch
    from mfar.data.pipeline import PipelinedSearcher
    rng = np.random.default_rng(15)
    F, D, E, Q = 3, 20000, 96, 64
    slab, _, W = _mk(rng, F, D, E, 1, mean=0.2, dup=5)
    ix = _load(idxmod, slab)
    dev = torch.device("cuda:0")
    Wd = torch.from_numpy(W).to(dev)
    qs = [(rng.standard_normal((Q if i != 3 else 17, E)) * 0.5 + 0.3).astype(np.float32) for i in range(9)]
    ix.set_screen(0)
    want = [ix.search(q, W, None) for q in qs]
    for eps_mult, expect_redo in ((1.0, False), (1e9, True)):
        ix.set_screen(2, eps_mult)
        ps = PipelinedSearcher(ix, Wd, None, max_batch=Q)
        tickets, got = [], []
        for i, q in enumerate(qs):
            tickets.append(ps.submit(torch.from_numpy(q).to(dev)))
            if i >= 1:
                got.append({k: v.clone() for k, v in ps.result(tickets[i - 1]).items()})
        got.append({k: v.clone() for k, v in ps.result(tickets[-1]).items()})
        torch.cuda.synchronize()
        # every batch fails with the impossible proof: the first launches are redone by result(); once the library has seen four failed
        # launches (the redone ones count) it repairs on the device and reports clean launches (include/mfar_hip.h "inline repair")
        inline = ix.auto_off_info()["inline_repair"]
        assert (1 <= ps.n_redone <= len(qs) and inline) if expect_redo else (ps.n_redone == 0 and not inline), (ps.n_redone, inline)
        assert ix.screen_setting == (2, pytest.approx(eps_mult))
        for w, g in zip(want, got):
            assert np.array_equal(g["ids"].cpu().numpy(), w["ids"])
            assert np.array_equal(g["scores"].cpu().numpy().view(np.uint32), w["scores"].view(np.uint32))
            assert np.array_equal(g["n_valid"].cpu().numpy(), w["n_valid"])
    ix.close()


def test_screen_anisotropic_embeddings_are_certified(idxmod):
    """Real sentence embeddings share a large common component (every doc . query score is dominated by the same offset).
    The screen slab is centred on the field mean, so its error bound scales with the spread of the rows, not with their
    norm: even with the common part 10x the spread every list is certified (no exact fall-back), bit-exact as always."""
    rng = np.random.default_rng(16)
    for Q in (33, 128):                     # the 64-column pass (two query terms) and the wide pass (one term: eps about doubles)
        F, D, E, k = 2, 30000, 96, 100
        common = rng.standard_normal(E).astype(np.float32)
        common *= np.float32(10.0 * np.sqrt(E) * 0.05 / np.linalg.norm(common))          # |common| = 10 x the noise norm
        slab = (rng.standard_normal((F, D, E)) * 0.05 + common).astype(np.float32)
        q = (rng.standard_normal((Q, E)) * 0.05 + common * 0.5).astype(np.float32)
        ix = _load(idxmod, slab)
        ix.set_screen(2)
        for sentinel in (True, False):
            _check_stage1(ix, slab, q, k, sentinel, ("anisotropic", sentinel, Q))
        st = ix.screen_stats()
        assert st["n_checked"] == 2 * Q * F and st["n_failed"] == 0, st
        # negative common component: every score is far below zero, the zero sentinel empties the lists
        _check_stage1(ix, slab, -q, k, True, "all negative")
        ix.close()


def test_screen_scans_unique_rows_and_expands_groups(idxmod):
    """Real fields are full of bit-identical rows: "" for every document that lacks the field (format.py:58-59), a handful of
    texts in low-cardinality fields (schema.py:11-53), every repeated text encoded once by on_eval_start.  The screen scans
    each distinct vector once and the certify step expands a selected vector to its documents (same score, ids ascending).
      field 0: FOUR large groups (9000 / 4000 / 2500 / 700 rows) that are every query's best matches, in that order
      field 1: one ordinary group of 5000 rows + a Zipf-like tail of small groups
      field 2: ten distinct vectors in the whole field
      field 3: no duplicates
    No list may fall back to the exact pass, every bit must match the oracle -- also across shards (each shard groups its own
    rows) and through the single-field entry point."""
    rng = np.random.default_rng(17)
    F, D, E, Q, k = 4, 24000, 96, 40, 100
    slab, q, W = _mk(rng, F, D, E, Q)
    top = (q.mean(0) * 8.0).astype(np.float32)               # far above every ordinary row, for every query
    perm = rng.permutation(D)
    sizes = [9000, 4000, 2500, 700]
    lo = 0
    for gi, n in enumerate(sizes):
        slab[0, perm[lo:lo + n]] = top * np.float32(1.0 - 0.05 * gi)       # best, second best, ... of every query
        lo += n
    g1 = np.sort(rng.choice(D, size=5000, replace=False))
    slab[1, g1] = slab[1, g1[0]]
    rest = np.setdiff1d(np.arange(D), g1)
    for gsz in (300, 120, 64, 30, 30, 9, 2, 2):                             # a tail of smaller groups
        pick = rng.choice(rest, size=gsz, replace=False)
        slab[1, pick] = slab[1, pick[0]]
    texts = rng.integers(0, 10, size=D)
    slab[2] = slab[2, :10][texts]
    ix = _load(idxmod, slab)
    ix.set_screen(2)
    for sentinel in (True, False):
        _check_stage1(ix, slab, q, k, sentinel, ("unique rows", sentinel))
    nu = [ix.screen_field_info(f) for f in range(F)]
    assert nu[0] == (D - sum(sizes) + 4, 9000), nu
    assert nu[2] == (10, int(np.bincount(texts).max())) and nu[3] == (D, 1), nu
    assert nu[1][1] == 5000 and nu[1][0] == len(np.unique(slab[1], axis=0)), nu
    st = ix.screen_stats()
    assert st["n_failed"] == 0 and st["unique_rows"] == [n for n, _ in nu], st      # no exact fall-back despite the ties
    assert st["screen_bytes"] < D * F * E * 2                                       # the slab holds unique rows only
    r = ix.search(q, W, None, return_fields=True)
    o = O.c_two_stage(slab, q, W, None)
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
    # field 0's lists: the 100 lowest ids of the best group (one score), field 2's: the best text's documents
    assert all(len(set(r["field_scores"][i, 0].tolist())) == 1 for i in range(Q))
    assert np.array_equal(r["field_ids"][:, 0], np.broadcast_to(np.sort(perm[:9000])[:100], (Q, 100)))
    # the single-field entry point (what DenseFlatIndex.retrieve_batch calls) gives the same lists
    for f in range(F):
        ids, sc = ix.retrieve_field(f, q, k, True)
        assert np.array_equal(ids, o["field_ids"][:, f]) and np.array_equal(sc.view(np.uint32), o["field_scores"][:, f].view(np.uint32)), f
    assert ix.screen_stats()["n_failed"] == 0
    ix.set_screen(0)
    for f in (0, 2):
        ids, sc = ix.retrieve_field(f, q, k, True)
        assert np.array_equal(ids, o["field_ids"][:, f]) and np.array_equal(sc.view(np.uint32), o["field_scores"][:, f].view(np.uint32)), f
    ix.close()
    # sharded: the groups are split over the shards, each shard finds and expands its own members
    bounds = [0, 7001, 15000, D]
    shards = [_load(idxmod, slab[:, bounds[i]:bounds[i + 1]], row_offset=bounds[i]) for i in range(3)]
    import torch
    qd = torch.from_numpy(q).cuda()
    for s_ in shards:
        s_.set_screen(2)
    payloads = torch.cat([s_.search_local(qd) for s_ in shards])
    rm = idxmod.merge_payloads(payloads, 3, qd, torch.from_numpy(W).cuda(), None, n_fields=F)
    torch.cuda.synchronize()
    assert np.array_equal(rm["ids"].cpu().numpy(), o["ids"])
    assert np.array_equal(rm["scores"].cpu().numpy().view(np.uint32), o["scores"].view(np.uint32))
    assert all(s_.screen_stats()["n_failed"] == 0 for s_ in shards)
    for s_ in shards:
        s_.close()


def test_screen_tied_scores_of_distinct_vectors(idxmod):
    """Distinct vectors with EXACTLY equal scores (values on a coarse grid: all arithmetic exact) around the k-th place:
    the expansion must interleave their documents by id; when more tied vectors exist than the certify step looks at, the
    exact pass decides.  Bits equal the oracle either way."""
    rng = np.random.default_rng(18)
    F, D, E, Q, k = 1, 20000, 32, 6, 100
    slab = (rng.integers(-8, 9, size=(F, D, E)) / 8.0).astype(np.float32)
    q = (rng.integers(-8, 9, size=(Q, E)) / 8.0).astype(np.float32)
    # 40 distinct vectors that all score the same against query 0: swap two coordinates where q[0] has equal entries
    base = slab[0, 0].copy()
    eq = [(a, b) for a in range(E) for b in range(a + 1, E) if q[0, a] == q[0, b] and base[a] != base[b]]
    rows = rng.choice(np.arange(1, D), size=1200, replace=False)
    for j, r_ in enumerate(rows):
        v = base.copy()
        a, b = eq[j % min(len(eq), 40)]
        v[a], v[b] = v[b], v[a]
        slab[0, r_] = v * 4.0                                    # far above everything else for query 0
    ix = _load(idxmod, slab)
    ix.set_screen(2)
    for sentinel in (True, False):
        _check_stage1(ix, slab, q, k, sentinel, ("tied distinct vectors", sentinel))
    ix.close()


# ------------------------------------------------------------------------------------------------ larger goldens (SURVEY 8(c) sizes)
def test_stage1_large_and_tie_heavy_goldens(golden_dir, idxmod):
    """G1 at D = 5000 x E = 768 (captured through the reference's 256-row chunk merge), and the tie-heavy grid case whose
    scores must equal the reference's bit for bit -- on both stage-1 paths of an fp32 index."""
    from test_oracle_golden import check_tie_grid_case
    z = np.load(os.path.join(golden_dir, "retrieve_batch_large.npz"))
    n = "g1_5000x768_chunk256"
    V, q = z[n + "__V16"].astype(np.float32), z[n + "__q16"].astype(np.float32)
    ix = _load(idxmod, V[None])
    for screen in (0, 2):
        ix.set_screen(screen)
        ids, sc = ix.retrieve_fields(q, 100, True)
        assert np.array_equal(ids[:, 0], z[n + "__ids"]), screen
        np.testing.assert_allclose(sc[:, 0], z[n + "__scores"], rtol=0, atol=TOL)
    ix.close()
    V, q = z["g1_ties_grid__Vg"].astype(np.float32) / 8.0, z["g1_ties_grid__qg"].astype(np.float32) / 8.0
    ix = _load(idxmod, V[None])
    for screen in (0, 2):
        ix.set_screen(screen)
        ids, sc = ix.retrieve_fields(q, 100, True)
        check_tie_grid_case(z, ids[:, 0], sc[:, 0])
    ix.close()


@pytest.mark.parametrize("name", ["L_f1_e768", "L_f4_e768", "L_f8_e768"])
def test_two_stage_large_goldens(golden_dir, idxmod, name):
    """G5 at E = 768, F in {1, 4, 8}: ids equal to the unmodified reference trec_eval_step, scores within 1e-4."""
    z = np.load(os.path.join(golden_dir, f"trec_eval_step_{name}.npz"))
    slab, q, W, mask = (z[k].astype(np.float32) for k in ("slab16", "q16", "W16", "mask"))
    ix = _load(idxmod, slab)
    for screen in (0, 2):
        ix.set_screen(screen)
        r = ix.search(q, W, mask)
        assert np.array_equal(r["ids"], z["ids"]), (name, screen)
        np.testing.assert_allclose(r["scores"], z["scores"], rtol=0, atol=TOL)
    ix.close()


# ------------------------------------------------------------------------------------------------ on-disk format of the vectors
def test_memmap_export_import_roundtrip(idxmod, tmp_path):
    """HbmFieldVectors.export_memmap writes the reference's raw {temp_dir}/{field}.npy layout (float32 [D, E], no header,
    data/util.py:35): the reference's MemoryMapDict reads it, and import_memmap of a MemoryMapDict-written file restores
    the slab bit for bit -- also when two row shards write the same file."""
    from mfar.data.util import HbmFieldVectors, MemoryMapDict
    rng = np.random.default_rng(40)
    F, D, E = 2, 333, 64
    slab, q, W = _mk(rng, F, D, E, 3)
    keys = [f"doc{i}" for i in range(D)]
    half = 150
    shards = [_load(idxmod, slab[:, :half]), _load(idxmod, slab[:, half:], row_offset=half)]
    for f in range(F):
        path = str(tmp_path / f"field{f}.npy")
        for sh in shards:
            HbmFieldVectors(sh, f, keys).export_memmap(path, n_total_rows=D)
        assert os.path.getsize(path) == D * E * 4
        mm = MemoryMapDict(path, keys, (D, E))
        assert np.array_equal(mm.file, slab[f]) and np.array_equal(mm["doc7"], slab[f, 7])
        mm["doc7"] = np.arange(E, dtype=np.float32)                  # the reference's writer (data/util.py:40-41)
        mm.close()
    full = idxmod.MultiFieldIndex(D, F, E, device=0)
    for f in range(F):
        HbmFieldVectors(full, f, keys).import_memmap(str(tmp_path / f"field{f}.npy"), D)
    want = slab.copy()
    want[:, 7] = np.arange(E, dtype=np.float32)
    for f in range(F):
        assert np.array_equal(full.read_rows(f), want[f])
    o = O.c_two_stage(want, q, W, None)
    r = full.search(q, W, None)
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))
    for ix in shards + [full]:
        ix.close()


def test_score_batch_refuses_rows_of_other_shards(idxmod):
    """DenseFlatIndex.score_batch on a row shard: a valid key owned by another rank raises KeyError instead of scoring NaN."""
    rng = np.random.default_rng(41)
    slab, q, _ = _mk(rng, 1, 200, 32, 1)
    keys = [str(i) for i in range(400)]
    sh = _load(idxmod, slab, row_offset=200)
    ix = idxmod.DenseFlatIndex(None, None, keys, {k: i for i, k in enumerate(keys)}, slab=sh, field_index=0)

    class Enc:
        def encode(self, texts, convert_to_tensor=False, **kw):
            import torch
            return torch.from_numpy(q[:len(texts)])
    ix.model = Enc()
    s = ix.score_batch(["q"], ["200", "399"])
    assert tuple(s.shape) == (1, 2) and np.isfinite(s.numpy()).all()
    with pytest.raises(KeyError):
        ix.score_batch(["q"], ["200", "5"])        # row 5 lives on another rank
    with pytest.raises(KeyError):
        ix.score_batch(["q"], ["nope"])            # unknown key (index.py:229)
    sh.close()


def test_wide_screen_pass_and_coalescing(idxmod):
    """Blocks of 65 .. 128 queries take the WIDE screened pass (one fp16 query term, 128 columns per scan; mfar_stage1_f16w_kernel,
    6-slot and 4-slot twins): same bits as the oracle, as the 64-column pass (set_wide(False)) and as the forced exact
    fall-back; random data is certified without a single redo; the pipeline coalesces two 64-query batches into one launch
    and hands out exactly what the plain search returns."""
    import torch
    from mfar.data.pipeline import PipelinedSearcher
    rng = np.random.default_rng(41)
    dev = torch.device("cuda:0")
    for F, D, E, Q in ((3, 40000, 768, 128), (2, 30000, 128, 100), (4, 20000, 96, 192), (1, 17000, 64, 65)):
        slab, q, W = _mk(rng, F, D, E, Q)
        ix = _load(idxmod, slab)
        ix.set_screen(2)
        assert ix.max_split_batch(100) == 128
        o = O.c_two_stage(slab, q, W, None)
        r = ix.search(q, W, None, return_fields=True)
        assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32)), (F, D, E, Q)
        st = ix.screen_stats()
        assert st["n_checked"] == Q * F and st["n_failed"] == 0, st
        ix.set_wide(False)
        assert ix.max_split_batch(100) == 64
        r64 = ix.search(q, W, None, return_fields=True)
        ix.set_wide(True)
        ix.set_screen(2, 1e9)                      # every certificate fails: the exact pass repairs both 64-query halves
        rx = ix.search(q, W, None, return_fields=True)
        ix.set_screen(2, 1.0)
        for key in ("ids", "scores", "field_ids", "field_scores", "n_cand"):
            assert np.array_equal(np.asarray(r[key]).view(np.uint8), np.asarray(r64[key]).view(np.uint8)), (key, "wide vs 64")
            assert np.array_equal(np.asarray(r[key]).view(np.uint8), np.asarray(rx[key]).view(np.uint8)), (key, "wide vs fall-back")
        # the pipeline: batches of <= 64 queries, two per launch
        Wd = torch.from_numpy(W).to(dev)
        ps = PipelinedSearcher(ix, Wd, None, max_batch=64)
        assert ps.coalesce == 2 and ps.lag == 2 * ps.depth - 1
        cuts = list(range(0, Q, 64)) * 4            # more launches than slots, the last one possibly a single (flushed) batch
        tickets, got = [], []
        for i, c in enumerate(cuts):
            tickets.append(ps.submit(torch.from_numpy(q[c:c + 64]).to(dev)))
            if i >= ps.lag:
                got.append({k: v.cpu().numpy().copy() for k, v in ps.result(tickets[i - ps.lag]).items()})
        for t in tickets[max(0, len(cuts) - ps.lag):]:
            got.append({k: v.cpu().numpy().copy() for k, v in ps.result(t).items()})
        assert len(got) == len(cuts) and ps.n_redone == 0
        for c, g in zip(cuts, got):
            assert np.array_equal(g["ids"], o["ids"][c:c + 64]) and np.array_equal(g["scores"].view(np.uint32), o["scores"][c:c + 64].view(np.uint32))
        with pytest.raises(ValueError):
            ps.result(tickets[0])
        ix.close()


def test_auto_screen_stays_off_for_very_wide_rows(idxmod):
    """Auto mode (the default) does not screen rows wider than 2560 dims (csrc/mfar_hip.hip screen_wanted: the certificate's
    accumulation term eats the k..k' margin there and every failure costs an exact pass); mode 2 still does, same bits."""
    rng = np.random.default_rng(61)
    F, D, E, Q = 1, 16500, 2592, 70
    slab, q, W = _mk(rng, F, D, E, Q)
    ix = _load(idxmod, slab)
    r = ix.search(q, W, None, return_fields=True)
    assert not ix.screen_stats()["built"] and ix.max_split_batch(100) == 64
    ix.set_screen(2)
    r2 = ix.search(q, W, None, return_fields=True)
    assert ix.screen_stats()["built"] and ix.max_split_batch(100) == 128
    for key in ("ids", "scores", "field_ids", "field_scores", "n_cand"):
        assert np.array_equal(np.asarray(r[key]).view(np.uint8), np.asarray(r2[key]).view(np.uint8)), key
    ix.close()


def test_long_chunks_grow_the_sample(idxmod):
    """Many fields x long chunks (32 fields share the grid: 16 chunks of ~38 tiles per field): the sample pass grows beyond its
    2 tiles per chunk so that a chunk expects ~130 appends per query (build_table in csrc/mfar_hip.hip; here 3 tiles, the
    1.25 M x 16 stress shape takes 6).  Thresholds only prune: wide screened pass == 64-column screened pass == exact fp32 pass."""
    import torch
    from mfar import synth
    D, F, E, Q = 170_000, 32, 64, 128
    corpus = synth.SyntheticCorpus(D, F, E, n_queries=Q, seed=77, device="cuda:0")
    ix = corpus.build_index(idxmod)
    q, W = corpus.queries(0, Q), corpus.W
    ix.set_screen(2)
    r = ix.search(q, W, None, return_fields=True)
    st = ix.screen_stats()
    assert st["built"] and st["n_checked"] == Q * F and st["n_failed"] == 0, st
    ix.set_wide(False)
    r64 = ix.search(q, W, None, return_fields=True)
    ix.set_wide(True)
    ix.set_screen(0)
    r0 = ix.search(q, W, None, return_fields=True)
    torch.cuda.synchronize()
    for key in ("ids", "scores", "field_ids", "field_scores", "n_cand"):
        assert torch.equal(r[key], r64[key]), (key, "wide vs 64")
        assert torch.equal(r[key], r0[key]), (key, "screened vs exact")
    ix.close()


def test_fused_mode_bit_exact_and_recall(idxmod):
    """mfar_search_fused: the exhaustive top-k of the gate-folded inner product equals the oracle bit for bit (companion of 8
    interleaved row groups, incl. the padding row of an uneven split), follows row updates, and -- the claim it makes against
    the two-stage scorer -- finds at least the same share of planted relevant documents in its top 20."""
    rng = np.random.default_rng(51)
    for F, D, E, Q, screen in ((3, 2003, 64, 9, 0), (4, 20001, 96, 70, 2), (8, 30000, 64, 128, 2), (2, 5, 32, 5, 2)):
        slab, q, W = _mk(rng, F, D, E, Q)
        mask = np.ones(F, np.float32)
        mask[F - 1] = 0
        ix = _load(idxmod, slab)
        ix.set_screen(screen)
        for m in (None, mask):
            oi, osc = O.c_search_fused(slab, q, W, m, 100)
            r = ix.search_fused(q, W, m, 100)
            assert np.array_equal(r["ids"], oi) and np.array_equal(r["scores"].view(np.uint32), osc.view(np.uint32)), (F, D, E, Q, screen)
        # rows change -> the companion is refilled
        r7 = min(7, D - 1)
        slab[0, r7] = q[0] * 5.0
        ix.write_rows(0, r7, slab[0, r7:r7 + 1])
        oi, osc = O.c_search_fused(slab, q, W, None, 100)
        r = ix.search_fused(q, W, None, 100)
        assert np.array_equal(r["ids"], oi) and np.array_equal(r["scores"].view(np.uint32), osc.view(np.uint32))
        # not query-conditioned: W is [F]
        w1 = rng.standard_normal(F).astype(np.float32)
        oi, osc = O.c_search_fused(slab, q, w1, None, 50, query_cond=False)
        r = ix.search_fused(q, w1, None, 50, query_cond=False)
        assert np.array_equal(r["ids"], oi) and np.array_equal(r["scores"].view(np.uint32), osc.view(np.uint32))
        ix.close()
    # Recall@20 against planted relevance (mfar/synth.py) at the headline embedding width: fused >= two-stage - 0.001 (SURVEY
    # 8d).  (On weak-signal data the two result sets differ by design -- at dim 128 the exhaustive mix finds 0.54 of the planted
    # documents where the per-field lists find 0.62 -- which is why the fused mode claims recall parity per corpus, not ids.)
    import torch
    from mfar import synth
    cp = synth.SyntheticCorpus(40000, 4, 768, n_queries=256, seed=7, device="cuda:0")
    ix = cp.build_index(idxmod)
    qs = cp.queries(0, 256)
    two = ix.search(qs, cp.W, None)["ids"].cpu().numpy()
    fu = ix.search_fused(qs, cp.W, None, 100)["ids"].cpu().numpy()
    rel = cp.qrels(0, 256)
    rec = lambda ids: float(np.mean([len(set(ids[j, :20].tolist()) & rel[j]) / len(rel[j]) for j in range(256)]))
    assert rec(fu) >= rec(two) - 0.001, (rec(fu), rec(two))
    ix.close()


def test_bf16_index_with_certified_screen(idxmod):
    """A bf16 index runs the CERTIFIED pass by default (from 16 384 rows, like an fp32 index): stage 1 scans the bf16 slab itself --
    no second copy of the rows -- with the query split into two bf16 terms (64 columns per scan, 128 for blocks of more than 64
    queries), re-scores the k' best unique rows of every list with the bf16 contract's exact natural-order chain and proves the
    lists, which makes them BIT-identical to the oracle (the plain three-term MFMA pass is only within 1e-4).  The pass scans every
    document but ranks unique rows (duplicate groups are masked down to their representative and expanded by the certificate);
    a forced fall-back (every certificate fails) stays within the plain pass's tolerance."""
    rng = np.random.default_rng(61)
    for F, D, E, Q, dup in ((3, 30000, 96, 70, 0), (2, 40000, 768, 128, 3000), (1, 20000, 64, 9, 0), (2, 17000, 128, 200, 500)):
        slab, q, W = _mk(rng, F, D, E, Q, dup=dup)
        rs = O.bf16_round(slab)
        ix = _load_bf16(idxmod, slab)
        assert ix.max_split_batch(100) == 128          # on by default: the wide pass serves blocks of more than 64 queries
        r = ix.search(q, W, None, return_fields=True)
        with O.chain("natural"):
            o = O.c_two_stage(rs, q, W, None)
        for key in ("field_ids", "ids"):
            assert np.array_equal(r[key], o[key]), (key, F, D, E, Q)
        for key in ("field_scores", "scores"):
            assert np.array_equal(r[key].view(np.uint32), o[key].view(np.uint32)), (key, F, D, E, Q)
        st = ix.screen_stats()
        assert st["built"] and st["n_checked"] == Q * F and st["n_failed"] == 0, st
        assert st["screen_bytes"] == 0 and st["scan_rows"] == D * F          # the scan reads the index's own rows
        if dup:
            assert min(st["unique_rows"]) <= D - dup + 1
        ix.set_screen(0)                               # the plain bf16 pass: same lists up to its 1e-4
        rp = ix.search(q, W, None, return_fields=True)
        assert ix.screen_stats()["n_checked"] == Q * F
        for f in range(F):
            O.assert_topk_equivalent(rp["field_ids"][:, f], rp["field_scores"][:, f], o["field_ids"][:, f], o["field_scores"][:, f],
                                     tol=TOL, what=f"plain bf16 pass f{f}")
        ix.set_screen(2, 1e9)                          # every certificate fails: the plain bf16 pass repairs
        rx = ix.search(q, W, None, return_fields=True)
        for f in range(F):
            O.assert_topk_equivalent(rx["field_ids"][:, f], rx["field_scores"][:, f], o["field_ids"][:, f], o["field_scores"][:, f],
                                     tol=TOL, what=f"bf16 forced fall-back f{f}")
        ix.close()


def test_bf16_certified_pass_with_duplicates_on_top(idxmod):
    """Groups of bit-identical rows at the TOP of every list (a low-cardinality field, Zipf duplicates, the planted best row
    repeated 300 times), rows written in several pieces and rewritten afterwards, two row shards: the certified bf16 pass ranks
    each distinct vector once, the certificate expands the groups (score desc, doc id asc) -- no list falls back, every bit is
    the oracle's."""
    rng = np.random.default_rng(62)
    F, D, E, Q = 3, 50000, 192, 96
    slab, q, W = _mk(rng, F, D, E, Q)
    texts = (rng.standard_normal((10, E)) * 0.5).astype(np.float32)
    slab[0] = texts[rng.integers(0, 10, D)]                            # ten distinct vectors in the whole field
    z = np.minimum((np.exp(rng.random(D) * np.log(D / 4)) - 1).astype(np.int64), D // 4 - 1)
    slab[1] = slab[1][z]                                               # Zipf-like duplicates
    best = (q[0] / np.linalg.norm(q[0]) * 6.0).astype(np.float32)
    slab[2, rng.choice(D, 300, replace=False)] = best                  # 300 copies of the row every query likes best
    rs = O.bf16_round(slab)
    with O.chain("natural"):
        o = O.c_two_stage(rs, q, W, None)
    ix = _load_bf16(idxmod, slab)
    r = ix.search(q, W, None, return_fields=True)
    st = ix.screen_stats()
    assert st["built"] and st["n_failed"] == 0 and st["unique_rows"][0] == 10 and st["unique_rows"][2] <= D - 299, st
    for key in ("field_ids", "ids"):
        assert np.array_equal(r[key], o[key]), key
    for key in ("field_scores", "scores"):
        assert np.array_equal(r[key].view(np.uint32), o[key].view(np.uint32)), key
    # rows rewritten: the tables follow
    slab[2, 7:11] = best * 1.5
    ix.write_rows(2, 7, slab[2, 7:11])
    rs = O.bf16_round(slab)
    with O.chain("natural"):
        o2 = O.c_two_stage(rs, q[:64], W, None)
    r2 = ix.search(q[:64], W, None, return_fields=True)                # 64 queries: the 64-column certified pass
    assert np.array_equal(r2["field_ids"], o2["field_ids"]) and np.array_equal(r2["scores"].view(np.uint32), o2["scores"].view(np.uint32))
    assert ix.screen_stats()["n_failed"] == 0
    ix.close()
    half = 23000
    shards = [_load_bf16(idxmod, slab[:, :half]), _load_bf16(idxmod, slab[:, half:], row_offset=half)]
    import torch
    qd, Wd = torch.from_numpy(q[:64]).cuda(), torch.from_numpy(W).cuda()
    rm = idxmod.merge_payloads(torch.cat([s_.search_local(qd) for s_ in shards]), 2, qd, Wd, None, n_fields=F)
    assert np.array_equal(rm["ids"].cpu().numpy(), o2["ids"]) and np.array_equal(rm["scores"].cpu().numpy().view(np.uint32), o2["scores"].view(np.uint32))
    for s_ in shards:
        s_.close()


def test_row_mode_certifies_heavy_tailed_fields(idxmod):
    """ROW MODE of the certified screen (csrc/mfar_screen.h): a field whose largest centred row norm is far above the typical one ranks
    its rows by approx + eps(row norm) -- an upper bound of the exact score -- so the certificate no longer pays for the field's worst
    row.  Three fields: Gaussian, Pareto(3)-scaled spread (norms up to 30 x typical), a handful of huge outliers.  With the mode on,
    every list of the wide pass is certified and every bit equals the oracle; with MFAR_SCREEN_ROW_MODE=0 semantics (set through the
    eps knob: the field-wide bound) lists fail and are repaired to the same bits.  The 64-column pass (field-wide bound) agrees."""
    rng = np.random.default_rng(77)
    F, D, E, Q = 3, 60000, 256, 128
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * 0.04 + mu).astype(np.float32)
    scale = np.minimum((1.0 - rng.random((D, 1))) ** (-1.0 / 3.0), 30.0).astype(np.float32)
    slab[1] = ((slab[1] - mu) * scale + mu).astype(np.float32)
    out = rng.choice(D, 12, replace=False)
    slab[2, out] = (mu + rng.standard_normal((12, E)) * 2.0).astype(np.float32)          # a dozen rows 50 x the typical spread
    q = (rng.standard_normal((Q, E)) * 0.04 + mu).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    o = O.c_two_stage(slab, q, W, None)
    ix = _load(idxmod, slab)
    ix.set_screen(2)
    r0 = ix.search(q, W, None, return_fields=True)       # auto: nothing activated yet -- field-wide bounds, failed lists repaired exactly
    info = ix.row_mode_info()
    assert info["eligible"] == [1, 2] and info["active"] == [], info
    n_failed_plain = ix.screen_stats()["n_failed"]
    ix.activate_row_mode()                               # (what PipelinedSearcher does when a launch reports a failed certificate)
    assert ix.row_mode_info()["active"] == [1, 2]
    r = ix.search(q, W, None, return_fields=True)
    st = ix.screen_stats()
    assert st["built"] and st["n_checked"] == 2 * Q * F, st
    for rr in (r0, r):
        for key in ("field_ids", "ids"):
            assert np.array_equal(rr[key], o[key]), key
        for key in ("field_scores", "scores"):
            assert np.array_equal(rr[key].view(np.uint32), o[key].view(np.uint32)), key
    assert st["n_failed"] == n_failed_plain, st           # the heavy-tailed fields certify: no new failure with per-row bounds
    assert n_failed_plain > 0, "the test corpus no longer defeats the field-wide bound: make it harsher"
    ix.set_row_mode(0)
    assert ix.row_mode_info()["active"] == []
    ix.set_row_mode(2)
    r64 = ix.search(q[:64], W, None, return_fields=True)      # the 64-column pass: field-wide bound, same bits (repairs allowed)
    assert np.array_equal(r64["field_ids"], o["field_ids"][:64]) and np.array_equal(r64["scores"].view(np.uint32), o["scores"][:64].view(np.uint32))
    ix.close()


def test_auto_off_switches_clustered_fields_to_the_exact_pass(idxmod):
    """The certified screen's worst case: two of eight fields are CLUSTERED -- ~235 near-duplicate, non-identical rows per
    cluster, so the k' = 192 approximate candidates of nearly every list tie inside the error bound and the certificate fails launch after
    launch.  The library reads the flags of finished launches and (1) repairs on the device instead of reporting, (2) switches exactly those
    two fields off: their lists then come from the exact fp32 pass, the screen scans the other six only.  Oracle bits before and after the
    switch; probes switch the fields back on once their rows are replaced by plain ones."""
    import torch
    from mfar import synth
    from mfar.data.pipeline import PipelinedSearcher
    D, F, E, Q = 60_000, 8, 128, 64
    kinds = ["plain", "plain", "clustered", "plain", "plain", "clustered", "plain", "plain"]
    corpus = synth.SyntheticCorpus(D, F, E, n_queries=40 * Q, seed=0xdeadbeef, device="cuda:0", field_kinds=kinds)
    ix = corpus.build_index(idxmod)
    ix.set_tier2(0)            # the policy on its own: with tier 2 (round 6, tests/test_gpu_tier2.py) these lists never reach the exact pass
    W = corpus.W
    mask = torch.ones(F, device="cuda:0")
    mask[3] = 0
    slab = np.stack([corpus.rows(f, 0, D).cpu().numpy() for f in range(F)])
    Wn, mn = W.cpu().numpy(), mask.cpu().numpy()

    def run(first, n):
        ps = PipelinedSearcher(ix, W, mask, max_batch=Q)
        out, tickets = [], []
        for j in range(n):
            tickets.append(ps.submit(corpus.queries((first + j) * Q, Q)))
            if j >= ps.lag:
                out.append({k: v.clone() for k, v in ps.result(tickets[j - ps.lag]).items()})
        for t in tickets[max(0, n - ps.lag):]:
            out.append({k: v.clone() for k, v in ps.result(t).items()})
        torch.cuda.synchronize()
        return ps, out

    def check(first, outs, which):
        for j in which:
            o = O.c_two_stage(slab, corpus.queries((first + j) * Q, Q).cpu().numpy(), Wn, mn)
            assert np.array_equal(outs[j]["ids"].cpu().numpy(), o["ids"]), ("ids", first + j)
            assert np.array_equal(outs[j]["scores"].cpu().numpy().view(np.uint32), o["scores"].view(np.uint32)), ("score bits", first + j)

    assert ix.auto_off_info()["off"] == []
    ps, outs = run(0, 36)                                     # 18 launches of 128 queries (+ the redone ones)
    info = ix.auto_off_info()
    assert info["off"] == [2, 5], info                        # exactly the clustered fields
    assert info["inline_repair"] or ps.n_redone <= 6, (info, ps.n_redone)
    check(0, outs, [0, 1, 34, 35])                            # before the switch (screen + repair) and after it (exact pass for 2, 5)
    # steady state: the screen runs over six fields and none of ITS lists fails; the lists of 2 and 5 are the exact pass's
    st0 = ix.screen_stats()
    r = ix.search(corpus.queries(36 * Q, 2 * Q), W, mask, return_fields=True)      # the synchronous entry point takes the same path
    torch.cuda.synchronize()
    st1 = ix.screen_stats()
    assert st1["n_failed"] == st0["n_failed"] and st1["n_checked"] - st0["n_checked"] == 2 * Q * 6, (st0, st1)
    o = O.c_two_stage(slab, corpus.queries(36 * Q, 2 * Q).cpu().numpy(), Wn, mn)
    assert np.array_equal(r["field_ids"].cpu().numpy(), o["field_ids"]) and np.array_equal(r["ids"].cpu().numpy(), o["ids"])
    assert np.array_equal(r["field_scores"].cpu().numpy().view(np.uint32), o["field_scores"].view(np.uint32))
    assert np.array_equal(r["scores"].cpu().numpy().view(np.uint32), o["scores"].view(np.uint32))
    # switching the policy off screens everything again (and repairs): same bits
    ix.set_auto_off(0)
    assert ix.auto_off_info()["off"] == []
    r0 = ix.search(corpus.queries(36 * Q, 2 * Q), W, mask)
    torch.cuda.synchronize()
    assert torch.equal(r0["ids"], r["ids"]) and torch.equal(r0["scores"], r["scores"])
    # back on: plain rows in the two fields, a probe every second launch -> two clean probes each switch them on again
    ix.set_auto_off(1, 0, 2)
    _, _ = run(0, 36)
    assert ix.auto_off_info()["off"] == [2, 5]
    plain = synth.SyntheticCorpus(D, F, E, n_queries=Q, seed=7, device="cuda:0")
    for f in (2, 5):
        rows = plain.rows(f, 0, D)
        ix.write_rows(f, 0, rows)
        slab[f] = rows.cpu().numpy()
    n_on0 = ix.auto_off_info()["n_switched_on"]
    ps, outs = run(40, 16)
    info = ix.auto_off_info()
    assert info["off"] == [] and info["n_switched_on"] == n_on0 + 2 and info["n_probes"] >= 2, info
    check(40, outs, [0, 15])
    ix.close()


def test_bf16_lists_carry_the_chain_bits_whatever_path_wrote_them(idxmod):
    """The bf16 contract: with the certified stage 1 in use -- the default at ANY index size -- every list of a bf16 index carries
    the natural-order chain's bits: certified by the screen, or written by the exhaustive chain pass (csrc/mfar_exact16.h) when a certificate
    fails.  Checked against the oracle bit for bit for: a tiny index (under the fp32 screen's 16 384-row threshold), every certificate forced
    to fail, a clustered field whose certificates fail for real, both sentinel modes, 64- and 128-column blocks; and a row-sharded run with a
    small last shard equals the unsharded one (a failing list in ONE shard no longer changes bits)."""
    rng = np.random.default_rng(77)
    for F, D, E, Q in ((3, 700, 64, 9), (2, 30000, 96, 100), (4, 20000, 128, 64)):
        slab, q, W = _mk(rng, F, D, E, Q, mean=0.2, dup=7)
        if D >= 20000:          # field 0: clusters of ~250 near-duplicates (not identical even after the bf16 rounding: distinct low bits)
            centres = slab[0][rng.integers(0, D, size=D // 250)]
            slab[0] = centres[rng.integers(0, centres.shape[0], size=D)] * (1.0 + rng.integers(-3, 4, size=(D, E)).astype(np.float32) * np.float32(2.0 ** -8))
        rs = O.bf16_round(slab)
        ix = _load_bf16(idxmod, slab)
        for sentinel in (True, False):
            with O.chain("natural"):
                o = O.c_two_stage(rs, q, W, None, sentinel=sentinel)
            for eps_mult in (1.0, 1e9):
                ix.set_screen(1, eps_mult)
                r = ix.search(q, W, None, sentinel=sentinel, return_fields=True)
                for key in ("field_ids", "ids"):
                    assert np.array_equal(r[key], o[key]), (F, D, E, Q, sentinel, eps_mult, key)
                for key in ("field_scores", "scores"):
                    assert np.array_equal(r[key].view(np.uint32), o[key].view(np.uint32)), (F, D, E, Q, sentinel, eps_mult, key)
        st = ix.screen_stats()
        assert st["built"] and st["n_failed"] >= Q * F, st           # the forced failures went through the chain pass
        ix.close()
        # two row shards, the second one small: same bits as the unsharded search
        cut = D - 300
        with O.chain("natural"):
            o = O.c_two_stage(rs, q, W, None)
        shards = [_load_bf16(idxmod, slab[:, :cut]), _load_bf16(idxmod, slab[:, cut:], row_offset=cut)]
        payloads = np.concatenate([sh.search_local(q) for sh in shards])
        rm = idxmod.merge_payloads(payloads, 2, q, W, None)
        assert np.array_equal(rm["ids"], o["ids"]) and np.array_equal(rm["scores"].view(np.uint32), o["scores"].view(np.uint32)), (F, D, E)
        for sh in shards:
            sh.close()


def test_limits_of_fields_and_list_depth(idxmod):
    """The widest call the ABI takes: MFAR_MAX_FIELDS = 32 fields x k1 = 128 = 4096 candidates per query (the union's and the mixer's LDS
    budget), k2 = 128 = SEL_MAX_K -- synchronous scorer and C-ABI pipeline against the oracle, screen on and off, a ragged 70-query call
    (one wide block + a rest); one past either limit is refused with an error, not truncated."""
    from mfar import _native
    from mfar.data.pipeline import NativePipeline
    rng = np.random.default_rng(44)
    F, D, E, Q, k = 32, 20000, 64, 70, 128
    slab, q, W = _mk(rng, F, D, E, Q, mean=0.2, dup=3)
    mask = (rng.random(F) < 0.8).astype(np.float32)
    ix = _load(idxmod, slab)
    o = O.c_two_stage(slab, q, W, mask, k1=k, k2=k, sentinel=True)
    assert (o["n_cand"] > 3000).any()                               # the candidate lists really are near the 4096 limit
    for screen in (2, 0):
        ix.set_screen(screen)
        r = ix.search(q, W, mask, k1=k, k2=k)
        assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32)), screen
        pl = NativePipeline(ix, W, mask, k1=k, k2=k, max_batch=64)
        t0, t1 = pl.submit(q[:64]), pl.submit(q[64:])
        for t, a, b in ((t0, 0, 64), (t1, 64, Q)):
            g = pl.result(t)
            assert np.array_equal(g["ids"], o["ids"][a:b]) and np.array_equal(g["scores"].view(np.uint32), o["scores"][a:b].view(np.uint32)), (screen, a)
        pl.close()
    st = ix.screen_stats()
    assert st["n_checked"] > 0 and st["n_failed"] == 0, st
    with pytest.raises(_native.MfarError):
        ix.search(q, W, mask, k1=129, k2=100)                       # 32 x 129 > 4096
    with pytest.raises(_native.MfarError):
        NativePipeline(ix, W, mask, k1=129, k2=100, max_batch=64)
    with pytest.raises(_native.MfarError):
        ix.search(q, W, mask, k1=100, k2=129)                       # k2 > SEL_MAX_K
    with pytest.raises(Exception):
        idxmod.MultiFieldIndex(100, 33, E, device=0)                # n_fields > MFAR_MAX_FIELDS
    ix.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_full_hbm_degrades_to_the_exact_pass(idxmod, dtype):
    """HBM nearly full when a query arrives.  (0) by the library's allocation-failure hook, deterministically, and (1) with a really full HBM:
    not even the call's scratch fits -- a clean MFAR_ERR_NOMEM, no stale HIP error left behind, the next call works; (1b) the same through
    the C-ABI pipeline: the launch that could not be enqueued keeps its batch, the error repeats while memory is short, the batch is run when
    its result is taken afterwards; (2) a little more room: the screen's big buffers (fp16 copy,
    gather slab) cannot all be allocated -- a clean error or the oracle's rows from whatever could be built, and the right rows once memory
    is back."""
    import contextlib
    import torch
    from mfar import _native
    from mfar.data.pipeline import NativePipeline
    rng = np.random.default_rng(71)
    F, D, E, Q = 4, 120_000, 256, 9
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E), dtype=np.float32) * 0.5 + 0.3 * mu * 4.0).astype(np.float32)
    q = (rng.standard_normal((Q, E)) * 0.5 + mu * 2.0).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    ref = O.bf16_round(slab) if dtype == "bf16" else slab
    with (O.chain("natural") if dtype == "bf16" else contextlib.nullcontext()):
        o = O.c_two_stage(ref, q, W, None)

    def right(r, what):
        if dtype == "f32":
            assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32)), what
        else:
            O.assert_topk_equivalent(r["ids"], r["scores"], o["ids"], o["scores"], tol=TOL, what=what)

    def fresh():
        ix = idxmod.MultiFieldIndex(D, F, E, device=0, dtype=dtype)
        for f in range(F):
            ix.write_rows(f, 0, slab[f])
        torch.cuda.synchronize()
        return ix

    import gc
    gc.collect()
    torch.cuda.empty_cache()
    room = F * D * E * 4 + (4 << 30)                 # one index at a time, its screen + gather slab, a pipeline's scratch (3 slots x ~0.3 GB)
    big = torch.empty(torch.cuda.mem_get_info(0)[0] - room, dtype=torch.uint8, device="cuda:0")     # (one huge allocation for the whole test)

    def squeeze(keep_mb):
        gc.collect()                 # (garbage of earlier tests -- an index in a reference cycle -- must not be freed between this and the call
        torch.cuda.empty_cache()     #  under test: its hipFree would hand the call the memory it is supposed to lack)
        return torch.empty(max(0, torch.cuda.mem_get_info(0)[0] - (keep_mb << 20)), dtype=torch.uint8, device="cuda:0")

    L = _native.lib()
    try:
        # (0) the same semantics, DETERMINISTICALLY: the library's test hook makes every scratch allocation of >= 64 MB fail like a hipMalloc
        #     that ran out of memory (with a really full HBM -- below -- whether a call fits depends on what the HIP runtime can still give back)
        ix = fresh()
        _native.check(L.mfar_debug_fail_allocations_above(64 << 20))
        try:
            with pytest.raises(_native.MfarError) as err:
                ix.search(q, W, None)
            assert "NOMEM" in str(err.value) or "out of memory" in str(err.value), str(err.value)
            pl = NativePipeline(ix, W, None, max_batch=16, coalesce=1)
            with pytest.raises(_native.MfarError) as err:
                pl.submit(q)
            ticket = err.value.ticket
            assert ticket is not None
            with pytest.raises(_native.MfarError):
                pl.result(ticket)                       # still no memory: the error repeats, nothing half-done comes back
        finally:
            _native.check(L.mfar_debug_fail_allocations_above(0))
        got = [pl.result(ticket), pl.result(pl.submit(q))]      # memory is back: the kept batch is run now; new batches flow
        assert pl.n_redone >= 1
        for g in got:
            right(g, "pipeline after a failed launch (test hook)")
        right(ix.search(q, W, None), "after a failed call (test hook)")
        pl.close()
        ix.close()
        # (1) a really full HBM
        ix = fresh()
        hog = squeeze(24)
        try:
            r = ix.search(q, W, None)
            right(r, "24 MB free, and the runtime found room")
        except _native.MfarError as e:
            assert "NOMEM" in str(e) or "out of memory" in str(e), str(e)
        del hog
        torch.cuda.empty_cache()
        right(ix.search(q, W, None), "after a failed call")
        ix.close()
        # (1b)
        ix = fresh()
        pl = NativePipeline(ix, W, None, max_batch=16, coalesce=1)
        hog = squeeze(24)
        # (in a long-lived process the HIP runtime may find the 150 MB of launch scratch after all -- a failed hipMalloc makes it give back
        #  what it caches -- so the launch either fails cleanly or goes through; alone, this test always takes the first branch)
        try:
            ticket = pl.submit(q)
            raised = False
        except _native.MfarError as e:
            ticket, raised = e.ticket, True
        assert ticket is not None
        if raised:
            try:                                        # still (almost) no memory: the error repeats -- or the kept batch now fits through
                right(pl.result(ticket), "kept batch under memory pressure")  # the degraded paths (optional scratch skipped): never half-done
            except _native.MfarError:
                pass
        del hog
        torch.cuda.empty_cache()
        got = [pl.result(ticket), pl.result(pl.submit(q))]      # memory is back: the kept batch is run now; new batches flow
        assert pl.n_redone >= 1 or not raised
        for g in got:
            right(g, "pipeline after a failed launch")
        pl.close()
        ix.close()
        # (2)
        for keep_mb in (260, 520):
            ix = fresh()
            hog = squeeze(keep_mb)
            r = None
            try:
                r = ix.search(q, W, None)
            except _native.MfarError as e:
                assert "NOMEM" in str(e) or "out of memory" in str(e), str(e)
            del hog
            torch.cuda.empty_cache()
            if r is not None:
                right(r, f"under memory pressure, {keep_mb} MB free")
            right(ix.search(q, W, None), f"after memory pressure, {keep_mb} MB")
            ix.close()
    finally:
        del big
        torch.cuda.empty_cache()


def test_degenerate_sizes_and_one_large_call(idxmod):
    """An EMPTY index, one row, two rows (fewer rows than k, lists shorter than k1, a final list shorter than k2) through the synchronous
    scorer and the C-ABI pipeline, both padding conventions; and 1000 queries in ONE synchronous call (eight wide blocks of 128): the oracle's
    rows on their valid prefix, its n_valid."""
    from mfar.data.pipeline import NativePipeline
    rng = np.random.default_rng(90)
    for D in (0, 1, 2):
        ix = idxmod.MultiFieldIndex(D, 2, 32, device=0)
        slab = rng.standard_normal((2, D, 32)).astype(np.float32)
        q = rng.standard_normal((5, 32)).astype(np.float32)
        W = (rng.standard_normal((32, 2)) * 0.05).astype(np.float32)
        for f in range(2):
            if D:
                ix.write_rows(f, 0, slab[f])
        for sentinel in (True, False):
            r = ix.search(q, W, None, k1=10, k2=10, sentinel=sentinel)
            pl = NativePipeline(ix, W, None, k1=10, k2=10, sentinel=sentinel, max_batch=8)
            g = pl.result(pl.submit(q))
            pl.close()
            if D == 0:
                assert (r["n_valid"] == 0).all() and (g["n_valid"] == 0).all()
                continue
            o = O.c_two_stage(slab, q, W, None, k1=10, k2=10, sentinel=sentinel)
            for res in (r, g):
                assert np.array_equal(res["n_valid"], o["n_valid"]), (D, sentinel)
                for i in range(5):
                    v = int(o["n_valid"][i])
                    assert np.array_equal(res["ids"][i, :v], o["ids"][i, :v]) and \
                        np.array_equal(res["scores"][i, :v].view(np.uint32), o["scores"][i, :v].view(np.uint32)), (D, sentinel, i)
        ix.close()
    F, D, E, Q = 3, 5000, 64, 1000
    slab, q, W = _mk(rng, F, D, E, Q)
    ix = _load(idxmod, slab)
    o = O.c_two_stage(slab, q, W, None)
    for screen in (0, 2):
        ix.set_screen(screen)
        r = ix.search(q, W, None)
        assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32)), screen
    ix.close()
