#!/usr/bin/env python3
"""Randomised parity stress of stage 1 (all scan kernels, screen on/off), of the whole scorer (with the screen on: certified
re-scoring prefix, two-level stage 2), of the C-ABI batch pipeline over ragged batches and of the multi-GPU data path over ragged row
shards (in one process) against the C oracle -- fp32 indexes, and bf16 indexes wherever their bit contract applies.
    python tests/stress.py [seconds] [seed] [big]   -- prints one line per configuration, exits non-zero on the first mismatch"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
sys.path.insert(0, ROOT)
import numpy as np


def _pipeline_section(rng, ix, slab, q, W, mask, o, k, sentinel, eps_mult, O, dtype, what):
    """The C-ABI batch pipeline (mfar_pipeline_*, the path bench.py times) over the same queries cut into RAGGED batches: random depth /
    coalescing / batch limit, host or device buffers, results taken late, at once or out of order, a weight change half-way (earlier tickets
    must keep the old weights' results), forced certificate failures (the redo path) -- every batch bit for bit the oracle's rows."""
    import torch
    from mfar.data.pipeline import NativePipeline
    Q = q.shape[0]
    max_batch = int(rng.choice([1, 3, 17, 64, 64, rng.integers(1, 65)]))
    depth, coalesce = int(rng.choice([0, 0, 2, 3, 4])), int(rng.choice([0, 0, 1, 2]))
    on_dev = bool(rng.integers(0, 2))
    ix.set_screen(int(rng.choice([0, 2, 2])) if dtype == "f32" else 2, eps_mult)
    if coalesce * max_batch > ix.max_split_batch(k):        # (no wide pass for this index: an explicit request would be refused)
        coalesce = int(rng.choice([0, 1]))
    cuts, at = [], 0
    while at < Q:
        nq = int(min(Q - at, rng.integers(1, max_batch + 1)))
        cuts.append((at, nq))
        at += nq
    change_at = int(rng.integers(1, len(cuts))) if len(cuts) > 1 and rng.random() < 0.4 else -1
    o2 = None
    if change_at >= 0:
        W2 = (rng.standard_normal(W.shape) * 0.05).astype(np.float32)
        mask2 = (rng.random(mask.shape[0]) < 0.7).astype(np.float32)
        if not mask2.any():
            mask2[0] = 1.0
        import contextlib
        with (O.chain("natural") if dtype == "bf16" else contextlib.nullcontext()):      # (`slab` holds the bf16-rounded rows then)
            o2 = O.c_two_stage(slab, q, W2, mask2, k1=k, k2=k, sentinel=sentinel)
    dev = torch.device("cuda", 0)
    put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)) if on_dev else (lambda a: np.ascontiguousarray(a))
    get = (lambda t: t.cpu().numpy()) if on_dev else (lambda a: a)
    pl = NativePipeline(ix, put(W), put(mask), k1=k, k2=k, sentinel=sentinel, max_batch=max_batch, depth=depth, coalesce=coalesce)
    # the validity rule of include/mfar_hip.h, modelled here: a batch belongs to the launch that was open when it was submitted; a launch
    # starts when it holds `coalesce` batches, or early -- flush / set_weights / the result of a batch still held; a ticket is valid until
    # `depth` LATER launches have started
    late = int(rng.integers(0, 2 * pl.depth * pl.coalesce))      # how long results are left lying (the model below takes them when it must)
    pending, ok = [], True
    st = dict(launched=0, held=0)

    def take(j):
        nonlocal ok
        t, a, nq, oo, _, launch = pending[j]
        if launch == st["launched"]:                  # still held: taking it starts its launch alone
            before_a_launch(skip=pending[j])
            st["launched"] += 1
            st["held"] = 0
        j = next(j_ for j_, x_ in enumerate(pending) if x_[0] == t)
        pending.pop(j)
        r = pl.result(t)
        ids, sc, nv = get(r["ids"]), get(r["scores"]), get(r["n_valid"])
        onv = oo["n_valid"][a:a + nq]
        same = np.array_equal(nv, onv)
        for j_ in range(nq if same else 0):          # (entries behind n_valid are padding)
            v = int(onv[j_])
            same = same and np.array_equal(ids[j_, :v], oo["ids"][a + j_, :v]) and \
                np.array_equal(sc[j_, :v].view(np.uint32), oo["scores"][a + j_, :v].view(np.uint32))
        if not same:
            ok = False
            print("MISMATCH pipeline", dict(what, batch=(a, nq), max_batch=max_batch, depth=pl.depth, coalesce=pl.coalesce, on_dev=on_dev, late=late,
                                            change_at=change_at), flush=True)

    def before_a_launch(skip=None):                   # the tickets the next launch's start would retire
        for e in [e for e in pending if e is not skip and e[5] <= st["launched"] - pl.depth]:
            take(next(j_ for j_, x_ in enumerate(pending) if x_ is e))

    for i, (a, nq) in enumerate(cuts):
        if i == change_at:
            if st["held"]:
                before_a_launch()
                st["launched"] += 1
                st["held"] = 0
            pl.set_weights(put(W2), put(mask2))       # flushes and drains; tickets submitted before it keep the old weights' rows
        if st["held"] + 1 == pl.coalesce:
            before_a_launch()
        pending.append((pl.submit(put(q[a:a + nq])), a, nq, o2 if (change_at >= 0 and i >= change_at) else o, i, st["launched"]))
        st["held"] += 1
        if st["held"] == pl.coalesce:
            st["launched"] += 1
            st["held"] = 0
        due = [e for e in pending if e[4] <= i - late]
        rng.shuffle(due)
        for e in due:
            if any(x_ is e for x_ in pending):
                take(next(j_ for j_, x_ in enumerate(pending) if x_ is e))
        if pending and rng.random() < 0.3:                       # and now and then one that is not due yet
            take(int(rng.integers(0, len(pending))))
    while pending:
        take(int(rng.integers(0, len(pending))))
    if on_dev:
        torch.cuda.synchronize()
    note = f"d{pl.depth}c{pl.coalesce}b{max_batch}n{len(cuts)}{'D' if on_dev else 'H'}{'w' if change_at >= 0 else ''}r{pl.n_redone}"
    pl.close()
    return ok, note


def _mask_sweep_section(rng, ix, ref, q, W, k, sentinel, eps_mult, O, dtype, what):
    """mask_fields' sweep (mfar_search_stage2_masks): the union and stage 2 once, the mixer once per mask, for M random masks (any sign,
    all-zero included), with and without the lists' exact scores as known pairs, stage-2 modes 0 / 1 / 2 (full gather / two-level up to two
    masks / two-level over the union of every mask's survivors) -- each mask's rows against the oracle run with that mask."""
    import contextlib
    import torch
    F = ref.shape[0]
    dev = torch.device("cuda", 0)
    M = int(rng.choice([1, 2, 3, 7]))
    masks = rng.choice(np.array([0.0, 1.0, 1.0, 1.0, -1.0, 0.5, 2.0], np.float32), size=(M, F)).astype(np.float32)
    if rng.random() < 0.2:
        masks[int(rng.integers(0, M))] = 0.0
    ix.set_screen(int(rng.choice([0, 2, 2])) if dtype == "f32" else 2, eps_mult)
    ix.set_stage2_mode(int(rng.choice([0, 1, 2])))
    ok = True
    for b0 in range(0, q.shape[0], 128):
        qb = np.ascontiguousarray(q[b0:b0 + 128])
        qd, Wd, md = torch.from_numpy(qb).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(masks).to(dev)
        fid, fsc = ix.retrieve_fields(qd, k, sentinel)
        known = bool(rng.integers(0, 2))
        sw = ix.search_stage2_masks(qd, Wd, fid, md, k1=k, k2=k, field_scores=fsc if known else None, sentinel=sentinel)
        torch.cuda.synchronize()
        for m in range(M):
            with (O.chain("natural") if dtype == "bf16" else contextlib.nullcontext()):
                oo = O.c_two_stage(ref, qb, W, masks[m], k1=k, k2=k, sentinel=sentinel)
            ids, sc, nv = sw["ids"][m].cpu().numpy(), sw["scores"][m].cpu().numpy(), sw["n_valid"][m].cpu().numpy()
            same = np.array_equal(nv, oo["n_valid"])
            for j_ in range(qb.shape[0] if same else 0):
                v = int(oo["n_valid"][j_])
                same = same and np.array_equal(ids[j_, :v], oo["ids"][j_, :v]) and np.array_equal(sc[j_, :v].view(np.uint32), oo["scores"][j_, :v].view(np.uint32))
            if not same:
                ok = False
                print("MISMATCH mask sweep", dict(what, M=M, mask=masks[m].tolist(), known=known, block=b0), flush=True)
    ix.set_stage2_mode(1)
    return ok, f"M{M}"


def _sharded_section(rng, idxmod, slab, q, W, mask, o, k, sentinel, eps_mult, dtype, what):
    """The multi-GPU data path in ONE process: the corpus cut into S RAGGED row shards (tiny ones included), one index per shard with its
    row offset, then the lists-first exchange exactly as the ranks run it -- every shard's stage-1 lists into the "all-gathered" buffer
    (mfar_retrieve_lists), every shard merges them and scores the candidates it owns (mfar_search_owned), the local top-k payloads are merged
    (mfar_merge_topk) -- and the single-payload variant (mfar_search_local + mfar_merge_payloads) for the first block: the oracle's rows of
    the UNSHARDED corpus, bit for bit."""
    import torch
    F, D, E = slab.shape
    S = int(rng.choice([2, 3, 5, 8]))
    if D < 2 * S:
        return True, "-"
    bounds = [0] + sorted(int(x) for x in rng.choice(np.arange(1, D), size=S - 1, replace=False)) + [D]
    dev = torch.device("cuda", 0)
    shards = []
    for g in range(S):
        lo, hi = bounds[g], bounds[g + 1]
        sh = idxmod.MultiFieldIndex(hi - lo, F, E, device=0, row_offset=lo, dtype=dtype)
        for f in range(F):
            sh.write_rows(f, 0, np.ascontiguousarray(slab[f, lo:hi]))
        sh.set_screen(int(rng.choice([0, 2, 2])) if dtype == "f32" else 2, eps_mult)      # (bf16: the contract holds under the certified stage 1)
        shards.append(sh)
    Wd, md = torch.from_numpy(W).to(dev), torch.from_numpy(mask).to(dev)
    ok = True
    Qb = 64
    for b0 in range(0, q.shape[0], Qb):
        qb = torch.from_numpy(np.ascontiguousarray(q[b0:b0 + Qb])).to(dev)
        nq = qb.shape[0]
        nl, nt = shards[0].lists_bytes(nq, k), shards[0].topk_bytes(nq, k)
        lists_all = torch.empty(S * nl, dtype=torch.uint8, device=dev)
        for g, sh in enumerate(shards):
            sh.retrieve_lists(qb, lists_all[g * nl:(g + 1) * nl], k, sentinel)
        topk_all = torch.empty(S * nt, dtype=torch.uint8, device=dev)
        for g, sh in enumerate(shards):
            sh.search_owned(lists_all, S, qb, Wd, topk_all[g * nt:(g + 1) * nt], md, k1=k, k2=k, sentinel=sentinel)
        rx = idxmod.merge_topk(topk_all, S, nq, k2=k)
        results = [("lists-first", rx)]
        if b0 == 0:
            npay = shards[0].payload_bytes(nq, k)
            pay = torch.empty(S * npay, dtype=torch.uint8, device=dev)
            for g, sh in enumerate(shards):
                sh.search_local(qb, k1=k, sentinel=sentinel, payload=pay[g * npay:(g + 1) * npay])
            results.append(("payloads", idxmod.merge_payloads(pay, S, qb, Wd, md, n_fields=F, k1=k, k2=k, sentinel=sentinel)))
        torch.cuda.synchronize()
        for name, r in results:
            ids, sc, nv = (r[x].cpu().numpy() for x in ("ids", "scores", "n_valid"))
            onv = o["n_valid"][b0:b0 + nq]
            same = np.array_equal(nv, onv)
            for j_ in range(nq if same else 0):
                v = int(onv[j_])
                same = same and np.array_equal(ids[j_, :v], o["ids"][b0 + j_, :v]) and \
                    np.array_equal(sc[j_, :v].view(np.uint32), o["scores"][b0 + j_, :v].view(np.uint32))
            if not same:
                ok = False
                print("MISMATCH sharded", name, dict(what, bounds=bounds, block=b0), flush=True)
    for sh in shards:
        sh.close()
    return ok, f"S{S}min{min(bounds[g + 1] - bounds[g] for g in range(S))}"


def one_config(rng, idxmod, O, n, seed, big=False, verbose=True):
    """One random configuration (shapes, data kind, knobs all drawn from `rng`) through stage 1 with the screen off and on, the whole
    scorer and -- sometimes -- the fused mode, against the C oracle.  Returns True when every comparison held."""
    F = int(rng.integers(1, 10))
    E = int(rng.choice([32, 64, 96, 128, 192, 384, 768]))
    D = int(rng.choice([rng.integers(1, 300), rng.integers(300, 5000), rng.integers(5000, 70000)]))
    if big:                            # long chunks, many compactions / drains
        D = int(rng.integers(20000, 250000))
        E = int(rng.choice([32, 96, 96, 192]))
        F = int(rng.integers(1, 4))
    D = min(D, int(3e7 // (F * E)))
    Q = int(rng.choice([rng.integers(1, 9), 64, rng.integers(9, 131), 128, rng.integers(129, 261)]))   # > 64: wide blocks of 128 + a rest
    k = int(rng.choice([1, 10, 100, 100, 128, rng.integers(1, 129)]))
    sentinel = bool(rng.integers(0, 2))
    mean = float(rng.choice([0.3, -0.4, 0.0, 2.0]))
    dtype = "bf16" if rng.random() < 0.2 else "f32"
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * 0.5 + mean * mu * 4.0).astype(np.float32)
    q = (rng.standard_normal((Q, E)) * 0.5 + mu * 2.0).astype(np.float32)
    kind = rng.integers(0, 6)
    eps_mult = 1e9 if rng.random() < 0.15 else 1.0      # forced fail: every certificate fails, the exact pass repairs (auto-off after 12 launches)
    if rng.random() < 0.15:
        eps_mult = float(rng.choice([5.0, 40.0, 400.0]))   # ... or many fail with a band tier 2 can still hold (mfar_set_tier2: threshold rescan)
    if kind == 1 and D > 8:          # duplicate group
        rows = rng.choice(D, size=min(D, int(rng.integers(2, 3000))), replace=False)
        slab[rng.integers(0, F), rows] = slab[0, rows[0]]
    if kind == 2 and D > 8:          # ascending scores for query 0 in one field
        f = int(rng.integers(0, F))
        ramp = np.linspace(0.0, 3.0, D, dtype=np.float32)[:, None] * (q[0] / np.dot(q[0], q[0]))[None, :]
        slab[f] = (slab[f] * 0.01 + ramp).astype(np.float32)
    if kind == 3:                    # tiny values
        slab *= np.float32(1e-12)
    if kind == 4:                    # heavy-tailed row norms in one field + a few huge outliers in another (ROW MODE territory)
        f = int(rng.integers(0, F))
        sc_ = np.minimum((1.0 - rng.random((D, 1))) ** (-1.0 / 3.0), 30.0).astype(np.float32)
        m_ = slab[f].mean(0)
        slab[f] = ((slab[f] - m_) * sc_ + m_).astype(np.float32)
        g = int(rng.integers(0, F))
        slab[g, rng.choice(D, size=min(D, 5), replace=False)] *= np.float32(20.0)
    if kind == 5 and D > 600:        # clusters of ~250 near-duplicate, non-identical rows in one field: certificates fail for real
        f = int(rng.integers(0, F))
        centres = slab[f][rng.integers(0, D, size=max(2, D // 250))]
        slab[f] = (centres[rng.integers(0, centres.shape[0], size=D)] + rng.standard_normal((D, E)).astype(np.float32) * np.float32(5e-5)).astype(np.float32)
    ix = idxmod.MultiFieldIndex(D, F, E, device=0, dtype=dtype)
    ix.set_row_mode(int(rng.choice([0, 1, 2, 2])))       # per-row bounds for heavy-tailed fields: never / auto / always
    ix.set_stage2_dump(int(rng.choice([0, 1, 2, 2])))    # the scan's score dump as stage 2's approximate level
    ix.set_tier2(int(rng.choice([0, 1, 2, 2, 5, 6])))    # tier 2 of the certified screen: never / armed by failures / always; + 4: every list takes the rescan
    ix.set_deep_scan(int(rng.choice([0, 1, 2, 2])))      # deep scan (no first certificate): never / learned per field / every field
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    ref = O.bf16_round(slab) if dtype == "bf16" else slab
    ok = True
    bf16_exact = False           # bf16: the certified stage 1 is in use for this shape -> every list carries the natural-order chain's bits
    n_updates = int(rng.choice([0, 0, 1, 2]))     # rows REPLACED after the first round of checks (a re-encoded field, a few edited rows)
    for update in range(n_updates + 1):
      if update:
        f_ = int(rng.integers(0, F))
        a_ = int(rng.integers(0, D))
        b_ = int(min(D, a_ + rng.choice([1, 7, 300, D])))
        new = (rng.standard_normal((b_ - a_, E)) * 0.5 + mean * mu * 4.0).astype(np.float32)
        if rng.random() < 0.3:
            new[:] = slab[f_, int(rng.integers(0, D))]                     # ... copies of one existing row: a duplicate group appears
        slab[f_, a_:b_] = new
        ix.write_rows(f_, a_, new)
        ref = O.bf16_round(slab) if dtype == "bf16" else slab
      for screen in (0, 2):
          ix.set_screen(screen, eps_mult if screen else 1.0)        # 2 = certified stage 1 whenever the shapes allow (bf16: over the slab itself)
          ix.set_wgs_per_cu(int(rng.choice([1, 2, 4])))
          ids, sc = ix.retrieve_fields(q, k, sentinel)
          for f in range(F):
              if dtype == "bf16":
                  with O.chain("natural"):
                      oi, osc = O.c_retrieve(ref[f], q, k, sentinel)
                  good = np.allclose(sc[:, f], osc, rtol=0, atol=1e-4 * max(1.0, float(np.abs(osc).max())))
                  st_ = ix.screen_stats()      # (dims whose k-steps divide by neither 4 nor 6 have no certified bf16 kernel: plain pass)
                  if screen == 2 and k + 64 <= 192 and st_["n_checked"] > 0:      # certified OR repaired by the chain pass: exact ids and bits
                      bf16_exact = True
                      good = good and np.array_equal(ids[:, f], oi) and np.array_equal(sc[:, f].view(np.uint32), osc.view(np.uint32))
              else:
                  oi, osc = O.c_retrieve(ref[f], q, k, sentinel)
                  good = np.array_equal(ids[:, f], oi) and np.array_equal(sc[:, f].view(np.uint32), osc.view(np.uint32))
              if not good:
                  ok = False
                  print("MISMATCH", dict(F=F, D=D, E=E, Q=Q, k=k, sentinel=sentinel, mean=mean, dtype=dtype, kind=int(kind), screen=screen, f=f,
                                         seed=seed, n=n, eps_mult=eps_mult), flush=True)
    pipe_ok, pipe_note = False, "-"
    screens = (0, 2) if dtype == "f32" else (2,)      # (a bf16 index in mode 0 answers from the plain pass: its 1e-4, no bit contract)
    if ok and (dtype == "f32" or bf16_exact) and D * F * E * Q < 3e9 and F * k <= 4096:    # the whole scorer, both stage-1 paths
        W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
        mask = (rng.random(F) < 0.8).astype(np.float32)
        if rng.random() < 0.3:           # masks of any sign (the two-level stage 2 swaps its interval ends under a negative entry)
            mask = rng.choice(np.array([0.0, 1.0, -1.0, 0.5, 2.0], np.float32), F)
        if dtype == "f32":
            o = O.c_two_stage(slab, q, W, mask, k1=k, k2=k, sentinel=sentinel)
        else:
            with O.chain("natural"):
                o = O.c_two_stage(ref, q, W, mask, k1=k, k2=k, sentinel=sentinel)
        for screen in screens:
            ix.set_screen(screen, eps_mult if screen else 1.0)
            try:
                r = ix.search(q, W, mask, k1=k, k2=k, sentinel=sentinel)
            except Exception as e:      # fewer than k2 candidates raises like torch.topk: compare the valid prefix instead
                r = None
            pipe_ok = r is not None
            if r is not None and not (np.array_equal(r["ids"], o["ids"]) and
                                      np.array_equal(r["scores"].view(np.uint32), o["scores"].view(np.uint32))):
                ok = False
                print("MISMATCH two-stage", dict(F=F, D=D, E=E, Q=Q, k=k, sentinel=sentinel, mean=mean, kind=int(kind), screen=screen,
                                                 seed=seed, n=n, eps_mult=eps_mult), flush=True)
    if ok and pipe_ok and rng.random() < 0.6:
        ok, pipe_note = _pipeline_section(rng, ix, ref, q, W, mask, o, k, sentinel, eps_mult, O, dtype,
                               dict(F=F, D=D, E=E, Q=Q, k=k, sentinel=sentinel, mean=mean, kind=int(kind), seed=seed, n=n, eps_mult=eps_mult))
    sweep_note = "-"
    if ok and pipe_ok and rng.random() < 0.3 and D * F * E * Q < 1e9:
        ok, sweep_note = _mask_sweep_section(rng, ix, ref, q, W, k, sentinel, eps_mult, O, dtype,
                                             dict(F=F, D=D, E=E, Q=Q, k=k, sentinel=sentinel, mean=mean, kind=int(kind), seed=seed, n=n, eps_mult=eps_mult))
    shard_note = "-"
    if ok and pipe_ok and rng.random() < 0.35:
        ok, shard_note = _sharded_section(rng, idxmod, slab, q, W, mask, o, k, sentinel, eps_mult, dtype,
                                          dict(F=F, D=D, E=E, Q=Q, k=k, sentinel=sentinel, mean=mean, kind=int(kind), seed=seed, n=n, eps_mult=eps_mult))
    if ok and dtype == "f32" and k < 128 and D * F * E * Q < 2e9 and rng.random() < 0.3:      # fused mode against its contract
        Wf = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
        oi, osc = O.c_search_fused(slab, q, Wf, None, k)
        r = ix.search_fused(q, Wf, None, k)
        if not (np.array_equal(r["ids"], oi) and np.array_equal(r["scores"].view(np.uint32), osc.view(np.uint32))):
            ok = False
            print("MISMATCH fused", dict(F=F, D=D, E=E, Q=Q, k=k, seed=seed, n=n, eps_mult=eps_mult), flush=True)
    st = ix.screen_stats()
    off = ix.auto_off_info()["off"]
    t2 = ix.tier2_stats()
    ix.close()
    if verbose or not ok:
        print(f"{n + 1:4d} ok={ok} F={F} D={D} E={E} Q={Q} k={k} sent={int(sentinel)} mean={mean} {dtype} kind={int(kind)} eps_mult={eps_mult:g} upd={n_updates} "
              f"checked={st.get('n_checked')} failed={st.get('n_failed')} t2={(str(t2['lists']) + '/' + str(t2['passed_on_to_exact']) + ' scan:' + str(t2['candidates_from_the_launch_scan']) + ' rescan:' + str(t2['lists_rescanned'])) if t2 else '-'} off={off} pipe={pipe_note} sweep={sweep_note} shards={shard_note}", flush=True)
    return ok


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    from mfar.data import index as idxmod
    from oracle import mfar_oracle as O
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = 0
    while time.time() < t_end:
        ok = one_config(rng, idxmod, O, n, seed, big=len(sys.argv) > 3)
        n += 1
        if not ok:
            sys.exit(1)
    print("stress ok:", n, "configurations")


if __name__ == "__main__":
    main()
