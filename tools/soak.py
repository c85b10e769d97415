#!/usr/bin/env python3
"""A soak of the C-ABI pipeline at the headline shape: `--seconds` of back-to-back 64-query batches (host buffers in and out), the same 32
batches over and over.  Reports queries/s per 5-second window (clock / thermal drift) and compares EVERY result with the first pass's
result of the same batch, ids and score bits -- a race anywhere in the pipeline (slots, streams, coalescing, the adaptive policy's
feedback) would show up as a run-to-run difference.
    python tools/soak.py [--seconds 60] [--docs 1000000 --fields 8 --dim 768 --dtype f32]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "multifield-adaptive-retrieval_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--fields", type=int, default=8)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--corpus", default="plain", choices=["plain", "clustered"], help="clustered: every first certificate fails, tier 2 "
                                                                                     "(the threshold rescan) finishes every list, launch after launch")
    a = ap.parse_args()
    import numpy as np
    from mfar import synth
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    import bench
    Q, NB = 64, 32
    corpus = synth.SyntheticCorpus(a.docs, a.fields, a.dim, n_queries=Q * NB, seed=0xdeadbeef, device="cuda:0",
                                   field_kinds=(["clustered"] * a.fields if a.corpus == "clustered" else None), cluster_noise=1e-3)
    ix = corpus.build_index(idxmod, dtype=a.dtype)
    W = corpus.W.cpu().numpy()
    qs = [corpus.queries(j * Q, Q).cpu().numpy() for j in range(NB)]
    pl = NativePipeline(ix, W, np.ones(a.fields, np.float32), max_batch=Q)
    ref = [None] * NB
    tickets = []
    windows, n_cmp, n_diff = [], 0, 0
    t0 = time.perf_counter()
    w_t, w_n, j = t0, 0, 0
    while True:
        now = time.perf_counter()
        if now - w_t >= 5.0:
            windows.append(round(w_n * Q / (now - w_t)))
            print(f"{now - t0:6.1f} s  {windows[-1]} q/s  compared {n_cmp} batches, {n_diff} different", flush=True)
            w_t, w_n = now, 0
            if now - t0 >= a.seconds:
                break
        tickets.append((pl.submit(qs[j % NB]), j % NB))
        j += 1
        if len(tickets) > pl.lag:
            t, b = tickets.pop(0)
            r = pl.result(t)
            w_n += 1
            key = (r["ids"].copy(), r["scores"].view(np.uint32).copy(), r["n_valid"].copy())
            if ref[b] is None:
                ref[b] = key
            else:
                n_cmp += 1
                if not all(np.array_equal(x, y) for x, y in zip(key, ref[b])):
                    n_diff += 1
    for t, b in tickets:
        pl.result(t)
    st = ix.screen_stats()
    out = {"shape": [a.docs, a.fields, a.dim], "dtype": a.dtype, "stage1_kernel": ix.last_stage1_kernel(), "seconds": round(time.perf_counter() - t0, 1), "batches": j, "queries": j * Q,
           "queries_per_s_per_5s_window": windows, "min_over_max": round(min(windows) / max(windows), 4),
           "batches_compared_with_their_first_result": n_cmp, "different": n_diff, "launches_redone": pl.n_redone,
           "lists_checked": st.get("n_checked"), "lists_failed": st.get("n_failed"), "tier2": ix.tier2_stats(), "fields_switched_off": ix.auto_off_info()["off"],
           "corpus": a.corpus, "source_hash": bench.source_hash()}
    pl.close()
    ix.close()
    print(json.dumps(out))
    sys.exit(1 if n_diff else 0)


if __name__ == "__main__":
    main()
