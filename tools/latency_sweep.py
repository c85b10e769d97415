#!/usr/bin/env python3
"""Throughput against latency of the C-ABI pipeline (mfar_pipeline_*) at the headline shape, host buffers in and out: one line per
(depth, coalesce) setting -- queries/s, and submit -> result-in-host-memory latency of a 64-query batch at full load (results taken
`lag` batches late) and for an isolated batch (nothing else in flight).
    python tools/latency_sweep.py [--docs 1000000 --fields 8 --dim 768]"""
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
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--fields", type=int, default=8)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batches", type=int, default=192)
    a = ap.parse_args()
    import numpy as np
    import torch
    from mfar import synth
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    Q = 64
    corpus = synth.SyntheticCorpus(a.docs, a.fields, a.dim, n_queries=Q * 32, seed=0xdeadbeef, device="cuda:0")
    ix = corpus.build_index(idxmod)
    W = corpus.W.cpu().numpy()
    mask = np.ones(a.fields, np.float32)
    qs = [corpus.queries((j % 32) * Q, Q).cpu().numpy() for j in range(a.batches)]
    ref = None
    for depth, coalesce in [(2, 1), (3, 1), (2, 2), (3, 2), (4, 2)]:
        pl = NativePipeline(ix, W, mask, max_batch=Q, depth=depth, coalesce=coalesce)
        lat, iso = [], []

        def run(bs):
            tk, ts, out = [], [], []
            for j, b in enumerate(bs):
                ts.append(time.perf_counter())
                tk.append(pl.submit(b))
                if j >= pl.lag:
                    out.append(pl.result(tk[j - pl.lag]))
                    lat.append(time.perf_counter() - ts[j - pl.lag])
            for j in range(max(0, len(bs) - pl.lag), len(bs)):
                out.append(pl.result(tk[j]))
            return out
        run(qs[:16])
        lat.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = run(qs)
        dt = time.perf_counter() - t0
        for b in qs[:24]:                       # isolated batches: submit, take the result at once
            t1 = time.perf_counter()
            pl.result(pl.submit(b))
            iso.append(time.perf_counter() - t1)
        ids = np.stack([g["ids"] for g in got[:32]])
        if ref is None:
            ref = ids
        same = bool((ids == ref).all())
        pl.close()
        lat_ms, iso_ms = sorted(x * 1e3 for x in lat), sorted(x * 1e3 for x in iso[4:])
        print(json.dumps({"depth": depth, "coalesce": coalesce, "batches_in_flight": depth * coalesce, "queries_per_s": round(len(qs) * Q / dt),
                          "latency_ms_full_load": {"p50": round(lat_ms[len(lat_ms) // 2], 2), "p99": round(lat_ms[int(len(lat_ms) * 0.99)], 2)},
                          "latency_ms_isolated_batch_p50": round(iso_ms[len(iso_ms) // 2], 2), "same_ids_as_first_setting": same}), flush=True)
        if not same:
            sys.exit(1)
    ix.close()


if __name__ == "__main__":
    main()
