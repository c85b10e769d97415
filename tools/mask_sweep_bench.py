#!/usr/bin/env python3
"""The field-masking sweep of mask_fields (2 F + 2 runs that differ in the mask only): one pipeline pass per mask, as the
reference's 2 F + 2 `trainer.test` calls do, against ONE pass with the mixer run once per mask (PipelinedSearcher(masks=...)).
python tools/mask_sweep_bench.py [--docs N --fields F --dim E --batches B]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--fields", type=int, default=8)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batches", type=int, default=32)
    a = ap.parse_args()
    import torch
    from mfar import synth
    from mfar.data import index as idxmod
    from mfar.data.pipeline import PipelinedSearcher
    dev = torch.device("cuda:0")
    corpus = synth.SyntheticCorpus(a.docs, a.fields, a.dim, n_queries=4096, seed=0xDEADBEEF, device="cuda:0")
    ix = corpus.build_index(idxmod)
    F = a.fields
    masks = torch.ones(2 * F + 2, F, device=dev)          # baseline, every field, all dense, every field name
    for f in range(F):
        masks[1 + f, f] = 0
        masks[2 + F + f, f] = 0
    masks[1 + F] = 0

    def run(ps):
        tickets = []
        for i in range(a.batches):
            tickets.append(ps.submit(corpus.queries(i * 64, 64)))
            if i >= ps.lag:
                ps.result(tickets[i - ps.lag])
        for t in tickets[max(0, a.batches - ps.lag):]:
            ps.result(t)
        torch.cuda.synchronize()

    one = PipelinedSearcher(ix, corpus.W, masks[0].contiguous(), max_batch=64)
    run(one)
    t0 = time.perf_counter()
    for m in range(masks.shape[0]):
        one.mask = masks[m].contiguous()
        run(one)
    t_seq = time.perf_counter() - t0
    sweep = PipelinedSearcher(ix, corpus.W, None, max_batch=64, masks=masks)
    run(sweep)
    t0 = time.perf_counter()
    run(sweep)
    t_sweep = time.perf_counter() - t0
    nq = a.batches * 64
    print(f"{a.docs} x {F} x {a.dim}, {masks.shape[0]} masks, {nq} queries: one pass per mask {t_seq * 1e3:.1f} ms, "
          f"one pass for all masks {t_sweep * 1e3:.1f} ms ({t_seq / t_sweep:.1f}x; {nq / t_sweep:.0f} queries/s for the whole sweep)")


if __name__ == "__main__":
    main()
