#!/usr/bin/env python3
"""Where index construction spends its time: python tools/build_timing.py [--docs N --fields F --dtype f32|bf16]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--fields", type=int, default=8)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    import torch
    from mfar import synth
    from mfar.data import index as idxmod
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cp = synth.SyntheticCorpus(a.docs, a.fields, a.dim, n_queries=4096, seed=0xDEADBEEF, device="cuda:0")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ix = cp.build_index(idxmod, dtype=a.dtype)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    q = cp.queries(0, 128)
    ix.max_split_batch(100)          # builds the screen / tables / gather slab synchronously
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ix.search(q, cp.W, None)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    ix.search(q, cp.W, None)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    # rebuild after a row update (what training-time validation pays per weight version)
    ix.write_rows(0, 0, cp.rows(0, 0, 64))
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    ix.max_split_batch(100)
    torch.cuda.synchronize()
    t7 = time.perf_counter()
    print(f"docs={a.docs} fields={a.fields} dtype={a.dtype}: corpus init {t1 - t0:.3f} s, rows generated + written {t2 - t1:.3f} s, "
          f"certified-stage-1 build {t3 - t2:.3f} s, first search {t4 - t3:.3f} s, second search {t5 - t4:.4f} s, REBUILD after a row update {t7 - t6:.3f} s; "
          f"resident {ix.resident_bytes()}")


if __name__ == "__main__":
    main()
