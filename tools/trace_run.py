#!/usr/bin/env python3
"""Workgroup-level timeline of the pipelined scorer (experiment builds with -DMFAR_TRACE only: MFAR_LIB_PATH points at one).
    MFAR_LIB_PATH=gpurun_exp/mt/libmfar_hip.so python tools/trace_run.py --docs 1250000 --fields 16 --dtype bf16 --out gpurun_out/trace.npz"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--fields", type=int, default=8)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--out", default="gpurun_out/trace.npz")
    a = ap.parse_args()
    import torch
    from mfar import _native, synth
    from mfar.data import index as idxmod
    from mfar.data.pipeline import PipelinedSearcher
    L = _native.lib()
    L.mfar_trace_dump.restype = ctypes.c_int
    L.mfar_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
    cp = synth.SyntheticCorpus(a.docs, a.fields, a.dim, n_queries=4096, seed=0xDEADBEEF, device="cuda:0")
    ix = cp.build_index(idxmod, dtype=a.dtype)
    ps = PipelinedSearcher(ix, cp.W, torch.ones(a.fields, device="cuda:0"), max_batch=64)

    def run(n):
        tk = []
        for i in range(n):
            tk.append(ps.submit(cp.queries(i * 64, 64)))
            if i >= ps.lag:
                ps.result(tk[i - ps.lag])
        for t in tk[max(0, n - ps.lag):]:
            ps.result(t)
        torch.cuda.synchronize()
    run(8)
    rec = np.zeros(1 << 17, dtype=[("kind", "i4"), ("blk", "i4"), ("where", "i4"), ("units", "i4"), ("t0", "u8"), ("t1", "u8")])
    L.mfar_trace_dump(rec.ctypes.data, len(rec))          # discard the warm-up
    import time
    t0 = time.perf_counter()
    run(a.steps)
    dt = time.perf_counter() - t0
    n = L.mfar_trace_dump(rec.ctypes.data, len(rec))
    rec = rec[:n]
    np.savez_compressed(a.out, rec=rec)
    print(f"{n} records, {a.steps} steps in {dt * 1e3:.2f} ms = {a.steps * 64 / dt:.0f} q/s")
    # summary: per kind, per launch cluster
    tmin = rec["t0"].min()
    tick = 1e-2   # wall_clock64: 100 MHz -> us
    for kind, name in ((1, "scan"), (2, "mix"), (3, "score_rows<f32>"), (4, "score_rows<f16g>"), (5, "score_rows<bf16g>")):
        r = rec[rec["kind"] == kind]
        if not len(r):
            continue
        s = (r["t0"] - tmin) * tick
        e = (r["t1"] - tmin) * tick
        o = np.argsort(s)
        s, e, u = s[o], e[o], r["units"][o]
        cuts = [0] + [i for i in range(1, len(s)) if s[i] - s[i - 1] > 1500] + [len(s)]
        print(f"-- {name}: {len(r)} records")
        for i in range(len(cuts) - 1):
            sl = slice(cuts[i], cuts[i + 1])
            d = e[sl] - s[sl]
            print(f"   group at {s[sl][0]:9.0f} us  n={cuts[i + 1] - cuts[i]:5d}  starts +[{np.percentile(s[sl] - s[sl][0], 50):7.0f} {np.percentile(s[sl] - s[sl][0], 90):7.0f} {(s[sl] - s[sl][0]).max():7.0f}]"
                  f"  ends +[{np.percentile(e[sl] - s[sl][0], 1):7.0f} {np.percentile(e[sl] - s[sl][0], 50):7.0f} {(e[sl] - s[sl][0]).max():7.0f}]  dur med {np.median(d):7.0f} max {d.max():7.0f}"
                  + (f"  units [{u[sl].min()} {int(np.median(u[sl]))} {u[sl].max()}]" if kind == 1 else ""))
    for kind, name in ((10, "merge: prefix of chunk counts"), (11, "merge: bisection + entry loads"), (12, "merge: selection"), (13, "merge: whole workgroup (units = m)")):
        r = rec[rec["kind"] == kind]
        if len(r):
            d = (r["t1"] - r["t0"]).astype(np.float64) * tick
            print(f"-- {name}: n={len(r)} dur us p10/50/90/max = {np.percentile(d, 10):.1f} {np.median(d):.1f} {np.percentile(d, 90):.1f} {d.max():.1f}  units med {int(np.median(r['units']))}")


if __name__ == "__main__":
    main()
