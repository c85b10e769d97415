import sys, os, time
sys.path.insert(0, "/root/repo/multifield-adaptive-retrieval_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mfar import synth
from mfar.data import index as idxmod
for (D, F) in ((1000000, 1), (700244, 5), (1000000, 3)):
    cp = synth.SyntheticCorpus(D, F, 768, n_queries=1024, seed=3, device="cuda:0")
    ix = cp.build_index(idxmod)
    q = cp.queries(0, 128)
    for _ in range(3): ix.retrieve_fields(q, 100, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): ix.retrieve_fields(cp.queries((i % 8) * 128, 128), 100, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    for _ in range(2): ix.retrieve_field(0, q, 100, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): ix.retrieve_field(0, cp.queries((i % 8) * 128, 128), 100, True)
    torch.cuda.synchronize(); dt1 = (time.perf_counter() - t0) / 10
    gb = D * F * 768 * 2 / 1e9
    print(f"D={D} F={F}: retrieve_fields(128 q) {dt*1e3:.2f} ms ({gb/dt/1e3:.2f} TB/s of fp16 rows), retrieve_field(0) {dt1*1e3:.2f} ms ({gb/F/dt1/1e3:.2f} TB/s)  n_failed={ix.screen_stats()['n_failed']}", flush=True)
    ix.close(); del cp
