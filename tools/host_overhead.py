#!/usr/bin/env python3
"""Host-side cost of the batch pipeline: wall time per 64-query batch vs. the time the host spends inside submit()
(enqueueing ~10 ctypes calls + a few torch ops per LAUNCH; a launch scans two coalesced batches when the wide screened pass
is available) and waiting in result().  The N = 8 proxy is one shard of 125 000 rows: the host must stay well below the
shard's GPU time per batch or the ranks become launch-bound.   usage: host_overhead.py [docs] [fields]"""
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(120, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd")); sys.path.insert(0, ROOT)
import json
import torch
from mfar import synth
from mfar.data import index as idxmod
from mfar.data.pipeline import PipelinedSearcher
D = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
EXCHANGE = os.environ.get("MFAR_BENCH_FORCE_EXCHANGE") == "1"       # the multi-GPU exchange path over a one-rank RCCL group
if EXCHANGE:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
corpus = synth.SyntheticCorpus(D, F, 768, n_queries=4096, seed=1, device="cuda:0")
ix = corpus.build_index(idxmod)
for coalesce in (None, 1):
    ps = PipelinedSearcher(ix, corpus.W, torch.ones(F, device="cuda:0"), max_batch=64, coalesce=coalesce, exchange=True if EXCHANGE else None)
    qs = [corpus.queries((i % 60) * 64, 64) for i in range(260)]
    tk = []
    for i in range(20):                       # warm-up: scratch allocation, screen build
        tk.append(ps.submit(qs[i]))
        if i >= ps.lag: ps.result(tk[i - ps.lag])
    for t in tk[-ps.lag:]: ps.result(t)
    torch.cuda.synchronize()
    tk, tsub, tres, n = [], 0.0, 0.0, 240
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter(); tk.append(ps.submit(qs[20 + i])); b = time.perf_counter(); tsub += b - a
        if i >= ps.lag:
            ps.result(tk[i - ps.lag]); tres += time.perf_counter() - b
    for t in tk[-ps.lag:]: ps.result(t)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(json.dumps({"docs": D, "fields": F, "exchange_path": EXCHANGE, "coalesce": ps.coalesce, "batches": n, "wall_ms_per_batch": tot / n * 1e3,
                      "host_in_submit_ms_per_batch": tsub / n * 1e3, "host_waiting_in_result_ms_per_batch": tres / n * 1e3,
                      "queries_per_s": n * 64 / tot}))
