"""Run by tests/test_gpu_multirank.py in a subprocess: the lists-first exchange of the pipelined searcher over a ONE-rank
nccl (= RCCL) process group on cuda:0 -- the multi-GPU code path (two all-gathers per launch on the side stream, owned
scoring, top-k merge with the certificate flag) with RCCL really executing -- must return exactly what the plain search
returns.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29571")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from mfar.data import index as idxmod
from mfar.data.pipeline import PipelinedSearcher

rng = np.random.default_rng(5)
out = {}
for dtype, (F, D, E) in (("f32", (4, 40000, 128)), ("bf16", (3, 20000, 64))):
    slab = (rng.standard_normal((F, D, E)) * 0.5 + 0.2).astype(np.float32)
    W = torch.from_numpy((rng.standard_normal((E, F)) * 0.05).astype(np.float32)).cuda()
    ix = idxmod.MultiFieldIndex(D, F, E, device=0, dtype=dtype)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    qs = [torch.from_numpy((rng.standard_normal((64, E)) * 0.5 + 0.2).astype(np.float32)).cuda() for _ in range(7)]
    plain = [ix.search(q, W, None) for q in qs]
    ps = PipelinedSearcher(ix, W, None, max_batch=64, exchange=True)
    tickets, got = [], []
    for i, q in enumerate(qs):
        tickets.append(ps.submit(q))
        if i >= ps.lag:
            got.append({k: v.clone() for k, v in ps.result(tickets[i - ps.lag]).items()})
    for t in tickets[max(0, len(qs) - ps.lag):]:
        got.append({k: v.clone() for k, v in ps.result(t).items()})
    same = all(torch.equal(g["ids"], p["ids"]) and torch.equal(g["scores"], p["scores"]) and torch.equal(g["n_valid"], p["n_valid"])
               for g, p in zip(got, plain))
    # a sweep of field masks through the same exchange: one local top-k payload per mask in the second all-gather
    masks = torch.ones(3, F, device="cuda")
    masks[1, 0] = 0
    masks[2, 1:] = 0
    want = [[ix.search(q, W, masks[m].contiguous()) for q in qs] for m in range(3)]
    pm = PipelinedSearcher(ix, W, None, max_batch=64, exchange=True, masks=masks)
    tickets, gm = [], []
    for i, q in enumerate(qs):
        tickets.append(pm.submit(q))
        if i >= pm.lag:
            gm.append({k: v.clone() for k, v in pm.result(tickets[i - pm.lag]).items()})
    for t in tickets[max(0, len(qs) - pm.lag):]:
        gm.append({k: v.clone() for k, v in pm.result(t).items()})
    sweep_same = all(torch.equal(g["ids"][m], want[m][i]["ids"]) and torch.equal(g["scores"][m], want[m][i]["scores"])
                     for i, g in enumerate(gm) for m in range(3))
    out[dtype] = dict(same=bool(same), coalesce=ps.coalesce, n=len(got), redone=ps.n_redone, sweep_same=bool(sweep_same))
    ix.close()
dist.barrier()
dist.destroy_process_group()
print(json.dumps(out))
