"""Run by tests/test_gpu_multirank.py under torch.distributed.run with 2 ranks (gloo, both on cuda:0): a sweep of field masks
through the row-sharded lists-first exchange (PipelinedSearcher(masks=...): one local top-k payload per mask and rank in the
second all-gather) must return, for every mask, what a search of the unsharded corpus with that mask returns.  Rank 0
prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from mfar import synth
from mfar.data import index as idxmod
from mfar.data.pipeline import PipelinedSearcher

D, F, E = 50_000, 4, 128
corpus = synth.SyntheticCorpus(D, F, E, n_queries=512, seed=11, device="cuda:0")
r0, r1 = D * rank // world, D * (rank + 1) // world
ix = corpus.build_index(idxmod, row0=r0, n=r1 - r0)
masks = torch.ones(4, F, device="cuda")
masks[1, 0] = 0
masks[2, 2:] = 0
masks[3] = 0
ps = PipelinedSearcher(ix, corpus.W, None, max_batch=64, masks=masks)
assert ps.sharded and ps.M == 4
qs = [corpus.queries(i * 64, 64) for i in range(5)]
tickets, got = [], []
for i, q in enumerate(qs):
    tickets.append(ps.submit(q))
    if i >= ps.lag:
        got.append({k: v.clone() for k, v in ps.result(tickets[i - ps.lag]).items()})
for t in tickets[max(0, len(qs) - ps.lag):]:
    got.append({k: v.clone() for k, v in ps.result(t).items()})
torch.cuda.synchronize()
ok = True
if rank == 0:
    full = corpus.build_index(idxmod)
    for i, q in enumerate(qs):
        for m in range(4):
            w = full.search(q, corpus.W, masks[m].contiguous())
            ok = ok and torch.equal(got[i]["ids"][m], w["ids"]) and torch.equal(got[i]["scores"][m], w["scores"])
    full.close()
dist.barrier()
ix.close()
if rank == 0:
    print(json.dumps(dict(same=bool(ok), n=len(got), world=world, redone=ps.n_redone)))
dist.destroy_process_group()
