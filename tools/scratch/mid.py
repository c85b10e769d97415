import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
import torch
from mfar import synth
from mfar.data import index as idxmod
def t(fn, n=6):
    fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
D, F, E = 1_000_000, 8, 768
corpus = synth.SyntheticCorpus(D, F, E, n_queries=1024, seed=7, device="cuda:0")
ix = corpus.build_index(idxmod)
q, W = corpus.queries(0, 128), corpus.W
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
for U in (0, 150, 300, 1000, 2000, 5000, 8000, 20000, 50000, 150000):
    if U:
        tab = corpus.rows(3, 0, U)
        for r0 in range(0, D, 100000):
            sel = torch.randint(0, U, (100000,), generator=g, device="cuda:0")
            ix.write_rows(3, r0, tab[sel].contiguous())
    ix.set_timing(True)
    ms1 = t(lambda: ix.retrieve_fields(q, 100, True))
    k, n = ix.stage1_timing()
    st = ix.screen_stats()
    print(f"field 3 with {U or 'all'} distinct rows: stage 1 {ms1:.3f} ms, main kernel {k / max(n, 1):.3f} ms, failed {st['n_failed']}, unique[3]={st['unique_rows'][3]}", flush=True)
