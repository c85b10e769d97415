import os, sys, time, faulthandler
faulthandler.dump_traceback_later(45, exit=True)
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd")); sys.path.insert(0, ROOT)
import torch
from mfar import synth
from mfar.data import index as idxmod
from mfar.data.pipeline import PipelinedSearcher
D=int(sys.argv[1]) if len(sys.argv)>1 else 125000
corpus = synth.SyntheticCorpus(D, 8, 768, n_queries=2048, seed=1, device="cuda:0")
ix = corpus.build_index(idxmod)
ps = PipelinedSearcher(ix, corpus.W, torch.ones(8, device="cuda:0"), max_batch=64)
qs=[corpus.queries(i*64,64) for i in range(100)]
prev=None
for i in range(10):
    t=ps.submit(qs[i]); 
    if prev is not None: ps.result(prev)
    prev=t
torch.cuda.synchronize()
# host cost of submit alone (GPU far behind -> measure enqueue time only, first 2 to avoid slot waits)
t0=time.perf_counter(); n=0
tsub=0.0; tres=0.0
prev=None
for i in range(10,90):
    a=time.perf_counter(); t=ps.submit(qs[i]); b=time.perf_counter(); tsub+=b-a
    if prev is not None:
        ps.result(prev); tres+=time.perf_counter()-b
    prev=t; n+=1
torch.cuda.synchronize()
tot=time.perf_counter()-t0
print(f"docs={D} steps={n} wall/step={tot/n*1e3:.3f} ms  submit host={tsub/n*1e3:.3f} ms  result wait={tres/n*1e3:.3f} ms")
