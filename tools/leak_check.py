"""Create / use / destroy in a loop: device memory (hipMemGetInfo) and host RSS must be flat after the first iterations.
    python tools/leak_check.py        (on the GPU box; ~1 minute)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, psutil
from mfar.data import index as idxmod
from mfar.data.pipeline import NativePipeline, PipelinedSearcher
rng = np.random.default_rng(0)
slab = rng.standard_normal((3, 20000, 96)).astype(np.float32)
q = rng.standard_normal((128, 96)).astype(np.float32)     # > 64 queries: the wide pass and its scratch
W = rng.standard_normal((96, 3)).astype(np.float32)
free0 = None
for it in range(60):
    ix = idxmod.MultiFieldIndex(20000, 3, 96, device=0)
    for f in range(3):
        ix.write_rows(f, 0, slab[f])
    ix.set_screen(2)
    ix.search(q, W, None)
    ix.search(q[:64], W, None)
    ix.retrieve_field(1, q, 100, True)              # single-field pass: two-level merge scratch
    ix.search_fused(q, W, None, 100)                # the fused companion index
    for cls in (NativePipeline, PipelinedSearcher):  # both pipelines: slots, events, pinned flags (the library's streams are per process)
        on_dev = cls is PipelinedSearcher or bool(it & 2)      # the C-ABI pipeline takes host or device buffers
        Wd, qd = (torch.from_numpy(W).cuda(), torch.from_numpy(q[:64]).cuda()) if on_dev else (W, q[:64])
        pl = cls(ix, Wd, None, max_batch=64)
        ts = [pl.submit(qd) for _ in range(4)]
        pl.result(ts[-1])
        if cls is NativePipeline and it % 2:
            pl.close()                               # (odd iterations: left to the index's close())
    ix.close()
    ib = idxmod.MultiFieldIndex(20000, 3, 96, device=0, dtype="bf16")     # bf16 index with the opt-in screen (fp32 staging copies)
    for f in range(3):
        ib.write_rows(f, 0, slab[f])
    ib.set_screen(2)
    ib.search(q, W, None)
    ib.close()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info(0)[0]
    rss = psutil.Process().memory_info().rss
    if it == 5: free0, rss0 = free, rss
print("device free after 5:", free0, "after 60:", free, "delta MB:", (free0 - free) / 2**20)
print("host RSS after 5: %.1f MB, after 60: %.1f MB, delta MB: %.1f" % (rss0 / 2**20, rss / 2**20, (rss - rss0) / 2**20))
assert free0 - free < (64 << 20) and rss - rss0 < (64 << 20), "leak"
