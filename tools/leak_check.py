import os, sys
sys.path.insert(0, "/root/repo/multifield-adaptive-retrieval_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mfar.data import index as idxmod
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
    ix.close()
    ib = idxmod.MultiFieldIndex(20000, 3, 96, device=0, dtype="bf16")     # bf16 index with the opt-in screen (fp32 staging copies)
    for f in range(3):
        ib.write_rows(f, 0, slab[f])
    ib.set_screen(2)
    ib.search(q, W, None)
    ib.close()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info(0)[0]
    if it == 5: free0 = free
print("free after 5:", free0, "after 60:", free, "delta MB:", (free0 - free) / 2**20)
