#!/usr/bin/env python3
"""Quick GPU-side diagnosis (not a test): stage-1 scores vs the oracle chain for a few shapes, and which
k-order hypothesis matches the hardware if the documented one does not."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd")); sys.path.insert(0, ROOT)
import numpy as np
from mfar.data.index import MultiFieldIndex
from oracle import mfar_oracle as O

rng = np.random.default_rng(0)
for (F, D, E, Q, k) in [(1, 64, 32, 4, 64), (1, 256, 32, 64, 100), (2, 1000, 768, 7, 100)]:
    slab = rng.standard_normal((F, D, E)).astype(np.float32)
    q = rng.standard_normal((Q, E)).astype(np.float32)
    ix = MultiFieldIndex(D, F, E)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
        assert np.array_equal(ix.read_rows(f), slab[f]), "layout roundtrip"
    ids, sc = ix.retrieve_fields(q, k, sentinel=False)
    for f in range(F):
        oi, osc = O.c_retrieve(slab[f], q, k, False)
        same_ids = np.array_equal(ids[:, f], oi)
        bit = np.array_equal(sc[:, f].view(np.uint32), osc.view(np.uint32))
        print(f"F{F} D{D} E{E} Q{Q} k{k} field{f}: ids_equal={same_ids} scores_bit_equal={bit} max|d|={np.abs(sc[:, f]-osc).max():.3e}")
        if not bit and same_ids and E == 16:
            # which order of the 16 dims reproduces the GPU bits?
            i, j = 0, 0
            target = sc[i, f, j]
            d = int(ids[i, f, j])
            a, b = q[i].astype(np.float32), slab[f, d]
            for name, perm in {"natural": list(range(16)), "doc": [0,4,1,5,2,6,3,7,8,12,9,13,10,14,11,15],
                               "swapped": [4,0,5,1,6,2,7,3,12,8,13,9,14,10,15,11]}.items():
                acc = np.float32(0)
                for p in perm:
                    acc = np.float32(np.float64(a[p]) * np.float64(b[p]) + np.float64(acc))  # fma via double (exact product)
                print("   hypothesis", name, acc == target, acc, target)
    ix.close()
print("probe done")
