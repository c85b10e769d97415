#!/usr/bin/env python3
"""Diagnostics: run the headline corpus once with MFAR_CERT_DEBUG=1 -> every failed certificate with its numbers on stderr."""
import os, sys
os.environ["MFAR_CERT_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
import torch
from mfar import synth
from mfar.data import index as idxmod
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
structured = len(sys.argv) > 2 and sys.argv[2] == "structured"
corpus = synth.SyntheticCorpus(D, 8, 768, n_queries=int(os.environ.get("NQ", "64")), seed=0xDEADBEEF, device="cuda:0", structured=structured)
ix = corpus.build_index(idxmod)
for b in range(2):
    r = ix.search(corpus.queries(b * 64, 64), corpus.W, None, return_fields=True)
    torch.cuda.synchronize()
print(ix.screen_stats())
