"""Host-side count asked for by VERDICT r02 item 1: of the stage-2 candidates (union of the per-field top-100 lists) of the
synthetic bench corpus, how many survive (a) the threshold-algorithm bound UB_c = sum_f w_f (known s_cf | tau_f) against the exact
k2-th mixed score, (b) a certified +-eps approximation of every (candidate, field) score.  CPU only (numpy / torch), no GPU.
Output kept in profiles/r03_stage2_bound_survival.txt."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multifield-adaptive-retrieval_amd"))
from mfar import synth
torch.set_num_threads(8)
def run(D, F, E=768, NQ=64, k=100):
    cp = synth.SyntheticCorpus(D, F, E, n_queries=max(NQ, 64), device="cpu")
    slab = torch.stack([cp.rows(f, 0, D) for f in range(F)])  # [F,D,E]
    q = cp.queries(0, NQ)
    S = torch.einsum("qe,fde->qfd", q, slab)  # [Q,F,D]
    w = torch.softmax(q @ cp.W, dim=1)  # [Q,F]
    tot_c = 0; surv_ta = 0; surv_eps = {0.5e-2:0, 1e-2:0,2e-2:0,5e-2:0,1e-1:0,2.5e-1:0}; pairs_unknown=0; pairs=0
    sig = []
    for i in range(NQ):
        s = S[i]  # [F,D]
        top = torch.topk(s, k, dim=1)
        tau = top.values[:, -1].clamp_min(0)  # [F]
        cand = torch.unique(top.indices.reshape(-1))
        x = s[:, cand]  # [F,C]
        known = torch.zeros_like(x, dtype=torch.bool)
        for f in range(F):
            known[f] = torch.isin(cand, top.indices[f][top.values[f] > 0])
        mixed = (w[i][:, None] * x).sum(0)
        T = torch.topk(mixed, k).values[-1]
        ub = (w[i][:, None] * torch.where(known, x, tau[:, None].expand_as(x))).sum(0)
        tot_c += cand.numel(); surv_ta += int((ub >= T).sum())
        pairs += x.numel(); pairs_unknown += int((~known).sum())
        sd = float(s.std())
        sig.append(sd)
        for e in surv_eps:
            eps = e * sd
            lb = mixed - eps; ubb = mixed + eps   # mixed eps = sum w eps = eps
            Tlb = torch.topk(lb, k).values[-1]
            surv_eps[e] += int((ubb >= Tlb).sum())
    print(f"D={D} F={F}: cands/query={tot_c/NQ:.0f}  TA-bound survivors={surv_ta/NQ:.0f} ({surv_ta/tot_c:.3f})  unknown pairs frac={pairs_unknown/pairs:.3f}  score sd={np.mean(sig):.4f}")
    for e,v in surv_eps.items():
        print(f"   certified-approx eps={e:g} sd: survivors/query={v/NQ:.0f}")
run(100000, 8)
run(40000, 22)
