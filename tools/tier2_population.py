#!/usr/bin/env python3
"""COUNT FIRST: what would a second certified tier between the fp16 screen and the exact fp32 pass buy, and which one?

For every (query, field) list of a corpus: eps = the certified screen's rigorous bound (csrc/mfar_screen.h, wide pass), e_k = the exact k-th
best DOCUMENT score, and
    P1 = distinct rows with exact score >= e_k - 2 eps     -- every row that could still reach the exact top-k after the fp16 scan.
The tier-1 certificate needs (about) P1 <= k' = 192.  Two candidate second tiers for the lists that fail it:
  (A) SPLIT PRECISION (hi + lo fp16 terms over the fp32 slab, VERDICT's proposal): same certificate with eps2 = the bound WITHOUT the
      fp16 rounding terms -- what remains is rigorous fp32 accumulation (4 K + 66) u32 |q||c| and the exact chain's own K u32 (|q||d| + |q||m|).
      It certifies a list when P2 = distinct rows with exact >= e_k - 2 eps2 is <= 192.
  (B) THRESHOLD RESCAN (same fp16 screen slab, HBM-bound on HALF the bytes of (A)): rescan the failed fields with the fixed threshold
      e_k - eps per list, gather EVERY row above it (P1 of them) from the fp32 slab, take the exact top-k.  Complete by construction; it
      works whenever P1 fits the candidate capacity (4096 here) and costs P1 x 3 KB of gathers per list.
Corpora: encoder-produced (tools/encode_bench.py: prime- and amazon-shaped texts through the BERT-base-shaped encoder) and mfar/synth.py
clustered corpora at noise 1e-2 / 1e-3 / 1e-4.  Scores by torch fp32 matmul on the GPU (a counting tool: not the chain's bits).

    python tools/tier2_population.py [--docs 50000] > profiles/r06_tier2_population.txt
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "multifield-adaptive-retrieval_amd"), ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

K, KP, CAP = 100, 192, 4096
U16, U32 = 2.0 ** -11, 2.0 ** -24


def population(rows_of_field, q, label):
    """rows_of_field(f) -> [D, E] fp32 device tensor; q [Q, E].  Returns per-field statistics."""
    import torch
    Q, E = q.shape
    qn = q.norm(dim=1) * 1.0001
    out = []
    f = 0
    while True:
        x = rows_of_field(f)
        if x is None:
            break
        m = x.mean(0)
        c = x - m
        dmax = float(c.norm(dim=1).max()) * 1.0001
        mnorm = float(m.norm()) * 1.0001
        Kf = float(E)
        c_rel1 = 2.04 * U16 + (4 * Kf + 66) * U32                 # the wide pass: one fp16 query term
        c_rel2 = (4 * Kf + 66) * U32 + 3 * 2.0 ** -22            # split precision: fp32 accumulation + the dropped lo x lo products
        e_rest = Kf * U32 * qn * (dmax + 2 * mnorm)
        eps1 = 1.25 * (c_rel1 * qn * dmax + e_rest)
        eps2 = 1.25 * (c_rel2 * qn * dmax + e_rest)
        s = q @ x.T                                               # [Q, D]
        P1, P2, gap = [], [], []
        for i in range(Q):
            si = s[i]
            pos = si[si > 0]
            if pos.numel() < K:
                continue
            ek = torch.topk(pos, K).values[-1]
            u = torch.unique(si[si >= ek - 2 * eps1[i]])          # distinct score values ~ distinct rows (identical rows score alike)
            P1.append(int(u.numel()))
            P2.append(int((u >= ek - 2 * eps2[i]).sum()))
            srt = torch.sort(torch.unique(pos), descending=True).values
            if srt.numel() > KP:
                gap.append(float(srt[K - 1] - srt[KP - 1]) / float(eps1[i]))
        import numpy as np
        P1, P2 = np.array(P1), np.array(P2)
        n = max(1, len(P1))
        out.append({"field": f, "rows": int(x.shape[0]), "lists": int(len(P1)), "eps1_mean": float(eps1.mean()), "eps2_over_eps1": float((eps2 / eps1).mean()),
                    "gap_k_to_kp_in_eps1_median": float(np.median(gap)) if gap else None,
                    "tier1_would_certify": float((P1 <= KP - 2).sum()) / n,
                    "P1_median": float(np.median(P1)) if len(P1) else None, "P1_p90": float(np.percentile(P1, 90)) if len(P1) else None,
                    "P1_max": int(P1.max()) if len(P1) else None,
                    "A_split_precision_certifies_of_tier1_failures": (float(((P1 > KP - 2) & (P2 <= KP - 2)).sum()) / max(1, int((P1 > KP - 2).sum()))),
                    "B_threshold_rescan_fits_of_tier1_failures": (float(((P1 > KP - 2) & (P1 <= CAP)).sum()) / max(1, int((P1 > KP - 2).sum()))),
                    "B_gather_rows_per_failed_list_mean": (float(P1[P1 > KP - 2].mean()) if (P1 > KP - 2).any() else 0.0),
                    "tier1_failures": int((P1 > KP - 2).sum())})
        f += 1
    fails = sum(o["tier1_failures"] for o in out)
    lists = sum(o["lists"] for o in out)
    summary = {"corpus": label, "lists": lists, "tier1_failure_share": fails / max(1, lists),
               "A_split_precision_certifies_of_failures": sum(o["A_split_precision_certifies_of_tier1_failures"] * o["tier1_failures"] for o in out) / max(1, fails),
               "B_threshold_rescan_fits_of_failures": sum(o["B_threshold_rescan_fits_of_tier1_failures"] * o["tier1_failures"] for o in out) / max(1, fails),
               "B_gather_rows_per_failed_list_mean": sum(o["B_gather_rows_per_failed_list_mean"] * o["tier1_failures"] for o in out) / max(1, fails),
               "fields_with_most_lists_failing": [o["field"] for o in out if o["tier1_failures"] > 0.5 * o["lists"]]}
    return {"summary": summary, "per_field": out}


def tier2_population(module, st):
    """encode_bench hook: the slab `on_eval_start` just wrote, the dataset's queries through the same encoder."""
    import torch
    dm = st.data_module
    dm.setup("test")
    with torch.no_grad():
        x = torch.cat([module.encode_query_batch(b) for loader in dm.test_dataloader()[:1] for b in loader]).contiguous()[:64]
    ix = module.slab
    rows = lambda f: torch.from_numpy(ix.read_rows(f)).to(x.device) if f < ix.n_fields else None
    return population(rows, x, "encoder-produced")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=50000)
    ap.add_argument("--synth-docs", type=int, default=250000)
    a = ap.parse_args()
    import torch
    import encode_bench
    res = {}
    for ds, n in (("prime", a.docs), ("amazon", a.docs)):
        r = encode_bench.run(n, 256, sweep=False, dataset=ds, modes=("bf16",), probe=False, hooks=(tier2_population,))
        res[f"encoder_{ds}_{n}x{r['fields']}"] = r["autocast_bf16"]["tier2_population"]
        print(json.dumps({f"encoder_{ds}": res[f"encoder_{ds}_{n}x{r['fields']}"]["summary"]}), flush=True)
    from mfar import synth
    for noise in (1e-2, 1e-3, 1e-4):
        cp = synth.SyntheticCorpus(a.synth_docs, 8, 768, n_queries=4096, seed=0xDEADBEEF, device="cuda:0", field_kinds=["clustered"] * 8, cluster_noise=noise)
        rows = lambda f: cp.rows(f, 0, cp.D) if f < 8 else None
        r = population(rows, cp.queries(0, 32), f"synth clustered, noise {noise:g}")
        res[f"clustered_{noise:g}"] = r
        print(json.dumps({f"clustered_{noise:g}": r["summary"]}), flush=True)
        del cp
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
