#!/usr/bin/env python3
"""VERDICT r04 item 8 (count before build): could the stage-1 scan read FEWER DIMS per row on anisotropic data?

Everything else about the headline scan is within ~15 % of a copy kernel; the only lever left on it is bytes per row.  Idea: rotate every
field into its PCA basis once (an orthogonal map: q . d is unchanged when the query is rotated too), scan only the leading E' dims, and
bound the rest by Cauchy-Schwarz with a per-row tail norm (4 B per row):   p - |q_t| |d_t|  <=  q . d  <=  p + |q_t| |d_t|,   p = q[:E'] . d[:E'].
A row must go to an exact pass when its upper bound reaches the k-th largest LOWER bound of its list.  This script counts those rows on the
host for synthetic fields with a power-law spectrum  lambda_i ~ i^-alpha  (alpha = 0: isotropic, the bench corpus; real sentence embeddings
sit around alpha = 0.5 .. 1), rows AND queries drawn from it (contriever embeds both with one model), 768 dims, k = 100.

Build criterion (VERDICT): survivors <= 2 k' = 384 per list at E' <= 384.   No GPU needed:  python tools/partial_dim_survival.py
"""
import sys
import time

import numpy as np


def run(D=200_000, E=768, Q=32, k=100, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for alpha in (0.0, 0.5, 1.0, 1.5, 2.0):
        lam = np.arange(1, E + 1, dtype=np.float64) ** (-alpha)
        lam *= E / lam.sum()                                  # same total variance for every alpha
        sd = np.sqrt(lam).astype(np.float32)
        docs = rng.standard_normal((D, E), dtype=np.float32) * sd          # already in the PCA basis (eigenvalues descending)
        qs = rng.standard_normal((Q, E), dtype=np.float32) * sd
        exact = qs @ docs.T                                    # [Q, D]
        kth = np.partition(exact, D - k, axis=1)[:, D - k]
        row = {"alpha": alpha, "variance_in_first_384": float(lam[:384].sum() / E), "variance_in_first_256": float(lam[:256].sum() / E)}
        for Ep in (512, 384, 256, 128):
            p = qs[:, :Ep] @ docs[:, :Ep].T
            qt = np.linalg.norm(qs[:, Ep:], axis=1)[:, None]
            dt = np.linalg.norm(docs[:, Ep:], axis=1)[None, :]
            slack = qt * dt
            ub, lb = p + slack, p - slack
            t_lb = np.partition(lb, D - k, axis=1)[:, D - k]   # what the scan itself can know: the k-th largest lower bound
            surv_lb = (ub >= t_lb[:, None]).sum(1)
            surv_oracle = (ub >= kth[:, None]).sum(1)          # (an oracle threshold: the exact k-th score -- a lower limit on the survivors)
            row[f"E'={Ep}"] = {"survivors_vs_kth_lower_bound": float(surv_lb.mean()), "survivors_vs_exact_kth (oracle)": float(surv_oracle.mean()),
                              "share_of_rows": float(surv_lb.mean() / D)}
        out.append(row)
    return out


if __name__ == "__main__":
    t0 = time.time()
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    res = run(D=D)
    print(f"partial-dimension screening: rows of a list that survive to an exact pass (D = {D} rows, E = 768, k = 100, 32 queries, mean per list)")
    print("criterion to build: <= 384 survivors at E' <= 384")
    for r in res:
        print(f"\nspectrum lambda_i ~ i^-{r['alpha']:.1f}   (variance in the first 384 / 256 dims: {r['variance_in_first_384']:.3f} / {r['variance_in_first_256']:.3f})")
        for key in ("E'=512", "E'=384", "E'=256", "E'=128"):
            v = r[key]
            print(f"  {key}:  {v['survivors_vs_kth_lower_bound']:10.0f} survivors ({100 * v['share_of_rows']:.2f} % of the rows);   "
                  f"with an oracle threshold (exact k-th score): {v['survivors_vs_exact_kth (oracle)']:10.0f}")
    print(f"\n({time.time() - t0:.0f} s on the host)")
