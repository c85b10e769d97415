"""Synthetic STaRK-shaped corpora for bench.py and the full-size GPU tests (no dataset or checkpoint can be
downloaded on either box).  Shapes follow the reference's datasets -- F = 5 / 22 / 8 dense fields for mag / prime /
amazon (reference mfar/data/schema.py:11-53), 768-d contriever embeddings (mfar/commands/train.py:36), query batches
of 64 (train.py:45), seed 0xdeadbeef (train.py:50) -- and its data quirks:

  * un-normalised embeddings with a shared mean, so most query.doc dot products are POSITIVE (this matters for the
    zero-initialised running top-k of the reference, mfar/data/index.py:192-193);
  * per field, a fraction of documents lack the field and are all encoded from the same empty string
    (mfar/data/format.py:58-59) -> blocks of identical rows -> exact score ties;
  * planted relevance: every query has 1-5 relevant documents whose vectors, in a random subset of fields, are
    pulled towards the query -> synthetic qrels, so Recall@20 is meaningful and the field weights matter;
  * W = 0.05 * N(0,1) [E, F]: a non-trivial query-conditioned gate (mfar/modeling/weighting.py:10-15 starts at ones).

`structured=True` gives three of every eight fields the duplicate / norm structure real STaRK fields have (the reference's
field presets, mfar/data/schema.py:11-53: `brand`, `type`, `source` ... are low-cardinality; a text that occurs more than once
is encoded to bit-identical rows):
  * "zipf"    every row draws one of D/4 distinct texts from a Zipf-like law (P(t) ~ 1/t): a few huge groups of identical
              rows, a long tail of singletons;
  * "lowcard" ten distinct texts in the whole field (STaRK-prime `type`);
  * "heavy"   distinct rows whose spread around the mean is Pareto(3)-scaled: a heavy tail of row norms (what a per-field
              error bound of the certified screen has to survive).

`corpus="clustered"` (or an explicit `field_kinds=[...]` with "clustered" entries) is the HOSTILE case of the certified screen:
  * "clustered"  D rows in D / 256 clusters (~235 non-empty members each): a row is its cluster's centre (a distinct text vector, as
              above) plus noise `cluster_noise` (default 1e-4) of the field's spread -- product variants, templated texts, one text
              encoded in different batches.
              The rows are NOT bit-identical (the unique-row build keeps every one), but at fp16 resolution the members of a cluster
              score alike: the best cluster fills the k' = 192 approximate candidates of nearly every list, the k-th .. k'-th gap is far inside the error
              bound and the certificate fails list after list (include/mfar_hip.h "AUTO-OFF and inline repair").

Rows are generated on the GPU in aligned chunks whose random stream depends only on (seed, field, chunk id), so any
row-sharding of the corpus produces bit-identical vectors.
"""
import numpy as np
import torch

CHUNK = 32768  # rows per generation chunk (aligned to global row numbers)


class SyntheticCorpus:
    def __init__(self, n_docs: int, n_fields: int, dim: int, n_queries: int = 4096, seed: int = 0xDEADBEEF,
                 device: str = "cuda:0", empty_frac: float = 0.08, sigma: float = 0.04, pull: float = 0.8,
                 structured: bool = False, field_kinds=None, cluster_noise: float = 1e-4, mu_scale: float = 1.0):
        self.D, self.F, self.E, self.NQ = int(n_docs), int(n_fields), int(dim), int(n_queries)
        self.seed, self.device = int(seed), torch.device(device)
        self.empty_frac, self.sigma, self.pull = float(empty_frac), float(sigma), float(pull)
        self.cluster_noise = float(cluster_noise)       # "clustered" fields: spread of a cluster's members, relative to the field's spread
        g = torch.Generator(device="cpu")
        g.manual_seed(self.seed)
        mu = torch.randn(self.E, generator=g)
        # mu_scale > 1: a NARROW CONE -- every row and query = a large common component + the same spread (cosine between two rows of a
        # field = s^2 / (s^2 + sigma^2 E): 0.45 at s = 1, 0.86 at s = 2.7 = what the BERT-base-shaped encoder's mean-pooled outputs measure,
        # tools/encode_bench.py `vector_geometry`).  Scores then carry a large common offset: the certificate's fp32-accumulation terms grow
        # with it while the spread that separates the k-th from the k'-th row does not.
        self.mu_scale = float(mu_scale)
        self.mu = (mu / mu.norm() * self.mu_scale).to(self.device)
        self.q_all = (self.mu.cpu() + self.sigma * torch.randn(self.NQ, self.E, generator=g)).to(self.device).contiguous()
        self.W = (0.05 * torch.randn(self.E, self.F, generator=g)).to(self.device).contiguous()
        self.empty_vec = (self.mu.cpu() * 0.6 + 0.02 * torch.randn(self.F, self.E, generator=g)).to(self.device)
        kinds = ["plain", "zipf", "plain", "lowcard", "plain", "heavy", "plain", "plain"]
        self.field_kinds = [kinds[f % 8] if structured else "plain" for f in range(self.F)]
        if field_kinds is not None:      # explicit kinds, one per field ("plain" | "zipf" | "lowcard" | "heavy" | "clustered")
            if len(field_kinds) != self.F or any(k not in ("plain", "zipf", "lowcard", "heavy", "clustered") for k in field_kinds):
                raise ValueError("field_kinds: one of plain / zipf / lowcard / heavy / clustered per field")
            self.field_kinds = list(field_kinds)
        if any(k in ("zipf", "lowcard", "clustered") for k in self.field_kinds):
            # text tables: the vector of text t is mu + sigma * (A[t % 4096] + B[t // 4096]) / sqrt(2)
            g2 = torch.Generator(device="cpu")
            g2.manual_seed(self.seed ^ 0x5EED)
            self._tab_a = torch.randn(4096, self.E, generator=g2).to(self.device)
            self._tab_b = torch.randn(4096, self.E, generator=g2).to(self.device)
        # planted relevance: (query, doc, field-subset)
        rng = np.random.default_rng(self.seed & 0x7FFFFFFF)
        n_rel = rng.integers(1, 6, size=self.NQ)
        self._rel = []
        rows, fields, qidx = [], [], []
        if int(n_rel.sum()) > self.D // 2:
            raise ValueError(f"{self.NQ} queries need {int(n_rel.sum())} distinct planted documents: too many for {self.D} docs")
        used = set()
        for qi in range(self.NQ):
            docs = set()
            for _ in range(int(n_rel[qi])):
                d = int(rng.integers(0, self.D))
                while d in used:
                    d = int(rng.integers(0, self.D))
                used.add(d)
                docs.add(d)
                for f in rng.choice(self.F, size=max(1, self.F // 2), replace=False):
                    rows.append(d)
                    fields.append(int(f))
                    qidx.append(qi)
            self._rel.append(docs)
        self._p_rows = np.asarray(rows, dtype=np.int64)
        self._p_fields = np.asarray(fields, dtype=np.int64)
        self._p_q = np.asarray(qidx, dtype=np.int64)

    # ------------------------------------------------------------------ data
    def queries(self, i0: int, n: int) -> torch.Tensor:
        i0 %= self.NQ
        if i0 + n <= self.NQ:
            return self.q_all[i0:i0 + n]       # a view: no copy, no host synchronisation
        idx = (torch.arange(i0, i0 + n) % self.NQ).to(self.device)
        return self.q_all.index_select(0, idx).contiguous()

    def qrels(self, i0: int, n: int):
        return [self._rel[(i0 + j) % self.NQ] for j in range(n)]

    def _chunk(self, f: int, cid: int) -> torch.Tensor:
        """All CHUNK rows of field f, chunk cid (global rows cid*CHUNK ...), fp32 [CHUNK, E] on the device."""
        g = torch.Generator(device=self.device)
        g.manual_seed((self.seed * 1000003 + f * 7919 + cid * 104729) & 0x7FFFFFFFFFFFFFFF)
        kind = self.field_kinds[f]
        if kind == "plain":
            x = torch.randn(CHUNK, self.E, generator=g, device=self.device) * self.sigma + self.mu
        elif kind == "heavy":
            u = torch.rand(CHUNK, 1, generator=g, device=self.device)
            scale = (1.0 - u).clamp_min(1e-6).pow(-1.0 / 3.0).clamp_max(30.0)            # Pareto(3) >= 1
            x = torch.randn(CHUNK, self.E, generator=g, device=self.device) * (self.sigma * scale) + self.mu
        elif kind == "clustered":
            n_cl = max(4, self.D // 256)
            t = torch.randint(0, n_cl, (CHUNK,), generator=g, device=self.device) + 104729 * (f + 1)      # other fields, other clusters
            x = (self._tab_a[t % 4096] + self._tab_b[(t // 4096) % 4096]) * (self.sigma * 0.70710678) + self.mu
            x = x + torch.randn(CHUNK, self.E, generator=g, device=self.device) * (self.sigma * self.cluster_noise)
        else:
            n_texts = 10 if kind == "lowcard" else max(16, self.D // 4)
            u = torch.rand(CHUNK, generator=g, device=self.device)
            t = (torch.exp(u * float(np.log(n_texts))).long() - 1).clamp_(0, n_texts - 1)    # P(t) ~ 1 / (t + 1)
            t = t + 7919 * (f + 1)                                                            # other fields, other texts
            x = (self._tab_a[t % 4096] + self._tab_b[(t // 4096) % 4096]) * (self.sigma * 0.70710678) + self.mu
        if self.empty_frac > 0:
            m = torch.rand(CHUNK, generator=g, device=self.device) < self.empty_frac
            x[m] = self.empty_vec[f]
        r0 = cid * CHUNK
        sel = np.nonzero((self._p_fields == f) & (self._p_rows >= r0) & (self._p_rows < r0 + CHUNK))[0]
        if sel.size:
            rr = torch.from_numpy(self._p_rows[sel] - r0).to(self.device)
            qq = self.q_all.index_select(0, torch.from_numpy(self._p_q[sel]).to(self.device))
            noise = torch.randn(sel.size, self.E, generator=g, device=self.device) * (self.sigma * 0.5)
            x[rr] = self.mu + self.pull * (qq - self.mu) + noise
        return x

    def rows(self, f: int, row0: int, n: int) -> torch.Tensor:
        """Rows [row0, row0+n) of field f (device tensor [n, E])."""
        out = torch.empty(n, self.E, device=self.device)
        r = row0
        while r < row0 + n:
            cid = r // CHUNK
            lo = r - cid * CHUNK
            m = min(CHUNK - lo, row0 + n - r)
            out[r - row0:r - row0 + m] = self._chunk(f, cid)[lo:lo + m]
            r += m
        return out

    def build_index(self, idxmod, row0: int = 0, n: int = None, dtype: str = "f32"):
        """A MultiFieldIndex holding rows [row0, row0+n) of every field (row_offset = row0)."""
        n = self.D - row0 if n is None else n
        ix = idxmod.MultiFieldIndex(n, self.F, self.E, device=self.device.index or 0, row_offset=row0, dtype=dtype)
        for f in range(self.F):
            r = row0
            while r < row0 + n:
                cid = r // CHUNK
                lo = r - cid * CHUNK
                m = min(CHUNK - lo, row0 + n - r)
                ix.write_rows(f, r - row0, self._chunk(f, cid)[lo:lo + m].contiguous())
                r += m
        torch.cuda.synchronize(self.device)
        return ix
