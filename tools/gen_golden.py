#!/usr/bin/env python3
"""Generate golden vectors by DRIVING THE REAL REFERENCE CODE in the authoring container.

This script only runs where /root/reference exists (never on the GPU box).  It puts the
reference on sys.path, installs import-time-only stubs for third-party packages that are not
installed here (bm25s, more_itertools, sentence_transformers, mashumaro, pytorch_lightning, fire)
and then calls the reference's own, unmodified functions:

  * mfar.data.index.DenseFlatIndex.retrieve_batch / score_batch   (index.py:181-232)
  * mfar.modeling.weighting.LinearWeights.forward                 (weighting.py:17-29)
  * mfar.modeling.contrastive.RetrievalTrainingModule.trec_eval_step (contrastive.py:669-704),
    called unbound with a SimpleNamespace `self`
  * mfar.data.schema.resolve_fields                               (schema.py:96-134)
  * mfar.data.trec.QRes.__str__/from_str                          (trec.py:35-59)
  * mfar.data.format.format_documents                             (format.py:7-61)
  * mfar.data.util.MemoryMapDict                                  (data/util.py:28-59)
  * inspect.signature of mfar.commands.{train,mask_fields}.main   (train.py:25-65, mask_fields.py:20-50)

Outputs: small .npz / .json fixtures under tests/golden/ (inputs + the reference's outputs).
A fixture is data only; no reference source text is written anywhere.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import inspect
import io
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Anything()

        def __getattr__(self, name):
            return _Anything()

    # bm25s
    mod("bm25s", BM25=_Anything, tokenize=lambda *a, **k: None)
    mod("Stemmer", Stemmer=_Anything)

    # more_itertools.chunked (needs to really work)
    def chunked(it, n):
        buf = []
        for x in it:
            buf.append(x)
            if len(buf) == n:
                yield buf
                buf = []
        if buf:
            yield buf

    mod("more_itertools", chunked=chunked)

    # sentence_transformers (+ models)
    st = mod("sentence_transformers", SentenceTransformer=_Anything)
    st.models = mod("sentence_transformers.models", Normalize=_Anything, Pooling=_Anything, Transformer=_Anything)

    # mashumaro
    class DataClassJSONMixin:
        def to_json(self):
            return json.dumps(self.__dict__)

    mod("mashumaro")
    mod("mashumaro.mixins")
    mod("mashumaro.mixins.json", DataClassJSONMixin=DataClassJSONMixin)

    # pytorch_lightning
    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

    class LightningDataModule:
        pass

    pl = mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=LightningDataModule,
             Trainer=_Anything, seed_everything=lambda *a, **k: None)
    pl.loggers = mod("pytorch_lightning.loggers", MLFlowLogger=_Anything, WandbLogger=_Anything, Logger=_Anything)
    pl.strategies = mod("pytorch_lightning.strategies", DDPStrategy=_Anything)
    pl.callbacks = mod("pytorch_lightning.callbacks", EarlyStopping=_Anything, ModelCheckpoint=_Anything,
                       LearningRateMonitor=_Anything)
    mod("pytorch_lightning.callbacks.early_stopping", EarlyStopping=_Anything)
    mod("pytorch_lightning.callbacks.model_checkpoint", ModelCheckpoint=_Anything)
    mod("fire", Fire=lambda *a, **k: None)
    mod("stark_qa", load_skb=_Anything, load_qa=_Anything)


# ----------------------------------------------------------------------------- fake encoder
class LookupEncoder:
    """Zero-cost stand-in for the SentenceTransformer: text 'q<i>' -> row i of a fixed matrix.
    The reference only needs .encode(texts, convert_to_tensor=True), __call__(features) and
    get_max_seq_length() (index.py:187,228; contrastive.py:688-693)."""

    def __init__(self, table: np.ndarray):
        self.table = torch.from_numpy(np.ascontiguousarray(table))

    def _idx(self, text):
        return int(text[1:])

    def encode(self, texts, convert_to_tensor=False, convert_to_numpy=False, batch_size=64, **kw):
        out = torch.stack([self.table[self._idx(t)] for t in texts])
        if convert_to_numpy and not convert_to_tensor:
            return out.numpy()
        return out

    def get_max_seq_length(self):
        return 512

    def get_sentence_embedding_dimension(self):
        return self.table.shape[1]

    def __call__(self, feats):
        # input_ids[0][0] carries the query index
        i = int(feats["input_ids"][0][0])
        return {"sentence_embedding": self.table[i].unsqueeze(0)}


def canon(ids, scores):
    """(score desc, id asc) ordering of one result list."""
    ids = np.asarray(ids, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    order = np.lexsort((ids, -scores.astype(np.float64)))
    return ids[order], scores[order]


def gauss(rng, shape, mean=0.0, std=1.0):
    return (mean + std * rng.standard_normal(shape)).astype(np.float32)


# ----------------------------------------------------------------------------- generators
def gen_retrieve(out, DenseFlatIndex):
    cases = {}
    specs = [
        # name, D, E, Q, k, chunk, mean (doc mean shifts scores positive/negative)
        ("g1_basic", 2500, 32, 8, 100, 1048576, 0.3),
        ("g1_chunked", 2500, 32, 8, 100, 256, 0.3),          # chunk-merge path index.py:194-212
        ("g1_negative", 1200, 32, 6, 100, 1048576, -0.6),    # mostly negative scores -> zero-sentinel padding
        ("g1_e768", 200, 768, 4, 100, 128, 0.05),
        ("g1_smallD", 60, 64, 3, 100, 1048576, 0.5),          # D < k: padded with (row 0, 0.0)
        ("g1_k10", 700, 64, 5, 10, 300, 0.2),
    ]
    for name, D, E, Q, k, chunk, mean in specs:
        rng = np.random.default_rng(sum(map(ord, name.replace("chunked", "basic"))))
        mu = gauss(rng, (E,), 0.0, 1.0)
        mu /= np.linalg.norm(mu)
        V = gauss(rng, (D, E)) * 0.5 + mean * mu * 4.0
        q = gauss(rng, (Q, E)) * 0.5 + mu * 2.0
        V = V.astype(np.float32)
        q = q.astype(np.float32)
        keys = [str(i) for i in range(D)]
        idx = DenseFlatIndex(None, V, keys, {k_: i for i, k_ in enumerate(keys)}, vector_batch_size=chunk)
        res = idx.retrieve_batch(q, top_k=k)
        ids = np.array([[int(key) for key, _ in r] for r in res], dtype=np.int64)
        sc = np.array([[s for _, s in r] for r in res], dtype=np.float32)
        cases[name] = dict(V=V, q=q, k=np.int64(k), chunk=np.int64(chunk), ids=ids, scores=sc)
    np.savez_compressed(os.path.join(out, "retrieve_batch.npz"),
                        **{f"{n}__{k}": v for n, c in cases.items() for k, v in c.items()})
    return list(cases)


def f16_exact(a):
    """Round to fp16-representable values: the fixture stores them as float16 (half the bytes), fp32 inputs are exact."""
    return np.asarray(a, dtype=np.float32).astype(np.float16)


def gen_retrieve_large(out, DenseFlatIndex):
    """SURVEY 8(c) G1 at its full size: D = 5000 x E = 768 through the chunk-merge path (vector_batch_size = 256), plus a
    TIE-HEAVY case on a coarse value grid (every product and every partial sum is exact in fp32, so ANY summation order
    -- MKL's included -- gives the same bits: scores must match the reference bit for bit, ids up to the reference's
    unspecified order inside runs of equal scores)."""
    cases = {}
    rng = np.random.default_rng(5000768)
    D, E, Q, k = 5000, 768, 8, 100
    mu = gauss(rng, (E,))
    mu /= np.linalg.norm(mu)
    V16 = f16_exact(gauss(rng, (D, E)) * 0.5 + 0.05 * mu * 4.0)
    q16 = f16_exact(gauss(rng, (Q, E)) * 0.5 + mu * 2.0)
    V, q = V16.astype(np.float32), q16.astype(np.float32)
    keys = [str(i) for i in range(D)]
    idx = DenseFlatIndex(None, V, keys, {k_: i for i, k_ in enumerate(keys)}, vector_batch_size=256)
    res = idx.retrieve_batch(q, top_k=k)
    cases["g1_5000x768_chunk256"] = dict(V16=V16, q16=q16, k=np.int64(k), chunk=np.int64(256),
                                         ids=np.array([[int(key) for key, _ in r] for r in res], dtype=np.int64),
                                         scores=np.array([[s_ for _, s_ in r] for r in res], dtype=np.float32))
    # tie-heavy: values on the grid j/8, |j| <= 24, E = 64: products are multiples of 1/64 below 9, sums stay exact
    rng = np.random.default_rng(777)
    D, E, Q = 3000, 64, 6
    Vg = rng.integers(-24, 25, size=(D, E)).astype(np.int8)
    dup = rng.choice(D, size=900, replace=False)
    Vg[dup[:500]] = Vg[dup[0]]                       # one block of 500 identical rows ("empty field", format.py:58-59)
    Vg[dup[500:]] = Vg[dup[500 + (np.arange(400) % 7)]]   # seven smaller groups
    qg = rng.integers(-24, 25, size=(Q, E)).astype(np.int8)
    qg[0] = Vg[dup[0]]                               # query 0 scores the big group highest
    V, q = Vg.astype(np.float32) / 8.0, qg.astype(np.float32) / 8.0
    keys = [str(i) for i in range(D)]
    idx = DenseFlatIndex(None, V, keys, {k_: i for i, k_ in enumerate(keys)}, vector_batch_size=1000)
    res = idx.retrieve_batch(q, top_k=100)
    ids = np.array([[int(key) for key, _ in r] for r in res], dtype=np.int64)
    sc = np.array([[s_ for _, s_ in r] for r in res], dtype=np.float32)
    for i in range(Q):
        ids[i], sc[i] = canon(ids[i], sc[i])         # canonical inside the set the reference happened to return
    cases["g1_ties_grid"] = dict(Vg=Vg, qg=qg, k=np.int64(100), chunk=np.int64(1000), ids=ids, scores=sc)
    np.savez_compressed(os.path.join(out, "retrieve_batch_large.npz"),
                        **{f"{n}__{k_}": v for n, c in cases.items() for k_, v in c.items()})
    return list(cases)


def gen_score_batch(out, DenseFlatIndex):
    rng = np.random.default_rng(1234)
    D, E, Q = 900, 64, 3
    V = gauss(rng, (D, E))
    qtab = gauss(rng, (Q, E))
    keys = [f"doc{i}" for i in range(D)]
    enc = LookupEncoder(qtab)
    idx = DenseFlatIndex(enc, V, keys, {k: i for i, k in enumerate(keys)})
    cand = rng.choice(D, size=257, replace=False).astype(np.int64)
    outs = []
    for qi in range(Q):
        s = idx.score_batch([f"q{qi}"], [keys[c] for c in cand])
        assert isinstance(s, torch.Tensor) and tuple(s.shape) == (1, len(cand))
        outs.append(s.numpy()[0])
    # unknown key -> KeyError (index.py:229)
    try:
        idx.score_batch(["q0"], ["nope"])
        raised = False
    except KeyError:
        raised = True
    np.savez_compressed(os.path.join(out, "score_batch.npz"), V=V, q=qtab, cand=cand,
                        scores=np.stack(outs).astype(np.float32), unknown_key_raises=np.bool_(raised))


def gen_linear_weights(out, LinearWeights):
    rng = np.random.default_rng(77)
    E, F = 64, 5
    W = gauss(rng, (E, F), std=0.2)
    d = {}
    # eval shape: x [C,F] 2-D, q [1,E]  (contrastive.py:694)
    x2 = gauss(rng, (37, F), std=3.0)
    q1 = gauss(rng, (1, E))
    lw = LinearWeights(E, F, query_cond=True)
    assert torch.equal(lw.weight.data, torch.ones(E, F))  # init ones (weighting.py:14)
    lw.weight.data = torch.from_numpy(W.copy())
    with torch.no_grad():
        d["eval_out"] = lw(torch.from_numpy(x2), torch.from_numpy(q1)).numpy()
    # train shape: x [B,S,F], q [B,E]
    x3 = gauss(rng, (4, 9, F), std=3.0)
    q4 = gauss(rng, (4, E))
    with torch.no_grad():
        d["train_out"] = lw(torch.from_numpy(x3), torch.from_numpy(q4)).numpy()
    # not query-conditioned: LinearWeights(num_fields, 1) (contrastive.py:286-287) -> weight [F,1]
    lw2 = LinearWeights(F, 1, query_cond=False)
    w2 = gauss(rng, (F, 1))
    lw2.weight.data = torch.from_numpy(w2.copy())
    with torch.no_grad():
        d["nocond_out"] = lw2(torch.from_numpy(x3), None).numpy()
    np.savez_compressed(os.path.join(out, "linear_weights.npz"), W=W, x2=x2, q1=q1, x3=x3, q4=q4, w2=w2, **d)


# SURVEY 8(c) G5 at E = 768, F in {1, 4, 8} (name, F, D, E, Q, mean, masked fields, std of W)
LARGE_TREC_SPECS = [
    ("L_f1_e768", 1, 2500, 768, 4, 0.05, [], 0.02),
    ("L_f4_e768", 4, 1000, 768, 4, 0.05, [2], 0.02),
    ("L_f8_e768", 8, 600, 768, 4, 0.05, [], 0.02),
]


def gen_trec_eval_step(out, DenseFlatIndex, LinearWeights, contrastive, FieldType, Query):
    """Drive the UNMODIFIED RetrievalTrainingModule.trec_eval_step (contrastive.py:669-704)."""
    torch.Tensor.cuda = lambda self, *a, **k: self  # reference hard-codes .cuda() (contrastive.py:685-686)
    cases, large_cases = {}, {}
    specs = [
        # name, F, D, E, Q, mean, mask (list of masked field idx), std_W
        ("t_f1", 1, 900, 32, 4, 0.3, [], 0.05),
        ("t_f4", 4, 1200, 32, 6, 0.3, [], 0.05),
        ("t_f4_mask1", 4, 1200, 32, 6, 0.3, [1], 0.05),
        ("t_f4_maskall_but_one", 4, 1200, 32, 6, 0.3, [0, 1, 3], 0.05),
        ("t_f8", 8, 700, 32, 5, 0.25, [], 0.1),
        ("t_f8_neg", 8, 700, 32, 5, -0.4, [2], 0.1),       # zero-sentinel padding flows into stage 2
        ("t_f3_e768", 3, 160, 768, 3, 0.05, [], 0.02),
        ("t_f22", 22, 260, 32, 3, 0.3, [5, 7], 0.05),
    ]
    for name, F, D, E, Q, mean, masked, stdw in specs + LARGE_TREC_SPECS:
        large = name.startswith("L_")
        rng = np.random.default_rng(sum(map(ord, name.split("_mask")[0])) + 1000)
        mu = gauss(rng, (E,))
        mu /= np.linalg.norm(mu)
        slab = (gauss(rng, (F, D, E)) * 0.5 + mean * mu * 4.0).astype(np.float32)
        # planted relevance: a few docs get alpha*q in some fields
        qtab = (gauss(rng, (Q, E)) * 0.5 + mu * 2.0).astype(np.float32)
        for qi in range(Q):
            for _ in range(3):
                d = int(rng.integers(0, D))
                for f in rng.choice(F, size=max(1, F // 2), replace=False):
                    slab[f, d] = (0.6 * qtab[qi] + 0.3 * gauss(rng, (E,))).astype(np.float32)
        W = gauss(rng, (E, F), std=stdw)
        if large:       # stored as float16 (exactly representable values): half the fixture bytes
            slab, qtab, W = (f16_exact(x).astype(np.float32) for x in (slab, qtab, W))
        keys = [str(i) for i in range(D)]
        k2n = {k: i for i, k in enumerate(keys)}
        enc = LookupEncoder(qtab)
        indices = {f"f{f:02d}_dense": DenseFlatIndex(enc, slab[f], keys, k2n) for f in range(F)}
        lw = LinearWeights(E, F, query_cond=True)
        lw.weight.data = torch.from_numpy(W.copy())
        mask = torch.ones([F, 1])
        if masked:
            mask[masked] = 0
        fake_self = SimpleNamespace(
            indices_dict=indices, mask=mask, encoder=enc,
            hybrid_contrastive_loss_fn=SimpleNamespace(mixture_of_fields_layer=lw),
        )
        instances = [Query(f"qid{qi}", f"q{qi}") for qi in range(Q)]
        input_ids = torch.arange(Q).unsqueeze(1).repeat(1, 3)
        batch = SimpleNamespace(
            instances=instances,
            query={FieldType.DENSE: {"input_ids": input_ids, "attention_mask": torch.ones_like(input_ids)}},
        )
        buf = io.StringIO()
        with torch.no_grad():
            contrastive.RetrievalTrainingModule.trec_eval_step(fake_self, batch, 0, buf)
        lines = [l for l in buf.getvalue().split("\n") if l]
        assert len(lines) == Q * 100, (name, len(lines))
        ids = np.zeros((Q, 100), dtype=np.int64)
        sc = np.zeros((Q, 100), dtype=np.float32)
        for li, line in enumerate(lines):
            qid, _it, doc, _rank, sim, _run = line.split("\t")
            qi, r = divmod(li, 100)
            assert qid == f"qid{qi}"
            ids[qi, r] = int(doc)
            sc[qi, r] = np.float32(float(sim))
        for qi in range(Q):
            ids[qi], sc[qi] = canon(ids[qi], sc[qi])
        cases[name] = dict(slab=slab, q=qtab, W=W, mask=mask.numpy()[:, 0].astype(np.float32), ids=ids, scores=sc)
        if large:
            c = cases.pop(name)
            large_cases[name] = dict(slab16=c["slab"].astype(np.float16), q16=c["q"].astype(np.float16), W16=c["W"].astype(np.float16),
                                     mask=c["mask"], ids=c["ids"], scores=c["scores"])
            continue
        base = name.split("_mask")[0]
        if base != name:  # mask variants share the base case's inputs: store them once
            assert np.array_equal(cases[base]["slab"], slab) and np.array_equal(cases[base]["W"], W)
            for k_ in ("slab", "q", "W"):
                del cases[name][k_]
            cases[name]["base"] = np.array(base)
        if name == "t_f4":
            cases[name]["first_line"] = np.array(lines[0])
    np.savez_compressed(os.path.join(out, "trec_eval_step.npz"),
                        **{f"{n}__{k}": v for n, c in cases.items() for k, v in c.items()})
    for n, c in large_cases.items():      # one file per case: each stays well below 20 MB
        np.savez_compressed(os.path.join(out, f"trec_eval_step_{n}.npz"), **c)
    return list(cases) + list(large_cases)


def gen_loss(out, LinearWeights):
    """HybridContrastiveLoss (losses.py:206-360) on tiny tensors, 1-rank gloo group (the reference only binds its
    gathered variables in the multi-GPU branch, losses.py:254-273), loss value + gradients."""
    import pickle
    import torch.distributed as dist
    from mfar.modeling.losses import HybridContrastiveLoss
    torch.Tensor.cuda = lambda self, *a, **k: self
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    res = {}
    for name, use_bn in (("plain", False), ("bn", True)):
        rng = np.random.default_rng(4242)
        B, F, E, N = 5, 3, 16, 1
        q = torch.tensor(gauss(rng, (B, E)), requires_grad=True)
        d_pos = torch.tensor(gauss(rng, (B, F, E)), requires_grad=True)
        d_neg = torch.tensor(gauss(rng, (B, F, N, E)), requires_grad=True)
        W0 = gauss(rng, (E, F), std=0.3)
        lw = LinearWeights(E, F, query_cond=True)
        lw.weight.data = torch.from_numpy(W0.copy())
        loss_fn = HybridContrastiveLoss(temperature=0.05, mixture_of_fields_layer=lw, sparse_indices_dict={}, num_fields=F,
                                        use_batchnorm=use_bn)
        loss_fn.train()
        dumps = lambda x: pickle.dumps(x)
        loss = loss_fn(q, dumps([f"q{i}" for i in range(B)]), d_pos, dumps([f"p{i}" for i in range(B)]), d_neg,
                       dumps([f"n{i}" for i in range(B)]), dumps(list(range(B))), {})
        loss.backward()
        res.update({f"{name}__q": q.detach().numpy(), f"{name}__d_pos": d_pos.detach().numpy(), f"{name}__d_neg": d_neg.detach().numpy(),
                    f"{name}__W": W0, f"{name}__loss": np.float32(loss.item()), f"{name}__grad_W": lw.weight.grad.numpy(),
                    f"{name}__grad_q": q.grad.numpy(), f"{name}__grad_d_pos": d_pos.grad.numpy()})
    # sparse score columns (losses.py:303-345): two fake sparse indices (only `score_batch(queries, doc_ids)` is called when the
    # `sparse_scores` cache is empty, losses.py:318-321) behind three dense fields, BatchNorm over all five columns
    class TableSparseIndex:
        def __init__(self, seed):
            self.seed = seed

        def score_batch(self, queries, doc_ids):
            # raw BM25-like non-negative scores, a pure function of (query text, doc id)
            out = torch.empty(len(queries), len(doc_ids))
            for i, qt in enumerate(queries):
                for j, d in enumerate(doc_ids):
                    h = np.random.default_rng([self.seed, sum(map(ord, qt)), sum(map(ord, d)) * 7 + len(d)]).random()
                    out[i, j] = float(np.float32(12.0 * h * h))
            return out

    rng = np.random.default_rng(777)
    B, Fd, Fs, E, N = 4, 3, 2, 16, 1
    q = torch.tensor(gauss(rng, (B, E)), requires_grad=True)
    d_pos = torch.tensor(gauss(rng, (B, Fd, E)), requires_grad=True)
    d_neg = torch.tensor(gauss(rng, (B, Fd, N, E)), requires_grad=True)
    W0 = gauss(rng, (E, Fd + Fs), std=0.3)
    lw = LinearWeights(E, Fd + Fs, query_cond=True)
    lw.weight.data = torch.from_numpy(W0.copy())
    sidx = {"s_a": TableSparseIndex(11), "s_b": TableSparseIndex(23)}
    loss_fn = HybridContrastiveLoss(temperature=0.05, mixture_of_fields_layer=lw, sparse_indices_dict=sidx, num_fields=Fd + Fs,
                                    use_batchnorm=True)
    loss_fn.train()
    qs, ps, ns = [f"q{i}" for i in range(B)], [f"p{i}" for i in range(B)], [f"n{i}" for i in range(B)]
    loss = loss_fn(q, pickle.dumps(qs), d_pos, pickle.dumps(ps), d_neg, pickle.dumps(ns), pickle.dumps(list(range(B))), {})
    loss.backward()
    sp = torch.stack([si.score_batch(qs, ps) for si in sidx.values()], dim=-1)        # the columns the loss saw (inputs of the fixture)
    sn = torch.stack([si.score_batch(qs, ns) for si in sidx.values()], dim=-1)
    res.update({"sparse__q": q.detach().numpy(), "sparse__d_pos": d_pos.detach().numpy(), "sparse__d_neg": d_neg.detach().numpy(),
                "sparse__W": W0, "sparse__sparse_pos": sp.numpy(), "sparse__sparse_neg": sn.numpy(), "sparse__sparse_rev": sp.numpy(),
                "sparse__loss": np.float32(loss.item()), "sparse__grad_W": lw.weight.grad.numpy(), "sparse__grad_q": q.grad.numpy(),
                "sparse__grad_d_pos": d_pos.grad.numpy()})
    res["temperature"] = np.float32(0.05)
    np.savez_compressed(os.path.join(out, "hybrid_loss.npz"), **res)
    dist.destroy_process_group()


def gen_schema(out, resolve_fields, FieldType):
    d = {}
    for ds in ["mag", "prime", "amazon"]:
        for spec in ["all_dense", "all_sparse", "all_dense,all_sparse", "single_dense"]:
            fi = resolve_fields(spec, ds)
            d[f"{ds}|{spec}"] = [
                dict(key=k, name=f.name, type=f.field_type.name, max_seq_length=f.max_seq_length, dataset=f.dataset)
                for k, f in fi.items()
            ]
    d["amazon|title_dense,brand_dense"] = [
        dict(key=k, name=f.name, type=f.field_type.name, max_seq_length=f.max_seq_length, dataset=f.dataset)
        for k, f in resolve_fields("title_dense,brand_dense", "data/amazon").items()
    ]
    # "." -> " " replacement (schema.py:110)
    d["prime|off-label.use_dense"] = [
        dict(key=k, name=f.name, type=f.field_type.name, max_seq_length=f.max_seq_length, dataset=f.dataset)
        for k, f in resolve_fields("off-label.use_dense,name_dense", "prime").items()
    ]
    errs = {}
    try:
        resolve_fields("all_dense", "nosuchdataset")
    except NotImplementedError as e:
        errs["bad_dataset"] = "NotImplementedError"
    try:
        resolve_fields("nosuchfield_dense", "mag")
    except ValueError as e:
        errs["bad_field"] = "ValueError"
    d["errors"] = errs
    with open(os.path.join(out, "schema.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


def gen_trec(out, trec):
    d = {}
    q = trec.QRes(query_id="q1", doc_id="d7", sim=float(np.float32(12.3456789)))
    d["qres_str"] = str(q)
    d["qres_roundtrip"] = str(trec.QRes.from_str(str(q)))
    r = trec.QRels("q1", "d7", 1.0)
    d["qrels_str"] = str(r)
    d["qrels_roundtrip"] = str(trec.QRels.from_str(str(r)))
    sample = "num_q\tall\t3\nmap\tall\t0.1234\nrecall_20\tall\t0.5000\nrunid\tall\t0\n"
    d["parse_sample_in"] = sample
    d["parse_sample_out"] = trec.parse_trec_eval_output(sample)
    # read_corpus on a small TSV
    import tempfile
    rows = ['1\t{"title": "A", "n": 3}', "2\tnot json\textra", "3"]
    with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as f:
        f.write("\n".join(rows) + "\n")
        path = f.name
    d["read_corpus_in"] = rows
    d["read_corpus_out"] = [[a, b] for a, b in trec.read_corpus(path)]
    os.unlink(path)
    with open(os.path.join(out, "trec.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


def gen_format(out, format_documents):
    docs = {
        "amazon": [
            ("a1", {"title": "Red shoe", "brand": "Acme", "feature": ["light", "durable"], "description": ["nice", "shoe"],
                    "also_buy": ["B2", "B3"], "review": [{"summary": "good", "reviewText": "I like it"}],
                    "qa": [{"question": "size?", "answer": "42"}], "price": 3.5}),
            ("a2", {"title": "Blue hat"}),
            ("a3", "plain string doc"),
        ],
        "mag": [
            ("m1", {"title": "Paper", "abstract": "We study.", "author___affiliated_with___institution": ["MIT", "CMU"],
                    "paper___cites___paper": ["P1", "P2"], "paper___has_topic___field_of_study": ["IR"]}),
            ("m2", {"title": "Other"}),
        ],
        "prime": [
            ("p1", {"name": "aspirin", "type": "drug", "source": "DB", "details": {"desc": "painkiller", "half_life": 3},
                    "interacts with": {"gene/protein": ["A", "B"]}, "side effect": {"effect/phenotype": ["nausea"]}}),
            ("p2", {"name": "x"}),
        ],
    }
    fields = {
        "amazon": ["title", "brand", "feature", "description", "also_buy", "also_view", "review", "qa"],
        "mag": ["title", "abstract", "author___affiliated_with___institution", "paper___cites___paper",
                "paper___has_topic___field_of_study"],
        "prime": ["name", "type", "source", "details", "interacts with", "side effect", "carrier"],
    }
    res = {}
    for ds, corpus in docs.items():
        for fld in fields[ds]:
            try:
                r = format_documents(corpus, fld, ds)
                res[f"{ds}|{fld}"] = [[a, b] for a, b in r]
            except Exception as e:  # record the behaviour, whatever it is
                res[f"{ds}|{fld}"] = {"raises": type(e).__name__}
    with open(os.path.join(out, "format_documents.json"), "w") as f:
        json.dump({"docs": docs, "out": res}, f, indent=1)      # key order of the documents matters: no sort_keys


def gen_format_single(out, format_documents):
    """The whole-document text of the `single_dense` / `single_sparse` field (format.py:20-22 -> format_stark,
    format.py:113-415), per dataset, on hand-made documents that walk every branch; crashes of the reference on
    documents it cannot format are recorded as such."""
    docs = {
        "amazon": [
            ("a1", {"title": "Red shoe", "brand": "Acme", "description": [" comfy ", "shoe "], "feature": ["light", "", "ASIN: B01", "durable"],
                    "review": [{"summary": "good", "reviewText": "I like it", "overall": 5}, {"summary": "meh", "reviewText": "ok"}],
                    "qa": [{"question": "size?", "answer": "42"}], "also_buy": ["B2", "B3"], "also_view": ["B9"]}),
            ("a2", {"title": "Blue hat", "also_buy": [], "also_view": []}),
            ("a3", {"title": "Plain", "description": ["", " "], "also_buy": ["X"], "also_view": [], "feature": []}),
            ("a4", {"title": "No relations at all"}),
        ],
        "mag": [
            ("m1", {"type": "paper", "title": "A Paper", "abstract": "We study things.\r\n\n",
                    "paper___cites___paper": ["P1", "P2"], "paper___has_topic___field_of_study": ["IR", "NLP"],
                    "author___affiliated_with___institution": {"Ann": ["MIT", "CMU"], "Bob": []}}),
            ("m2", {"type": "paper", "title": "Bare", "abstract": ""}),
            ("m3", {"type": "author", "title": "Not a paper"}),
        ],
        "prime": [
            ("p1", {"name": "aspirin", "type": "drug", "source": "DB",
                    "details": {"description": "painkiller", "half_life": 3, "_private": "x", "drug_id": "D1", "empty": "", "nan": float("nan")},
                    "interacts with": {"gene/protein": ["A", "B"], "drug": ["C"]}, "side effect": {"effect/phenotype": ["nausea"]}}),
            ("p2", {"name": "TP53", "type": "gene/protein", "source": "NCBI",
                    "details": {"name": "tumor protein", "alias": ["p53", "LFS1"], "interpro": {"desc": "P53 family", "id": "IPR1"},
                                "generif": [{"text": "role in cancer", "pubmed": 1}, {"text": "binds DNA", "pubmed": 2}],
                                "genomic_pos": [{"chr": "17", "start": 7}, {"chr": "X", "start": 1}], "summary": "guardian"},
                    "ppi": {"gene/protein": ["MDM2"]}}),
            ("p3", {"name": "x", "type": "disease", "source": "S"}),
            ("p4", {"type": "disease", "source": "S"}),
        ],
        "whatsthatbook": [
            ("b1", {"title": "Dune", "author": "F. Herbert", "author_url": "http://a", "description": "sand", "isbn": "123",
                    "parsed_dates": ["1965", None, "1984"], "image_link": "http://i", "num_ratings": 10, "num_reviews": 2,
                    "genres": ["sf", "classic"], "id": "77"}),
            ("b2", {"title": "Bare", "parsed_dates": None, "genres": []}),
            ("b3", {}),
        ],
    }
    res = {}
    for ds, corpus in docs.items():
        rows = []
        for doc in corpus:
            try:
                (i, text), = format_documents([doc], "single", ds)
                rows.append([i, text])
            except Exception as e:
                rows.append([doc[0], {"raises": type(e).__name__}])
        res[ds] = rows
    try:
        format_documents(docs["amazon"][:1], "single", "nosuch")
        res["bad_dataset"] = None
    except Exception as e:
        res["bad_dataset"] = type(e).__name__
    with open(os.path.join(out, "format_single.json"), "w") as f:
        json.dump({"docs": docs, "out": res}, f, indent=1)


def gen_memmap(out, MemoryMapDict):
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "title.npy")
        open(p, "w").close()
        keys = ["a", "b", "c"]
        m = MemoryMapDict(p, keys, (3, 4))
        m["b"] = np.array([1, 2, 3, 4], dtype=np.float32)
        m["c"] = np.array([5, 6, 7, 8], dtype=np.float32)
        m.close()
        raw = open(p, "rb").read()
        d = dict(size=len(raw), raw_hex=raw.hex(), len=len(m), contains_b=("b" in m), contains_z=("z" in m),
                 iter=list(iter(m)), b=m["b"].tolist())
    with open(os.path.join(out, "memmap.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


def gen_cli(out):
    import importlib
    d = {}
    for name in ["train", "mask_fields"]:
        m = importlib.import_module(f"mfar.commands.{name}")
        sig = inspect.signature(m.main)
        params = []
        for p in sig.parameters.values():
            default = None if p.default is inspect._empty else p.default
            params.append(dict(name=p.name, kind=p.kind.name, required=(p.default is inspect._empty),
                               default=default if isinstance(default, (int, float, str, bool, type(None))) else repr(default)))
        d[name] = params
    with open(os.path.join(out, "cli_signatures.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default=None, choices=[None, "loss"], help="regenerate one fixture only")
    args = ap.parse_args()
    only = args.only
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; goldens can only be regenerated in the authoring container")
    _install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(4)

    from mfar.data.index import DenseFlatIndex
    from mfar.modeling.weighting import LinearWeights
    from mfar.data.schema import resolve_fields
    from mfar.data.typedef import FieldType, Query
    from mfar.data import trec
    from mfar.data.format import format_documents
    from mfar.data.util import MemoryMapDict
    from mfar.modeling import contrastive

    if only == "loss":          # one fixture only (the others stay byte for byte what they are)
        gen_loss(out, LinearWeights)
        return
    print("retrieve_batch:", gen_retrieve(out, DenseFlatIndex))
    print("retrieve_batch_large:", gen_retrieve_large(out, DenseFlatIndex))
    gen_score_batch(out, DenseFlatIndex)
    gen_linear_weights(out, LinearWeights)
    print("trec_eval_step:", gen_trec_eval_step(out, DenseFlatIndex, LinearWeights, contrastive, FieldType, Query))
    gen_schema(out, resolve_fields, FieldType)
    gen_trec(out, trec)
    gen_format(out, format_documents)
    gen_format_single(out, format_documents)
    gen_memmap(out, MemoryMapDict)
    try:
        gen_loss(out, LinearWeights)
    except Exception as e:
        import traceback; traceback.print_exc()
        print("loss golden failed:", repr(e))
    try:
        gen_cli(out)
    except Exception as e:
        print("cli signature dump failed:", repr(e))
    meta = dict(torch=torch.__version__, numpy=np.__version__,
                note="generated by tools/gen_golden.py from the reference snapshot at /root/reference (2025-05-09)")
    with open(os.path.join(out, "META.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(out)):
        print(f"{fn:32s} {os.path.getsize(os.path.join(out, fn)):>10d} B")


if __name__ == "__main__":
    main()
