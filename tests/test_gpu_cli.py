"""End-to-end on the GPU: `train` (2 epochs on a toy TREC dataset, random-init encoder) -> checkpoint -> `mask_fields`
sweep, checking the files the reference writes and that the evaluation equals the oracle run on the same embeddings."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_dataset(root, n_docs=300, n_q=24, seed=0, relations=False):
    rng = np.random.default_rng(seed)
    words = ["red", "blue", "shoe", "hat", "acme", "zen", "light", "heavy", "wool", "cotton", "alpha", "beta", "gamma", "delta"]
    os.makedirs(root, exist_ok=True)
    docs = []
    with open(f"{root}/corpus", "w") as f:
        for i in range(n_docs):
            body = {"title": " ".join(rng.choice(words, 3)), "brand": str(rng.choice(words)),
                    "feature": [str(w) for w in rng.choice(words, 2)]}
            if relations:                    # the whole-document ("single") rendering reads both lists (format.py:196-207)
                body["also_buy"] = [str(w) for w in rng.choice(words, 2)]
                body["also_view"] = []
            if i % 7 == 0:
                del body["brand"]            # missing field -> "" -> identical vectors -> ties
            docs.append(body)
            f.write(f"{i}\t{json.dumps(body)}\n")
    for part in ("train", "val", "test"):
        with open(f"{root}/{part}.queries", "w") as fq, open(f"{root}/{part}.qrels", "w") as fr:
            for j in range(n_q):
                qid = f"{part[0]}{j}"
                d = int(rng.integers(0, n_docs))
                fq.write(f"{qid}\t{docs[d]['title']} {docs[d].get('brand', '')}\n")
                fr.write(f"{qid}\t0\t{d}\t1\n")
    return docs


def _data_module(module, data, tmp, dataset, dev_batch_size):
    """The query side exactly as the CLIs build it (same tokenizer, same batching)."""
    from mfar.modeling.contrastive import RetrievalDataModule
    dm = RetrievalDataModule(tokenizer=module.encoder.tokenizer, queries_path=data, corpus=module.corpus, temp_path=tmp,
                             dev_partition="val", additional_partition=None, lexical_index="unused",
                             negative_sampling_params=(100, 50, 1), dataset_name=dataset, dev_batch_size=dev_batch_size,
                             field_info=module.field_info, indices_dict=module.indices_dict)
    dm.setup("test")
    return dm


def _oracle_eval(module, dm, mask=None):
    """{query id: (ids[100], scores[100])} from the C oracle, fed the rows read back from the module's slab and the query
    embeddings computed exactly as the evaluation computes them (same padded batches through the same encoder)."""
    import torch
    from oracle import mfar_oracle as O
    F = module.slab.n_fields
    slab = np.stack([module.slab.read_rows(f) for f in range(F)])
    W = module.mixture_of_fields_layer.weight.detach().cpu().numpy()
    want = {}
    module.eval()
    for batch in dm.test_dataloader()[0]:
        with torch.no_grad():
            qe = module.encode_query_batch(batch).cpu().numpy()
        o = O.c_two_stage(slab, qe, W, mask)
        for i, inst in enumerate(batch.instances):
            want[inst._id] = (o["ids"][i], o["scores"][i])
    return want


def _read_qres(path):
    got = {}
    for l in open(path).read().strip().split("\n"):
        p = l.split("\t")
        got.setdefault(p[0], []).append((int(p[2]), np.float32(float(p[4]))))
    return got


def _assert_qres_equals_oracle(path, want, bits=True):
    got = _read_qres(path)
    assert set(got) == set(want)
    for qid, rows in got.items():
        assert [d for d, _ in rows] == want[qid][0].tolist(), qid                 # exact ids, all 100 ranks
        if bits:
            assert np.array_equal(np.array([s for _, s in rows], np.float32).view(np.uint32), want[qid][1].view(np.uint32)), qid


def test_train_then_mask_fields(tmp_path):
    import torch
    from mfar.commands import mask_fields, train
    data = str(tmp_path / "data")
    _write_dataset(data)
    out, tmp = str(tmp_path / "out"), str(tmp_path / "tmp")
    module = train.main(dataset_name="amazon", lexical_index="unused", out=out, temp_dir=tmp, data=data,
                        model_name="random-init:64x2", field_names="title_dense,brand_dense,feature_dense", weights_lr=1e-2,
                        encoder_lr=1e-4, train_batch_size=8, dev_batch_size=16, max_epochs=2, precision="32",
                        additional_partition="test", trec_val_freq=1)
    best = open(f"{out}/best.txt").read().strip()
    assert os.path.exists(best) and os.path.exists(f"{out}/last.ckpt")
    ck = torch.load(best, map_location="cpu", weights_only=False)
    assert "mixture_of_fields_layer.weight" in ck["state_dict"] and any(k.startswith("encoder.0.auto_model.") for k in ck["state_dict"])
    assert set(ck["hyper_parameters"]["field_info"]) == {"brand_dense", "feature_dense", "title_dense"}
    for fn in ("0.qres", "additional_0.qres", "final-all-0.qres", "final-additional-all-0.qres", "results_dicts-all-0.jsonl"):
        assert os.path.getsize(f"{out}/{fn}") > 0, fn
    lines = open(f"{out}/final-all-0.qres").read().strip().split("\n")
    assert len(lines) == 24 * 100 and len(lines[0].split("\t")) == 6
    rows = [json.loads(l) for l in open(f"{out}/results_dicts-all-0.jsonl")]
    assert rows[-1]["additional"] == "test" and rows[-2]["additional"] == "val" and "recall_20" in rows[-1]

    # the evaluation the CLI ran == the oracle on the very same slab rows and query embeddings: exact ids, exact score bits
    dm = _data_module(module, data, tmp, "amazon", 16)
    _assert_qres_equals_oracle(f"{out}/final-all-0.qres", _oracle_eval(module, dm))
    W = module.mixture_of_fields_layer.weight.detach().cpu().numpy()

    out2 = str(tmp_path / "out2")
    m2 = mask_fields.main(dataset_name="amazon", lexical_index="unused", out=out2, temp_dir=tmp, data=data,
                          model_name="random-init:64x2", field_names="title_dense,brand_dense,feature_dense",
                          checkpoint_dir=out, dev_batch_size=16, additional_partition="test")
    rows = [json.loads(l) for l in open(f"{out2}/results_dicts-all-0.jsonl")]
    # baseline + 3 single-field masks + all-dense + 3 per-name masks, each for val and test
    assert len(rows) == 2 * (1 + 3 + 1 + 3)
    assert [r["masked_fields"] for r in rows[::2]] == ["", "brand_dense", "feature_dense", "title_dense",
                                                        "brand_dense,feature_dense,title_dense", "brand_dense", "feature_dense", "title_dense"]
    np.testing.assert_allclose(m2.mixture_of_fields_layer.weight.detach().cpu().numpy(), W)
    # masking every field zeroes all mixed scores
    assert rows[8]["recall_20"] <= rows[0]["recall_20"]
    # that ran as ONE pass over the queries (test_sweep: the mixer once per mask); one test() per mask, as the reference does,
    # leaves the same files behind, byte for byte
    out3 = str(tmp_path / "out3")
    os.environ["MFAR_MASK_SWEEP"] = "0"
    try:
        mask_fields.main(dataset_name="amazon", lexical_index="unused", out=out3, temp_dir=tmp, data=data,
                         model_name="random-init:64x2", field_names="title_dense,brand_dense,feature_dense",
                         checkpoint_dir=out, dev_batch_size=16, additional_partition="test")
    finally:
        del os.environ["MFAR_MASK_SWEEP"]
    for fn in ("results_dicts-all-0.jsonl", "0.qres", "additional_0.qres", "final-all-0.qres", "final-additional-all-0.qres"):
        assert open(f"{out2}/{fn}").read() == open(f"{out3}/{fn}").read(), fn
    assert not [fn for fn in os.listdir(out2) if fn.startswith(".sweep_")]


# ------------------------------------------------------------------------------------------------ BASELINE.json configs[0]
_PRIME_REL = ["associated with", "carrier", "contraindication", "enzyme", "expression absent", "expression present", "indication",
              "interacts with", "linked to", "off-label use", "parent-child", "phenotype absent", "phenotype present", "ppi",
              "side effect", "synergistic interaction", "target", "transporter"]


def _write_prime_dataset(root, n_docs=2000, n_q=40, seed=1):
    """STaRK-prime shaped records (22 fields: name / type / source / details + 18 relation dicts, schema.py:19-42): a
    handful of distinct `type` / `source` values, most relation fields missing from most records (-> "" -> big groups of
    identical rows per field), nested dict values formatted by format_dict (format.py:64-110)."""
    rng = np.random.default_rng(seed)
    vocab = [f"w{i}" for i in range(300)]
    types = ["gene/protein", "drug", "disease", "effect/phenotype", "anatomy", "pathway", "exposure", "molecular_function",
             "biological_process", "cellular_component"]
    sources = ["NCBI", "DrugBank", "MONDO", "HPO", "UBERON", "REACTOME"]
    os.makedirs(root, exist_ok=True)
    docs = []
    with open(f"{root}/corpus", "w") as f:
        for i in range(n_docs):
            t = types[int(rng.zipf(1.6)) % len(types)]
            body = {"name": " ".join(rng.choice(vocab, 2)), "type": t, "source": sources[int(rng.integers(0, len(sources)))],
                    "details": {"description": " ".join(rng.choice(vocab, 6)), "half_life": int(rng.integers(1, 9))}}
            for rel in _PRIME_REL:
                if rng.random() < 0.25:
                    body[rel] = {str(rng.choice(types)): [str(w) for w in rng.choice(vocab, int(rng.integers(1, 4)))]}
            docs.append(body)
            f.write(f"{i}\t{json.dumps(body)}\n")
    for part in ("train", "val", "test"):
        with open(f"{root}/{part}.queries", "w") as fq, open(f"{root}/{part}.qrels", "w") as fr:
            for j in range(n_q):
                d = int(rng.integers(0, n_docs))
                fq.write(f"{part[0]}{j}\t{docs[d]['name']} {docs[d]['details']['description']}\n")
                fr.write(f"{part[0]}{j}\t0\t{d}\t1\n")
    return docs


def test_prime_2000_all_dense_end_to_end(tmp_path, monkeypatch):
    """BASELINE.json configs[0]: STaRK-prime shaped corpus, --max_docs 2000 sized, field_names=all_dense (F = 22), through the
    CLIs (one training iteration, then the evaluation; then the mask sweep).  The .qres the CLI wrote must hold EXACTLY the
    ids the C oracle computes from the same slab rows and the same query embeddings, and the same score bits."""
    from mfar.commands import mask_fields, train
    monkeypatch.setenv("MFAR_SCREEN", "2")      # 2000 rows are below the automatic threshold: force the certified screen on
    data = str(tmp_path / "prime")
    _write_prime_dataset(data)
    out, tmp = str(tmp_path / "out"), str(tmp_path / "tmp")
    module = train.main(dataset_name="prime", lexical_index="unused", out=out, temp_dir=tmp, data=data,
                        model_name="random-init:256x1", field_names="all_dense", weights_lr=5e-2, encoder_lr=1e-4,
                        train_batch_size=4, dev_batch_size=16, max_epochs=1, run_one_iteration=True, precision="32")
    F = len(module.field_info)
    assert F == 22 and module.slab.n_fields == 22 and module.slab.n_rows == 2000
    assert list(module.field_info) == sorted(module.field_info)                   # schema.py:131-134 order == slab field order
    # every field is full of bit-identical rows (missing relation -> ""; 10 types, 6 sources): the screen scans unique rows
    slab = np.stack([module.slab.read_rows(f) for f in range(F)])
    n_unique = [len(np.unique(slab[f], axis=0)) for f in range(F)]
    assert n_unique[list(module.field_info).index("type_dense")] <= 10 and max(n_unique) > 1900
    W = module.mixture_of_fields_layer.weight.detach().cpu().numpy()
    assert not np.allclose(W, 1.0)                                                # the training step moved the gate

    dm = _data_module(module, data, tmp, "prime", 16)
    want = _oracle_eval(module, dm)
    assert len(want) == 40
    _assert_qres_equals_oracle(f"{out}/final-all-0.qres", want)
    st = module.slab.screen_stats()
    assert st["built"] and st["n_checked"] > 0 and st["n_failed"] == 0, st        # screened, and no list fell back to the exact pass

    out2 = str(tmp_path / "out2")
    m2 = mask_fields.main(dataset_name="prime", lexical_index="unused", out=out2, temp_dir=tmp, data=data, model_name="random-init:256x1",
                          field_names="all_dense", checkpoint_dir=out, dev_batch_size=16)
    rows = [json.loads(l) for l in open(f"{out2}/results_dicts-all-0.jsonl")]
    assert len(rows) == 1 + 22 + 1 + 22                                          # baseline, per field, all-dense, per name (mask_fields.py:143-170)
    # the last evaluation of the sweep masked the field named "type": its .qres equals the oracle with that mask
    last = rows[-1]["masked_fields"]
    assert last == "type_dense"
    mask = np.ones(F, np.float32)
    mask[list(m2.field_info).index("type_dense")] = 0
    _assert_qres_equals_oracle(f"{out2}/final-all-0.qres", _oracle_eval(m2, _data_module(m2, data, tmp, "prime", 16), mask), bits=False)


def test_sparse_fields_and_bm25_negatives(tmp_path):
    """a field set with sparse (BM25) fields through the CLIs.  `create_bm25s_index` builds the whole-document
    index that hard-negative mining reads; `train` mines its negatives from it, feeds the BM25 score columns to the loss and
    evaluates with the hybrid step (dense lists / scores from the HBM slab, sparse lists / scores from the host index, one
    mixer over all columns).  The .qres must equal the same algorithm restated on the host from the oracle's pieces."""
    import torch
    from mfar.commands import create_bm25s_index, mask_fields, train
    from mfar.data.typedef import FieldType
    from oracle import mfar_oracle as O
    data, lex = str(tmp_path / "data"), str(tmp_path / "lex")
    _write_dataset(data, relations=True)
    create_bm25s_index.main(data_path=data, dataset_name="amazon", output_path=lex, fields_str="single_sparse")
    assert os.path.exists(f"{lex}/single_sparse_sparse_index/keys.json")
    out, tmp = str(tmp_path / "out"), str(tmp_path / "tmp")
    module = train.main(dataset_name="amazon", lexical_index=lex, out=out, temp_dir=tmp, data=data, model_name="random-init:64x2",
                        field_names="title_dense,brand_dense,title_sparse,feature_sparse", weights_lr=1e-2, encoder_lr=1e-4,
                        train_batch_size=8, dev_batch_size=16, max_epochs=1, precision="32", negative_sampling_params=(20, 5, 1))
    fields = list(module.field_info.items())
    assert [k for k, _ in fields] == ["brand_dense", "title_dense", "feature_sparse", "title_sparse"]      # schema.py:131-134
    assert module.slab.n_fields == 2 and module.has_sparse
    W = module.mixture_of_fields_layer.weight.detach().cpu().numpy()
    assert W.shape == (64, 4) and not np.allclose(W, 1.0)

    # the same algorithm from the oracle's pieces (contrastive.py:669-704 with sparse indices in indices_dict)
    slab = np.stack([module.slab.read_rows(f) for f in range(2)])
    dm = _data_module(module, data, tmp, "amazon", 16)
    keys = module.numeric_ids_to_keys
    want = {}
    module.eval()
    for batch in dm.test_dataloader()[0]:
        with torch.no_grad():
            qe = module.encode_query_batch(batch).cpu().numpy()
        for i, inst in enumerate(batch.instances):
            lists = [O.c_retrieve(slab[f], qe[i:i + 1], 100, True)[0][0] for f in range(2)]
            for key, f in fields:
                if f.field_type == FieldType.SPARSE:
                    lists.append(np.array([module.keys_to_numeric_ids[k] for k, _ in module.indices_dict[key].retrieve(inst.text, 100)]))
            cand = np.unique(np.concatenate(lists))
            xs = np.zeros((len(cand), 4), np.float32)
            xs[:, :2] = O.c_score_candidates(slab, qe[i:i + 1], cand[None])[0]
            for col, (key, f) in enumerate(fields):
                if f.field_type == FieldType.SPARSE:
                    xs[:, col] = module.indices_dict[key].score(inst.text, [keys[d] for d in cand])
            ids, sc = O.canon(cand, O.c_mix(xs, O.c_gate(qe[i], W, True), None))
            want[inst._id] = (ids[:100], sc[:100])
    _assert_qres_equals_oracle(f"{out}/final-all-0.qres", want)
    # sparse scores matter: the ranking differs from the dense-only ranking for at least one query
    dense_only = O.c_two_stage(slab, qe, W[:, :2].copy(), None)
    assert any(not np.array_equal(want[inst._id][0], dense_only["ids"][i]) for i, inst in enumerate(batch.instances))

    # mask_fields: baseline, 4 single fields, all-sparse, all-dense, then per field NAME (brand, feature, title)
    out2 = str(tmp_path / "out2")
    mask_fields.main(dataset_name="amazon", lexical_index=lex, out=out2, temp_dir=tmp, data=data, model_name="random-init:64x2",
                     field_names="title_dense,brand_dense,title_sparse,feature_sparse", checkpoint_dir=out, dev_batch_size=16)
    rows = [json.loads(l) for l in open(f"{out2}/results_dicts-all-0.jsonl")]
    assert [r["masked_fields"] for r in rows][:7] == ["", "brand_dense", "title_dense", "feature_sparse", "title_sparse",
                                                      "feature_sparse,title_sparse", "brand_dense,title_dense"]


def test_corpus_encode_replays_captured_forwards(tmp_path, monkeypatch):
    """mfar/modeling/graphed.py: the corpus-encode forwards of a shape seen before are replayed from a captured graph -- the same bits as the
    eager forward of that shape, under fp32 and under autocast; a replay reads the CURRENT weights (an in-place update between two encodes
    is seen); MFAR_ENCODE_GRAPHS=0 = the eager pipeline, rows equal to rounding (other batch shapes)."""
    import torch
    from mfar.commands import train
    data = str(tmp_path / "data")
    _write_dataset(data, n_docs=400, n_q=101)

    def encode(m):
        m.mark_encoder_updated()
        m.on_eval_start()
        m.qres_output.close()
        torch.cuda.synchronize()
        return np.stack([m.slab.read_rows(f) for f in range(2)])
    for mode in ("", "fp16"):
        monkeypatch.setenv("MFAR_ENCODE_AUTOCAST", mode)
        monkeypatch.setenv("MFAR_ENCODE_GRAPHS", "1")
        m = train.main(dataset_name="amazon", lexical_index="unused", out=str(tmp_path / f"out{mode}"), temp_dir=str(tmp_path / f"tmp{mode}"),
                       data=data, model_name="random-init:64x2", field_names="title_dense,brand_dense", weights_lr=1e-2, max_epochs=0,
                       dev_batch_size=16, precision="32")
        encode(m)
        g = m._graphed
        assert g is not None and not g.failed
        g.reset()
        cap, g.max_graphs = g.max_graphs, 0     # nothing captured: the eager forwards at the graph shapes
        r0 = g.n_replays
        first = encode(m)
        assert g.n_replays == r0 and not g.graphs
        g.max_graphs = cap
        encode(m)                               # every shape of this corpus seen twice by now: captured
        r0, e0 = g.n_replays, g.n_eager
        again = encode(m)
        assert g.n_replays > r0 and g.n_eager == e0 and len(g.graphs) >= 1, (g.n_replays, g.n_eager, len(g.graphs))
        assert np.array_equal(first, again)     # replays = the eager forwards of the same shapes, bit for bit
        monkeypatch.setenv("MFAR_ENCODE_GRAPHS", "0")
        eager = encode(m)
        assert m._graphed is None
        scale = float(np.abs(eager).max())
        assert np.abs(eager - again).max() <= (2e-5 if mode == "" else 4e-3) * scale
        assert len(np.unique(again[1], axis=0)) == len(np.unique(eager[1], axis=0))       # brand: the same duplicate structure
        # an in-place weight update: the graphs stay (same addresses) and replay the new weights
        monkeypatch.setenv("MFAR_ENCODE_GRAPHS", "1")
        encode(m)
        encode(m)
        g = m._graphed
        n_graphs = len(g.graphs)
        assert n_graphs >= 1
        with torch.no_grad():
            for p_ in m.encoder.parameters():
                p_.mul_(1.25)
        r0 = g.n_replays
        moved = encode(m)
        assert m._graphed is g and len(g.graphs) == n_graphs and g.n_replays > r0
        monkeypatch.setenv("MFAR_ENCODE_GRAPHS", "0")
        want = encode(m)
        assert np.abs(moved - want).max() <= (2e-5 if mode == "" else 4e-3) * float(np.abs(want).max())
        assert np.abs(moved - again).max() > 1e-2 * scale
        # parameters that MOVED: the graphs are dropped, not replayed against stale addresses
        monkeypatch.setenv("MFAR_ENCODE_GRAPHS", "1")
        encode(m)
        g = m._graphed
        m.encoder.float()                        # (a no-op: same storage) ...
        encode(m)
        assert len(g.graphs) >= 1
        for p_ in m.encoder.parameters():        # ... a real move
            p_.data = p_.data.clone()
        encode(m)
        assert g.n_eager > 0 and np.isfinite(m.slab.read_rows(0)).all()
        assert np.abs(np.stack([m.slab.read_rows(f) for f in range(2)]) - want).max() <= (2e-5 if mode == "" else 4e-3) * float(np.abs(want).max())


def test_captured_forward_at_the_encoders_full_length():
    """A BERT-base-wide encoder at L = 512 (truncated documents: the commonest shape of a long field).  There a captured graph replays the
    old mean pooling `(tok * m).sum(1)` wrong from the SECOND replay on (profiles/r06_k_graph_reduce_probe.txt); SentenceEncoder.forward
    pools by a batched product instead, and GraphedForward checks every graph against the eager forward before it trusts it.
    Replays 1..5 on fresh inputs = the eager forward, bit for bit (fp32)."""
    import torch
    from mfar.modeling.graphed import GraphedForward
    from mfar.modeling.util import prepare_model
    dev = torch.device("cuda:0")
    enc = prepare_model("random-init:768x1")[1].to(dev).eval()
    g = GraphedForward(enc, dev)
    g.begin()
    gen = torch.Generator().manual_seed(3)

    def batch():
        ids = torch.randint(5, 60, (16, 512), generator=gen)
        lens = torch.randint(300, 513, (16,), generator=gen)
        return {"input_ids": ids, "attention_mask": (torch.arange(512)[None, :] < lens[:, None]).long()}

    def eager(f):
        with torch.no_grad():
            d = {k: v.to(dev) for k, v in f.items()}
            d["token_type_ids"] = torch.zeros_like(d["input_ids"])
            return enc(d)["sentence_embedding"].float()
    f = batch()
    assert torch.equal(g(f, None), eager(f)) and not g.graphs                   # first sight: eager
    f = batch()
    assert torch.equal(g(f, None).clone(), eager(f))                            # second sight: captured, checked, replayed
    assert len(g.graphs) == 1 and not g.rejected and not g.failed
    for _ in range(5):
        f = batch()
        out = g(f, None).clone()
        torch.cuda.synchronize()
        assert torch.equal(out, eager(f))
    assert g.n_replays >= 6 and g.n_eager == 1
    # the hazard itself (informational: printed, so that a run with -s says when the stack stops having it)
    x = torch.randn(16, 512, 768, device=dev)
    m = (torch.rand(16, 512, 1, device=dev) > 0.3).float()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        (x * m).sum(1)
    torch.cuda.current_stream().wait_stream(s)
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=s):
        y = (x * m).sum(1)
    cg.replay()
    x.normal_()
    cg.replay()
    torch.cuda.synchronize()
    print("captured (x * m).sum(1) at L = 512, second replay: max difference to the eager result", float((y - (x * m).sum(1)).abs().max()))


def test_corpus_encode_under_autocast(tmp_path, monkeypatch):
    """MFAR_ENCODE_AUTOCAST=bf16: the corpus encode runs under bf16 autocast, the slab still holds fp32 rows, and
    they stay close to the fp32 encode's (same texts -> bit-identical duplicate rows either way)."""
    from mfar.commands import train
    data = str(tmp_path / "data")
    _write_dataset(data, n_docs=120, n_q=101)      # >= 100 candidates per query are needed by the top-100
    rows = {}
    for mode in ("", "bf16"):
        monkeypatch.setenv("MFAR_ENCODE_AUTOCAST", mode)
        m = train.main(dataset_name="amazon", lexical_index="unused", out=str(tmp_path / f"out{mode}"), temp_dir=str(tmp_path / f"tmp{mode}"),
                       data=data, model_name="random-init:64x2", field_names="title_dense,brand_dense", weights_lr=1e-2, max_epochs=0,
                       dev_batch_size=16, precision="32")
        rows[mode] = np.stack([m.slab.read_rows(f) for f in range(2)])
    a, b = rows[""], rows["bf16"]
    cos = (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))
    assert cos.min() > 0.995 and not np.array_equal(a, b)
    assert len(np.unique(b[1], axis=0)) == len(np.unique(a[1], axis=0))       # brand: the same duplicate structure
