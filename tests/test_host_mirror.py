"""The host-side mirror of the reference's Python surface, against the golden fixtures captured from the reference
(tools/gen_golden.py): field resolution/order, TREC line formats, document formatting, memmap layout, CLI signatures,
training loss."""
import inspect
import json
import math
import os

import numpy as np
import pytest

NL = chr(10)


def _j(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_resolve_fields_golden(golden_dir):
    from mfar.data.schema import resolve_fields
    g = _j(golden_dir, "schema.json")
    for case, want in g.items():
        if case == "errors":
            continue
        ds, spec = case.split("|")
        if spec == "off-label.use_dense":
            spec = "off-label.use_dense,name_dense"          # how tools/gen_golden.py produced this case
        got = resolve_fields(spec, "data/amazon" if "title_dense,brand" in spec else ds)
        assert [dict(key=k, name=f.name, type=f.field_type.name, max_seq_length=f.max_seq_length, dataset=f.dataset)
                for k, f in got.items()] == want, case
    with pytest.raises(NotImplementedError):
        resolve_fields("all_dense", "nosuchdataset")
    with pytest.raises(ValueError):
        resolve_fields("nosuchfield_dense", "mag")
    assert list(resolve_fields(["all_dense"], "amazon")) == [w["key"] for w in g["amazon|all_dense"]]


def test_field_serialisation_roundtrip():
    from mfar.data.typedef import Field, FieldType
    f = Field("title_dense", "title", FieldType.DENSE, 64, "mag")
    assert Field.deserialize(f.serialize()).serialize() == f.serialize()
    assert f.__dict__() == {"name": "title", "field_type": "DENSE", "max_seq_length": 64}
    assert json.loads(str(f)) == f.__dict__()


def test_trec_formats_golden(golden_dir, tmp_path):
    from mfar.data import trec
    g = _j(golden_dir, "trec.json")
    q = trec.QRes(query_id="q1", doc_id="d7", sim=float(np.float32(12.3456789)))
    assert str(q) == g["qres_str"] and str(trec.QRes.from_str(str(q))) == g["qres_roundtrip"]
    r = trec.QRels("q1", "d7", 1.0)
    assert str(r) == g["qrels_str"] and str(trec.QRels.from_str(str(r))) == g["qrels_roundtrip"]
    assert trec.parse_trec_eval_output(g["parse_sample_in"]) == g["parse_sample_out"]
    p = tmp_path / "c.tsv"
    p.write_text(NL.join(g["read_corpus_in"]) + NL)
    assert [[a, b] for a, b in trec.read_corpus(str(p))] == g["read_corpus_out"]


def test_format_documents_golden(golden_dir):
    from mfar.data.format import format_documents
    g = _j(golden_dir, "format_documents.json")
    for case, want in g["out"].items():
        ds, fld = case.split("|")
        corpus = [(a, b) for a, b in g["docs"][ds]]
        if isinstance(want, dict):
            with pytest.raises(Exception) as ei:
                format_documents(corpus, fld, ds)
            assert type(ei.value).__name__ == want["raises"], case
        else:
            assert [[a, b] for a, b in format_documents(corpus, fld, ds)] == want, case


def test_memmap_dict_golden(golden_dir, tmp_path):
    from mfar.data.util import MemoryMapDict
    g = _j(golden_dir, "memmap.json")
    p = tmp_path / "title.npy"
    p.write_bytes(b"")
    m = MemoryMapDict(str(p), ["a", "b", "c"], (3, 4))
    m["b"] = np.array([1, 2, 3, 4], dtype=np.float32)
    m["c"] = np.array([5, 6, 7, 8], dtype=np.float32)
    m.close()
    raw = p.read_bytes()
    assert len(raw) == g["size"] and raw.hex() == g["raw_hex"]      # raw, headerless float32 rows
    assert len(m) == g["len"] and ("b" in m) == g["contains_b"] and ("z" in m) == g["contains_z"]
    assert list(iter(m)) == g["iter"] and m["b"].tolist() == g["b"]
    m.reopen()
    assert m["c"].tolist() == [5, 6, 7, 8]


def test_cli_signatures_golden(golden_dir):
    from mfar.commands import mask_fields, train
    g = _j(golden_dir, "cli_signatures.json")
    for name, mod in (("train", train), ("mask_fields", mask_fields)):
        sig = inspect.signature(mod.main)
        got = []
        for p in sig.parameters.values():
            default = None if p.default is inspect._empty else p.default
            got.append(dict(name=p.name, kind=p.kind.name, required=(p.default is inspect._empty),
                            default=default if isinstance(default, (int, float, str, bool, type(None))) else repr(default)))
        assert got == g[name], name


def test_metrics_definitions():
    from mfar.data import trec
    qrels = [trec.QRels("q1", "a", 1.0), trec.QRels("q1", "b", 1.0), trec.QRels("q2", "x", 1.0), trec.QRels("q3", "n", 0.0)]
    qres = [trec.QRes("q1", d, s) for d, s in [("z", 9.0), ("a", 8.0), ("y", 7.0), ("b", 6.0)]] + \
           [trec.QRes("q2", d, s) for d, s in [("x", 5.0), ("w", 4.0)]] + [trec.QRes("q3", "n", 1.0)]
    m = trec.compute_metrics(qrels, qres)
    assert m["recip_rank"] == pytest.approx((0.5 + 1.0) / 2)
    assert m["map"] == pytest.approx(((1 / 2 + 2 / 4) / 2 + 1.0) / 2)
    assert m["recall_5"] == pytest.approx(1.0) and m["success_1"] == pytest.approx(0.5)
    assert m["Rprec"] == pytest.approx((0.5 + 1.0) / 2)
    dcg1 = 1 / math.log2(3) + 1 / math.log2(5)
    idcg1 = 1 + 1 / math.log2(3)
    assert m["ndcg_cut_10"] == pytest.approx((dcg1 / idcg1 + 1.0) / 2)


def test_query_dataset_short_query_rule():
    from mfar.data.dataset import QueryDataset
    ds = QueryDataset(tokenizer=None, queries={"1": "hi", "2": "a real query"})
    assert ds[0].text == "what" and ds[1].text == "a real query" and ds[0]._id == "1" and len(ds) == 2




def test_hybrid_contrastive_loss_golden(golden_dir):
    """Training-time scorer + loss (losses.py:176-188, 275-360) against loss value and gradients captured from the
    reference's HybridContrastiveLoss (1-rank group): plain, with BatchNorm over fields, and with sparse score columns behind
    the dense ones (tests/helpers/loss_check.py; the `-m gpu` twin runs the same check on the device)."""
    from helpers.loss_check import check_hybrid_loss_golden
    assert check_hybrid_loss_golden(golden_dir, "cpu") == ["plain", "bn", "sparse"]


def test_fire_like_cli_parsing():
    from mfar.commands._cli import parse

    def main(*, dataset_name: str, out: str, temp_dir: str = "/tmp", dev_batch_size: int = 64, weights_lr=None,
             negative_sampling_params=(100, 50, 1), query_cond: bool = True, field_names=None):
        pass

    kw = parse(["--dataset-name", "amazon", "--out=/x", "--temp-dir", "/t", "--dev_batch_size", "32", "--weights_lr=1e-1",
                "--negative-sampling-params", "(10,5,1)", "--query_cond", "False", "--field_names", "all_dense"], main)
    assert kw == dict(dataset_name="amazon", out="/x", temp_dir="/t", dev_batch_size=32, weights_lr=0.1,
                      negative_sampling_params=(10, 5, 1), query_cond=False, field_names="all_dense")
    with pytest.raises(SystemExit):
        parse(["--out", "/x"], main)              # missing required flag
    with pytest.raises(SystemExit):
        parse(["--dataset_name", "a", "--out", "/x", "--nope", "1"], main)


def test_sentence_encoder_structure():
    """prepare_model's encoder = HF model -> mask-weighted mean pooling (modeling/util.py:38-52); checked against a
    hand-written pooling on a randomly initialised BERT (no checkpoint can be downloaded: encoder parity is structural)."""
    import torch
    from mfar.modeling.util import prepare_model
    tok, enc, dec = prepare_model("random-init:64x2")
    assert dec is None and enc.get_sentence_embedding_dimension() == 64 and enc.get_max_seq_length() == 512
    texts = ["a red shoe", "blue", "", "the quick brown fox jumps over the lazy dog"]
    emb = enc.encode(texts, batch_size=2, convert_to_numpy=True)
    assert emb.shape == (4, 64) and emb.dtype == np.float32
    feats = enc.tokenize(texts)
    with torch.no_grad():
        hid = enc.auto_model(**feats).last_hidden_state
        m = feats["attention_mask"].unsqueeze(-1).float()
        want = ((hid * m).sum(1) / m.sum(1).clamp(min=1e-9)).numpy()
        got = enc(feats)["sentence_embedding"].numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(emb, want, rtol=1e-4, atol=1e-5)      # encode() sorts by length and restores the order
    assert [k for k in enc.state_dict() if k.startswith("0.auto_model.")]      # sentence-transformers key layout
    _, encn, _ = prepare_model("random-init:64x2", normalize=True)
    np.testing.assert_allclose(np.linalg.norm(encn.encode(texts[:2]), axis=1), 1.0, rtol=1e-5)
    with pytest.raises(ValueError):
        prepare_model("/definitely/not/a/model")


def test_training_step_encodes_all_fields_in_few_forwards(monkeypatch):
    """train._encode_fields: the F x B field texts of a training batch go through the encoder in as few forwards as a token
    budget allows (per-field truncation lengths kept) -- same [B, F, E] embeddings as one forward per field
    (contrastive.py:412-414), gradients reach the encoder, and the number of forwards drops."""
    import json
    from types import SimpleNamespace
    import torch
    from mfar.commands import train
    from mfar.data.schema import resolve_fields
    from mfar.modeling.util import prepare_model
    tok, enc, _ = prepare_model("random-init:64x2")
    enc.eval()
    rng = np.random.default_rng(3)
    words = ["red", "blue", "shoe", "hat", "acme", "zen", "light", "heavy", "wool", "cotton"]
    docs = [(str(i), {"title": " ".join(rng.choice(words, int(rng.integers(1, 9)))), "brand": str(rng.choice(words)),
                      "feature": [str(w) for w in rng.choice(words, int(rng.integers(0, 5)))]}) for i in range(7)]
    docs[3] = ("3", {"title": "only a title"})                                    # missing fields -> ""
    module = SimpleNamespace(field_info=resolve_fields("title_dense,brand_dense,feature_dense", "amazon"), prefix=True, encoder=enc)
    calls = []
    orig = enc.forward
    monkeypatch.setattr(enc, "forward", lambda feats: (calls.append(feats["input_ids"].shape), orig(feats))[1])
    fused = train._encode_fields(module, tok, docs, 512, torch.device("cpu"))
    n_fused = len(calls)
    monkeypatch.setenv("MFAR_TRAIN_FUSED_ENCODE", "0")
    calls.clear()
    per_field = train._encode_fields(module, tok, docs, 512, torch.device("cpu"))
    assert fused.shape == per_field.shape == (7, 3, 64) and len(calls) == 3 and n_fused == 1
    np.testing.assert_allclose(fused.detach().numpy(), per_field.detach().numpy(), rtol=1e-4, atol=1e-5)
    fused.sum().backward()
    assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in enc.parameters())
    # a field's own truncation length is kept (brand: 16 tokens, schema.py) when texts of all fields share a forward
    long_brand = [("0", {"title": "t", "brand": "x " * 200, "feature": []})]
    monkeypatch.setenv("MFAR_TRAIN_FUSED_ENCODE", "1")
    calls.clear()
    a = train._encode_fields(module, tok, long_brand, 512, torch.device("cpu"))
    assert sorted(s[1] for s in calls)[-1] <= 16                                   # nothing longer than brand's 16 tokens was fed
    monkeypatch.setenv("MFAR_TRAIN_FUSED_ENCODE", "0")
    b = train._encode_fields(module, tok, long_brand, 512, torch.device("cpu"))
    np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=1e-4, atol=1e-5)


def test_format_single_golden(golden_dir):
    """The whole-document text of the single_dense field (format.py:20-22, 113-415), per dataset, incl. the records the
    reference cannot format (it raises UnboundLocalError; so does the mirror)."""
    from mfar.data.format import format_documents
    d = json.load(open(os.path.join(golden_dir, "format_single.json")))
    n = 0
    for ds, rows in d["out"].items():
        if ds == "bad_dataset":
            continue
        for doc, want in zip(d["docs"][ds], rows):
            doc = (doc[0], doc[1])
            if isinstance(want[1], dict):
                with pytest.raises(Exception) as ei:
                    format_documents([doc], "single", ds)
                assert type(ei.value).__name__ == want[1]["raises"], (ds, doc[0])
            else:
                assert [list(x) for x in format_documents([doc], "single", ds)] == [want], (ds, doc[0])
            n += 1
    assert n == 14
    with pytest.raises(ValueError):
        format_documents([("x", {"title": "t"})], "single", "nosuch")
    # resolve_fields("single_dense") names this field (schema.py:123-124)
    from mfar.data.schema import resolve_fields
    f = resolve_fields("single_dense", "amazon")["single_dense"]
    assert f.name == "single" and f.max_seq_length == 512


class _FakeEncoder:
    """Records every encode() call; the embedding of a text is [len(text), #call, position in call]."""

    def __init__(self):
        self.calls = []

    def encode(self, texts, batch_size=64, convert_to_numpy=False, convert_to_tensor=False, **kw):
        self.calls.append((list(texts), batch_size, convert_to_numpy, convert_to_tensor))
        arr = np.array([[len(t), len(self.calls) - 1, j] for j, t in enumerate(texts)], dtype=np.float32)
        if convert_to_tensor:
            import torch
            return torch.from_numpy(arr)
        return arr


def test_candidate_encoding_stream_order_and_batches():
    """index.py:234-258 (the non-multiprocess branch the eval path uses, contrastive.py:487): input order is kept, the
    corpus is cut into chunks of batch_size with a short tail batch, and each chunk is one encoder.encode() call."""
    from mfar.data.index import candidate_encoding_stream
    corpus = [(f"d{i}", "x" * (i % 5)) for i in range(11)]
    enc = _FakeEncoder()
    out = list(candidate_encoding_stream(enc, iter(corpus), batch_size=4, multiprocess=False, show_progress=False))
    assert [i for i, _ in out] == [i for i, _ in corpus]
    assert [len(c[0]) for c in enc.calls] == [4, 4, 3]                       # tail batch
    assert all(c[1] == 4 and c[2] is True for c in enc.calls)                # batch_size passed on, numpy rows (index.py:257)
    for (i, v), (_, t) in zip(out, corpus):
        assert isinstance(v, np.ndarray) and v.shape == (3,) and v[0] == len(t)
    assert [int(v[1]) for _, v in out] == [0] * 4 + [1] * 4 + [2] * 3
    assert list(candidate_encoding_stream(enc, [], batch_size=4, show_progress=False)) == []
    # the device-resident variant used by on_eval_start yields tensors
    import torch
    out_t = list(candidate_encoding_stream(_FakeEncoder(), corpus[:5], batch_size=2, show_progress=False, as_tensor=True))
    assert all(torch.is_tensor(v) for _, v in out_t) and [i for i, _ in out_t] == [f"d{i}" for i in range(5)]


def _reference_layout_state_dict(encoder, E, F, with_bn=False):
    """A state_dict with exactly the key set a checkpoint of the REFERENCE module holds (contrastive.py:279-293, 634-645):
    `encoder.0.auto_model.*` (SentenceTransformer Sequential index 0), the field-weight matrix under both of its names
    (the same Parameter is registered on the module and inside the loss), optional BatchNorm1d(F) of the loss."""
    import torch
    g = torch.Generator().manual_seed(5)
    sd = {f"encoder.0.auto_model.{k}": torch.randn(v.shape, generator=g).to(v.dtype) if v.dtype.is_floating_point else v.clone()
          for k, v in encoder.auto_model.state_dict().items()}
    w = torch.randn(E, F, generator=g)
    sd["mixture_of_fields_layer.weight"] = w
    sd["hybrid_contrastive_loss_fn.mixture_of_fields_layer.weight"] = w.clone()
    if with_bn:
        sd.update({"hybrid_contrastive_loss_fn.bn.weight": torch.ones(F), "hybrid_contrastive_loss_fn.bn.bias": torch.zeros(F),
                   "hybrid_contrastive_loss_fn.bn.running_mean": torch.zeros(F), "hybrid_contrastive_loss_fn.bn.running_var": torch.ones(F),
                   "hybrid_contrastive_loss_fn.bn.num_batches_tracked": torch.tensor(3)})
    return sd


def test_load_checkpoint_with_reference_key_layout(tmp_path):
    """mask_fields.py:110-121 -> load_from_checkpoint: a Lightning .ckpt with the reference's exact key set loads into the
    mirror (encoder weights, W, BatchNorm state); layouts that do not line up raise instead of silently loading nothing."""
    import torch
    from mfar.data.schema import resolve_fields
    from mfar.modeling.contrastive import RetrievalTrainingModule
    from mfar.modeling.util import prepare_model
    fields = resolve_fields("title_dense,brand_dense,feature_dense", "amazon")
    _, enc, _ = prepare_model("random-init:64x2")
    corpus = [("0", {"title": "a"}), ("1", {"title": "b"})]
    sd = _reference_layout_state_dict(enc, 64, 3, with_bn=True)
    hp = dict(model_id="random-init:64x2", dataset_name="amazon", corpus_path="/c", query_cond=True, weights_learning_rate=0.1,
              field_info={k: f.serialize() for k, f in fields.items()}, indices_list=[], vectors_list=[], corpus=[],
              precomputed_sparse_scores=[], some_future_hparam=1)
    path = str(tmp_path / "ref.ckpt")
    torch.save({"state_dict": sd, "hyper_parameters": hp, "pytorch-lightning_version": "2.0.0", "epoch": 3, "global_step": 7}, path)
    kw = dict(corpus=corpus, indices_dict={}, vectors_dict={}, encoder=enc, dev_qrels_path="/q", field_info=fields, out_dir=str(tmp_path))
    m = RetrievalTrainingModule.load_from_checkpoint(path, **kw)
    assert torch.equal(m.mixture_of_fields_layer.weight.data, sd["mixture_of_fields_layer.weight"])
    k0 = "embeddings.word_embeddings.weight"
    assert torch.equal(m.encoder.auto_model.state_dict()[k0], sd["encoder.0.auto_model." + k0])
    assert set(m.bn_state) == {"weight", "bias", "running_mean", "running_var", "num_batches_tracked"}
    # what the mirror writes back carries the same keys the reference wrote
    assert set(m.checkpoint_state()["state_dict"]) == set(sd)
    # position_ids is persisted by some transformers versions only: its absence is fine
    sd2 = {k: v for k, v in sd.items() if not k.endswith("position_ids")}
    torch.save({"state_dict": sd2, "hyper_parameters": hp}, path)
    RetrievalTrainingModule.load_from_checkpoint(path, **kw)
    # a different encoder key layout (e.g. the gtr-t5 module stack): nothing would load -> must raise
    bad = {k.replace("encoder.0.auto_model.", "encoder.0.model."): v for k, v in sd.items()}
    torch.save({"state_dict": bad, "hyper_parameters": hp}, path)
    with pytest.raises(RuntimeError):
        RetrievalTrainingModule.load_from_checkpoint(path, **kw)
    torch.save({"state_dict": {k: v for k, v in sd.items() if not k.startswith("encoder.")}, "hyper_parameters": hp}, path)
    with pytest.raises(RuntimeError, match="encoder"):
        RetrievalTrainingModule.load_from_checkpoint(path, **kw)
    lost = {k: v for k, v in sd.items() if "layer.1.output.dense.weight" not in k}
    torch.save({"state_dict": lost, "hyper_parameters": hp}, path)
    with pytest.raises(RuntimeError, match="lacks"):
        RetrievalTrainingModule.load_from_checkpoint(path, **kw)
    # W trained for another field set
    wrong = dict(sd)
    wrong["mixture_of_fields_layer.weight"] = wrong["hybrid_contrastive_loss_fn.mixture_of_fields_layer.weight"] = torch.ones(64, 8)
    torch.save({"state_dict": wrong, "hyper_parameters": hp}, path)
    with pytest.raises(RuntimeError, match="field set"):
        RetrievalTrainingModule.load_from_checkpoint(path, **kw)


def test_gtr_t5_branch_and_decoder(tmp_path):
    """prepare_model's gtr-t5 branch (modeling/util.py:22-36): T5 encoder -> mean pooling -> bias-free Dense [-> Normalize],
    sentence-transformers checkpoint keys (`0.auto_model.*`, `2.linear.weight`), with_decoder = a T5ForConditionalGeneration
    that shares the encoder's parameters; loading from a local directory in the sentence-transformers layout."""
    import torch
    from mfar.modeling.util import prepare_model
    tok, enc, dec = prepare_model("random-init-t5:64x2", with_decoder=True)
    assert enc.get_sentence_embedding_dimension() == 64 and enc.get_max_seq_length() == 512
    keys = list(enc.state_dict())
    assert "2.linear.weight" in keys and "2.linear.bias" not in keys and any(k.startswith("0.auto_model.encoder.") for k in keys)
    texts = ["a red shoe", "blue", "the quick brown fox"]
    feats = enc.tokenize(texts)
    with torch.no_grad():
        hid = enc.auto_model(input_ids=feats["input_ids"], attention_mask=feats["attention_mask"]).last_hidden_state
        m = feats["attention_mask"].unsqueeze(-1).float()
        want = ((hid * m).sum(1) / m.sum(1).clamp(min=1e-9)) @ enc.dense.linear.weight.t()
        got = enc(feats)["sentence_embedding"]
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    assert not np.allclose(np.linalg.norm(got.numpy(), axis=1), 1.0)                 # Normalize is dropped without --normalize
    _, encn, _ = prepare_model("random-init-t5:64x2", normalize=True)
    np.testing.assert_allclose(np.linalg.norm(encn.encode(texts), axis=1), 1.0, rtol=1e-5)
    # the decoder model runs over the SAME encoder parameters
    assert dec is not None and dec.encoder is enc.auto_model.encoder
    p0 = next(enc.auto_model.encoder.parameters())
    assert any(p is p0 for p in dec.encoder.parameters())
    with torch.no_grad():
        out = dec(input_ids=feats["input_ids"], attention_mask=feats["attention_mask"], decoder_input_ids=feats["input_ids"][:, :2])
    assert out.logits.shape[:2] == (3, 2)
    with pytest.raises(UnboundLocalError):
        prepare_model("random-init:64x2", with_decoder=True)
    # a local sentence-transformers directory: config + encoder weights at the root, 2_Dense/model.safetensors
    from safetensors.torch import save_file
    root = tmp_path / "gtr-t5-tiny"
    enc.auto_model.save_pretrained(root)
    tok.save_pretrained(root)
    os.makedirs(root / "2_Dense")
    save_file({"linear.weight": enc.dense.linear.weight.detach().clone()}, str(root / "2_Dense" / "model.safetensors"))
    _, enc2, dec2 = prepare_model(str(root))
    assert dec2 is None and enc2.dense is not None
    np.testing.assert_allclose(enc2.encode(texts), enc.encode(texts), rtol=1e-5, atol=1e-6)


def test_prepare_model_from_local_directory(tmp_path):
    """The local-directory branch of prepare_model (modeling/util.py:54-71): a HF model directory (what a fine-tuned contriever
    is on disk) -> AutoModel + mean pooling [+ Normalize]; unknown paths raise ValueError like the reference."""
    import torch
    from mfar.modeling.util import prepare_model
    tok, enc, _ = prepare_model("random-init:64x2")
    root = tmp_path / "bert-tiny"
    enc.auto_model.save_pretrained(root)
    tok.save_pretrained(root)
    tok2, enc2, dec2 = prepare_model(str(root), freeze_encoder=True)
    assert dec2 is None and not any(p.requires_grad for p in enc2.parameters())
    texts = ["a red shoe", "the quick brown fox jumps", ""]
    np.testing.assert_allclose(enc2.encode(texts), enc.encode(texts), rtol=1e-5, atol=1e-6)
    assert set(enc2.state_dict()) == set(enc.state_dict())


def test_merge_of_shard_lists_equals_the_oracle_merge():
    """modeling/contrastive.py `_merge_shard_lists` (the hybrid step with several ranks: per-shard zero-sentinel lists ->
    global per-field top-k) against the oracle's list merge, ties and padding included."""
    import torch
    from mfar.modeling.contrastive import _merge_shard_lists
    from oracle import mfar_oracle as O
    rng = np.random.default_rng(9)
    world, Q, F, k = 3, 4, 2, 10
    ids = np.zeros((world, Q, F, k), np.int64)
    sc = np.zeros((world, Q, F, k), np.float32)
    for w in range(world):
        for q in range(Q):
            for f in range(F):
                n = int(rng.integers(0, k + 1))                       # short lists are padded with (0, 0.0)
                s = np.sort(rng.choice(np.array([0.5, 1.0, 1.5, 2.0, 2.5, 3.0], np.float32), n))[::-1]      # many ties
                ids[w, q, f, :n] = 100 * w + np.sort(rng.choice(90, n, replace=False)) + 1
                order = np.lexsort((ids[w, q, f, :n], -s))
                sc[w, q, f, :n], ids[w, q, f, :n] = s[order], ids[w, q, f, :n][order]
    gi, gs = _merge_shard_lists(torch.from_numpy(ids.reshape(world * Q, F, k)), torch.from_numpy(sc.reshape(world * Q, F, k)), k, world)
    for q in range(Q):
        for f in range(F):
            wi, ws = O.c_merge_lists(ids[:, q, f], sc[:, q, f], True)
            assert np.array_equal(gi[q, f].numpy(), wi) and np.array_equal(gs[q, f].numpy(), ws), (q, f)


def test_prefetched_corpus_encode_equals_the_generic_path(monkeypatch):
    """RetrievalTrainingModule._encode_texts_prefetched (the corpus encode with tokenisation on a producer thread, Rust tokenizer called
    directly): the same tokens as `encoder.tokenize` -- special tokens, truncation at the encoder's limit -- hence, text by text, the rows of
    `encoder.encode`; every row written exactly once whatever the token budget; off by MFAR_ENCODE_PREFETCH=0 or without a Rust backend."""
    import types
    import torch
    from mfar.modeling.contrastive import RetrievalTrainingModule
    from mfar.modeling.util import prepare_model
    _, enc, _ = prepare_model("random-init:32x1")
    enc.max_seq_length = 24
    rng = np.random.default_rng(5)
    words = ["alpha", "beta", "gamma", "delta", "x", "retrieval", "field", "a.b,c"]
    uniq = [""] + [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(1, 30, size=150)]      # some far beyond 24 tokens
    uniq = list(dict.fromkeys(uniq))
    order = sorted(range(len(uniq)), key=lambda i: len(uniq[i]))
    class Stub:                                                       # the methods under test on an object that is not a LightningModule
        pass
    for name in ("_prefetch_backend", "_token_batches", "_consume", "_forward_rows", "_encode_texts_prefetched", "_encode_fields_prefetched",
                 "_use_graphs"):
        setattr(Stub, name, getattr(RetrievalTrainingModule, name))
    Stub._write_field = staticmethod(RetrievalTrainingModule._write_field)
    stub = Stub()
    stub._graphed = None                                              # (captured forwards are a GPU matter: tests/test_gpu_cli.py)
    stub.encoder, stub.device = enc, torch.device("cpu")
    E = enc.get_sentence_embedding_dimension()
    want = torch.stack([enc.encode([t], batch_size=1, convert_to_tensor=True)[0] for t in uniq])           # every text alone
    alone = torch.full((len(uniq), E), float("nan"))
    assert stub._encode_texts_prefetched(uniq, order, alone, 1, 0, 24, None)      # budget 0, batch size 1: alone
    assert torch.equal(alone, want)
    for bs, budget in ((4, 0), (4, 4 * 24), (16, 16 * 24)):
        got = torch.full((len(uniq), E), float("nan"))
        assert stub._encode_texts_prefetched(uniq, order, got, bs, budget, 24, None)
        assert not torch.isnan(got).any()
        assert torch.allclose(got, want, atol=2e-5, rtol=0), float((got - want).abs().max())
    # several producer chunks (1024, 2048, then 8192 texts): every row still written once, to the generic path's values
    many = list(dict.fromkeys(" ".join(rng.choice(words, size=int(n))) + f" {i}" for i, n in enumerate(rng.integers(1, 6, size=3400))))
    order_m = sorted(range(len(many)), key=lambda i: len(many[i]))
    got = torch.full((len(many), E), float("nan"))
    assert stub._encode_texts_prefetched(many, order_m, got, 64, 64 * 24, 24, None)
    ref = enc.encode(many, batch_size=64, convert_to_tensor=True)
    assert not torch.isnan(got).any() and torch.allclose(got, ref, atol=2e-5, rtol=0)
    # the shape family of captured forwards (mfar/modeling/graphed.py; the GPU half is tests/test_gpu_cli.py): texts per batch from the ladder,
    # lengths in steps of 8 tokens, the budget kept, padding rows attend one token, every text exactly once, same rows as above
    import threading as _th
    from mfar.modeling.graphed import shape_ladder
    assert shape_ladder(64)[0] == 4096 and shape_ladder(64)[-1] == 64 and shape_ladder(64) == sorted(shape_ladder(64), reverse=True)
    assert shape_ladder(8192) == [8192]
    for bs, budget in ((8, 8 * 24), (4, 0), (64, 64 * 24)):
        seen, shapes = [], set()
        got = torch.full((len(many), E), float("nan"))
        for feats, rows in stub._token_batches(many, order_m, bs, budget, 24, _th.Event(), True):
            n, L = feats["input_ids"].shape
            assert n in shape_ladder(bs) and (L % 8 == 0 or L == 24) and L <= 24 and len(rows) <= n
            assert n == bs or n * L <= budget
            assert (feats["attention_mask"].sum(1) >= 1).all() and (feats["attention_mask"][len(rows):].sum(1) == 1).all()
            shapes.add((n, L))
            seen += rows.tolist()
            got[rows] = enc(dict(feats))["sentence_embedding"].float()[:len(rows)].detach()
        assert sorted(seen) == list(range(len(many)))
        assert len(shapes) <= 8
        assert torch.allclose(got, ref, atol=2e-5, rtol=0)
    # all fields through ONE producer: rows land in corpus order in each field's vectors, repeated texts get the one row of their text
    written = {}
    stub.slab = types.SimpleNamespace(dim=E)
    stub.vectors_dict = {k: types.SimpleNamespace(write_block=lambda first, block, k=k: written.setdefault(k, []).append((first, block.clone())))
                         for k in ("a", "b")}
    corpus = {"a": [uniq[i % 7] for i in range(40)], "b": [uniq[(3 * i) % len(uniq)] for i in range(40)]}

    def prepare(field):
        docs = [(100 + i, t) for i, t in enumerate(corpus[field])]
        texts = [t for _, t in docs]
        u = list(dict.fromkeys(texts))
        return docs, texts, u, {t: i for i, t in enumerate(u)}, sorted(range(len(u)), key=lambda i: len(u[i]))
    stub._encode_fields_prefetched([("a", "a"), ("b", "b")], prepare, 1, 0, 24, None)
    index_of = {t: i for i, t in enumerate(uniq)}
    for k in ("a", "b"):
        assert [first for first, _ in written[k]] == [100]
        assert torch.equal(written[k][0][1], torch.stack([want[index_of[t]] for t in corpus[k]])), k
    # a failing forward surfaces here and the producer thread is gone when it does (no batch left behind a full queue)
    import threading

    class Boom(RuntimeError):
        pass
    calls = {"n": 0}
    real_forward = enc.forward

    def failing(features):
        calls["n"] += 1
        if calls["n"] == 3:
            raise Boom("forward failed")
        return real_forward(features)
    enc.forward = failing
    with pytest.raises(Boom):
        stub._encode_texts_prefetched(many, order_m, torch.empty(len(many), E), 8, 0, 24, None)
    enc.forward = real_forward
    assert not [t for t in threading.enumerate() if t.name == "mfar-encode-prefetch"]
    # the tokenizer's own settings are not left changed for the generic path
    ids = enc.tokenize([uniq[-1], "x"])["input_ids"]
    assert ids.shape[1] <= 24 and ids.shape[0] == 2
    monkeypatch.setenv("MFAR_ENCODE_PREFETCH", "0")
    assert not stub._encode_texts_prefetched(uniq, order, alone, 1, 0, 24, None)
    monkeypatch.delenv("MFAR_ENCODE_PREFETCH")
    stub.encoder = types.SimpleNamespace(tokenizer=object())                                               # no Rust backend
    assert not stub._encode_texts_prefetched(uniq, order, alone, 1, 0, 24, None)
