"""End-to-end on the GPU: `train` (2 epochs on a toy TREC dataset, random-init encoder) -> checkpoint -> `mask_fields`
sweep, checking the files the reference writes and that the evaluation equals the oracle run on the same embeddings."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_dataset(root, n_docs=300, n_q=24, seed=0):
    rng = np.random.default_rng(seed)
    words = ["red", "blue", "shoe", "hat", "acme", "zen", "light", "heavy", "wool", "cotton", "alpha", "beta", "gamma", "delta"]
    os.makedirs(root, exist_ok=True)
    docs = []
    with open(f"{root}/corpus", "w") as f:
        for i in range(n_docs):
            body = {"title": " ".join(rng.choice(words, 3)), "brand": str(rng.choice(words)),
                    "feature": [str(w) for w in rng.choice(words, 2)]}
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

    # the evaluation the CLI ran == the oracle on the very same embeddings
    from oracle import mfar_oracle as O
    slab = np.stack([module.slab.read_rows(f) for f in range(3)])
    W = module.mixture_of_fields_layer.weight.detach().cpu().numpy()
    qs = dict(__import__("mfar.data.trec", fromlist=["x"]).read_corpus(f"{data}/val.queries"))
    qids = list(qs)[:5]
    qe = module.encoder.encode([qs[i] for i in qids], convert_to_numpy=True)
    o = O.c_two_stage(slab, qe, W, None)
    got = {}
    for l in lines:
        p = l.split("\t")
        got.setdefault(p[0], []).append((int(p[2]), float(p[4])))
    for i, qid in enumerate(qids):
        ids = [d for d, _ in got[qid]]
        sims = np.array([s for _, s in got[qid]], dtype=np.float32)
        # the CLI encodes the query inside a padded batch, the check encodes it alone: scores agree to fp32 noise
        np.testing.assert_allclose(sims, o["scores"][i], rtol=1e-4, atol=1e-4)
        # (documents that share a text now share one embedding bit for bit, so this tiny corpus has many exact score ties;
        #  the two query embeddings differ in the last bits and resolve a few more boundary near-ties differently)
        assert len(set(ids) & set(o["ids"][i].tolist())) >= 90

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
