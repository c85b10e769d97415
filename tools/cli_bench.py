#!/usr/bin/env python3
"""Wall-clock profile of the CLI path around the scorer (corpus-encode pipeline and mask_fields reuse), on a synthetic
STaRK-prime-shaped TREC dataset (no dataset or checkpoint can be downloaded on either box; the encoder is a randomly initialised
BERT of the asked size):

  encode   `train.main(max_epochs=0)` = corpus encode + one evaluation, with the token-budget batching of on_eval_start on and off
           (MFAR_ENCODE_TOKEN_BUDGET), reference mfar/modeling/contrastive.py:465-496;
  sweep    `mask_fields.main` over all fields with the one-pass mask sweep on and off (MFAR_MASK_SWEEP), reference
           mfar/commands/mask_fields.py:143-170 (2 F + 2 evaluations, each re-encoding the corpus in the reference).

Prints one JSON line per measurement, each carrying the kernel source hash (bench.source_hash()), so the numbers quoted in
DESIGN.md can be traced to a file under profiles/.
    python tools/cli_bench.py [--docs 20000 --queries 256 --model random-init:768x12 --fields details_dense,name_dense,source_dense,type_dense]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
sys.path.insert(0, ROOT)


def write_dataset(root, n_docs, n_q, seed=0):
    import numpy as np
    rng = np.random.default_rng(seed)
    vocab = [f"w{i}" for i in range(4000)]
    types = ["gene/protein", "drug", "disease", "effect/phenotype", "pathway", "anatomy", "molecular_function", "biological_process",
             "cellular_component", "exposure"]
    sources = ["NCBI", "DrugBank", "MONDO", "HPO", "REACTOME", "UBERON", "GO", "CTD"]
    os.makedirs(root, exist_ok=True)
    docs = []
    with open(f"{root}/corpus", "w") as f:
        for i in range(n_docs):
            body = {"name": " ".join(rng.choice(vocab, int(rng.integers(1, 5)))), "type": types[int(rng.integers(0, len(types)))],
                    "source": sources[int(rng.integers(0, len(sources)))]}
            if rng.random() < 0.6:          # most STaRK-prime records carry a long free-text `details` dict; the rest lack it
                body["details"] = {"summary": " ".join(rng.choice(vocab, int(rng.integers(20, 300)))),
                                   "alias": [str(w) for w in rng.choice(vocab, int(rng.integers(0, 6)))]}
            docs.append(body)
            f.write(f"{i}\t{json.dumps(body)}\n")
    for part in ("train", "val", "test"):
        with open(f"{root}/{part}.queries", "w") as fq, open(f"{root}/{part}.qrels", "w") as fr:
            for j in range(n_q):
                d = int(rng.integers(0, n_docs))
                fq.write(f"{part[0]}{j}\twhich {docs[d]['type']} is {docs[d]['name']}\n")
                fr.write(f"{part[0]}{j}\t0\t{d}\t1\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--model", default="random-init:768x12")
    ap.add_argument("--fields", default="details_dense,name_dense,source_dense,type_dense")
    ap.add_argument("--skip-sweep", action="store_true")
    a = ap.parse_args()
    import torch
    import bench
    from mfar.commands import mask_fields, train
    tmp = tempfile.mkdtemp(prefix="mfar_cli_bench_")
    data = f"{tmp}/prime"
    write_dataset(data, a.docs, a.queries)
    common = dict(dataset_name="prime", lexical_index="unused", data=data, model_name=a.model, field_names=a.fields, weights_lr=5e-2,
                  encoder_lr=1e-5, train_batch_size=8, dev_batch_size=64, precision="32")
    meta = {"source_hash": bench.source_hash(), "docs": a.docs, "queries": a.queries, "model": a.model, "fields": a.fields,
            "gpu": torch.cuda.get_device_name(0)}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    # corpus encode + one evaluation, token-budget batching on / off (a first run warms the encoder build and the allocator)
    for i, budget in enumerate(("1", "1", "0")):
        os.environ["MFAR_ENCODE_TOKEN_BUDGET"] = budget
        dt, _ = timed(lambda: train.main(out=f"{tmp}/enc{i}", temp_dir=f"{tmp}/t{i}", max_epochs=0, **common))
        if i:
            print(json.dumps({"what": "train.main(max_epochs=0): corpus encode + one evaluation", "token_budget_batching": budget == "1",
                              "seconds": round(dt, 3), **meta}), flush=True)
    os.environ["MFAR_ENCODE_TOKEN_BUDGET"] = "1"
    if a.skip_sweep:
        return
    # one epoch of training for a checkpoint, then the mask sweep both ways
    dt, _ = timed(lambda: train.main(out=f"{tmp}/ck", temp_dir=f"{tmp}/tck", max_epochs=1, **common))
    print(json.dumps({"what": "train.main(max_epochs=1)", "seconds": round(dt, 3), **meta}), flush=True)
    for sweep in ("1", "0"):
        os.environ["MFAR_MASK_SWEEP"] = sweep
        dt, _ = timed(lambda: mask_fields.main(out=f"{tmp}/mf{sweep}", temp_dir=f"{tmp}/tmf{sweep}", checkpoint_dir=f"{tmp}/ck",
                                               **{k: v for k, v in common.items() if k not in ("weights_lr", "encoder_lr")}))
        n_runs = sum(1 for _ in open(f"{tmp}/mf{sweep}/results_dicts-all-0.jsonl"))
        print(json.dumps({"what": "mask_fields.main", "one_pass_sweep": sweep == "1", "evaluations": n_runs, "seconds": round(dt, 3), **meta}),
              flush=True)
    same = open(f"{tmp}/mf1/results_dicts-all-0.jsonl").read() == open(f"{tmp}/mf0/results_dicts-all-0.jsonl").read()
    print(json.dumps({"what": "sweep on/off leave the same results_dicts-all-0.jsonl", "same": same, **meta}), flush=True)


if __name__ == "__main__":
    main()
