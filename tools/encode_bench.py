#!/usr/bin/env python3
"""SURVEY 8(f1) in numbers: the corpus-encode pipeline (`on_eval_start`, reference mfar/modeling/contrastive.py:465-496) and the reuse of
its result by a field-masking sweep (reference mfar/commands/mask_fields.py:143-170), on a synthetic STaRK-prime-shaped TREC dataset with
the FULL prime field set (22 dense fields, schema.py:11-53) and a BERT-base-shaped encoder (12 layers x 768, randomly initialised: no
checkpoint can be downloaded on either box; character-level toy tokenizer, mfar/modeling/util.py `random-init:768x12`).

    python tools/encode_bench.py [--docs 50000 --queries 256]          -> one JSON object on stdout
    bench.py imports `run()` for its `encode_pipeline` leg (same numbers inside the driver-run line).

Reported: documents / distinct texts (= sequences encoded: every distinct text of a field once) / tokens per second of one corpus encode in
fp32 and under MFAR_ENCODE_AUTOCAST=bf16 (rows written to the slab stay fp32), the largest difference the bf16 forward makes to a query's
mixed scores, and the wall clock of the 2 F + 2 = 46 evaluations of the sweep in ONE pass over the queries (`test_sweep`) against one
`test()` per mask -- both over the corpus encoded ONCE (the reference re-encodes it for every one of the 46 `trainer.test` calls).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "multifield-adaptive-retrieval_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

RELATIONS = ["associated with", "carrier", "contraindication", "enzyme", "expression absent", "expression present", "indication",
             "interacts with", "linked to", "off-label use", "parent-child", "phenotype absent", "phenotype present", "ppi", "side effect",
             "synergistic interaction", "target", "transporter"]


def write_dataset(root, n_docs, n_q, seed=0):
    """STaRK-prime-shaped records: `name` (short), `type` / `source` (ten / eight distinct values), `details` (a dict of free text, on 40 % of
    the records, up to a few hundred words), and the 18 relation fields -- each present on 5 .. 25 % of the records as {node type: [names]}.
    A record that lacks a field is formatted to "" (format.py:58-59)."""
    import random
    rng = random.Random(seed)
    vocab = [f"w{i}" for i in range(4000)]
    types = ["gene/protein", "drug", "disease", "effect/phenotype", "pathway", "anatomy", "molecular_function", "biological_process",
             "cellular_component", "exposure"]
    sources = ["NCBI", "DrugBank", "MONDO", "HPO", "REACTOME", "UBERON", "GO", "CTD"]
    rel_p = {r: rng.uniform(0.05, 0.25) for r in RELATIONS}
    words = lambda lo, hi: " ".join(rng.choices(vocab, k=rng.randrange(lo, hi)))
    os.makedirs(root, exist_ok=True)
    docs = []
    with open(f"{root}/corpus", "w") as f:
        for i in range(n_docs):
            body = {"name": words(1, 5), "type": rng.choice(types), "source": rng.choice(sources)}
            if rng.random() < 0.4:
                body["details"] = {"summary": words(20, 200), "alias": rng.choices(vocab, k=rng.randrange(0, 6))}
            for r in RELATIONS:
                if rng.random() < rel_p[r]:
                    body[r] = {rng.choice(types): [words(1, 4) for _ in range(rng.randrange(1, 12))]}
            docs.append((body["type"], body["name"]))
            f.write(f"{i}\t{json.dumps(body)}\n")
    for part in ("train", "val", "test"):
        with open(f"{root}/{part}.queries", "w") as fq, open(f"{root}/{part}.qrels", "w") as fr:
            for j in range(n_q):
                d = rng.randrange(n_docs)
                fq.write(f"{part[0]}{j}\twhich {docs[d][0]} is {docs[d][1]}\n")
                fr.write(f"{part[0]}{j}\t0\t{d}\t1\n")


def run(n_docs=50000, n_queries=256, model="random-init:768x12", sweep=True, quiet=True):
    import contextlib
    import io
    import torch
    from mfar.commands import _setup
    from mfar.data.format import format_documents
    from mfar.data.typedef import FieldType
    from mfar.modeling.contrastive import RetrievalTrainingModule
    tmp = tempfile.mkdtemp(prefix="mfar_encode_bench_")
    sink = io.StringIO()
    try:
        with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
            data = f"{tmp}/prime"
            t0 = time.perf_counter()
            write_dataset(data, n_docs, n_queries)
            t_data = time.perf_counter() - t0
            flags = dict(dataset_name="prime", lexical_index="unused", out=f"{tmp}/out", temp_dir=f"{tmp}/t", partition="val", data=data, queries=None,
                         corpus=None, additional_partition=None, model_name=model, model_path=None, normalize=False,
                         negative_sampling_params=(100, 50, 1), train_batch_size=8, dev_batch_size=64, train_max_length=512, dev_max_length=512,
                         seed=0xdeadbeef, field_names="all_dense", trec_val_freq=0, prefix=False)
            st = _setup.build(flags)
            module = RetrievalTrainingModule(
                encoder=st.encoder, model_id=st.model_id, decoder=None, contrastive_temp=0.05, dev_qrels_path=st.dev_qrels,
                additional_qrels_path=None, corpus_path=f"{st.corpus_dir}/corpus", sparse_scores=None, corpus=st.corpus, dataset_name="prime",
                encoder_learning_rate=1e-5, weights_learning_rate=5e-2, weight_decay=0.0, dev_batch_size=64, out_dir=f"{tmp}/out",
                field_info=st.field_info, indices_dict=st.indices_dict, vectors_dict=st.vectors_dict, trec_val_freq=0, freeze_encoder=False,
                query_cond=True, prefix=False, use_batchnorm=False)
            module.to(st.device)
            module.eval()
            # a non-trivial gate, so that masking a field changes the ranking (the layer starts at ones: weighting.py:14)
            with torch.no_grad():
                g = torch.Generator().manual_seed(1)
                module.mixture_of_fields_layer.weight.copy_(0.05 * torch.randn(module.mixture_of_fields_layer.weight.shape, generator=g))
            fields = [f for f in st.field_info.values() if f.field_type == FieldType.DENSE]
            # what one encode has to do: distinct texts per field, their tokens (each field truncates at the encoder's limit)
            tok, max_len = st.tokenizer, int(st.encoder.get_max_seq_length())
            n_seq, n_tok, per_field = 0, 0, {}
            for f in fields:
                uniq = list(dict.fromkeys(t for _, t in format_documents(st.corpus, f.name, f.dataset)))
                lens = [len(x) for x in tok(uniq, padding=False, truncation=True, max_length=max_len)["input_ids"]]
                per_field[f.name] = {"distinct_texts": len(uniq), "tokens": int(sum(lens))}
                n_seq += len(uniq)
                n_tok += int(sum(lens))

            def encode(autocast):
                os.environ["MFAR_ENCODE_AUTOCAST"] = autocast
                module.mark_encoder_updated()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                module.on_eval_start()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                module.qres_output.close()
                return dt

            t_f32 = encode("")                     # (the first forward also builds the encoder's kernels: < 1 % of this encode)
            slab = module.slab
            probe_rows = slab.read_rows(3, 0, min(2048, n_docs)).copy()          # field 3 = `details`: the long texts
            t_bf16 = encode("bf16")
            rows_bf16 = slab.read_rows(3, 0, min(2048, n_docs))
            import numpy as np
            denom = float(np.abs(probe_rows).max()) or 1.0
            row_diff = float(np.abs(rows_bf16 - probe_rows).max()) / denom
            os.environ["MFAR_ENCODE_AUTOCAST"] = ""
            out = {"docs": n_docs, "fields": len(fields), "encoder": model + " (BERT-base shape, random init), toy character-level tokenizer",
                   "sequences_per_encode": n_seq, "tokens_per_encode": n_tok, "document_fields": n_docs * len(fields),
                   "distinct_text_share": n_seq / float(n_docs * len(fields)),
                   "dataset_written_s": t_data,
                   "fp32": {"seconds": t_f32, "docs_per_s": n_docs / t_f32, "sequences_per_s": n_seq / t_f32, "tokens_per_s": n_tok / t_f32},
                   "autocast_bf16": {"seconds": t_bf16, "docs_per_s": n_docs / t_bf16, "sequences_per_s": n_seq / t_bf16, "tokens_per_s": n_tok / t_bf16,
                                     "speedup": t_f32 / t_bf16, "max_abs_row_difference_relative_to_max_abs_value": row_diff},
                   "per_field": per_field,
                   "what": "on_eval_start: every DISTINCT text of a field encoded once (length-sorted, token-budget batches), rows written straight "
                           "into the HBM slab; the slab rows are fp32 in both modes"}
            if sweep:
                dm = st.data_module
                n_f = len(st.field_info)
                runs = [[]] + [[i] for i in range(n_f)] + [list(range(n_f))] + \
                       [[i for i, f in enumerate(st.field_info.values()) if f.name == name] for name in sorted({f.name for f in st.field_info.values()})]
                torch.cuda.synchronize()              # (both sweeps reuse the rows of the last encode)
                t0 = time.perf_counter()
                ok = module.test_sweep(dm, runs)
                torch.cuda.synchronize()
                t_one = time.perf_counter() - t0
                one = open(f"{tmp}/out/results_dicts-all-0.jsonl").read() if os.path.exists(f"{tmp}/out/results_dicts-all-0.jsonl") else ""
                if os.path.exists(f"{tmp}/out/results_dicts-all-0.jsonl"):
                    os.remove(f"{tmp}/out/results_dicts-all-0.jsonl")
                t0 = time.perf_counter()
                for r in runs:
                    module.mask_field(r)
                    module.test(dm)
                torch.cuda.synchronize()
                t_each = time.perf_counter() - t0
                each = open(f"{tmp}/out/results_dicts-all-0.jsonl").read() if os.path.exists(f"{tmp}/out/results_dicts-all-0.jsonl") else ""
                out["mask_sweep"] = {"evaluations": len(runs), "queries": n_queries, "one_pass_sweep_s": t_one, "one_test_per_mask_s": t_each,
                                     "sweep_path_taken": bool(ok), "same_results_file": bool(one) and one == each,
                                     "corpus_encodes": "1 for all evaluations (the reference: one per evaluation = %d x %.1f s)" % (len(runs), t_f32)}
            return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=50000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--model", default="random-init:768x12")
    ap.add_argument("--no-sweep", action="store_true")
    a = ap.parse_args()
    import bench
    res = run(a.docs, a.queries, a.model, sweep=not a.no_sweep, quiet=False)
    res["source_hash"] = bench.source_hash()
    print(json.dumps(res))
