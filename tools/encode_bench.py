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


def write_amazon_dataset(root, n_docs, n_q, seed=0):
    """STaRK-amazon-shaped records (the 8 amazon fields, schema.py:43-53): products come in FAMILIES of 1 .. 12 variants that share a
    template -- title, description, feature bullets, the also_buy / also_view lists -- and differ in ONE token of it (colour / size / pack),
    so a field holds clusters of near-duplicate TEXTS (hence near-duplicate vectors that are not bit-identical) the way a product catalogue
    does; `brand` repeats a few thousand names; every field but title is missing on 30 .. 70 % of the records (-> "", format.py:58-59);
    reviews and Q&A are lists of dicts with the bookkeeping keys format.py drops."""
    import random
    rng = random.Random(seed)
    vocab = [f"w{i}" for i in range(6000)]
    colours = ["black", "white", "red", "blue", "green", "silver", "pink", "grey", "navy", "beige", "gold", "clear"]
    sizes = ["small", "medium", "large", "xl", "2-pack", "4-pack", "8oz", "16oz", "32oz", "twin", "queen", "king"]
    brands = [f"brand{i}" for i in range(max(50, n_docs // 25))]
    p_missing = {"also_buy": 0.6, "also_view": 0.5, "brand": 0.3, "description": 0.4, "feature": 0.35, "qa": 0.7, "review": 0.45}
    words = lambda lo, hi: " ".join(rng.choices(vocab, k=rng.randrange(lo, hi)))
    os.makedirs(root, exist_ok=True)
    docs, titles = [], []
    with open(f"{root}/corpus", "w") as f:
        i = 0
        while i < n_docs:
            fam = min(n_docs - i, 1 + min(11, int(rng.expovariate(0.35))))
            base_title, brand = words(3, 9), rng.choice(brands)
            desc, feats = [words(30, 160)], [words(4, 14) for _ in range(rng.randrange(2, 7))]
            near = [rng.choice(titles) if titles else words(3, 8) for _ in range(rng.randrange(2, 10))]
            have = {k: rng.random() >= p for k, p in p_missing.items()}
            for v in range(fam):
                variant = rng.choice(colours) if v % 2 == 0 else rng.choice(sizes)
                title = f"{brand} {base_title} {variant}"
                body = {"title": title}
                if have["brand"]:
                    body["brand"] = brand
                if have["description"]:
                    body["description"] = [variant + " " + desc[0]]                      # the family's text, one token changed (in front:
                                                                                         # the toy tokenizer is character level and truncates at 512)
                if have["feature"]:
                    body["feature"] = [variant + " " + feats[0]] + feats[1:]
                if have["also_buy"]:
                    body["also_buy"] = near[: max(1, len(near) // 2)]
                if have["also_view"]:
                    body["also_view"] = near
                if rng.random() >= p_missing["review"]:
                    body["review"] = [{"reviewerID": f"r{rng.randrange(10**6)}", "summary": words(2, 8), "reviewText": words(8, 80),
                                       "overall": rng.randrange(1, 6), "verified": True} for _ in range(rng.randrange(1, 4))]
                if rng.random() >= p_missing["qa"]:
                    body["qa"] = [{"questionType": "open-ended", "question": words(4, 14) + "?", "answer": words(3, 30)} for _ in range(rng.randrange(1, 3))]
                docs.append(title)
                if len(titles) < 5000:
                    titles.append(title)
                f.write(f"{i}\t{json.dumps(body)}\n")
                i += 1
    for part in ("train", "val", "test"):
        with open(f"{root}/{part}.queries", "w") as fq, open(f"{root}/{part}.qrels", "w") as fr:
            for j in range(n_q):
                d = rng.randrange(n_docs)
                fq.write(f"{part[0]}{j}\tlooking for {docs[d]}\n")
                fr.write(f"{part[0]}{j}\t0\t{d}\t1\n")


def screen_probe(module, st, warm_passes=10, timed_passes=10):
    """VERDICT r05 items 1 / 'missing 2': the certified screen on ENCODER-PRODUCED vectors.  The slab holds what `on_eval_start` just wrote
    (mean-pooled transformer outputs of the formatted field texts), the queries are the dataset's query texts through the same encoder.
    The default pipeline (screen auto, AUTO-OFF, ROW MODE, dumps: nothing forced) serves them for `warm_passes` passes -- the adaptive
    policy learns from finished launches -- then `timed_passes` are timed; the same with the screen off; bits compared."""
    import numpy as np
    import torch
    from mfar.data.pipeline import NativePipeline
    dm = st.data_module
    dm.setup("test")
    with torch.no_grad():
        x = torch.cat([module.encode_query_batch(b) for loader in dm.test_dataloader()[:1] for b in loader]).contiguous()
    ix, W = module.slab, module._weights()
    Q = 64
    batches = [x[b:b + Q].contiguous() for b in range(0, x.shape[0], Q)]
    torch.cuda.synchronize()

    def serve(passes, keep):
        pl = NativePipeline(ix, W, None, max_batch=Q)
        out = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in range(passes):
            tickets = []
            for j, xb in enumerate(batches):
                tickets.append(pl.submit(xb))
                if j >= pl.lag:
                    r = pl.result(tickets[j - pl.lag])
                    if keep and p == passes - 1:
                        out.append(r)
            for t in tickets[max(0, len(batches) - pl.lag):]:
                r = pl.result(t)
                if keep and p == passes - 1:
                    out.append(r)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        red = pl.n_redone
        pl.close()
        return dt, out, red

    mode0, _ = ix.screen_setting
    # (MFAR_PROBE_EPS_MULT: probes only -- the certificate's bound scaled, e.g. ~0 = every first certificate passes, UNSOUND: what the
    # same vectors would cost without tier 2; the bits line then says whether any result differed)
    ix.set_screen(1, float(os.environ.get("MFAR_PROBE_EPS_MULT", "1")))
    s0, t20 = ix.screen_stats(), ix.tier2_stats()
    t_learn, _, red_learn = serve(warm_passes, False)
    s1, ao1, t21 = ix.screen_stats(), ix.auto_off_info(), ix.tier2_stats()
    t_on, res_on, red_on = serve(timed_passes, True)
    s2, ao2, rm, t22 = ix.screen_stats(), ix.auto_off_info(), ix.row_mode_info(), ix.tier2_stats()
    kern = ix.last_stage1_kernel()
    uniq = [ix.screen_field_info(f)[0] for f in range(ix.n_fields)] if s2["built"] else None
    s2stats = ix.stage2_stats()
    ix.set_screen(0)
    serve(1, False)
    t_off, res_off, _ = serve(timed_passes, True)
    ix.set_screen(mode0)
    same = len(res_on) == len(res_off) and all(torch.equal(a["ids"], b["ids"]) and torch.equal(a["scores"], b["scores"]) for a, b in zip(res_on, res_off))
    nq = x.shape[0]
    F = ix.n_fields
    cone = torch.nn.functional.normalize(x, dim=1)
    rows = torch.from_numpy(ix.read_rows(min(3, F - 1), 0, min(4096, ix.n_rows))).to(x.device)
    rn = torch.nn.functional.normalize(rows, dim=1)
    out = {
        "queries": int(nq), "docs": int(ix.n_rows), "fields": int(F), "screened": bool(s2["built"]), "scan_kernel_steady": kern,
        "learning": {"passes": warm_passes, "lists_checked": s1["n_checked"] - s0["n_checked"], "lists_redone_exactly": s1["n_failed"] - s0["n_failed"],
                     "lists_finished_by_tier2": (t21["lists"] - t20["lists"]) - (t21["passed_on_to_exact"] - t20["passed_on_to_exact"]),
                     "launches_redone": red_learn, "queries_per_s": warm_passes * nq / t_learn, "fields_switched_off": len(ao1["off"])},
        "steady": {"passes": timed_passes, "lists_checked": s2["n_checked"] - s1["n_checked"], "lists_redone_exactly": s2["n_failed"] - s1["n_failed"],
                   "lists_finished_by_tier2": (t22["lists"] - t21["lists"]) - (t22["passed_on_to_exact"] - t21["passed_on_to_exact"]),
                   "tier2_armed": t22["armed"],
                   "launches_redone": red_on, "queries_per_s": timed_passes * nq / t_on,
                   "fields_switched_off": len(ao2["off"]), "inline_repair": bool(ao2["inline_repair"]),
                   "row_mode_fields_eligible": len(rm["eligible"]), "row_mode_fields_active": len(rm["active"])},
        "screen_off": {"queries_per_s": timed_passes * nq / t_off},
        "ratio_to_screen_off": t_off / t_on,
        "ids_and_score_bits_identical_to_screen_off": bool(same),
        "unique_rows_per_field": uniq,
        "stage2_two_level": bool(s2stats["two_level"]),
        "vector_geometry": {"mean_cosine_between_queries": float((cone @ cone.T).mean()),
                            "mean_cosine_between_rows_of_one_field": float((rn[:512] @ rn[:512].T).mean()),
                            "row_norm_mean": float(rows.norm(dim=1).mean()), "query_norm_mean": float(x.norm(dim=1).mean())},
    }
    denom = max(1, out["steady"]["lists_checked"])
    out["steady"]["certified_fraction"] = 1.0 - out["steady"]["lists_redone_exactly"] / denom if out["steady"]["lists_checked"] else None
    out["steady"]["certified_by_the_first_certificate"] = 1.0 - (t22["lists"] - t21["lists"] + out["steady"]["lists_redone_exactly"]) / denom \
        if out["steady"]["lists_checked"] else None
    out["what"] = ("lists_checked = (query, field) lists the screen answered; lists_redone_exactly = of those, sent to the exact fp32 pass; "
                   "lists_finished_by_tier2 = first certificate failed, finished by the threshold rescan (include/mfar_hip.h mfar_set_tier2); "
                   "certified_fraction = 1 - redone / checked")
    return out


def run(n_docs=50000, n_queries=256, model="random-init:768x12", sweep=True, quiet=True, dataset="prime", modes=("fp32", "bf16", "fp16"),
        probe=True, hooks=()):
    """dataset: "prime" (22 fields, `write_dataset`) or "amazon" (8 fields, templated product families: `write_amazon_dataset`);
    modes: the corpus encodes to run, in order ("fp32" = the reference's precision for the corpus encode, "bf16" / "fp16" =
    MFAR_ENCODE_AUTOCAST); probe: search every encoded slab (`screen_probe`); hooks: callables (module, st) -> dict run on every encoded
    slab, reported under their __name__ (tools/tier2_population.py)."""
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
            data = f"{tmp}/{dataset}"
            t0 = time.perf_counter()
            (write_amazon_dataset if dataset == "amazon" else write_dataset)(data, n_docs, n_queries)
            t_data = time.perf_counter() - t0
            flags = dict(dataset_name=dataset, lexical_index="unused", out=f"{tmp}/out", temp_dir=f"{tmp}/t", partition="val", data=data, queries=None,
                         corpus=None, additional_partition=None, model_name=model, model_path=None, normalize=False,
                         negative_sampling_params=(100, 50, 1), train_batch_size=8, dev_batch_size=64, train_max_length=512, dev_max_length=512,
                         seed=0xdeadbeef, field_names="all_dense", trec_val_freq=0, prefix=False)
            st = _setup.build(flags)
            module = RetrievalTrainingModule(
                encoder=st.encoder, model_id=st.model_id, decoder=None, contrastive_temp=0.05, dev_qrels_path=st.dev_qrels,
                additional_qrels_path=None, corpus_path=f"{st.corpus_dir}/corpus", sparse_scores=None, corpus=st.corpus, dataset_name=dataset,
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
                os.environ["MFAR_ENCODE_AUTOCAST"] = autocast or "fp32"
                module.mark_encoder_updated()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                module.on_eval_start()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                module.qres_output.close()
                return dt

            import numpy as np
            slab = None
            probe_field = min(3, len(fields) - 1)           # prime: field 3 = `details`, amazon: `description` -- the long texts
            ref_rows = None
            out = {"dataset": dataset + "-shaped synthetic TREC records", "docs": n_docs, "fields": len(fields),
                   "encoder": model + " (BERT-base shape, RANDOM init: no checkpoint exists on either box), toy character-level tokenizer "
                                      "(tokens/s are tokens of THAT tokenizer)",
                   "sequences_per_encode": n_seq, "tokens_per_encode": n_tok, "document_fields": n_docs * len(fields),
                   "distinct_text_share": n_seq / float(n_docs * len(fields)), "dataset_written_s": t_data, "per_field": per_field,
                   "what": "on_eval_start: every DISTINCT text of a field encoded once (length-sorted, token-budget batches), rows written straight "
                           "into the HBM slab; the slab rows are fp32 in every mode.  certified_screen: the encoded slab searched with "
                           "encoder-produced query vectors by the default pipeline (screen auto, adaptive policy on, nothing forced)"}
            for mode in modes:
                dt = encode("" if mode == "fp32" else mode)
                slab = module.slab
                rows = slab.read_rows(probe_field, 0, min(2048, n_docs)).copy()
                key = "fp32" if mode == "fp32" else "autocast_" + mode
                while key in out:                # the same mode again (--modes bf16,bf16,bf16: what a RE-encode costs once forwards are captured)
                    key += "_again"
                out[key] = {"seconds": dt, "docs_per_s": n_docs / dt, "sequences_per_s": n_seq / dt, "tokens_per_s": n_tok / dt}
                gf = getattr(module, "_graphed", None)
                out[key]["captured_forwards"] = None if gf is None else {
                    "graphs_held": len(gf.graphs), "replays_so_far": gf.n_replays, "eager_forwards_so_far": gf.n_eager, "capture_failed": gf.failed,
                    "what": "mfar/modeling/graphed.py: forwards of a shape seen before are hipGraph replays (captures are inside `seconds`)"}
                if mode == "fp32":
                    ref_rows = rows
                elif ref_rows is not None:
                    denom = float(np.abs(ref_rows).max()) or 1.0
                    out[key]["speedup"] = out["fp32"]["seconds"] / dt
                    out[key]["max_abs_row_difference_relative_to_max_abs_value"] = float(np.abs(rows - ref_rows).max()) / denom
                    c = rows - rows.mean(0)
                    cr = ref_rows - ref_rows.mean(0)
                    out[key]["max_abs_row_difference_relative_to_max_abs_CENTRED_value"] = float(np.abs(c - cr).max()) / (float(np.abs(cr).max()) or 1.0)
                if probe:
                    out[key]["certified_screen"] = screen_probe(module, st)
                for h in hooks:
                    out[key][h.__name__] = h(module, st)
            os.environ["MFAR_ENCODE_AUTOCAST"] = ""
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
                first = out.get("fp32") or out.get("autocast_" + modes[0])
                out["mask_sweep"] = {"evaluations": len(runs), "queries": n_queries, "one_pass_sweep_s": t_one, "one_test_per_mask_s": t_each,
                                     "sweep_path_taken": bool(ok), "same_results_file": bool(one) and one == each,
                                     "corpus_encodes": "1 for all evaluations (the reference: one per evaluation = %d x %.1f s)" % (len(runs), first["seconds"])}
            return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=50000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--model", default="random-init:768x12")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--dataset", choices=["prime", "amazon"], default="prime")
    ap.add_argument("--modes", default="fp32,bf16,fp16")
    ap.add_argument("--no-probe", action="store_true")
    a = ap.parse_args()
    import bench
    res = run(a.docs, a.queries, a.model, sweep=not a.no_sweep, quiet=False, dataset=a.dataset, modes=tuple(a.modes.split(",")), probe=not a.no_probe)
    res["source_hash"] = bench.source_hash()
    print(json.dumps(res))
