"""`python -m mfar.commands.train ...` -- contrastive fine-tuning + final TREC evaluation (reference
mfar/commands/train.py), without PyTorch-Lightning.

Same keyword-only flags as the reference `main` (pinned by tests/golden/cli_signatures.json) and the same output files:
`{out}/epoch=E-valid_loss=X.ckpt`, `last.ckpt` (Lightning key layout, readable by `mask_fields`), `best.txt`,
`{rank}.qres`, `final-all-0.qres`, `results_dicts-all-0.jsonl`.  The evaluation after (and, with --trec_val_freq, during)
training is the accelerated path: corpus encode into the HBM slab + `mfar_search_two_stage`.

Training itself is stock PyTorch-ROCm (out of scope for kernels): one process per GPU, two AdamW optimisers
(encoder / field weights, contrastive.py:305-374), `HybridContrastiveLoss` with in-batch negatives and the autograd-aware
all-gather, early stopping on the proxy validation loss (train.py:225-240).  Hard negatives are mined like the reference
does (contrastive.py:71-77: `IndexNegativeSampler` over the BM25 index `{lexical_index}/single_sparse_sparse_index`, built by
`python -m mfar.commands.create_bm25s_index`) when that index exists; without it the negative of an instance is a random
corpus document.  Sparse fields contribute BM25 score columns to the loss (`--sparse_scores_path`: precomputed scores in
the reference's `{field}_keys_bm25.npy / {field}_vals_bm25.npy` layout, else computed on the fly).
"""
import json
import os
import random
import time
from typing import *  # noqa: F401,F403

import torch

from mfar.commands import _setup
from mfar.commands._cli import run
from mfar.data import trec
from mfar.data.format import format_documents
from mfar.data.typedef import FieldType, Query
from mfar.modeling.contrastive import RetrievalTrainingModule
from mfar.modeling.losses import HybridContrastiveLoss


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _autocast(precision: str, device):
    if precision.startswith("bf16"):
        return torch.autocast(device_type="cuda", dtype=torch.bfloat16)
    if precision.startswith("16"):
        return torch.autocast(device_type="cuda", dtype=torch.float16)
    return torch.autocast(device_type="cuda", enabled=False)


class _Instances:
    """(query, positive doc, one negative doc) triples from `<partition>.queries` / `<partition>.qrels`."""

    def __init__(self, queries_path, partition, corpus, seed, sampler=None):
        self.queries = dict(trec.read_corpus(f"{queries_path}/{partition}.queries"))
        with open(f"{queries_path}/{partition}.qrels") as f:
            self.qrels = [r for r in trec.QRels.from_text_io(f) if r.relevance > 0 and r.query_id in self.queries]
        self.corpus = corpus
        self.key_to_row = {k: i for i, (k, _) in enumerate(corpus)}
        self.qrels = [r for r in self.qrels if r.doc_id in self.key_to_row]
        self.seed = seed
        self.rng = random.Random(seed)
        # BM25-mined hard negatives (negative_sampler.py:40-60); positives of a query = all its relevant documents
        self.sampler = sampler
        self.pos_for_each_qid = {}
        for r in self.qrels:
            self.pos_for_each_qid.setdefault(r.query_id, set()).add(r.doc_id)

    def batches(self, batch_size, rank, world, shuffle, fixed_negatives=False):
        """This rank's share of every global batch: (qrels rows, negative corpus rows).

        Every instance is covered: like Lightning's DistributedSampler (drop_last=False) the instance list is padded to a multiple
        of the world size by repeating instances from its start, so all ranks run the SAME number of steps (the loss all-gathers
        and all-reduces) and the last step is simply smaller; nothing of train.qrels / val.qrels is silently dropped.
        Negatives are mined for THIS rank's slice only (the reference mines inside each rank's dataloader workers,
        negative_sampler.py:40-60): the draw of an instance comes from its own generator seeded with (seed, pass, position in the
        pass), so it does not depend on the number of ranks or on what other ranks drew; the BM25 candidate list of a query
        (retrieve, drop positives, keep the n_bottom lowest) is a pure function of the query and is computed once."""
        order = list(range(len(self.qrels)))
        if not order:
            return
        if shuffle:
            self.rng.shuffle(order)                      # same seed on every rank -> same order
        n_pass = getattr(self, "_n_pass", 0)
        if not fixed_negatives:
            self._n_pass = n_pass + 1
        pad = (-len(order)) % world
        order = order + [order[i % len(order)] for i in range(pad)]
        # validation passes `fixed_negatives`: the same draws every epoch, so valid_loss values are comparable between epochs
        # (early stopping and best-checkpoint selection rest on them)
        pass_key = -1 if fixed_negatives else n_pass
        G = batch_size * world
        for g in range(0, len(order), G):
            cur = order[g:g + G]
            per = len(cur) // world
            lo = rank * per
            rows = [self.qrels[i] for i in cur[lo:lo + per]]
            negs = [self._negative(r, random.Random(((self.seed * 1000003 + pass_key) * 1000003 + g + lo + j) & 0xFFFFFFFFFFFF))
                    for j, r in enumerate(rows)]
            yield rows, negs

    def _negative(self, r, rng) -> int:
        """Corpus row of one negative for qrels row `r`, drawn from `rng`."""
        sampler = getattr(self, "sampler", None)
        if sampler is not None:
            cache = self.__dict__.setdefault("_bottom_cache", {})
            bottom = cache.get(r.query_id)
            if bottom is None:
                bottom = cache[r.query_id] = sampler.bottom_candidates(Query(r.query_id, self.queries[r.query_id]), self.pos_for_each_qid)
            return self.key_to_row[sampler.pick(bottom, rng)[0]]
        n = rng.randrange(len(self.corpus))
        while self.corpus[n][0] == r.doc_id and len(self.corpus) > 1:
            n = rng.randrange(len(self.corpus))
        return n


def _encode_fields(module, tokenizer, docs, max_length, device):
    """[(id, json)] -> [B, F, E] (contrastive.py:412-414 runs one encoder forward per dense field).  Here the F x B texts are
    tokenised per field (each field keeps its own truncation length), sorted by length and encoded in as few forwards as a
    token budget allows: the fields of STaRK records are mostly short texts, and 2 F forwards of B short texts per training
    step are launch-bound (22 fields: 44 forwards + 44 small backward graphs per step).  A text's embedding does not depend
    on what it is batched with (padding is masked), so the loss is the reference's; MFAR_TRAIN_FUSED_ENCODE=0 restores one
    forward per field."""
    fields = [f for f in module.field_info.values() if f.field_type == FieldType.DENSE]
    if not fields:                                 # an all-sparse field set: no dense columns
        return torch.zeros(len(docs), 0, module.encoder.get_sentence_embedding_dimension(), device=device)
    B, F = len(docs), len(fields)
    enc_max = module.encoder.get_max_seq_length()
    per_field = []
    for field in fields:
        texts = [t for _, t in format_documents(docs, field.name, field.dataset)]
        if module.prefix:
            texts = [field.name + ": " + t for t in texts]
        per_field.append((texts, min(max_length, field.max_seq_length, enc_max)))
    if os.environ.get("MFAR_TRAIN_FUSED_ENCODE", "1") == "0":
        outs = []
        for texts, L in per_field:
            toks = tokenizer(texts, padding=True, truncation=True, max_length=L, return_tensors="pt")
            outs.append(module.encoder({k: v.to(device) for k, v in toks.items()})["sentence_embedding"])
        return torch.stack(outs, dim=1)
    seqs = []                                      # token ids of text (field f, document b) at index f * B + b
    for texts, L in per_field:
        seqs += tokenizer(texts, padding=False, truncation=True, max_length=L)["input_ids"]
    order = sorted(range(len(seqs)), key=lambda i: len(seqs[i]))
    budget = max(B * enc_max, 8192)                # padded tokens per forward: what ONE per-field forward may already take
    out = [None] * len(seqs)
    pos = 0
    while pos < len(order):
        n = 1
        while pos + n < len(order) and (n + 1) * len(seqs[order[pos + n]]) <= budget:
            n += 1
        idx = order[pos:pos + n]
        toks = tokenizer.pad({"input_ids": [seqs[i] for i in idx]}, padding=True, return_tensors="pt")
        emb = module.encoder({k: v.to(device) for k, v in toks.items() if k in ("input_ids", "attention_mask", "token_type_ids")})["sentence_embedding"]
        for j, i in enumerate(idx):
            out[i] = emb[j]
        pos += n
    return torch.stack(out).view(F, B, -1).transpose(0, 1)


def _sparse_columns(module, inst, rows, negs, device, sparse_scores):
    """BM25 score columns of the loss for this rank's batch (losses.py:303-324): (pos [B, ws*B, Fs], neg [B, ws*B, Fs],
    rev [ws*B, B, Fs]) or (None, None, None) without sparse fields.  Texts / keys of the other ranks come through
    `all_gather_object` like in the reference (losses.py:262-273)."""
    sparse = [(k, module.indices_dict[k]) for k, f in module.field_info.items() if f.field_type == FieldType.SPARSE]
    if not sparse:
        return None, None, None
    import torch.distributed as dist
    queries = [inst.queries[r.query_id] for r in rows]
    qids = [r.query_id for r in rows]
    pos_keys = [r.doc_id for r in rows]
    neg_keys = [inst.corpus[n][0] for n in negs]
    all_q, all_qids, all_pos, all_neg = queries, qids, pos_keys, neg_keys
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        def gather(x):
            out = [None] * dist.get_world_size()
            dist.all_gather_object(out, x)
            return [y for part in out for y in part]
        all_q, all_qids, all_pos, all_neg = gather(queries), gather(qids), gather(pos_keys), gather(neg_keys)

    def cols(qs, ids, keys):
        per_field = []
        for key, si in sparse:
            cache = (sparse_scores or {}).get(key)
            if cache and all(_qkey(i) in cache for i in ids):
                per_field.append(si.score_batch_with_cache([_qkey(i) for i in ids], keys, cache).float())
            else:
                per_field.append(si.score_batch(qs, keys).float())
        return torch.stack(per_field, dim=-1).to(device)
    return cols(queries, qids, all_pos), cols(queries, qids, all_neg), cols(all_q, all_qids, pos_keys)


def _qkey(qid):
    """Precomputed sparse scores are keyed by the integer query id (precompute_bm25s_scores.py:56)."""
    try:
        return int(qid)
    except (TypeError, ValueError):
        return qid


def read_sparse_scores(scores_path, field_info):
    """{field_key: {query id: {doc number: score}}} from `{scores_path}/{field_key}_keys_bm25.npy` ([n, 2] int: query id,
    doc number) and `{field_key}_vals_bm25.npy` ([n] float16) -- reference modeling/util.py:151-173."""
    import numpy as np
    out = {}
    for key, f in field_info.items():
        if f.field_type != FieldType.SPARSE:
            continue
        keys = np.load(f"{scores_path}/{key}_keys_bm25.npy")
        vals = np.load(f"{scores_path}/{key}_vals_bm25.npy")
        assert len(keys) == len(vals)
        d = {}
        for (qid, doc), v in zip(keys.tolist(), vals.tolist()):
            d.setdefault(qid, {})[doc] = v
        out[key] = d
    return out


def _loss_on_batch(module, loss_fn, tokenizer, inst, rows, negs, max_length, device, precision, sparse_scores=None):
    qt = tokenizer([inst.queries[r.query_id] for r in rows], padding=True, truncation=True,
                   max_length=module.encoder.get_max_seq_length(), return_tensors="pt")
    with _autocast(precision, device):
        q = module.encoder({k: v.to(device) for k, v in qt.items()})["sentence_embedding"]
        d_pos = _encode_fields(module, tokenizer, [inst.corpus[inst.key_to_row[r.doc_id]] for r in rows], max_length, device)
        d_neg = _encode_fields(module, tokenizer, [inst.corpus[n] for n in negs], max_length, device).unsqueeze(2)
    sp_pos, sp_neg, sp_rev = _sparse_columns(module, inst, rows, negs, device, sparse_scores)
    return loss_fn(q.float(), d_pos.float(), d_neg.float(), sp_pos, sp_neg, sp_rev)


def _sync_grads(params, world):
    """Average the gradients over the ranks.  Called on the still-SCALED gradients, before `scaler.unscale_`: the
    GradScaler records inf/NaN per rank at unscale time, so reducing first makes every rank see the same overflow,
    skip the same step and keep the same loss scale (weights stay bit-equal across ranks)."""
    if world == 1:
        return
    import torch.distributed as dist
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)      # a rank whose batch did not touch p must still join the collective
        dist.all_reduce(p.grad)
        p.grad /= world


def _train_step_sync(scaler, opts, params, world):
    """all-reduce (scaled) -> unscale -> step -> update, in that order on every rank."""
    _sync_grads(params, world)
    for o in opts:
        scaler.unscale_(o)
    for o in opts:
        scaler.step(o)
    scaler.update()


def main(
        *,
        dataset_name: str, lexical_index: str, out: str, temp_dir: str, partition: str = "val",
        data: Optional[str] = None, queries: Optional[str] = None, corpus: Optional[str] = None,
        sparse_scores_path: Optional[str] = None, additional_partition: Optional[str] = None,
        model_name: str = "facebook/contriever-msmarco", model_path: Optional[str] = None, normalize: bool = False,
        temperature: float = 0.05, negative_sampling_params: Tuple[int, int, int] = (100, 50, 1),
        encoder_lr: float = 1e-4, weights_lr: Optional[float] = None, regularizer: float = 0.0,
        train_batch_size: int = 16, dev_batch_size: int = 64, train_max_length: int = 512, dev_max_length: int = 512,
        max_epochs: int = 50, patience: int = 10, seed: int = 0xdeadbeef, precision: str = "16-mixed",
        num_gpus: int = -1, dev_by_iter: bool = False, logger: Optional[str] = None, freeze_encoder: bool = False,
        wandb_name: str = None, wandb_dir: str = None, experiment_name: str = None, field_names: List = None,
        trec_val_freq: int = 0, query_cond: bool = True, prefix: bool = False, run_one_iteration=False,
        use_batchnorm: bool = False,
):
    st = _setup.build(locals(), freeze_encoder=freeze_encoder)
    rank, world = _rank_world()
    device, field_info, tokenizer, data_module = st.device, st.field_info, st.tokenizer, st.data_module
    queries, corpus_contents, model_name = st.queries, st.corpus, st.model_id
    os.makedirs(out, exist_ok=True)
    if rank == 0:
        print(f"Starting training: model={model_name} queries={queries} corpus={st.corpus_dir} dataset={dataset_name} "
              f"fields={json.dumps({k: v.__dict__() for k, v in field_info.items()})} prefix={prefix} encoder_lr={encoder_lr} "
              f"weights_lr={weights_lr} seed={seed} time={time.strftime('%Y-%m-%d %H:%M:%S')}")
    print(f"Indices are created for all {len(st.indices_dict)} fields, including {field_info.keys()}")
    module = RetrievalTrainingModule(
        encoder=st.encoder, model_id=model_name, decoder=None, contrastive_temp=temperature, dev_qrels_path=st.dev_qrels,
        additional_qrels_path=st.additional_qrels, corpus_path=f"{st.corpus_dir}/corpus", sparse_scores=None,
        corpus=corpus_contents, dataset_name=dataset_name, encoder_learning_rate=encoder_lr, weights_learning_rate=weights_lr,
        weight_decay=regularizer, dev_batch_size=dev_batch_size, out_dir=out, field_info=field_info,
        indices_dict=st.indices_dict, vectors_dict=st.vectors_dict, trec_val_freq=trec_val_freq, freeze_encoder=freeze_encoder,
        query_cond=query_cond, prefix=prefix, use_batchnorm=use_batchnorm)
    module.to(device)
    module.encode_precision = _setup.encode_precision_for(precision)
    sparse_indices = {k: st.indices_dict[k] for k, f in field_info.items() if f.field_type == FieldType.SPARSE}
    loss_fn = HybridContrastiveLoss(temperature=temperature, mixture_of_fields_layer=module.mixture_of_fields_layer,
                                    sparse_indices_dict=sparse_indices, num_fields=len(field_info), use_batchnorm=use_batchnorm).to(device)
    sparse_scores = read_sparse_scores(sparse_scores_path, field_info) if (sparse_scores_path and sparse_indices) else None
    # hard negatives: BM25 over the whole-document text, as the reference's data module sets up (contrastive.py:71-77)
    sampler = None
    neg_index = f"{lexical_index}/single_sparse_sparse_index"
    if max_epochs > 0 and os.path.exists(f"{neg_index}/keys.json"):
        from mfar.data.index import BM25sSparseIndex
        from mfar.data.negative_sampler import IndexNegativeSampler
        if int(negative_sampling_params[2]) != 1:
            raise NotImplementedError("one negative per instance (n_sample = 1, the reference default)")
        sampler = IndexNegativeSampler(BM25sSparseIndex.load(neg_index), dict(corpus_contents), n_retrieve=int(negative_sampling_params[0]),
                                       n_bottom=int(negative_sampling_params[1]), n_sample=1)
    elif max_epochs > 0 and rank == 0:
        print(f"No BM25 index at {neg_index}: training negatives are random corpus documents")

    enc_params = [p for p in module.encoder.parameters() if p.requires_grad]
    lin_params = [p for p in list(module.mixture_of_fields_layer.parameters()) + list(loss_fn.bn.parameters()) if p.requires_grad]
    opts = []
    if enc_params:
        opts.append(torch.optim.AdamW(enc_params, lr=encoder_lr, weight_decay=regularizer))
    opts.append(torch.optim.AdamW(lin_params, lr=weights_lr))
    scaler = torch.amp.GradScaler("cuda", enabled=precision.startswith("16"))

    best_path, best_loss, bad_epochs, step = "", float("inf"), 0, 0
    train_inst = _Instances(queries, "train", corpus_contents, seed, sampler) if max_epochs > 0 else None
    val_inst = _Instances(queries, "val", corpus_contents, seed + 1, sampler) if max_epochs > 0 else None
    print(f"Starting training: {time.strftime('%Y-%m-%d %H:%M:%S')}")
    for epoch in range(max_epochs):
        module.train()
        loss_fn.train()
        for rows, negs in train_inst.batches(train_batch_size, rank, world, shuffle=True):
            for o in opts:
                o.zero_grad(set_to_none=True)
            loss = _loss_on_batch(module, loss_fn, tokenizer, train_inst, rows, negs, train_max_length, device, precision, sparse_scores)
            scaler.scale(loss).backward()
            _train_step_sync(scaler, opts, enc_params + lin_params, world)
            step += 1
            if rank == 0:
                print(f"Training loss: {loss.item()}")
            if run_one_iteration:
                break
        module.mark_encoder_updated()
        # proxy validation (contrastive.py:647-667): the same loss on the dev qrels
        module.eval()
        loss_fn.eval()
        tot, cnt = 0.0, 0
        with torch.no_grad():
            for rows, negs in val_inst.batches(dev_batch_size if dev_batch_size < 32 else 16, rank, world, shuffle=False,
                                                fixed_negatives=True):
                tot += float(_loss_on_batch(module, loss_fn, tokenizer, val_inst, rows, negs, dev_max_length, device, precision, sparse_scores))
                cnt += 1
                if run_one_iteration:
                    break
        t = torch.tensor([tot, cnt], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(t)
        valid_loss = float(t[0] / max(1.0, float(t[1])))
        module.bn_state = {k: v.detach().cpu() for k, v in loss_fn.bn.state_dict().items()}
        if rank == 0:
            print(f"Validation loss: {valid_loss}")
            ckpt = f"{out}/epoch={epoch}-valid_loss={valid_loss:.3f}.ckpt"
            module.save_checkpoint(ckpt)
            module.save_checkpoint(f"{out}/last.ckpt")
            if valid_loss < best_loss:
                best_path = ckpt
        if trec_val_freq > 0 and (epoch + 1) % trec_val_freq == 0:
            module.test(data_module)
            if rank == 0 and os.path.exists(f"{out}/final-all-0.qres"):
                os.replace(f"{out}/final-all-0.qres", f"{out}/epoch-{step}-all-0.qres")      # contrastive.py:525
        if valid_loss < best_loss:
            best_loss, bad_epochs = valid_loss, 0
        else:
            bad_epochs += 1
            if bad_epochs >= patience:
                break

    if world > 1:
        holder = [best_path]
        torch.distributed.broadcast_object_list(holder, src=0)
        best_path = holder[0]
    if best_path:                                                                            # trainer.test(ckpt_path="best")
        module.load_reference_state_dict(torch.load(best_path, map_location="cpu", weights_only=False)["state_dict"])
    elif rank == 0:
        best_path = f"{out}/last.ckpt"
        module.save_checkpoint(best_path)
    module.test(data_module)
    if rank == 0:
        with open(f"{out}/best.txt", "w") as f:
            f.write(str(best_path))
    return module


if __name__ == "__main__":
    run(main)
