"""Evaluation orchestration of the path (reference mfar/modeling/contrastive.py), without PyTorch-Lightning.

`RetrievalDataModule` keeps the query-side loading of the reference's data module (contrastive.py:31-96, 119-135);
`RetrievalTrainingModule` keeps the hooks the CLIs drive -- `on_eval_start` (:465-496), `trec_eval_step` (:669-704),
`mask_field` (:706-714), `merge_qres_and_score` (:566-613), `on_test_epoch_start/end` (:553-556, 616-631) -- plus
`test()`, the loop `trainer.test(...)` ran.  What changed underneath:

  * the corpus is encoded ONCE into the on-HBM slab and re-encoded only when the encoder weights changed
    (`mark_encoder_updated()`); the reference re-encodes the whole corpus for every `trainer.test`, 46 times in a prime
    mask sweep (contrastive.py:553-554, mask_fields.py:143-170);
  * each query batch is encoded once (the reference runs 2F+1 encoder forwards per query: index.py:187,228 and
    contrastive.py:693) and scored by `mfar_search_two_stage` on the GPU;
  * with several ranks every rank holds a ROW SHARD of the corpus and all ranks score the same query batches
    (`ShardedSearcher`); rank 0 writes the result lines.  The reference instead shards the queries over ranks and has every
    rank search the full corpus on its CPU (contrastive.py:184,200,207).  File names and formats are unchanged.
"""
import json
import os
import shutil
from collections import deque
from types import SimpleNamespace
from typing import Dict, List, Optional, TextIO, Tuple

import torch

from mfar.data import trec
from mfar.data.dataset import QueryDataset
from mfar.data.format import format_documents
from mfar.data.index import candidate_encoding_stream
from mfar.data.sharded import shard_bounds
from mfar.data.typedef import Field, FieldType
from mfar.modeling.weighting import LinearWeights

TOP_K = 100  # hard-coded in the reference for both stages (contrastive.py:673,696)


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


class RetrievalDataModule:
    """Query / qrels side of the reference's data module.  The BM25 negative sampler and the contrastive training
    datasets (contrastive.py:71-77, 137-160) are training-data plumbing and are not built here."""

    def __init__(self, tokenizer, queries_path: str, corpus, temp_path: str, dev_partition: str,
                 additional_partition: Optional[str], lexical_index: str, negative_sampling_params: Tuple[int, int, int],
                 dataset_name: str, train_batch_size: int = 64, dev_batch_size: int = 64, query_max_length: int = 64,
                 train_max_length: int = 384, dev_max_length: int = 512, dim: int = 768, field_info: Dict[str, Field] = None,
                 indices_dict=None, prefix: bool = False, trec_val_freq: int = 0):
        self.tokenizer, self.queries_path, self.corpus = tokenizer, queries_path, corpus
        self.dev_partition, self.additional_partition = dev_partition, additional_partition
        self.lexical_index, self.negative_sampling_params = lexical_index, negative_sampling_params
        self.train_batch_size, self.dev_batch_size = train_batch_size, dev_batch_size
        self.query_max_length, self.train_max_length, self.dev_max_length = query_max_length, train_max_length, dev_max_length
        self.field_info, self.indices_dict, self.prefix, self.trec_val_freq = field_info, indices_dict, prefix, trec_val_freq
        self.field_types = {f.field_type for f in field_info.values()}
        self.dev_queries_dict = dict(trec.read_corpus(f"{queries_path}/{dev_partition}.queries"))
        self.additional_queries_dict = ([dict(trec.read_corpus(f"{queries_path}/{additional_partition}.queries"))]
                                        if additional_partition else [])
        self.dev_queries: Optional[QueryDataset] = None
        self.additional_queries: List[QueryDataset] = []

    def setup(self, stage: str = "test") -> None:
        mk = lambda d: QueryDataset(tokenizer=self.tokenizer, queries=d, max_length=self.query_max_length, field_types=self.field_types)
        self.dev_queries = mk(self.dev_queries_dict)
        self.additional_queries = [mk(d) for d in self.additional_queries_dict]

    @staticmethod
    def batches(dataset: QueryDataset, batch_size: int):
        for b in range(0, len(dataset), batch_size):
            yield dataset.collate([dataset[i] for i in range(b, min(len(dataset), b + batch_size))])

    def test_dataloader(self):
        return [self.batches(ds, self.dev_batch_size) for ds in [self.dev_queries] + self.additional_queries]


def _all_gather_cat(t: torch.Tensor) -> torch.Tensor:
    """All ranks' tensors concatenated along dim 0 (rank-major).  gloo stages through the host."""
    import torch.distributed as dist
    world = dist.get_world_size()
    src = t.contiguous()
    if dist.get_backend() == "gloo":
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)
    return out.to(t.device)


def _merge_shard_lists(fid: torch.Tensor, fsc: torch.Tensor, k: int, world: int):
    """[world * Q, F, k] zero-sentinel lists of the row shards -> the global per-field top-k [Q, F, k] in canonical order (score
    desc, doc id asc), padded with (0, 0.0) like index.py:192-193.  A global top-k member is in its shard's local top-k, so the
    merge of the shard lists is the list of the unsharded corpus."""
    Q = fid.shape[0] // world
    ids = fid.view(world, Q, *fid.shape[1:]).permute(1, 2, 0, 3).reshape(Q, fid.shape[1], world * k)
    sc = fsc.view(world, Q, *fsc.shape[1:]).permute(1, 2, 0, 3).reshape(Q, fsc.shape[1], world * k)
    real = sc > 0                                                     # everything else is sentinel padding
    key_s = torch.where(real, sc, torch.full_like(sc, -1.0))
    order = torch.argsort(ids.masked_fill(~real, 2 ** 62), dim=-1, stable=True)           # id asc first ...
    key_s, ids, real = (torch.gather(a, -1, order) for a in (key_s, ids, real))
    order = torch.argsort(key_s, dim=-1, descending=True, stable=True)[..., :k]           # ... then score desc, stably
    key_s, ids, real = (torch.gather(a, -1, order) for a in (key_s, ids, real))
    return ids.masked_fill(~real, 0), key_s.masked_fill(~real, 0.0)


class RetrievalTrainingModule(torch.nn.Module):
    def __init__(self, encoder, model_id: str, decoder, corpus_path: str, corpus: List[Tuple[str, str]], dataset_name: str,
                 dev_qrels_path: str, out_dir: str, sparse_scores: Optional[Dict] = None, contrastive_temp: float = 0.01,
                 encoder_learning_rate: float = 1e-5, weights_learning_rate: Optional[float] = None, weight_decay: float = 0.0,
                 dev_batch_size: int = 32, field_info: Dict = None, indices_dict: Dict = None, vectors_dict: Dict = None,
                 trec_val_freq: int = 0, freeze_encoder: bool = False, query_cond: bool = True, prefix: bool = False,
                 additional_qrels_path: Optional[str] = None, use_batchnorm: bool = False):
        super().__init__()
        if field_info is None:
            raise NotImplementedError("No fields passed in!")
        if weights_learning_rate is None:
            raise ValueError("Need to specify a learning weight for the weights!")
        self.encoder, self.model_id, self.decoder = encoder, model_id, decoder
        self.dataset_name, self.corpus, self.corpus_path = dataset_name, corpus, corpus_path
        self.numeric_ids_to_keys = [x[0] for x in corpus]
        self.keys_to_numeric_ids = {k: i for i, k in enumerate(self.numeric_ids_to_keys)}
        self.indices_dict, self.vectors_dict = indices_dict, vectors_dict
        self.encoder_learning_rate, self.weights_learning_rate, self.weight_decay = encoder_learning_rate, weights_learning_rate, weight_decay
        self.contrastive_temp, self.use_batchnorm = contrastive_temp, use_batchnorm
        self.dev_batch_size, self.dev_qrels_path, self.additional_qrels_path = dev_batch_size, dev_qrels_path, additional_qrels_path
        self.out_dir, self.n_docs, self.field_info = out_dir, len(corpus), field_info
        self.trec_val_freq, self.query_cond, self.prefix, self.freeze_encoder = trec_val_freq, query_cond, prefix, freeze_encoder
        num_fields = len(field_info)
        if query_cond:
            self.mixture_of_fields_layer = LinearWeights(encoder.get_sentence_embedding_dimension(), num_fields, query_cond=True)
        else:
            self.mixture_of_fields_layer = LinearWeights(num_fields, 1)      # contrastive.py:286-287
        # the reference reaches the mixer through the loss object (contrastive.py:694); keep that attribute path
        self.hybrid_contrastive_loss_fn = SimpleNamespace(mixture_of_fields_layer=self.mixture_of_fields_layer)
        self.mask = torch.ones([num_fields, 1])                                # contrastive.py:270
        self.masked_fields_string = ""
        self._qrels_cache = {}                                                   # qrels path -> parsed records (a mask sweep scores 2 F + 2 runs)
        self.best_score = 0.0
        self.qres_output: Optional[TextIO] = None
        self.additional_qres_output: Optional[TextIO] = None
        self._corpus_encoded = False
        self._searcher = None
        self._graphed = None                # captured corpus-encode forwards (mfar/modeling/graphed.py), kept across encodes
        # precision of the CORPUS encode forwards: "fp32" | "fp16" | "bf16" (rows written to the slab are fp32 either way).  A module built
        # directly encodes in fp32; the CLIs set it from their `precision` flag (commands/_setup.py `encode_precision_for`): the reference's
        # default "16-mixed" (train.py:51) -> fp16 autocast.  MFAR_ENCODE_AUTOCAST=fp32|fp16|bf16 overrides either.
        self.encode_precision = "fp32"
        # `hybrid_contrastive_loss_fn.bn.*` of a use_batchnorm run: trained by commands/train.py's loss object, unused at
        # evaluation (contrastive.py:685-694 applies no BatchNorm), carried through checkpoints under the reference's keys
        self.bn_state: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self):
        return next(self.encoder.parameters()).device

    @property
    def slab(self):
        """The HBM row shard shared by the dense fields (None for an all-sparse field set)."""
        for ix in self.indices_dict.values():
            if getattr(ix, "slab", None) is not None:
                return ix.slab
        return None

    @property
    def has_sparse(self) -> bool:
        return any(f.field_type == FieldType.SPARSE for f in self.field_info.values())

    def mark_encoder_updated(self):
        """Call after the encoder weights changed (a training epoch, a checkpoint load): the next evaluation re-encodes."""
        self._corpus_encoded = False

    def _weights(self):
        W = self.mixture_of_fields_layer.weight.detach().float()
        return (W if self.query_cond else W.reshape(-1)).contiguous().to(self.device)

    # ------------------------------------------------------------------ corpus encode (contrastive.py:465-496)
    @torch.no_grad()
    def on_eval_start(self) -> None:
        rank, n = _dist()
        os.makedirs(self.out_dir, exist_ok=True)
        self.qres_output = open(f"{self.out_dir}/{rank}.qres", "w")
        # only the fields we care about, in field_info order (contrastive.py:496)
        self.indices_dict = {k: self.indices_dict[k] for k in self.field_info.keys()}
        if self._corpus_encoded:
            return      # same weights as the last encode: the slab in HBM is still valid (the mask only changes the mixer)
        r0, r1 = shard_bounds(self.n_docs, rank, n)                             # contrastive.py:470
        segment = self.corpus[r0:r1]
        bs = self.dev_batch_size
        dense = [(key, field) for key, field in self.field_info.items() if field.field_type == FieldType.DENSE]   # sparse fields were indexed at start-up (BM25)

        def prepare(field):
            """A field's texts, host side only (on the producer thread of the prefetched path: beside the previous field's forwards)."""
            docs = format_documents(segment, field.name, field.dataset)          # contrastive.py:473-475
            if self.prefix:
                docs = [(i, field.name + ": " + t) for i, t in docs]            # :476-481
            # Encode every DISTINCT text of the field once, shortest first (length-bucketed batches: no padding to the
            # longest document of an arbitrary corpus slice), then gather the rows in corpus order.  A text that occurs more
            # than once -- above all "": documents that lack the field, format.py:58-59 -- gets bit-identical rows, which
            # lets the index treat them as one duplicate group (csrc/mfar_screen.h) instead of thousands of near-ties.
            texts = [t for _, t in docs]
            uniq = list(dict.fromkeys(texts))
            slot = {t: i for i, t in enumerate(uniq)}
            order = sorted(range(len(uniq)), key=lambda i: len(uniq[i]))
            return docs, texts, uniq, slot, order

        # Batches by TOKEN budget, not by count: `dev_batch_size` texts of `max_seq_length` tokens is the largest forward the
        # caller sized memory for; short texts (names, types, the relation fields of STaRK-prime) are launch-bound at 64 per
        # forward, so a batch takes as many of them as fit that budget (at least dev_batch_size; MFAR_ENCODE_TOKEN_BUDGET=0:
        # always dev_batch_size).  Lengths ascend, so the last text of a batch is its longest.
        max_len = int(self.encoder.get_max_seq_length())
        tb = os.environ.get("MFAR_ENCODE_TOKEN_BUDGET", "1")             # 0: always dev_batch_size texts; 1: the default; n > 1: n tokens
        budget = 0 if tb == "0" else (int(tb) if tb.isdigit() and int(tb) > 1 else bs * max_len)
        # Precision of the corpus-encode forwards: `self.encode_precision`, overridden by MFAR_ENCODE_AUTOCAST=fp32|fp16|bf16.
        # The reference encodes the corpus outside its precision plugin (which wraps the steps, not on_test_epoch_start) but with
        # torch.set_float32_matmul_precision("high") (train.py:67, mask_fields.py:52): fp32 matmuls may run with 10-bit-mantissa inputs on
        # its hardware.  gfx950 has no such fp32 mode; fp16 autocast IS that precision here (11-bit significand into fp32 accumulators):
        # measured on the BERT-base-shaped encoder, rows differ by 6e-4 of the largest value (bf16: 6e-3) at 3.1x the fp32 rate
        # (bench.py encode_pipeline).  The rows written to the slab are fp32 either way.
        mode = (os.environ.get("MFAR_ENCODE_AUTOCAST", "") or self.encode_precision or "fp32").lower()
        ac = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(mode)
        if dense and self._prefetch_backend() is not None:
            self._encode_fields_prefetched(dense, prepare, bs, budget, max_len, ac)
            dense = []                                       # (nothing left for the generic loop below)
        for key, field in dense:                             # generic path: a tokenizer without a Rust backend, or MFAR_ENCODE_PREFETCH=0
            docs, texts, uniq, slot, order = prepare(field)
            vec = self.vectors_dict[key]
            emb_u = torch.empty(len(uniq), self.slab.dim, device=self.device)

            def batches():
                pos = 0
                while pos < len(order):
                    n = min(bs, len(order) - pos)
                    if budget:
                        n = min(max(4096, budget // 8), len(order) - pos)
                        while n > bs and n * min(max_len, len(uniq[order[pos + n - 1]]) // 3 + 8) > budget:
                            n = max(bs, (n * 3) // 4)
                        # len // 3 + 8 is an estimate for English prose; digit-heavy, CJK or byte-level texts run ~1 token per
                        # character.  A group that could exceed the budget even then (every character a token) is measured with
                        # the tokenizer and clamped to n * L_real <= budget -- short-text groups never pay this second tokenisation
                        tok = getattr(self.encoder, "tokenizer", None)
                        if n > bs and tok is not None and n * min(max_len, len(uniq[order[pos + n - 1]]) + 2) > budget:
                            lens = [len(x) for x in tok([uniq[i] for i in order[pos:pos + n]], padding=False, truncation=True,
                                                        max_length=max_len)["input_ids"]]
                            longest = 0
                            for j, L in enumerate(lens):                 # largest prefix whose padded size fits (at least bs texts)
                                longest = max(longest, L)
                                if j + 1 > bs and (j + 1) * longest > budget:
                                    n = j
                                    break
                    yield [(i, uniq[i]) for i in order[pos:pos + n]]
                    pos += n

            def stream_all():
                for group in batches():                                          # one forward per group (contrastive.py:483-489)
                    yield from candidate_encoding_stream(self.encoder, group, batch_size=len(group), multiprocess=False,
                                                         show_progress=False, as_tensor=True)
            stream = stream_all()
            got, vecs = [], []
            with torch.autocast(device_type=self.device.type, dtype=ac, enabled=ac is not None):
                for i, v in stream:
                    got.append(i)
                    vecs.append(v)
                    if len(got) == 4096:
                        emb_u[torch.tensor(got, device=self.device)] = torch.stack(vecs).float()
                        got, vecs = [], []
            if got:
                emb_u[torch.tensor(got, device=self.device)] = torch.stack(vecs).float()
            self._write_field(vec, docs, emb_u, torch.tensor([slot[t] for t in texts], device=self.device), bs)
            del emb_u
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        _barrier()                                                               # :491 (no memmap reopen needed)
        self._corpus_encoded = True

    # -- the corpus encode with the host side off the critical path ---------------------------------------------------------------------
    # profiles/r05_p_encode_*: under bf16 autocast the GPU was busy 40 % of an encode; the rest was `PreTrainedTokenizerFast.__call__`'s
    # per-sequence Python (16.5 of 29 s, 3 s of them in Rust), synchronous copies and the preparation of each field's texts.  A PRODUCER
    # thread now does all of that: it formats a field's texts, tokenises chunks of them (ordered by characters) with the Rust tokenizer
    # directly (`backend_tokenizer.encode_batch`: releases the GIL, every text tokenised exactly ONCE), cuts them into batches by TRUE
    # token counts under the same token budget as the generic path, pads with numpy into pinned buffers and queues them -- across field
    # boundaries, so the GPU does not drain between fields; the calling thread copies asynchronously, runs the forwards and scatters the
    # rows.  Same tokens as `encoder.tokenize` (special tokens, longest-first truncation at `max_len`); batch composition differs from
    # the generic path as it does between any two budgets.
    def _prefetch_backend(self):
        """The Rust tokenizer behind `encoder.tokenizer`, or None (no fast tokenizer, or MFAR_ENCODE_PREFETCH=0): generic path."""
        tok = getattr(self.encoder, "tokenizer", None)
        be = getattr(tok, "backend_tokenizer", None)
        if be is None or os.environ.get("MFAR_ENCODE_PREFETCH", "1") == "0" or not hasattr(be, "encode_batch"):
            return None
        return be

    def _token_batches(self, uniq, order, bs, budget, max_len, stop, graph_shapes=False):
        """Producer side: yields (features on the host, rows of `uniq` they belong to) for every text of `uniq`, each exactly once.
        graph_shapes: batches are cut to the shape family captured forwards replay (mfar/modeling/graphed.py): texts per batch from
        `shape_ladder(bs)`, lengths rounded up to 8 tokens; a short last batch is padded with rows of one [PAD] token (features have more
        rows than `rows`: the consumer drops the surplus)."""
        import itertools
        import numpy as np
        from mfar.modeling.graphed import round_len, shape_ladder
        ladder = shape_ladder(bs) if graph_shapes else None
        tok = self.encoder.tokenizer
        be = tok.backend_tokenizer
        pad_id = int(tok.pad_token_id or 0)
        pin = self.device.type == "cuda"
        be.enable_truncation(max_length=max_len)            # (PreTrainedTokenizerFast resets both on its next __call__)
        be.no_padding()
        # Chunks of the (character-ordered) texts, tokenised one chunk at a time.  While a chunk is in the tokenizer no batch is produced, so
        # a chunk must not be more work than the queue ahead of the consumer covers: at most 8192 texts AND about 32 forwards (real word-piece vocabularies: ~8)
        # of characters (profiles/r06_j_encode_halves.txt: 8192 texts of the long `details` field were 0.4 s in the tokenizer behind a
        # queue of 0.1 s -- the GPU drained in every chunk).  A short first chunk: the GPU starts early.
        chars = np.cumsum(np.fromiter((len(uniq[i]) for i in order), np.int64, len(order)))
        per_chunk = max(1, int(os.environ.get("MFAR_ENCODE_CHUNK_BATCHES", "32"))) * max(budget, bs * 32)
        c1 = 0
        while c1 < len(order) and not stop.is_set():
            c0 = c1
            done = int(chars[c0 - 1]) if c0 else 0
            c1 = int(np.searchsorted(chars, done + (per_chunk // 4 if c0 == 0 else per_chunk), side="right"))
            c1 = min(len(order), max(c0 + min(4 * bs, 256), min(c1, c0 + (1024 if c0 == 0 else 8192))))
            idxs = np.asarray(order[c0:c1], np.int64)
            ids = [e.ids for e in be.encode_batch([uniq[i] for i in idxs], add_special_tokens=True)]
            lens = np.fromiter((len(x) for x in ids), np.int64, len(ids))
            flat = np.fromiter(itertools.chain.from_iterable(ids), np.int64, int(lens.sum()))
            starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
            o2 = np.argsort(lens, kind="stable")
            pos = 0
            while pos < len(o2) and not stop.is_set():
                rem = len(o2) - pos
                n = min(bs, rem)
                n_shape = 0
                if ladder is not None:
                    # the largest ladder size whose padded batch fits the budget (the floor `bs` always does: the caller sized memory for
                    # it); fewer texts left than that: the smallest ladder size that holds them
                    for n_shape in (ladder if budget else [bs]):
                        m = min(n_shape, rem)
                        if n_shape == bs or n_shape * round_len(int(lens[o2[pos + m - 1]]), max_len) <= budget:
                            break
                    n = min(n_shape, rem)
                    n_shape = min(v for v in ladder if v >= n and v <= n_shape)
                elif budget:
                    n = min(max(4096, budget // 8), rem)
                    while n > bs and n * int(lens[o2[pos + n - 1]]) > budget:     # lengths ascend: the last text is the longest
                        n = max(bs, (n * 3) // 4)
                sel = o2[pos:pos + n]
                L = max(1, int(lens[sel[-1]]))
                rows_n = n
                if ladder is not None:
                    L, rows_n = round_len(L, max_len), n_shape
                valid = np.zeros((rows_n, L), bool)
                valid[:n] = np.arange(L)[None, :] < lens[sel][:, None]
                arr = np.full((rows_n, L), pad_id, np.int64)
                arr[valid] = flat[(starts[sel][:, None] + np.arange(L)[None, :])[valid[:n]]]
                valid[n:, 0] = True                                               # padding rows attend their one [PAD] token (no 0 / 0)
                feats = {"input_ids": torch.from_numpy(arr), "attention_mask": torch.from_numpy(valid.astype(np.int64))}
                rows = torch.from_numpy(idxs[sel])
                if pin:
                    feats = {k: v.pin_memory() for k, v in feats.items()}
                    rows = rows.pin_memory()
                yield feats, rows
                pos += n

    def _consume(self, produce, on_item, ac) -> None:
        """Runs `produce(put, stop)` on a thread and hands every item it puts to `on_item` on this one, under the autocast setting."""
        import queue
        import threading
        q_out: "queue.Queue" = queue.Queue(maxsize=int(os.environ.get("MFAR_ENCODE_QUEUE", "48")))   # (0.5 MB of pinned tokens per batch)
        stop = threading.Event()

        def run():
            try:
                produce(q_out.put, stop)
                q_out.put(None)
            except BaseException as e:                                # surfaces in the consumer
                q_out.put(e)

        th = threading.Thread(target=run, name="mfar-encode-prefetch", daemon=True)
        th.start()
        was_training = self.encoder.training
        self.encoder.eval()
        try:
            with torch.autocast(device_type=self.device.type, dtype=ac, enabled=ac is not None):
                while True:
                    item = q_out.get()
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    on_item(item)
        finally:
            stop.set()
            while th.is_alive():                                      # (an exception above: free a queue slot so that the producer sees `stop`)
                try:
                    q_out.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
            if was_training:
                self.encoder.train()

    def _forward_rows(self, feats, rows, emb_u, ac=None) -> None:
        dev = self.device
        if self._graphed is not None:
            # (a replay returns the graph's static output: the scatter below reads it on this stream before the next forward is enqueued)
            emb_u[rows.to(dev, non_blocking=True)] = self._graphed(feats, ac)[:rows.shape[0]]
            return
        f = {k: v.to(dev, non_blocking=True) for k, v in feats.items()}
        if "token_type_ids" in (getattr(self.encoder.tokenizer, "model_input_names", None) or ()):
            f["token_type_ids"] = torch.zeros_like(f["input_ids"])
        emb_u[rows.to(dev, non_blocking=True)] = self.encoder(f)["sentence_embedding"].float()[:rows.shape[0]]

    def _use_graphs(self) -> bool:
        """Captured forwards for this encode?  (GPU, eval-mode forwards, MFAR_ENCODE_GRAPHS != 0.)  Creates / revalidates the graph cache."""
        from mfar.modeling.graphed import GraphedForward
        if not GraphedForward.enabled(self.device):
            self._graphed = None
            return False
        if self._graphed is None or self._graphed.encoder is not self.encoder:
            self._graphed = GraphedForward(self.encoder, self.device)
        self._graphed.begin()
        return True

    @torch.no_grad()
    def _encode_texts_prefetched(self, uniq, order, emb_u, bs, budget, max_len, ac) -> bool:
        """One list of distinct texts -> rows of `emb_u` through the producer / consumer pair above.  Returns False (nothing done) when the
        tokenizer has no Rust backend or MFAR_ENCODE_PREFETCH=0: the caller then takes the generic path (`candidate_encoding_stream`)."""
        if self._prefetch_backend() is None or not uniq:
            return False
        graphs = self._use_graphs()

        def produce(put, stop):
            for item in self._token_batches(uniq, order, bs, budget, max_len, stop, graphs):
                put(item)
        self._consume(produce, lambda item: self._forward_rows(item[0], item[1], emb_u, ac), ac)
        return True

    @torch.no_grad()
    def _encode_fields_prefetched(self, dense, prepare, bs, budget, max_len, ac) -> None:
        """Every dense field through ONE producer: field j + 1 is formatted and tokenised while field j's forwards run, and the queue holds
        batches across the boundary -- the GPU does not drain between fields."""
        import numpy as np
        cur = {}
        graphs = self._use_graphs()

        def produce(put, stop):
            for key, field in dense:
                if stop.is_set():
                    return
                docs, texts, uniq, slot, order = prepare(field)
                put(("field", key, docs, len(uniq), np.fromiter((slot[t] for t in texts), np.int64, len(texts))))
                for feats, rows in self._token_batches(uniq, order, bs, budget, max_len, stop, graphs):
                    put(("batch", feats, rows))
                put(("end",))

        def on_item(item):
            if item[0] == "field":
                _, cur["key"], cur["docs"], n_uniq, cur["rows"] = item
                cur["emb"] = torch.empty(n_uniq, self.slab.dim, device=self.device)
            elif item[0] == "batch":
                self._forward_rows(item[1], item[2], cur["emb"], ac)
            else:
                self._write_field(self.vectors_dict[cur["key"]], cur["docs"], cur["emb"], torch.from_numpy(cur["rows"]).to(self.device), bs)
                cur.clear()
        self._consume(produce, on_item, ac)

    @staticmethod
    def _write_field(vec, docs, emb_u, rows, bs) -> None:
        """Rows of the distinct texts -> the field's rows in corpus order, straight into the HBM slab."""
        step = max(bs, 65536)
        for b in range(0, len(docs), step):
            vec.write_block(docs[b][0], emb_u.index_select(0, rows[b:b + step]).contiguous())

    # ------------------------------------------------------------------ scoring (contrastive.py:669-704)
    def _get_searcher(self):
        """The two-deep batch pipeline over this rank's slab (mfar/data/pipeline.py): stage 1 of batch i+1 runs beside the
        tail of batch i; with several ranks it uses the lists-first exchange (two small RCCL all-gathers per batch)."""
        from mfar.data.pipeline import PipelinedSearcher
        qmax = min(64, max(1, int(self.dev_batch_size)))
        if self._searcher is None or self._searcher.ix is not self.slab or self._searcher.Qb != qmax:
            self._searcher = PipelinedSearcher(self.slab, self._weights(), None, k1=TOP_K, k2=TOP_K, sentinel=True,
                                               query_cond=self.query_cond, max_batch=qmax)
        return self._searcher

    @torch.no_grad()
    def encode_query_batch(self, batch) -> torch.Tensor:
        """[Q, E] fp32 embeddings of a collated query batch: ONE encoder forward over the batch-padded tokens sliced to the
        encoder's max length (contrastive.py:688-693).  The reference additionally re-encodes every query 2F times
        through `encoder.encode` (index.py:187,228); here this single embedding serves retrieval, re-scoring and gating."""
        toks = batch.query[FieldType.DENSE]
        L = self.encoder.get_max_seq_length()
        feats = {k: v[:, :L].to(self.device) for k, v in toks.items() if k in ("input_ids", "attention_mask", "token_type_ids")}
        return self.encoder(feats)["sentence_embedding"].float().contiguous()

    def _submit(self, batch, qres_output):
        """Enqueue one query batch (asynchronous); returns what `_collect` needs.  Batches above the pipeline's width are
        split; with several ranks a short (last) batch is padded to the fixed exchange size."""
        ps = self._get_searcher()
        ps.W, ps.mask = self._weights(), self.mask[:, 0].float().contiguous().to(self.device)
        x = self.encode_query_batch(batch)
        parts = []
        for b in range(0, x.shape[0], ps.Qb):
            xb = x[b:b + ps.Qb]
            n = xb.shape[0]
            if ps.sharded and n < ps.Qb:       # fixed exchange size: pad with copies of the last query (results are dropped)
                xb = torch.cat([xb, xb[-1:].expand(ps.Qb - n, -1)])
            parts.append((ps.submit(xb.contiguous()), batch.instances[b:b + n]))
        return parts, qres_output

    def _collect(self, pending) -> None:
        parts, qres_output = pending
        ps = self._get_searcher()
        rank, _ = _dist()
        for ticket, data in parts:
            res = ps.result(ticket)
            n = len(data)
            n_valid = res["n_valid"][:n].cpu().tolist()
            if min(n_valid) < TOP_K:                                             # what torch.topk raises at :696
                raise RuntimeError(f"selected index k out of range: only {min(n_valid)} candidates for k={TOP_K}")
            if rank != 0 or qres_output is None:
                continue
            ids, sims = res["ids"][:n].cpu().tolist(), res["scores"][:n].cpu().tolist()
            keys = self.numeric_ids_to_keys
            # the lines `print(QRes(...), file=...)` writes (trec.py:49-50: "{qid}\t0\t{doc}\t0\t{sim}\t0"), formatted in one
            # pass: 6400 dataclass constructions + prints per batch were the slowest step of an evaluation
            qres_output.write("".join(f"{q._id}\t0\t{keys[d]}\t0\t{s}\t0\n" for q, row_ids, row_sims in zip(data, ids, sims)
                                      for d, s in zip(row_ids, row_sims)))

    @torch.no_grad()
    def _hybrid_step(self, batch, qres_output) -> None:
        """trec_eval_step (contrastive.py:669-704) for a field set WITH sparse fields, one batch, synchronously: the dense
        per-field lists and candidate scores come from the HBM slab through the C ABI (`mfar_retrieve_fields`,
        `mfar_score_candidates`), the sparse lists and score columns from the host BM25 indices (index.py:97-124), and the
        mask + field-weight softmax + top-k runs in `mfar_mix_topk` over all F columns in `field_info` order."""
        import numpy as np
        from mfar.data import index as idxmod
        rank, world = _dist()
        # Several ranks: the dense rows are sharded (contrastive.py:470), the BM25 indices are host objects every rank holds in full
        # -- like the reference, which evaluates sparse fields on every rank (index.py:97-124, contrastive.py:672-683).  Every rank
        # scores the SAME batch: the per-shard dense lists are all-gathered and merged to the global per-field top-k, the sparse
        # lists are computed redundantly (deterministic), the candidate union is therefore identical everywhere; a rank scores the
        # dense columns of the candidates whose rows it owns (`mfar_score_candidates` returns NaN for foreign rows) and one more
        # all-gather hands every rank all columns.
        x = self.encode_query_batch(batch)
        Q = x.shape[0]
        texts = [q.text for q in batch.instances]
        fields = list(self.field_info.items())
        dense_cols = [i for i, (_, f) in enumerate(fields) if f.field_type == FieldType.DENSE]
        lists = [[] for _ in range(Q)]                                           # per query: arrays of numeric doc ids
        if dense_cols:
            fid, fsc = self.slab.retrieve_fields(x, TOP_K, True)                 # zero-sentinel lists incl. their (0, 0.0) padding
            if world > 1:
                fid, fsc = _merge_shard_lists(_all_gather_cat(fid), _all_gather_cat(fsc), TOP_K, world)
            fid = fid.cpu().numpy()
            for i in range(Q):
                lists[i].append(fid[i].reshape(-1))
        for key, f in fields:
            if f.field_type == FieldType.SPARSE:
                for i, row in enumerate(self.indices_dict[key].retrieve_batch(texts, TOP_K)):          # :672-674
                    lists[i].append(np.array([self.keys_to_numeric_ids[k] for k, _ in row], dtype=np.int64))
        cands = [np.unique(np.concatenate(l)) for l in lists]                    # :678-679 (set union)
        cands = [c[c >= 0] for c in cands]
        C = max(len(c) for c in cands)
        cand = np.full((Q, C), -1, np.int64)
        n_cand = np.zeros(Q, np.int32)
        for i, c in enumerate(cands):
            cand[i, :len(c)], n_cand[i] = c, len(c)
        xs = np.zeros((Q, C, len(fields)), np.float32)
        if dense_cols:
            cd = torch.from_numpy(cand).to(self.device)
            xd = self.slab.score_candidates(x, cd)                                                     # :681-683, dense; NaN = not my row
            if world > 1:
                parts = _all_gather_cat(xd.unsqueeze(0))                                               # [world, Q, C, Fd]
                own = ~torch.isnan(parts)
                xd = torch.where(own, parts, torch.zeros_like(parts)).sum(0)                           # exactly one owner per real candidate
                xd[~own.any(0)] = float("nan")                                                         # padding slots stay NaN
            xs[:, :, dense_cols] = xd.cpu().numpy()
        for col, (key, f) in enumerate(fields):
            if f.field_type == FieldType.SPARSE:
                for i in range(Q):
                    keys = [self.numeric_ids_to_keys[d] for d in cands[i]]
                    xs[i, :len(keys), col] = self.indices_dict[key].score_batch([texts[i]], keys)[0].numpy()
        xs[np.isnan(xs) & (np.arange(C)[None, :, None] >= n_cand[:, None, None])] = 0.0                # padding slots: never mixed
        res = idxmod.mix_topk(xs, cand, x.cpu().numpy(), self._weights().cpu().numpy(), self.mask[:, 0].float().numpy(),
                              n_cand=n_cand, k=TOP_K, query_cond=self.query_cond, device=self.device.index or 0)   # :685-696
        if int(res["n_valid"].min()) < TOP_K:
            raise RuntimeError(f"selected index k out of range: only {int(res['n_valid'].min())} candidates for k={TOP_K}")
        if rank == 0 and qres_output is not None:
            for q, row_ids, row_sims in zip(batch.instances, res["ids"].tolist(), res["scores"].tolist()):
                for d, sim in zip(row_ids, row_sims):
                    print(trec.QRes(query_id=q._id, doc_id=self.numeric_ids_to_keys[d], sim=sim), file=qres_output)

    @torch.no_grad()
    def trec_eval_step(self, batch, batch_idx: int, qres_output) -> None:
        """One batch, synchronously (the reference's hook signature).  `test()` overlaps consecutive batches instead."""
        if self.has_sparse:
            return self._hybrid_step(batch, qres_output)
        if len(batch.instances) > self._get_searcher().Qb:
            for b in range(0, len(batch.instances), self._searcher.Qb):           # one ticket at a time
                sub = SimpleNamespace(instances=batch.instances[b:b + self._searcher.Qb],
                                      query={k: {n: t[b:b + self._searcher.Qb] for n, t in v.items()} for k, v in batch.query.items()})
                self._collect(self._submit(sub, qres_output))
            return
        self._collect(self._submit(batch, qres_output))

    def mask_field(self, field_idx_list) -> None:                                # contrastive.py:706-714
        names = list(self.field_info.keys())
        masked = [names[i] for i in field_idx_list]
        if _dist()[0] == 0:
            print(f"Masking fields: {masked}")
        self.masked_fields_string = ",".join(masked)
        mask = torch.ones([len(self.field_info), 1])
        mask[field_idx_list] = 0
        self.mask = mask

    # ------------------------------------------------------------------ the test loop (trainer.test)
    def on_test_epoch_start(self) -> None:
        self.on_eval_start()
        self.additional_qres_output = open(f"{self.out_dir}/additional_{_dist()[0]}.qres", "w")

    def test_step(self, batch, batch_idx: int, dataloader_idx: int = 0) -> None:
        self.trec_eval_step(batch, batch_idx, self.qres_output if dataloader_idx == 0 else self.additional_qres_output)

    def on_test_epoch_end(self) -> None:
        rank, n = _dist()
        self.qres_output.close()
        self.additional_qres_output.close()
        has_additional = os.path.getsize(f"{self.out_dir}/additional_{rank}.qres") > 0
        _barrier()
        self.merge_qres_and_score([f"{self.out_dir}/{i}.qres" for i in range(n)], self.dev_qrels_path)
        if has_additional:
            self.merge_qres_and_score([f"{self.out_dir}/additional_{i}.qres" for i in range(n)], self.additional_qrels_path,
                                      additional="additional-")

    def test(self, data_module: RetrievalDataModule) -> None:
        """What `trainer.test(module, data_module)` does for this module (train.py:260, mask_fields.py:143-170)."""
        data_module.setup("test")
        was_training = self.training
        self.eval()
        self.on_test_epoch_start()
        # `lag` batches stay in flight while the next one is tokenised, encoded and submitted (the launches of the pipeline;
        # with the wide screened pass a launch scans two coalesced batches, mfar/data/pipeline.py)
        pending = deque()
        with torch.no_grad():
            for li, loader in enumerate(data_module.test_dataloader()):
                out = self.qres_output if li == 0 else self.additional_qres_output
                for batch in loader:
                    if self.has_sparse:
                        self.trec_eval_step(batch, 0, out)
                        continue
                    if len(batch.instances) > self._get_searcher().Qb:
                        while pending:
                            self._collect(pending.popleft())
                        self.trec_eval_step(batch, 0, out)
                        continue
                    pending.append(self._submit(batch, out))
                    while len(pending) > self._searcher.lag:
                        self._collect(pending.popleft())
            while pending:
                self._collect(pending.popleft())
        self.on_test_epoch_end()
        if was_training:
            self.train()

    def test_sweep(self, data_module: RetrievalDataModule, field_idx_lists) -> bool:
        """Every evaluation of a field-masking sweep (mask_fields.py:143-170: baseline, one run per field, per field type, per
        field name) in ONE pass over the queries.  The runs differ in the mask only, and the mask only enters the mixer
        (contrastive.py:685-686): queries are encoded once, stage 1, the candidate union and stage 2 run once per batch, the
        mixer once per mask (`mfar_search_stage2_masks`).  Afterwards the runs are replayed in order -- `mask_field`, the
        rank's .qres files, `merge_qres_and_score` -- so the output files, the printed metrics and results_dicts-*.jsonl
        are what the 2 F + 2 separate `test()` calls leave behind.  `field_idx_lists`: one list of masked field indices per
        run (empty = baseline).  With several ranks the rows stay sharded and the second all-gather of the exchange carries one
        local top-k payload per mask.  Returns False without doing anything when the sweep path does not apply (sparse
        fields: scored on the host) -- the caller then runs the masks one by one."""
        from mfar.data.pipeline import PipelinedSearcher
        rank, world = _dist()
        if self.has_sparse or not field_idx_lists:
            return False
        data_module.setup("test")
        was_training = self.training
        self.eval()
        self.on_eval_start()
        self.qres_output.close()
        F, M = len(self.field_info), len(field_idx_lists)
        masks = torch.ones(M, F)
        for m, idx in enumerate(field_idx_lists):
            masks[m, list(idx)] = 0
        qmax = min(64, max(1, int(self.dev_batch_size)))
        ps = PipelinedSearcher(self.slab, self._weights(), None, k1=TOP_K, k2=TOP_K, sentinel=True, query_cond=self.query_cond,
                               max_batch=qmax, masks=masks.to(self.device))
        tmp = [[f"{self.out_dir}/.sweep_{m}_{li}_{rank}.qres" for li in range(2)] for m in range(M)]
        files = [[open(fn, "w") for fn in pair] for pair in tmp] if rank == 0 else None
        keys = self.numeric_ids_to_keys

        def collect(item):
            ticket, data, li = item
            res = ps.result(ticket)
            n = len(data)
            if int(res["n_valid"][:, :n].min()) < TOP_K:                         # what torch.topk raises at :696
                raise RuntimeError(f"selected index k out of range: fewer than k={TOP_K} candidates")
            if rank != 0:
                return
            ids, sims = res["ids"][:, :n].cpu().tolist(), res["scores"][:, :n].cpu().tolist()
            for m in range(M):
                files[m][li].write("".join(f"{q._id}\t0\t{keys[d]}\t0\t{s}\t0\n" for q, row_ids, row_sims in zip(data, ids[m], sims[m])
                                           for d, s in zip(row_ids, row_sims)))

        pending = deque()
        with torch.no_grad():
            for li, loader in enumerate(data_module.test_dataloader()):
                for batch in loader:
                    x = self.encode_query_batch(batch)
                    for b in range(0, x.shape[0], ps.Qb):
                        xb = x[b:b + ps.Qb]
                        n = xb.shape[0]
                        if ps.sharded and n < ps.Qb:       # fixed exchange size: pad with copies of the last query (results are dropped)
                            xb = torch.cat([xb, xb[-1:].expand(ps.Qb - n, -1)])
                        pending.append((ps.submit(xb.contiguous()), batch.instances[b:b + n], min(li, 1)))
                        while len(pending) > ps.lag:
                            collect(pending.popleft())
            while pending:
                collect(pending.popleft())
        for pair in files or []:
            for f in pair:
                f.close()
        for m, idx in enumerate(field_idx_lists):                                # replay the runs in order
            if idx:
                self.mask_field(list(idx))
            if rank != 0:
                continue                                                         # rank 0 holds every query's results
            os.replace(tmp[m][0], f"{self.out_dir}/{rank}.qres")
            os.replace(tmp[m][1], f"{self.out_dir}/additional_{rank}.qres")
            has_additional = os.path.getsize(f"{self.out_dir}/additional_{rank}.qres") > 0
            self.merge_qres_and_score([f"{self.out_dir}/{rank}.qres"], self.dev_qrels_path)
            if has_additional:
                self.merge_qres_and_score([f"{self.out_dir}/additional_{rank}.qres"], self.additional_qrels_path, additional="additional-")
        _barrier()
        if was_training:
            self.train()
        return True

    def merge_qres_and_score(self, qres_files, qrels_path, additional=""):       # contrastive.py:566-613
        rank, _ = _dist()
        if rank != 0:
            return None
        seen = set()
        merged = f"{self.out_dir}/final-{additional}all-{rank}.qres"
        kept = []                                                                # (query_id, doc_id, sim) of the merged run
        with open(merged, "w") as out:
            for fn in qres_files:
                if not os.path.exists(fn):
                    continue
                with open(fn) as f:
                    here = set()
                    buf = []
                    for line in f:       # what `for r in QRes.from_text_io(f): print(r, file=out)` does, without 100 dataclasses per query
                        q, it, d, rk, sim, run = line.split()
                        if q not in seen:                                        # first-seen dedup across ranks (:570-581)
                            here.add(q)
                            sim = float(sim)
                            buf.append(f"{q}\t{it}\t{d}\t{int(rk)}\t{sim}\t{run}\n")
                            kept.append((q, d, sim))
                    out.write("".join(buf))
                seen.update(here)
        if shutil.which("trec_eval"):
            metrics = trec.call_trec_eval_and_get_metrics(qrels=qrels_path, qres=merged)
        else:                                                                    # same numbers from the records just written
            cache = self.__dict__.setdefault("_qrels_cache", {})
            key = (qrels_path, os.path.getmtime(qrels_path))
            if key not in cache:
                with open(qrels_path) as f:
                    cache[key] = trec.QRels.from_text_io(f)
            metrics = trec.compute_metrics(cache[key], kept)
        keys = ["success_1", "success_5", "recall_5", "recall_10", "recall_15", "recall_20", "ndcg", "ndcg_cut_10", "recip_rank", "map"]
        print("\t".join(keys))
        print("\t".join(f"{metrics[k]:.3f}" for k in keys))
        row = {k: f"{metrics[k]:.3f}" for k in keys}
        row["masked_fields"] = self.masked_fields_string
        row["additional"] = "test" if additional != "" else "val"
        line = json.dumps(row)
        print(line)
        with open(f"{self.out_dir}/results_dicts-all-{rank}.jsonl", "a") as f:
            f.write(line + "\n")
        self.last_metrics = metrics
        return metrics

    # ------------------------------------------------------------------ checkpoints (Lightning .ckpt key layout)
    def checkpoint_state(self) -> dict:
        sd = {f"encoder.{k}": v for k, v in self.encoder.state_dict().items()}
        w = self.mixture_of_fields_layer.weight.detach()
        sd["mixture_of_fields_layer.weight"] = w
        sd["hybrid_contrastive_loss_fn.mixture_of_fields_layer.weight"] = w      # registered twice in the reference (:279-293)
        for k, v in self.bn_state.items():
            sd[f"hybrid_contrastive_loss_fn.bn.{k}"] = v
        hp = dict(model_id=self.model_id, dataset_name=self.dataset_name, corpus_path=self.corpus_path, query_cond=self.query_cond,
                  prefix=self.prefix, contrastive_temp=self.contrastive_temp, encoder_learning_rate=self.encoder_learning_rate,
                  weights_learning_rate=self.weights_learning_rate, weight_decay=self.weight_decay, dev_batch_size=self.dev_batch_size,
                  trec_val_freq=self.trec_val_freq, freeze_encoder=self.freeze_encoder, use_batchnorm=self.use_batchnorm,
                  field_info={k: f.serialize() for k, f in self.field_info.items()},      # on_save_checkpoint (:634-640)
                  indices_list=[], vectors_list=[], corpus=[], precomputed_sparse_scores=[])
        return {"state_dict": sd, "hyper_parameters": hp}

    def save_checkpoint(self, path: str) -> None:
        torch.save(self.checkpoint_state(), path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, **kwargs) -> "RetrievalTrainingModule":
        """Reads a Lightning-style checkpoint (`state_dict` + `hyper_parameters`), as mask_fields.py:110-121 does."""
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        hp = dict(ckpt.get("hyper_parameters", {}))
        for drop in ("indices_list", "vectors_list", "corpus", "precomputed_sparse_scores", "field_info"):
            hp.pop(drop, None)
        accepted = cls.__init__.__code__.co_varnames[1:cls.__init__.__code__.co_argcount]
        args = {k: v for k, v in hp.items() if k in accepted}
        args.update(kwargs)
        args.setdefault("model_id", hp.get("model_id", ""))
        args.setdefault("decoder", None)
        args.setdefault("corpus_path", hp.get("corpus_path", ""))
        args.setdefault("dataset_name", hp.get("dataset_name", ""))
        if args.get("weights_learning_rate") is None:
            args["weights_learning_rate"] = 0.0
        module = cls(**args)
        module.load_reference_state_dict(ckpt["state_dict"])
        module.eval()       # a module restored for evaluation: dropout off until the caller asks for .train()
        return module

    # buffers some transformers versions persist and others do not: their absence is not a layout mismatch
    _NON_PERSISTENT = ("embeddings.position_ids", "embeddings.token_type_ids")

    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Load a Lightning `state_dict` with the reference module's key layout (contrastive.py:279-293):
        `encoder.0.auto_model.*` (SentenceTransformer Sequential index 0), `mixture_of_fields_layer.weight` and its alias
        `hybrid_contrastive_loss_fn.mixture_of_fields_layer.weight` (the same Parameter registered twice), optional
        `hybrid_contrastive_loss_fn.bn.*`.  Anything that does not line up raises: a checkpoint that silently loads
        nothing would make `mask_fields` evaluate the base encoder."""
        enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
        if not enc:
            raise RuntimeError("checkpoint has no `encoder.*` keys: not a RetrievalTrainingModule checkpoint")
        missing, unexpected = self.encoder.load_state_dict(enc, strict=False)
        if unexpected:
            raise RuntimeError(f"unexpected encoder keys in checkpoint: {unexpected[:5]}")
        missing = [k for k in missing if not k.endswith(self._NON_PERSISTENT)]
        if missing:
            raise RuntimeError(f"checkpoint lacks {len(missing)} encoder tensors (first: {missing[:5]}): the encoder key "
                               f"layout does not match `encoder.0.auto_model.*`")
        names = ("mixture_of_fields_layer.weight", "hybrid_contrastive_loss_fn.mixture_of_fields_layer.weight")
        have = [n for n in names if n in sd]
        if not have:
            raise RuntimeError("checkpoint has no mixture_of_fields_layer.weight")
        w = sd[have[0]]
        if len(have) == 2 and not torch.equal(sd[names[0]], sd[names[1]]):
            raise RuntimeError("the two aliases of mixture_of_fields_layer.weight differ in this checkpoint")
        want = tuple(self.mixture_of_fields_layer.weight.shape)       # [E, F] query-conditioned, [F, 1] otherwise
        if tuple(w.shape) != want:
            raise RuntimeError(f"mixture_of_fields_layer.weight is {tuple(w.shape)}, the current field set needs {want}")
        self.mixture_of_fields_layer.weight.data = w.clone().float().to(self.mixture_of_fields_layer.weight.device)
        pre = "hybrid_contrastive_loss_fn.bn."
        self.bn_state = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        self.mark_encoder_updated()
