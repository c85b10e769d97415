"""Encoder construction and index creation (reference mfar/modeling/util.py:16-108).

`prepare_model` returns `(tokenizer, encoder, None)` like the reference; the encoder is `SentenceEncoder`, a small
re-statement of the sentence-transformers stack the reference assembles (`Transformer` -> `Pooling(mean)` [->
`Normalize`], modeling/util.py:38-52): the HF `AutoModel` forward runs on PyTorch-ROCm (host code, as the north-star
asks), its state-dict keys (`0.auto_model.*`) match the reference's so Lightning checkpoints load.  sentence-transformers
itself is not installed on either box and its arithmetic is un-vendored third-party code: encoder parity is
structural only.

`read_and_create_indices` allocates ONE on-HBM slab for this rank's row shard of all dense fields instead of one
np.memmap file + CPU DenseFlatIndex per field (modeling/util.py:84-101).
"""
import os
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from mfar.data import trec
from mfar.data.format import format_documents
from mfar.data.index import BM25sSparseIndex, DenseFlatIndex, MultiFieldIndex
from mfar.data.sharded import shard_bounds
from mfar.data.typedef import Corpus, FieldType
from mfar.data.util import HbmFieldVectors


class _Transformer(torch.nn.Module):
    """Holds the HF model as `.auto_model` (the attribute name sentence-transformers uses, so checkpoint keys match)."""

    def __init__(self, auto_model):
        super().__init__()
        self.auto_model = auto_model

    def forward(self, features):
        out = self.auto_model(input_ids=features["input_ids"], attention_mask=features["attention_mask"],
                              **({"token_type_ids": features["token_type_ids"]} if "token_type_ids" in features else {}))
        return out[0] if not hasattr(out, "last_hidden_state") else out.last_hidden_state


class SentenceEncoder(torch.nn.Sequential):
    """tokens -> mean-pooled sentence embedding.  Interface used by the path: `.encode(texts, ...)` (index.py:187,257),
    `__call__(features)["sentence_embedding"]` (contrastive.py:693), `.get_sentence_embedding_dimension()`,
    `.get_max_seq_length()`, `.tokenizer`."""

    def __init__(self, auto_model, tokenizer, normalize: bool = False, max_seq_length: Optional[int] = None,
                 dense: Optional[torch.nn.Module] = None):
        super().__init__(_Transformer(auto_model))
        if dense is not None:
            # sentence-transformers numbers its modules: 0 Transformer, 1 Pooling (no parameters), 2 Dense -> the projection's
            # checkpoint key is `2.linear.weight` (gtr-t5: Dense(768 -> 768, bias=False, identity activation))
            self.add_module("2", dense)
        self.tokenizer = tokenizer
        self.normalize = normalize
        cfg = auto_model.config
        limit = getattr(cfg, "max_position_embeddings", None) or getattr(cfg, "n_positions", None) or 512
        self.max_seq_length = min(max_seq_length or limit, limit)
        self.train(auto_model.training)      # the wrapper starts in the wrapped model's mode (from_pretrained: eval)

    @property
    def auto_model(self):
        return self[0].auto_model

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dense(self):
        return self._modules.get("2")

    def get_sentence_embedding_dimension(self) -> int:
        if self.dense is not None:
            return int(self.dense.linear.out_features)
        return int(self.auto_model.config.hidden_size)

    def get_max_seq_length(self) -> int:
        return int(self.max_seq_length)

    def forward(self, features) -> Dict[str, torch.Tensor]:
        tok = self[0](features)
        # Pooling(mean): sum of unmasked tokens / their count, as ONE batched product [n, 1, L] x [n, L, E] in fp32 (outside autocast).
        # Not `(tok * m).sum(1)`: at L >= 512 a captured graph replays that expression WRONG on this stack from the second replay on
        # (profiles/r06_k_graph_reduce_probe.txt: difference ~100 on N(0,1) inputs at L = 512 / 1024 / 4096, none below 512; the product
        # alone, the sum alone and the eager expression are all right) -- and mfar/modeling/graphed.py replays these forwards.
        m = features["attention_mask"].unsqueeze(1).to(torch.float32)
        with torch.autocast(device_type=tok.device.type, enabled=False):
            emb = (torch.bmm(m, tok.float()) / torch.bmm(m, m.transpose(1, 2)).clamp(min=1e-9)).squeeze(1).to(tok.dtype)
        if self.dense is not None:
            emb = self.dense(emb)
        if self.normalize:
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
        return {"token_embeddings": tok, "sentence_embedding": emb, **features}

    def tokenize(self, texts: List[str]):
        return self.tokenizer(list(texts), padding=True, truncation="longest_first", max_length=self.max_seq_length,
                              return_tensors="pt")

    @torch.no_grad()
    def encode(self, sentences, batch_size: int = 32, convert_to_numpy: bool = True, convert_to_tensor: bool = False,
               show_progress_bar: bool = False, device=None, **_):
        was_training = self.training
        self.eval()
        single = isinstance(sentences, str)
        if single:
            sentences = [sentences]
        dev = torch.device(device) if device is not None else self.device
        order = np.argsort([-len(s) for s in sentences], kind="stable")      # longest first: less padding per batch
        out = [None] * len(sentences)
        for b in range(0, len(sentences), batch_size):
            idx = order[b:b + batch_size]
            feats = {k: v.to(dev) for k, v in self.tokenize([sentences[i] for i in idx]).items()}
            emb = self.forward(feats)["sentence_embedding"].float()
            for j, i in enumerate(idx):
                out[i] = emb[j]
        if was_training:
            self.train()
        emb = torch.stack(out) if out else torch.empty(0, self.get_sentence_embedding_dimension(), device=dev)
        if convert_to_tensor:
            return emb[0] if single else emb
        arr = emb.cpu().numpy()
        return arr[0] if single else arr


def _tiny_random_model(spec: str):
    """'random-init:<hidden>x<layers>' -> a randomly initialised BERT + a character-level WordPiece tokenizer.
    For plumbing tests and smoke runs on boxes without checkpoints (there is no network)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import BertConfig, BertModel, PreTrainedTokenizerFast
    dims = spec.split(":", 1)[1] if ":" in spec else "64x2"
    hidden, layers = (int(x) for x in dims.split("x"))
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789.,:;!?-_'\"()/{}[]")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + chars + ["##" + c for c in chars]
    ids = {t: i for i, t in enumerate(vocab)}
    tk = Tokenizer(models.WordPiece(vocab=ids, unk_token="[UNK]", max_input_chars_per_word=200))
    tk.normalizer = normalizers.BertNormalizer(lowercase=True)
    tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                      special_tokens=[("[CLS]", ids["[CLS]"]), ("[SEP]", ids["[SEP]"])])
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]",
                                  mask_token="[MASK]")
    # heads of 64 dims from 128 up -- "768x12" is BERT-base / contriever's shape exactly (12 layers, 12 heads x 64, 3072 in the FFN); the
    # tiny plumbing models keep heads of 32.  (Until round 6 every size had heads of 32: 24 heads at 768, whose attention kernel took a
    # third of a forward's GPU time -- an artefact of the stand-in, not of the encoder the reference runs.)
    heads = hidden // 64 if hidden >= 128 and hidden % 64 == 0 else max(1, hidden // 32)
    cfg = BertConfig(vocab_size=len(vocab), hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                     intermediate_size=hidden * 4, max_position_embeddings=512)
    torch.manual_seed(0)
    return tok, BertModel(cfg).eval()       # from_pretrained() also hands models out in eval mode


class _Dense(torch.nn.Module):
    """sentence-transformers `models.Dense` with its defaults for gtr-t5: `linear` without bias, identity activation."""

    def __init__(self, in_features: int, out_features: int, bias: bool = False):
        super().__init__()
        self.linear = torch.nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        return self.linear(x)


def _is_gtr_t5(model_id: str) -> bool:
    return model_id.startswith("sentence-transformers/gtr-t5") or model_id.startswith("random-init-t5") or \
        (os.path.isdir(model_id) and os.path.isdir(os.path.join(model_id, "2_Dense")))


def _prepare_gtr_t5(model_id: str, with_decoder: bool, normalize: bool):
    """The gtr-t5 branch (modeling/util.py:22-36): `SentenceTransformer(model_id)` = T5 encoder -> mean pooling -> Dense(d -> d,
    no bias) -> Normalize, the Normalize dropped unless --normalize; with_decoder: a `T5ForConditionalGeneration` whose
    encoder IS the sentence encoder's (shared parameters).  Weights come from a local directory in the sentence-transformers
    layout (`config.json` + T5 encoder weights at the root, `2_Dense/{model.safetensors|pytorch_model.bin}`): there is no
    network here, so a hub id resolves only through the local HF cache.  `random-init-t5:<d>x<layers>` builds the same
    structure with random weights (plumbing tests)."""
    from transformers import T5Config, T5EncoderModel, T5ForConditionalGeneration
    if model_id.startswith("random-init-t5"):
        dims = model_id.split(":", 1)[1] if ":" in model_id else "64x2"
        d, layers = (int(x) for x in dims.split("x"))
        tokenizer, _ = _tiny_random_model(f"random-init:{d}x1")
        cfg = T5Config(vocab_size=len(tokenizer), d_model=d, d_kv=max(8, d // 4), d_ff=2 * d, num_layers=layers, num_decoder_layers=layers,
                       num_heads=4, decoder_start_token_id=0, pad_token_id=tokenizer.pad_token_id)
        torch.manual_seed(0)
        t5 = T5EncoderModel(cfg).eval()
        dense = _Dense(d, d)
        full = T5ForConditionalGeneration(cfg).eval() if with_decoder else None
    else:
        from transformers import AutoTokenizer
        try:
            tokenizer = AutoTokenizer.from_pretrained(model_id)
            t5 = T5EncoderModel.from_pretrained(model_id)
        except Exception as e:
            raise ValueError(f"Unsupported model_id or unable to find: {model_id}") from e
        d = int(t5.config.d_model)
        dense = _Dense(d, d)
        root = model_id if os.path.isdir(model_id) else None
        if root is None:
            try:
                from huggingface_hub import snapshot_download
                root = snapshot_download(model_id, local_files_only=True)
            except Exception as e:
                raise ValueError(f"unable to find the Dense module of {model_id} locally") from e
        st_path, bin_path = os.path.join(root, "2_Dense", "model.safetensors"), os.path.join(root, "2_Dense", "pytorch_model.bin")
        if os.path.exists(st_path):
            from safetensors.torch import load_file
            dense.load_state_dict(load_file(st_path))
        elif os.path.exists(bin_path):
            dense.load_state_dict(torch.load(bin_path, map_location="cpu"))
        else:
            raise ValueError(f"{root}/2_Dense holds no weights")
        full = None
        if with_decoder:
            size = model_id.rstrip("/").split("-")[-1]                       # util.py:33-34
            try:
                full = T5ForConditionalGeneration.from_pretrained(f"google-t5/t5-{size}")
            except Exception as e:
                raise ValueError(f"unable to find google-t5/t5-{size} locally") from e
    encoder = SentenceEncoder(t5, tokenizer, normalize=normalize, dense=dense, max_seq_length=512)
    if full is not None:
        full.encoder = t5.encoder                                            # util.py:35: the decoder model reads THE encoder
    return tokenizer, encoder, full


def prepare_model(model_id: str, with_decoder: bool = False, normalize: bool = False, freeze_encoder: bool = False):
    """(tokenizer, encoder, decoder|None) -- reference modeling/util.py:16-71.  The contriever branch and the
    local-directory branch are the same thing here (HF AutoModel + mean pooling); the gtr-t5 branch (`_prepare_gtr_t5`) adds
    the Dense projection and, with_decoder, the T5 decoder over the shared encoder."""
    if _is_gtr_t5(model_id):
        return _prepare_gtr_t5(model_id, with_decoder, normalize)
    if with_decoder:
        raise UnboundLocalError("with_decoder is only defined for the gtr-t5 branch (the reference's other branches return no decoder)")
    if model_id.startswith("random-init"):
        tokenizer, model = _tiny_random_model(model_id)
    else:
        try:
            from transformers import AutoModel, AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(model_id)
            model = AutoModel.from_pretrained(model_id)
        except Exception as e:
            raise ValueError(f"Unsupported model_id or unable to find: {model_id}") from e
    if freeze_encoder:
        for p in model.parameters():
            p.requires_grad = False
        model.eval()
    return tokenizer, SentenceEncoder(model, tokenizer, normalize=normalize), None


def _dist_info() -> Tuple[int, int, int]:
    import torch.distributed as dist
    share = os.environ.get("MFAR_SHARE_GPU") == "1"        # several ranks rehearsed on ONE GPU (mfar/commands/_setup.py)
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), 0 if share else int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    return 0, 1, 0 if share else int(os.environ.get("LOCAL_RANK", "0"))


def read_and_create_indices(corpus_path, dataset_name, field_info, temp_dir, encoder):
    """-> (corpus [(id, json)], vectors_dict {field_key: HbmFieldVectors}, indices_dict {field_key: DenseFlatIndex | BM25sSparseIndex}).
    All dense fields share one `MultiFieldIndex` holding this rank's row shard (contrastive.py:470); the field order
    inside the slab is the order of the dense keys in `field_info` (schema.py:131-134: dense keys first, then sparse).
    Sparse fields get a host-side BM25 index over the field's formatted text (modeling/util.py:102-106)."""
    corpus = list(trec.read_corpus(corpus_path))
    dense = [k for k, f in field_info.items() if f.field_type == FieldType.DENSE]
    keys = [x[0] for x in corpus]
    key_to_id = {k: i for i, k in enumerate(keys)}
    Path(temp_dir).mkdir(parents=True, exist_ok=True)
    rank, world, local_rank = _dist_info()
    r0, r1 = shard_bounds(len(corpus), rank, world)
    vectors_dict, indices_dict = {}, {}
    if dense:
        slab = MultiFieldIndex(r1 - r0, len(dense), encoder.get_sentence_embedding_dimension(), device=local_rank, row_offset=r0)
        for fi, key in enumerate(dense):
            field = field_info[key]
            vectors_dict[key] = HbmFieldVectors(slab, fi, keys, path=f"{temp_dir}/{field.name}.npy")
            indices_dict[key] = DenseFlatIndex(encoder, None, numeric_ids_to_keys=keys, keys_to_numeric_ids=key_to_id, slab=slab,
                                               field_index=fi)
    for key, field in field_info.items():
        if field.field_type == FieldType.SPARSE:
            formatted = format_documents(corpus, field.name, field.dataset)
            docs = Corpus.from_docs_dict({item[0]: item[1] for item in formatted})
            indices_dict[key] = BM25sSparseIndex.create(docs, dataset_name=dataset_name)
            indices_dict[key].name = field.name
    return corpus, vectors_dict, indices_dict
