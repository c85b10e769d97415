"""Shared construction code of the two CLIs: flags (a dict of main()'s keyword arguments) -> fields, encoder, on-HBM
index, data module.  Mirrors the sequence both reference mains walk through (train.py:67-160, mask_fields.py:52-104)."""
import os
from types import SimpleNamespace

import torch

from mfar.data.schema import resolve_fields
from mfar.modeling.contrastive import RetrievalDataModule
from mfar.modeling.util import prepare_model, read_and_create_indices


def init_distributed_from_env() -> None:
    """One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK from torch.distributed.run); no-op for a single process."""
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = _local_device()
        torch.cuda.set_device(local)
        # MFAR_DIST_BACKEND=gloo + MFAR_SHARE_GPU=1: several ranks on ONE GPU (RCCL refuses two ranks on a device) -- how the
        # row-sharded CLIs are rehearsed on a one-GPU box (tests/test_gpu_multirank.py)
        backend = os.environ.get("MFAR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)


def _local_device() -> int:
    return 0 if os.environ.get("MFAR_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))


def encode_precision_for(precision: str) -> str:
    """The CLI's `precision` flag (reference train.py:51, default "16-mixed") -> precision of the corpus-encode forwards: the run's own
    reduced precision where it asked for one, fp32 for "32" (RetrievalTrainingModule.encode_precision)."""
    p = str(precision)
    return "bf16" if p.startswith("bf16") else ("fp16" if p.startswith("16") else "fp32")


def build(flags: dict, freeze_encoder: bool = False) -> SimpleNamespace:
    """-> namespace(field_info, tokenizer, encoder, corpus, vectors_dict, indices_dict, data_module, queries, corpus_dir,
    device).  `flags` uses the CLI names (dataset_name, data / queries / corpus, temp_dir, model_name, ...)."""
    f = dict(flags)
    torch.manual_seed(int(f["seed"]) & 0x7FFFFFFF)
    init_distributed_from_env()
    if f.get("data"):
        f["queries"] = f["corpus"] = f["data"]
    device = torch.device(f"cuda:{_local_device()}")
    field_info = resolve_fields(f["field_names"], f["dataset_name"])
    model_id = f.get("model_path") or f["model_name"]
    tokenizer, encoder, _ = prepare_model(model_id, normalize=f["normalize"], with_decoder=False, freeze_encoder=freeze_encoder)
    encoder.to(device)
    corpus, vectors_dict, indices_dict = read_and_create_indices(f"{f['corpus']}/corpus", f["dataset_name"], field_info,
                                                                  f["temp_dir"], encoder)
    shared = ("lexical_index", "negative_sampling_params", "train_batch_size", "dev_batch_size", "train_max_length",
              "dev_max_length", "dataset_name", "prefix", "trec_val_freq", "additional_partition")
    data_module = RetrievalDataModule(tokenizer=tokenizer, queries_path=f["queries"], corpus=corpus, temp_path=f["temp_dir"],
                                      dev_partition=f["partition"], field_info=field_info, indices_dict=indices_dict,
                                      **{k: f[k] for k in shared})
    return SimpleNamespace(field_info=field_info, tokenizer=tokenizer, encoder=encoder, model_id=model_id, corpus=corpus,
                           vectors_dict=vectors_dict, indices_dict=indices_dict, data_module=data_module, queries=f["queries"],
                           corpus_dir=f["corpus"], device=device,
                           dev_qrels=f"{f['queries']}/{f['partition']}.qrels",
                           additional_qrels=(f"{f['queries']}/{f['additional_partition']}.qrels" if f.get("additional_partition") else None))
