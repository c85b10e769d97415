"""`python -m mfar.commands.mask_fields ...` -- field-masking ablation sweep (reference mfar/commands/mask_fields.py).

Same keyword-only flags as the reference `main` (pinned by tests/golden/cli_signatures.json), same evaluation sequence
(baseline, one run per field, all-sparse, all-dense, one run per field NAME; mask_fields.py:143-170) and the same output
files ({out}/{rank}.qres, additional_{rank}.qres, final-*all-0.qres, results_dicts-all-0.jsonl).  The corpus is encoded
into HBM once and reused by every masked run -- only the mixer changes with the mask -- instead of being re-encoded by
each of the 1+F+1+F `trainer.test` calls (contrastive.py:553-554); with dense fields the whole sweep is ONE
pass over the queries (queries encoded once, stages 1 and 2 once per batch, the mixer once per mask).  Launch one process per GPU with torch.distributed.run
for row-sharded multi-GPU evaluation; a single process uses GPU 0.
"""
import os
import time
from typing import *  # noqa: F401,F403

import torch

from mfar.commands import _setup
from mfar.commands._cli import run
from mfar.commands._setup import init_distributed_from_env  # noqa: F401  (re-exported)
from mfar.data.typedef import FieldType
from mfar.modeling.contrastive import RetrievalTrainingModule


def main(
        *,
        dataset_name: str, lexical_index: str, out: str, temp_dir: str, data: Optional[str] = None,
        queries: Optional[str] = None, corpus: Optional[str] = None, partition: str = "val",
        additional_partition: Optional[str] = None, model_name: str = "facebook/contriever-msmarco",
        normalize: bool = False, negative_sampling_params: Tuple[int, int, int] = (100, 50, 1),
        train_batch_size: int = 16, dev_batch_size: int = 64, train_max_length: int = 512, dev_max_length: int = 512,
        max_epochs: int = 50, seed: int = 0xdeadbeef, precision: str = "16-mixed", num_gpus: int = -1,
        logger: Optional[str] = None, wandb_name: str = None, wandb_dir: str = None, experiment_name: str = None,
        field_names: List = None, trec_val_freq: int = 0, prefix: bool = False, checkpoint_dir: Optional[str] = None,
        debug: bool = False,
):
    st = _setup.build(locals())
    field_info, data_module = st.field_info, st.data_module
    with open(f"{checkpoint_dir}/best.txt") as f:                                  # mask_fields.py:106-108
        checkpoint_path = f"{checkpoint_dir}/{f.read().strip().split('/')[-1]}"
    print(f"PATH IS: {checkpoint_path}")
    module = RetrievalTrainingModule.load_from_checkpoint(
        checkpoint_path, corpus=st.corpus, indices_dict=st.indices_dict, vectors_dict=st.vectors_dict, encoder=st.encoder,
        dev_qrels_path=st.dev_qrels, additional_qrels_path=st.additional_qrels, field_info=field_info, out_dir=out,
        dev_batch_size=dev_batch_size)
    module.encoder.to(st.device)
    module.encode_precision = _setup.encode_precision_for(precision)
    print(f"Starting re-testing of {checkpoint_path}: {time.strftime('%Y-%m-%d %H:%M:%S')}")

    # the evaluation sequence of mask_fields.py:143-170: baseline, every field, all sparse, all dense, every field NAME
    fields = list(field_info.values())
    sparse_idx = [i for i, f in enumerate(fields) if f.field_type == FieldType.SPARSE]
    dense_idx = [i for i, f in enumerate(fields) if f.field_type == FieldType.DENSE]
    runs = [[]]
    if not debug:
        runs += [[idx] for idx in range(len(fields))]
        runs += [sparse_idx] if sparse_idx else []
        runs += [dense_idx] if dense_idx else []
        runs += [[i for i, f in enumerate(fields) if f.name == name] for name in sorted({f.name for f in fields})]
    print("Baseline Evaluation")
    # dense fields: the whole sweep in ONE pass over the queries (the mask only enters the mixer; contrastive.py
    # test_sweep; MFAR_MASK_SWEEP=0 switches it off); otherwise one `test()` per run as the reference does
    if os.environ.get("MFAR_MASK_SWEEP", "1") != "0" and module.test_sweep(data_module, runs):
        if not debug and not sparse_idx:
            print("No sparse fields")
        if not debug and not dense_idx:
            print("No dense fields")
        return module
    module.test(data_module)
    if not debug:
        for idx in range(len(fields)):
            module.mask_field([idx])
            module.test(data_module)
        if sparse_idx:
            module.mask_field(sparse_idx)
            module.test(data_module)
        else:
            print("No sparse fields")
        if dense_idx:
            module.mask_field(dense_idx)
            module.test(data_module)
        else:
            print("No dense fields")
        for name in sorted({f.name for f in fields}):
            module.mask_field([i for i, f in enumerate(fields) if f.name == name])
            module.test(data_module)
    return module


if __name__ == "__main__":
    run(main)
