"""Run by tests/test_gpu_multirank.py under torch.distributed.run with 2 ranks (MFAR_DIST_BACKEND=gloo, MFAR_SHARE_GPU=1: both
on cuda:0): `mask_fields.main` over two row shards -- every rank encodes and holds its half of the corpus, the evaluation
runs through the lists-first exchange, the mask sweep carries one local top-k payload per mask -- must leave the files a
single process leaves (dev_batch_size = 1: every text is encoded alone, so the embeddings do not depend on how the corpus is
cut into shards and batches).  argv: data dir, temp dir, checkpoint dir, out dir [, field names [, lexical index dir]] -- with
sparse fields in the field set every evaluation goes through the hybrid step over the row shards (replicated BM25 indices)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
sys.path.insert(0, ROOT)

from mfar.commands import mask_fields

data, tmp, ckpt, out = sys.argv[1:5]
fields = sys.argv[5] if len(sys.argv) > 5 else "title_dense,brand_dense,feature_dense"
lex = sys.argv[6] if len(sys.argv) > 6 else "unused"
rank = int(os.environ.get("RANK", "0"))
m = mask_fields.main(dataset_name="amazon", lexical_index=lex, out=out, temp_dir=f"{tmp}_{rank}", data=data,
                     model_name="random-init:64x2", field_names=fields, checkpoint_dir=ckpt,
                     dev_batch_size=1, additional_partition="test")
assert m.slab.n_rows == (150 if rank == 0 else 150), m.slab.n_rows      # 300 documents over two row shards
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
