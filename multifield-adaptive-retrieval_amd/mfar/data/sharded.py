"""Row-sharded search across the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm), one all-gather per query batch.

Replaces the reference's file-based exchange: every rank writes its row slice of the shared memmaps, `barrier()`,
every rank then searches the FULL replicated corpus on its own CPU and rank 0 merges per-rank `.qres` files
(reference mfar/modeling/contrastive.py:470,491-494,519-536,566-581).  Here a rank keeps only the rows it encoded,
searches them on its GPU (`search_local`), and the fixed-size per-shard payloads are exchanged with ONE
`all_gather_into_tensor`; every rank then computes the same merged answer (`merge`).
"""
from typing import Optional, Tuple


def shard_bounds(n_docs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Row range of a rank: corpus[n*rank//ws : n*(rank+1)//ws]  (reference contrastive.py:470)."""
    return n_docs * rank // world_size, n_docs * (rank + 1) // world_size


class HipShardBackend:
    """The product backend: local half and merge both run in libmfar_hip.so on this rank's GPU."""

    def __init__(self, index):
        self.index = index

    def search_local(self, q, k1, sentinel, payload=None):
        return self.index.search_local(q, k1=k1, sentinel=sentinel, payload=payload)

    def merge(self, gathered, n_shards, q, W, mask, k1, k2, sentinel, query_cond):
        from mfar.data.index import merge_payloads
        return merge_payloads(gathered, n_shards, q, W, mask, n_fields=self.index.n_fields, k1=k1, k2=k2,
                              sentinel=sentinel, query_cond=query_cond, device=self.index.device)


class ShardedSearcher:
    """search() == MultiFieldIndex.search() over the concatenation of all ranks' shards."""

    def __init__(self, backend, group=None):
        import torch.distributed as dist
        self.backend = backend
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gathered = None

    def search(self, q, W, mask=None, k1: int = 100, k2: int = 100, sentinel: bool = True, query_cond: bool = True):
        import torch
        import torch.distributed as dist
        payload = self.backend.search_local(q, k1, sentinel)
        if self.world_size == 1:
            gathered = payload
        else:
            if not torch.is_tensor(payload):
                payload = torch.from_numpy(payload)
            n = payload.numel()
            if self._gathered is None or self._gathered.numel() != n * self.world_size or self._gathered.device != payload.device:
                self._gathered = torch.empty(n * self.world_size, dtype=payload.dtype, device=payload.device)
            dist.all_gather_into_tensor(self._gathered, payload, group=self.group)   # shard-major concatenation
            gathered = self._gathered
        return self.backend.merge(gathered, self.world_size, q, W, mask, k1, k2, sentinel, query_cond)
