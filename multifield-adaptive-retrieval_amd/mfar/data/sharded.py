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


class ReplicaLayout:
    """N ranks = G replica groups x R row shards.

    The reference evaluates with a query-sharded search over a REPLICATED corpus: a `DistributedSampler` deals the query
    batches to the ranks and every rank holds all rows (reference mfar/modeling/contrastive.py:184,200,207,470,672-674).  The
    north-star layout row-shards the corpus over all N ranks and exchanges per-shard lists (R = N, one group).  Both are corners
    of one layout: rank r belongs to replica group g = r // R and holds row shard s = r % R of the corpus,
    rows [D*s/R, D*(s+1)/R) (the reference's encode split, contrastive.py:470, inside the group); query batch i is served by group
    i % G; the lists-first exchange (two all-gathers per launch) runs inside a group only.  R = N: today's north-star path,
    every batch crosses xGMI.  R = 1: the reference's query-sharded search, no exchange at all.  In between: the corpus no longer
    fits one GPU (BASELINE.json configs[4]: 245.8 GB of bf16 rows) but does fit R of them.
    """

    def __init__(self, world_size: int, rank: int, row_shards: int):
        if row_shards < 1 or world_size % row_shards:
            raise ValueError(f"row_shards = {row_shards} must divide the world size {world_size}")
        self.world, self.rank, self.R = int(world_size), int(rank), int(row_shards)
        self.G = self.world // self.R
        self.group_index, self.shard_index = self.rank // self.R, self.rank % self.R
        self.group = None                      # torch.distributed process group of this rank's replica group (None: WORLD / no exchange)

    def rows(self, n_docs: int) -> Tuple[int, int]:
        return shard_bounds(n_docs, self.shard_index, self.R)

    def group_ranks(self, g: int):
        return list(range(g * self.R, (g + 1) * self.R))

    def serves(self, batch_index: int) -> bool:
        """Query batches are dealt round-robin to the replica groups (the DistributedSampler's interleaving, contrastive.py:200)."""
        return batch_index % self.G == self.group_index

    def my_batches(self, first: int, n: int):
        """The global batch indices in [first, first + n) this rank's group serves, ascending."""
        return [i for i in range(first, first + n) if self.serves(i)]

    def make_groups(self):
        """Create the G process groups (collective: every rank of WORLD must call it).  With one group the default group is
        used; with R = 1 there is nothing to exchange and no group is created."""
        import torch.distributed as dist
        if self.R == 1 or self.G == 1 or not dist.is_initialized():
            self.group = None
            return None
        groups = [dist.new_group(ranks=self.group_ranks(g)) for g in range(self.G)]    # same order on every rank
        self.group = groups[self.group_index]
        return self.group

    @property
    def exchanges(self) -> bool:
        return self.R > 1

    def describe(self) -> str:
        if self.world == 1:
            return "single shard"
        if self.R == self.world:
            return f"row-shard x{self.R} (one group): lists-first exchange over RCCL, two all-gathers per launch"
        if self.R == 1:
            return f"{self.G} full replicas, query batches dealt round-robin (the reference's query-sharded search): no exchange"
        return f"{self.G} replica groups x {self.R} row shards: batches dealt round-robin to the groups, lists-first exchange inside a group"


def choose_row_shards(world_size: int, index_bytes: int, free_hbm_bytes: int, headroom: float = 0.8) -> int:
    """Smallest R (a divisor of the world size) whose per-GPU share of the index fits `headroom` of the free HBM: fewer row
    shards = more replica groups = fewer batches crossing xGMI.  `index_bytes` = everything an index of the WHOLE corpus holds
    (slab + screen slab + gather slab + tables)."""
    for r in range(1, world_size + 1):
        if world_size % r == 0 and index_bytes / r <= headroom * free_hbm_bytes:
            return r
    return world_size


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
