"""Two-deep software pipeline over query batches (HIP streams + events through torch).

The reference scores one batch at a time, synchronously (reference mfar/modeling/contrastive.py:559-563 -> 669-704).
On the GPU the per-batch work splits into a long MFMA-bound part (stage 1: per-field exhaustive top-k) and a short
memory/latency-bound tail (candidate union, stage-2 re-scoring, mixer; with several GPUs also the payload all-gather
and the merge).  `PipelinedSearcher` runs the tail of batch i on a side stream BESIDE the full stage-1 kernel of batch
i+1 (high-priority stream): the tail kernels are small enough in LDS to be co-resident with the two stage-1 workgroups
of a CU.  Results are identical to `MultiFieldIndex.search` / `ShardedSearcher.search`.

    ps = PipelinedSearcher(index, W, mask)
    t0 = ps.submit(q0)            # returns immediately (everything is enqueued asynchronously)
    t1 = ps.submit(q1)            # ... the tail of batch 0 is enqueued here, next to stage 1 of batch 1
    r0 = ps.result(t0)            # dict(ids, scores, n_valid): valid until two more batches have been submitted
"""
import torch

from mfar import _native
from mfar.data import index as _index


class PipelinedSearcher:
    def __init__(self, index, W, mask=None, k1: int = 100, k2: int = 100, sentinel: bool = True, query_cond: bool = True,
                 max_batch: int = 64, group=None):
        self.ix, self.W, self.mask = index, W, mask
        self.k1, self.k2, self.sentinel, self.query_cond = k1, k2, sentinel, query_cond
        self.group = group
        dist = torch.distributed
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.dev = torch.device(f"cuda:{index.device}")
        self.main = torch.cuda.Stream(device=self.dev, priority=-1)   # stage 1: dispatched ahead of the tail kernels
        self.side = torch.cuda.Stream(device=self.dev)
        self.Qmax = int(max_batch)
        F, E = index.n_fields, index.dim
        self.slots = []
        for _ in range(2):
            s = dict(q=torch.empty(self.Qmax, E, device=self.dev),
                     ids=torch.empty(self.Qmax, k2, dtype=torch.int64, device=self.dev),
                     scores=torch.empty(self.Qmax, k2, device=self.dev),
                     n_valid=torch.empty(self.Qmax, dtype=torch.int32, device=self.dev),
                     stage1=torch.cuda.Event(), done=torch.cuda.Event(), Q=0, tail_pending=False)
            if self.world == 1:
                s["fid"] = torch.empty(self.Qmax, F, k1, dtype=torch.int64, device=self.dev)
                s["fsc"] = torch.empty(self.Qmax, F, k1, device=self.dev)
            else:       # lists-first exchange: two small all-gathers per batch (include/mfar_hip.h)
                nl, nt = index.lists_bytes(self.Qmax, k1), index.topk_bytes(self.Qmax, k2)
                s["lists"] = torch.empty(nl, dtype=torch.uint8, device=self.dev)
                s["lists_all"] = torch.empty(nl * self.world, dtype=torch.uint8, device=self.dev)
                s["topk"] = torch.empty(nt, dtype=torch.uint8, device=self.dev)
                s["topk_all"] = torch.empty(nt * self.world, dtype=torch.uint8, device=self.dev)
            s["done"].record(torch.cuda.current_stream(self.dev))
            self.slots.append(s)
        self.n_submitted = 0

    # ---- the tail of one batch, on the side stream ----
    def _enqueue_tail(self, t: int, beside_next_stage1: bool):
        s = self.slots[t & 1]
        if not s["tail_pending"]:
            return
        s["tail_pending"] = False
        Q = s["Q"]
        qk = s["q"][:Q]
        with torch.cuda.stream(self.side):
            if beside_next_stage1:
                # start when the NEXT batch's full stage-1 kernel starts, so this tail runs beside it
                _native.check(_native.lib().mfar_stream_wait_stage1_start(self.ix._h, self.side.cuda_stream))
            self.side.wait_event(s["stage1"])
            out = dict(ids=s["ids"][:Q], scores=s["scores"][:Q], n_valid=s["n_valid"][:Q])
            if self.world == 1:
                self.ix.search_stage2(qk, self.W, s["fid"][:Q], self.mask, self.k1, self.k2, self.query_cond, slot=t & 1, out=out)
            else:
                dist = torch.distributed
                dist.all_gather_into_tensor(s["lists_all"], s["lists"], group=self.group)
                self.ix.search_owned(s["lists_all"], self.world, qk, self.W, s["topk"], self.mask, self.k1, self.k2, self.sentinel,
                                     self.query_cond, slot=t & 1)
                dist.all_gather_into_tensor(s["topk_all"], s["topk"], group=self.group)
                _index.merge_topk(s["topk_all"], self.world, Q, self.k2, device=self.ix.device, out=out)
            s["done"].record(self.side)

    def submit(self, q) -> int:
        """q: [Q, E] float32 CUDA tensor (Q == max_batch for the sharded path: fixed payload size)."""
        t = self.n_submitted
        self.n_submitted += 1
        s = self.slots[t & 1]
        Q = q.shape[0]
        if Q > self.Qmax or (self.world > 1 and Q != self.Qmax):
            raise ValueError("batch size does not fit the pipeline's buffers")
        if t >= 2:
            self._enqueue_tail(t - 2, False)          # normally already enqueued by submit(t - 1)
        cur = torch.cuda.current_stream(self.dev)
        self.main.wait_stream(cur)                    # q may have been produced on the caller's stream
        self.main.wait_event(s["done"])               # the slot's previous tail has finished with these buffers
        s["Q"] = Q
        with torch.cuda.stream(self.main):
            qk = s["q"][:Q]
            qk.copy_(q)
            if self.world == 1:
                _native.check(_native.lib().mfar_retrieve_fields(
                    self.ix._h, qk.data_ptr(), Q, int(self.k1), int(bool(self.sentinel)), s["fid"].data_ptr(), s["fsc"].data_ptr(), 1,
                    self.main.cuda_stream))
            else:
                self.ix.retrieve_lists(qk, s["lists"], self.k1, self.sentinel)
            s["stage1"].record(self.main)
        s["tail_pending"] = True
        if t >= 1:
            self._enqueue_tail(t - 1, True)           # previous batch's tail runs beside this batch's stage 1
        return t

    def result(self, ticket: int):
        if ticket < self.n_submitted - 2 or ticket >= self.n_submitted:
            raise ValueError("ticket is no longer (or not yet) in flight")
        self._enqueue_tail(ticket, False)
        s = self.slots[ticket & 1]
        torch.cuda.current_stream(self.dev).wait_event(s["done"])
        Q = s["Q"]
        return dict(ids=s["ids"][:Q], scores=s["scores"][:Q], n_valid=s["n_valid"][:Q])
