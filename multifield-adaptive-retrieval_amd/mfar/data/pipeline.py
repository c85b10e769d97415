"""Software pipeline over query batches, three launches deep (HIP streams + events through torch).

The reference scores one batch at a time, synchronously (reference mfar/modeling/contrastive.py:559-563 -> 669-704).
On the GPU the per-batch work is one long scan (stage 1: per-field exhaustive top-k) and a chain of short, latency- or
gather-bound kernels around it.  `PipelinedSearcher` keeps the scans of consecutive launches back to back on a
high-priority stream (`mfar_stage1_begin`: query prep, sample pass, scan, list merge) and runs everything that follows a
scan -- exact re-scoring + certificate of the screened lists (`mfar_stage1_finish`), candidate union, stage 2, mixer;
with several GPUs the two small all-gathers (the second also carries the certificate flag), `mfar_search_owned` and
`mfar_merge_topk` -- on a side stream BESIDE the next launch's scan: the 16-bit scan kernels keep their doc tiles in
registers, so those small kernels fit next to them on every CU.  Results are identical to `MultiFieldIndex.search` /
`ShardedSearcher.search`.

COALESCING.  The screened scan is HBM-bound: it reads the screen slab once per launch whatever the number of query columns
(up to 128, include/mfar_hip.h "the WIDE screened pass").  The reference's batch is 64 queries (dev_batch_size,
train.py:45); when the index offers the wide pass the searcher therefore holds a submitted batch until the next one
arrives and scans both with ONE launch -- half the scan bytes per query, same result bits per query (a query's lists do
not depend on its neighbours).  `result()` of a batch that is still waiting launches it alone.

    ps = PipelinedSearcher(index, W, mask)
    t0 = ps.submit(q0)            # returns immediately (everything is enqueued asynchronously, or held for coalescing)
    t1 = ps.submit(q1)
    r0 = ps.result(t0)            # dict(ids, scores, n_valid): valid until ps.depth LATER launches started (a launch = ps.coalesce batches,
                                  # fewer when a result / flush / set_weights cut it short: include/mfar_hip.h)

To keep the pipeline full, ask for results `ps.lag` (= depth * coalesce - 1) submissions late: submit(i); result(i - lag).
`W` / `mask` are read when a launch is issued: call `flush()` before replacing them (mask_fields sweeps).

A screened launch whose certificate failed (include/mfar_hip.h, `any_fail`) is detected in `result()` and redone there
through the non-split entry points (which repair the failed fields with the exact pass), so what `result()` returns is
always the exact answer.  Data on which certificates keep failing does not keep paying for that: the library switches to
repairing on the device, and switches fields whose lists fail launch after launch off altogether (exact pass only, re-probed
now and then) -- include/mfar_hip.h "AUTO-OFF and inline repair".
"""
import torch

from mfar import _native
from mfar.data import index as _index


_STREAMS = {}


def _shared_stream(torch, dev, role: str, priority: int):
    """ONE stream per (device, role, priority) and process.  Searchers that are alive at the same time share them: their launches are
    ordered on the scan stream and simply take turns (two searchers do not overlap each other; each still overlaps its own scans and
    tails).  The device is normalised to an explicit index, so 'cuda' and 'cuda:0' name the same set."""
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (int(index), role, priority)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", int(index)), priority=priority)
    return _STREAMS[key]


class PipelinedSearcher:
    def __init__(self, index, W, mask=None, k1: int = 100, k2: int = 100, sentinel: bool = True, query_cond: bool = True,
                 max_batch: int = 64, group=None, coalesce=None, exchange=None, masks=None, depth=None):
        self.ix, self.W, self.mask = index, W, mask
        # masks [M, F]: a SWEEP of field masks (mask_fields.py:143-170) -- stage 1, the candidate union and stage 2 run once per
        # launch, the mixer once per mask; results then carry a leading mask dimension.  With several ranks the second all-gather
        # carries one local top-k payload per mask.
        self.masks = None if masks is None else masks.float().contiguous()
        self.M = 0 if masks is None else int(self.masks.shape[0])
        self.k1, self.k2, self.sentinel, self.query_cond = k1, k2, sentinel, query_cond
        self.group = group
        dist = torch.distributed
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # exchange=True runs the lists-first exchange (the two all-gathers + owned scoring + merge) even with ONE rank: the
        # RCCL code path of the multi-GPU pipeline can then be exercised on a one-GPU box (tests/test_gpu_multirank.py)
        self.sharded = self.world > 1 if exchange is None else bool(exchange)
        if self.sharded and not (dist.is_available() and dist.is_initialized()):
            raise ValueError("the exchange path needs an initialised torch.distributed process group")
        self.dev = torch.device(f"cuda:{index.device}")
        # stream priorities: by default the scans are dispatched ahead of the small kernels.  MFAR_TAIL_PRIORITY=1 flips it (the
        # tail's workgroups then take the CU slots a finely cut scan grid frees while it runs; measured)
        import os
        flip = os.environ.get("MFAR_TAIL_PRIORITY", "0") == "1"
        # ONE set of streams per device and process, shared by every searcher: HIP maps streams onto a handful of hardware queues round-robin,
        # and the streams of the sixth searcher a process creates land on queues its own other streams already use -- scan and tail then
        # serialise (bench.py's BASELINE-config legs, run after five other legs: 62 k q/s against 70 k for the same shape in a fresh
        # process, the scan kernel itself unchanged).  Searchers alive at the same time merely take turns.
        self.main = _shared_stream(torch, self.dev, "main", 0 if flip else -1)
        # DEPTH: launches in flight = slots of scratch (include/mfar_hip.h: up to 4).  The wide scan fills the register file, so a
        # tail only runs in the gaps between scans.  With two slots scan i+2 has to wait for tail i, which scan i+1 kept off the
        # machine: every gap is as long as one whole tail.  With three (default) the next scan starts as soon as its own sample
        # pass is through, and the tails of consecutive launches -- on alternating streams -- share the gaps: 50.4 -> 52.0 k q/s at
        # 1 M x 8, 43.9 -> 47.9 k at 129 k x 22, 159 -> 172 k at the 125 k-row shard; four slots measured below three
        # (MFAR_PIPE_DEPTH).
        self.depth = int(depth if depth is not None else os.environ.get("MFAR_PIPE_DEPTH", "3"))
        if not 2 <= self.depth <= 4:
            raise ValueError("pipeline depth must be 2, 3 or 4")
        self.sides = [_shared_stream(torch, self.dev, f"side{i}", -1 if flip else 0) for i in range(2 if self.depth > 2 else 1)]
        if os.environ.get("MFAR_PIPE_SERIAL", "0") == "1":      # diagnostic: tails on the scan stream (no kernel of a tail beside a scan)
            self.sides = [self.main]
        self.side = self.sides[0]
        index.set_repair_mode(True)   # repairs are launched here only after a failure was reported (or when they are frequent)
        self.Qb = int(max_batch)      # queries per submitted batch (at most)
        cap = index.max_split_batch(k1)                       # 128 with the wide screened pass, else 64
        if self.sharded:
            # every rank must cut the query stream into the same launches (fixed-size payloads are all-gathered): shards
            # whose sizes straddle the screen's row threshold, or a rank that could not allocate its screen slab, would
            # otherwise disagree -- take the smallest answer
            t = torch.tensor([cap], dtype=torch.int32, device=torch.device(f"cuda:{index.device}"))
            if dist.get_backend(group) == "gloo":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            cap = int(t.item())
        if self.Qb > cap:
            raise ValueError(f"the split-phase stage 1 takes at most {cap} queries per batch on this index")
        if coalesce is None:
            coalesce = max(1, min(2, cap // self.Qb))
        if coalesce < 1 or coalesce * self.Qb > cap:
            raise ValueError(f"coalesce x max_batch must not exceed {cap} on this index")
        self.coalesce = int(coalesce)
        self.lag = self.depth * self.coalesce - 1
        self.Qmax = self.Qb * self.coalesce                   # queries per launch
        F, E = index.n_fields, index.dim
        self.n_redone = 0             # launches whose screen certificate failed and that were redone exactly
        self.slots = []
        for _ in range(self.depth):
            lead = (self.M,) if self.M else ()
            s = dict(q=torch.zeros(self.Qmax, E, device=self.dev),
                     ids=torch.empty(*lead, self.Qmax, k2, dtype=torch.int64, device=self.dev),
                     scores=torch.empty(*lead, self.Qmax, k2, device=self.dev),
                     n_valid=torch.empty(*lead, self.Qmax, dtype=torch.int32, device=self.dev),
                     W=torch.empty_like(W, device=self.dev), mask=torch.ones(F, device=self.dev),
                     fail=torch.zeros(1, dtype=torch.int32, device=self.dev),
                     fail_host=torch.zeros(1, dtype=torch.int32).pin_memory(),
                     stage1=torch.cuda.Event(), done=torch.cuda.Event(), Q=0, checked=True, launch=-1)
            if not self.sharded:
                s["fid"] = torch.empty(self.Qmax, F, k1, dtype=torch.int64, device=self.dev)
                s["fsc"] = torch.empty(self.Qmax, F, k1, device=self.dev)
            else:       # lists-first exchange: two small all-gathers per launch (include/mfar_hip.h)
                nl, nt = index.lists_bytes(self.Qmax, k1), index.topk_bytes(self.Qmax, k2)
                s["lists"] = torch.empty(nl, dtype=torch.uint8, device=self.dev)
                s["lists_all"] = torch.empty(nl * self.world, dtype=torch.uint8, device=self.dev)
                s["topk"] = torch.empty(nt * max(1, self.M), dtype=torch.uint8, device=self.dev)
                s["topk_all"] = torch.empty(nt * max(1, self.M) * self.world, dtype=torch.uint8, device=self.dev)
                self._nt = nt
            s["done"].record(torch.cuda.current_stream(self.dev))
            self.slots.append(s)
        self.n_submitted = 0          # batches (tickets)
        self.n_launched = 0           # launches issued
        self._pending = []            # batches copied into the next launch's slot, not launched yet: (ticket, row offset, Q)
        self._where = {}              # ticket -> (launch, row offset, Q) for the pending batches and the launches still in flight

    # where stage 1 leaves the per-field lists of a slot: (ids, scores) as tensors or raw device addresses
    def _list_targets(self, s):
        if not self.sharded:
            Q = s["Q"]
            return s["fid"][:Q], s["fsc"][:Q]
        base = s["lists"].data_ptr()                     # lists layout: ids | scores (256-byte aligned), include/mfar_hip.h
        ids_bytes = self.Qmax * self.ix.n_fields * self.k1 * 8
        return base, base + ((ids_bytes + 255) & ~255)

    # everything after the lists of a launch are final, on the current stream
    def _tail(self, s, slot: int):
        Q = s["Q"]
        qk = s["q"][:Q]
        out = dict(ids=s["ids"][:Q], scores=s["scores"][:Q], n_valid=s["n_valid"][:Q])
        if self.M:
            # (the kernels write [M, Q, k2] densely: hand them a dense scratch view when the launch is short)
            if Q == self.Qmax:
                dense = dict(ids=s["ids"], scores=s["scores"], n_valid=s["n_valid"])
            else:
                dense = dict(ids=s["ids"].view(-1)[:self.M * Q * self.k2].view(self.M, Q, self.k2),
                             scores=s["scores"].view(-1)[:self.M * Q * self.k2].view(self.M, Q, self.k2),
                             n_valid=s["n_valid"].view(-1)[:self.M * Q].view(self.M, Q))
            s["dense_Q"] = Q
            if not self.sharded:
                self.ix.search_stage2_masks(qk, s["W"], s["fid"][:Q], self.masks, self.k1, self.k2, self.query_cond, slot=slot, out=dense,
                                            field_scores=s["fsc"][:Q], sentinel=self.sentinel)
            else:       # (sharded launches always hold Qmax queries)
                dist = torch.distributed
                dist.all_gather_into_tensor(s["lists_all"], s["lists"], group=self.group)
                self.ix.search_owned_masks(s["lists_all"], self.world, qk, s["W"], s["topk"], self.masks, self.k1, self.k2, self.sentinel,
                                           self.query_cond, slot=slot, any_fail=s["fail"])
                dist.all_gather_into_tensor(s["topk_all"], s["topk"], group=self.group)
                per_rank = s["topk_all"].view(self.world, self.M, self._nt)
                for m in range(self.M):     # the payloads of mask m from every rank, back to back, as merge_topk reads them
                    _index.merge_topk(per_rank[:, m].contiguous().view(-1), self.world, Q, self.k2, device=self.ix.device,
                                      out=dict(ids=dense["ids"][m], scores=dense["scores"][m], n_valid=dense["n_valid"][m]),
                                      any_fail=s["fail"])
        elif not self.sharded:
            self.ix.search_stage2(qk, s["W"], s["fid"][:Q], s["mask"], self.k1, self.k2, self.query_cond, slot=slot, out=out,
                                  field_scores=s["fsc"][:Q], sentinel=self.sentinel)
        else:
            dist = torch.distributed
            dist.all_gather_into_tensor(s["lists_all"], s["lists"], group=self.group)
            # the rank's certificate flag travels inside the top-k payload; merge_topk ORs the ranks' flags back into
            # s["fail"], so every rank takes the same redo decision without a third collective
            self.ix.search_owned(s["lists_all"], self.world, qk, s["W"], s["topk"], s["mask"], self.k1, self.k2, self.sentinel,
                                 self.query_cond, slot=slot, any_fail=s["fail"])
            dist.all_gather_into_tensor(s["topk_all"], s["topk"], group=self.group)
            _index.merge_topk(s["topk_all"], self.world, Q, self.k2, device=self.ix.device, out=out, any_fail=s["fail"])

    def submit(self, q) -> int:
        """q: [Q, E] float32 CUDA tensor, Q <= max_batch (Q == max_batch for the sharded path: fixed payload size)."""
        t = self.n_submitted
        Q = q.shape[0]
        if Q > self.Qb or (self.sharded and Q != self.Qb):
            raise ValueError("batch size does not fit the pipeline's buffers")
        slot = self.n_launched % self.depth
        s = self.slots[slot]
        cur = torch.cuda.current_stream(self.dev)
        if not self._pending:
            if not s["checked"]:
                self._check(self.n_launched - self.depth)      # the slot's previous launch must be verified before its buffers go
            self.main.wait_event(s["done"])           # the slot's previous tail has finished with these buffers
            cur.wait_event(s["done"])
        off = sum(p[2] for p in self._pending)
        # the batch is copied into the slot on the CALLER's stream (the scan stream waits for that stream when the launch is
        # issued): the copy then runs in the first gap it finds instead of queueing behind the previous launch's list merge on the
        # scan stream, where ~60 us of tiny copies sat between two scans
        s["q"][off:off + Q].copy_(q)
        self.main.wait_stream(cur)                    # (whatever stream the caller is on for THIS batch)
        self._pending.append((t, off, Q))
        self._where[t] = (self.n_launched, off, Q)
        self.n_submitted += 1
        if len(self._pending) == self.coalesce:
            self._launch()
        return t

    def flush(self):
        """Issue the launch of a batch that is being held for coalescing (no-op when nothing is held)."""
        if self._pending:
            self._launch()

    def _launch(self):
        slot = self.n_launched % self.depth
        s = self.slots[slot]
        side = self.sides[self.n_launched % len(self.sides)]
        Q = sum(p[2] for p in self._pending)
        if self.sharded and Q != self.Qmax:         # fixed payload size: the missing batch repeats the last real query
            with torch.cuda.stream(self.main):       # (an all-zero query cannot be certified -- every row ties at 0 -- and
                s["q"][Q:].copy_(s["q"][Q - 1:Q].expand(self.Qmax - Q, -1))   # would send the launch through the exact redo)
            Q = self.Qmax
        self._pending = []
        for tk in [tk for tk, (L, _, _) in self._where.items() if L <= self.n_launched - self.depth]:
            del self._where[tk]                       # this launch overwrites the results of the slot's previous launch
        s["Q"], s["checked"], s["launch"] = Q, False, self.n_launched
        s["failed"] = True            # until everything below is enqueued: a launch that raises part-way (out of memory ...) is redone by
        self.n_launched += 1          # _check when its result is taken, never read as is
        cur = torch.cuda.current_stream(self.dev)
        self.main.wait_stream(cur)                    # W / mask may have been produced on the caller's stream
        with torch.cuda.stream(self.main):
            qk = s["q"][:Q]
            # W / mask are snapshotted per slot -- but only when they changed since the slot's last snapshot (same tensor object,
            # same in-place version counter: the slot keeps a reference, so the address cannot be recycled under it).  Three tiny
            # copies per launch sat on the scan stream between two scans (~70 us of the ~400 us between them).
            wkey = (self.W._version, None if self.mask is None else self.mask._version)
            if s.get("wref") is not self.W or s.get("mref") is not self.mask or s.get("wkey") != wkey:
                s["W"].copy_(self.W)
                if self.mask is None:
                    s["mask"].fill_(1.0)
                else:
                    s["mask"].copy_(self.mask.reshape(-1))
                for t_ in (self.W, self.mask):            # sources allocated on the caller's stream, read on this one
                    if t_ is not None and t_.is_cuda:
                        t_.record_stream(self.main)
                s["wref"], s["mref"], s["wkey"] = self.W, self.mask, wkey
            fid, fsc = self._list_targets(s)
            self.ix.stage1_begin(qk, slot, fid, fsc, self.k1, self.sentinel)
            s["stage1"].record(self.main)
        with torch.cuda.stream(side):
            side.wait_event(s["stage1"])
            # finish REPORTS a failed certificate (the flag is read in _check, which then redoes the launch).  When failures are frequent on
            # this data the library repairs on the device instead and reports a clean launch; fields that keep failing are switched off
            # altogether (include/mfar_hip.h "AUTO-OFF and inline repair"): both decisions are taken inside the library from the flags of
            # finished launches and both are reversible, so nothing here latches.
            self.ix.stage1_finish(qk, slot, fid, fsc, self.k1, self.sentinel, any_fail=s["fail"])
            self._tail(s, slot)
            s["fail_host"].copy_(s["fail"], non_blocking=True)
            s["done"].record(side)
        s["failed"] = False

    # host side of the certificate: wait for the launch, redo it exactly if its screen could not be proven
    def _check(self, launch: int):
        if launch < 0:
            return
        slot = launch % self.depth
        s = self.slots[slot]
        if s["checked"] or s["launch"] != launch:
            return
        if s.get("failed"):           # never made it onto the streams (the error went to the caller of submit): run it now, or raise again
            failed = True
        else:
            s["done"].synchronize()
            failed = int(s["fail_host"][0]) != 0
        if not failed:
            s["checked"] = True
            return
        # A redo costs a pipeline drain plus a second pass: rare by construction (see _launch).
        self.n_redone += 1
        # a certificate failed on this data: fields with heavy-tailed row norms switch to per-row bounds from the next launch on
        # (include/mfar_hip.h "ROW MODE"; a no-op when no field is eligible; the library does the same from its own feedback)
        self.ix.activate_row_mode()
        torch.cuda.synchronize(self.dev)              # the redo uses the index's slot-0 scratch: nothing else may be in flight
        # the non-split entry points repair a failed certificate themselves: screened pass again, then the exact fp32 pass
        # for the failed fields only (cheaper than switching the screen off for the whole launch)
        Q = s["Q"]
        qk = s["q"][:Q]
        if not self.sharded:
            _native.check(_native.lib().mfar_retrieve_fields(
                self.ix._h, qk.data_ptr(), Q, int(self.k1), int(bool(self.sentinel)), s["fid"].data_ptr(), s["fsc"].data_ptr(), 1,
                torch.cuda.current_stream(self.dev).cuda_stream))
        else:
            self.ix.retrieve_lists(qk, s["lists"], self.k1, self.sentinel)
        self._tail(s, slot)
        torch.cuda.current_stream(self.dev).synchronize()
        s["checked"], s["failed"] = True, False

    def lists(self, ticket: int):
        """The stage-1 lists of a batch whose `result()` has been taken and is still valid: (field_ids [Q, F, k1] int64, field_scores
        [Q, F, k1] f32), views into the launch's slot.  Single-shard searchers only (a sharded launch keeps its lists in the exchange payload)."""
        if self.sharded:
            raise ValueError("lists() is for single-shard searchers")
        w = self._where.get(ticket)
        if w is None:
            raise ValueError("ticket is no longer (or not yet) in flight")
        launch, off, Q = w
        s = self.slots[launch % self.depth]
        if s["launch"] != launch or not s["checked"]:
            raise ValueError("take result(ticket) first")
        return s["fid"][off:off + Q], s["fsc"][off:off + Q]

    def result(self, ticket: int):
        w = self._where.get(ticket)
        if w is None or ticket >= self.n_submitted:
            raise ValueError("ticket is no longer (or not yet) in flight")
        launch, off, Q = w
        if launch == self.n_launched:                 # still held for coalescing: launch it alone
            self._launch()
        if launch < self.n_launched - self.depth:
            raise ValueError("ticket is no longer in flight")
        self._check(launch)
        s = self.slots[launch % self.depth]
        if self.M:      # [M, Q, k2]: the launch's results are laid out densely over its own query count
            QL = s["dense_Q"]
            ids = s["ids"].view(-1)[:self.M * QL * self.k2].view(self.M, QL, self.k2)
            sc = s["scores"].view(-1)[:self.M * QL * self.k2].view(self.M, QL, self.k2)
            nv = s["n_valid"].view(-1)[:self.M * QL].view(self.M, QL)
            return dict(ids=ids[:, off:off + Q], scores=sc[:, off:off + Q], n_valid=nv[:, off:off + Q])
        return dict(ids=s["ids"][off:off + Q], scores=s["scores"][off:off + Q], n_valid=s["n_valid"][off:off + Q])


class NativePipeline:
    """The same pipeline behind the C ABI (`mfar_pipeline_*`, include/mfar_hip.h; csrc/mfar_pipeline.h): streams, slots, coalescing and the
    redo of failed certificates live inside libmfar_hip.so, so a host in any language reaches the pipelined rate (INTEGRATION.md section 2
    binds exactly these entry points).  One row shard, one mask.  `submit` takes a CUDA tensor (copied on torch's current stream) or a
    numpy array (copied before the call returns); `result` returns tensors on the index's device / numpy arrays accordingly.

        np_ = NativePipeline(index, W, mask)
        t = np_.submit(q)                 # returns at once
        r = np_.result(t)                 # dict(ids, scores, n_valid); take results `np_.lag` submissions late to keep the pipe full
    """

    def __init__(self, index, W, mask=None, k1: int = 100, k2: int = 100, sentinel: bool = True, query_cond: bool = True,
                 max_batch: int = 64, depth: int = 0, coalesce: int = 0):
        import ctypes
        self.ix, self.k1, self.k2 = index, int(k1), int(k2)
        Wa = _index._Arg(W, __import__("numpy").float32, index.device)
        ma = _index._Arg(mask, __import__("numpy").float32, index.device, allow_none=True)
        on_dev = _index._same_side([Wa, ma])
        self._p = ctypes.c_void_p()
        import os
        depth = int(depth or os.environ.get("MFAR_PIPE_DEPTH", "0"))      # (0: the library's default, 3)
        _native.check(_native.lib().mfar_pipeline_create(ctypes.byref(self._p), index._h, Wa.ptr, int(bool(query_cond)), ma.ptr, self.k1, self.k2,
                                                         int(bool(sentinel)), int(max_batch), int(depth), int(coalesce), int(on_dev)))
        d, c, n, lag = (ctypes.c_int() for _ in range(4))
        _native.check(_native.lib().mfar_pipeline_info(self._p, ctypes.byref(d), ctypes.byref(c), ctypes.byref(n), ctypes.byref(lag), None))
        self.depth, self.coalesce, self.Qmax, self.lag = d.value, c.value, n.value, lag.value
        self._meta = {}
        import weakref
        if not hasattr(index, "_pipelines"):
            index._pipelines = []
        index._pipelines.append(weakref.ref(self))       # the index closes its pipelines before it goes

    @property
    def n_redone(self) -> int:
        import ctypes
        v = ctypes.c_int64()
        _native.check(_native.lib().mfar_pipeline_info(self._p, None, None, None, None, ctypes.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_p", None):
            _native.lib().mfar_pipeline_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, W, mask=None):
        import numpy as np
        Wa = _index._Arg(W, np.float32, self.ix.device)
        ma = _index._Arg(mask, np.float32, self.ix.device, allow_none=True)
        _native.check(_native.lib().mfar_pipeline_set_weights(self._p, Wa.ptr, ma.ptr, int(_index._same_side([Wa, ma]))))

    def flush(self):
        _native.check(_native.lib().mfar_pipeline_flush(self._p))

    def submit(self, q) -> int:
        import ctypes
        import numpy as np
        qa = _index._Arg(q, np.float32, self.ix.device)
        t = ctypes.c_int64(-1)
        rc = _native.lib().mfar_pipeline_submit(self._p, qa.ptr, int(qa.keep.shape[0]), int(qa.on_device),
                                                _index._current_stream(self.ix.device, qa.on_device), ctypes.byref(t))
        if t.value >= 0:                 # (registered even when the launch itself failed: the library runs it when the result is taken)
            self._meta[t.value] = (int(qa.keep.shape[0]), bool(qa.on_device))
        try:
            _native.check(rc)
        except _native.MfarError as e:
            e.ticket = t.value if t.value >= 0 else None
            raise
        for old in [k for k in self._meta if k <= t.value - 2 * self.depth * self.coalesce]:
            del self._meta[old]
        return t.value

    def result(self, ticket: int):
        import numpy as np
        Q, on_dev = self._meta[ticket]
        ids = _index._empty_like_side(on_dev, self.ix.device, (Q, self.k2), np.int64)
        sc = _index._empty_like_side(on_dev, self.ix.device, (Q, self.k2), np.float32)
        nv = _index._empty_like_side(on_dev, self.ix.device, (Q,), np.int32)
        ia, sa, na = (_index._Arg(a, d, self.ix.device) for a, d in ((ids, np.int64), (sc, np.float32), (nv, np.int32)))
        _native.check(_native.lib().mfar_pipeline_result(self._p, int(ticket), ia.ptr, sa.ptr, na.ptr, int(on_dev),
                                                         _index._current_stream(self.ix.device, on_dev)))
        return dict(ids=ids, scores=sc, n_valid=nv)

    def lists(self, ticket: int):
        """The stage-1 lists of a batch whose result is still valid: (field_ids [Q, F, k1] int64, field_scores [Q, F, k1] f32), copies."""
        import numpy as np
        Q, on_dev = self._meta[ticket]
        shape = (Q, self.ix.n_fields, self.k1)
        fid = _index._empty_like_side(on_dev, self.ix.device, shape, np.int64)
        fsc = _index._empty_like_side(on_dev, self.ix.device, shape, np.float32)
        fa, sa = _index._Arg(fid, np.int64, self.ix.device), _index._Arg(fsc, np.float32, self.ix.device)
        _native.check(_native.lib().mfar_pipeline_lists(self._p, int(ticket), fa.ptr, sa.ptr, int(on_dev), _index._current_stream(self.ix.device, on_dev)))
        return fid, fsc
