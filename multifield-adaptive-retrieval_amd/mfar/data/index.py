"""Index / vector store of the dense multi-field scoring path, MI355X-native.

Mirrors the reference's `mfar.data.index` surface (reference mfar/data/index.py):
  * `Index`                     index.py:21-37   (abstract retrieve / retrieve_batch)
  * `DenseFlatIndex`            index.py:160-232 (same constructor, `.vectors` re-assignable, `retrieve_batch`,
                                                  `score_batch`, KeyError on unknown keys)
  * `candidate_encoding_stream` index.py:234-258
  * `BM25sSparseIndex`          index.py:39-157  (sparse fields and hard-negative mining; host CPU, mfar/data/bm25.py)
and adds `MultiFieldIndex`, the on-HBM row shard of ALL fields that replaces the per-field np.memmap files
(data/util.py:28-59) and runs the whole of `trec_eval_step` (modeling/contrastive.py:669-704) on the GPU through
the C ABI of libmfar_hip.so (include/mfar_hip.h).  There is no CPU fallback: without the HIP library these
classes raise.
"""
import ctypes
from abc import ABC, abstractmethod
from typing import Dict, Generic, Iterable, List, Optional, Sequence, Tuple, TypeVar, Union

import numpy as np

from mfar import _native

Key = TypeVar("Key")
Query = TypeVar("Query")


class Index(ABC, Generic[Key, Query]):
    """Anything that can be searched (reference index.py:21-37)."""

    @abstractmethod
    def retrieve(self, query: Query, top_k: int) -> Sequence[Tuple[Key, float]]:
        raise NotImplementedError

    def retrieve_batch(self, queries: Sequence[Query], top_k: int) -> Sequence[Sequence[Tuple[Key, float]]]:
        return [self.retrieve(q, top_k) for q in queries]


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class _Arg:
    """A host (numpy) or device (torch.cuda) buffer handed to the C ABI."""

    def __init__(self, x, dtype, device_index: int, allow_none=False):
        self.keep = None
        self.ptr = None
        self.on_device = None
        if x is None:
            if not allow_none:
                raise ValueError("missing array")
            return
        if _is_torch(x):
            import torch
            want = {np.float32: torch.float32, np.int64: torch.int64, np.int32: torch.int32, np.uint8: torch.uint8}[dtype]
            if x.dtype != want:
                raise TypeError(f"expected {want}, got {x.dtype}")
            if not x.is_contiguous():
                raise ValueError("tensor must be contiguous")
            if x.is_cuda:
                if x.device.index != device_index:
                    raise ValueError(f"tensor lives on {x.device}, index is on cuda:{device_index}")
                self.on_device = True
                self.keep = x
                self.ptr = x.data_ptr()
                return
            x = x.numpy()
        a = np.ascontiguousarray(x, dtype=dtype)
        self.keep = a
        self.ptr = a.ctypes.data
        self.on_device = False


def _same_side(args):
    sides = {a.on_device for a in args if a.on_device is not None}
    if len(sides) > 1:
        raise ValueError("all arrays of one call must be on the same side (all host or all on the index's GPU)")
    return bool(sides.pop()) if sides else False


def _dev_ptr(x):
    """CUDA tensor -> its device address; an int is taken as a device address already."""
    return int(x) if isinstance(x, int) else int(x.data_ptr())


def _current_stream(device_index: int, on_device: bool):
    if not on_device:
        return None
    import torch
    return torch.cuda.current_stream(device_index).cuda_stream


def _empty_like_side(on_device, device_index, shape, dtype):
    if on_device:
        import torch
        td = {np.float32: torch.float32, np.int64: torch.int64, np.int32: torch.int32, np.uint8: torch.uint8}[dtype]
        return torch.empty(shape, dtype=td, device=f"cuda:{device_index}")
    return np.empty(shape, dtype=dtype)


class MultiFieldIndex:
    """Rows [row_offset, row_offset + n_rows) of every dense field, resident in HBM as one tiled fp32 slab.

    Replaces `read_and_create_indices`' per-field memmaps + DenseFlatIndex objects (reference modeling/util.py:73-108)
    and the row split of `on_eval_start` (contrastive.py:470).  Arrays may be numpy (host path: copied, synchronous)
    or torch CUDA tensors on the index's device (device path: zero-copy, asynchronous on torch's current stream).
    """

    def __init__(self, n_rows: int, n_fields: int, dim: int, device: int = 0, row_offset: int = 0, dtype: str = "f32"):
        L = _native.lib()
        h = ctypes.c_void_p()
        dt = {"f32": _native.DTYPE_F32, "bf16": _native.DTYPE_BF16}[dtype]
        _native.check(L.mfar_index_create(ctypes.byref(h), int(device), int(n_rows), int(row_offset), int(n_fields), int(dim), dt))
        self._h = h
        self.n_rows, self.n_fields, self.dim = int(n_rows), int(n_fields), int(dim)
        self.device, self.row_offset, self.dtype = int(device), int(row_offset), dtype

    def close(self):
        if getattr(self, "_h", None):
            for ref in getattr(self, "_pipelines", []):      # pipelines over this handle go first (mfar.data.pipeline.NativePipeline)
                pl = ref()
                if pl is not None:
                    pl.close()
            self._pipelines = []
            _native.lib().mfar_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def slab_bytes(self) -> int:
        v = ctypes.c_int64()
        _native.check(_native.lib().mfar_index_info(self._h, None, None, None, None, None, ctypes.byref(v)))
        return v.value

    # ---- rows in / out (MemoryMapDict.__setitem__ / .file, data/util.py:37-41) ----
    def resident_bytes(self) -> dict:
        """HBM the index keeps between searches: rows (slab), fp16 screen slab, 16-bit gather slab, unique-row tables + row norms, the
        score dumps of the pipeline slots that use one; `ratio` = all of it over the rows alone (include/mfar_hip.h)."""
        v = [ctypes.c_int64() for _ in range(5)]
        _native.check(_native.lib().mfar_index_resident_bytes(self._h, *[ctypes.byref(x) for x in v]))
        rows, screen, gather, tables, dumps = (x.value for x in v)
        total = rows + screen + gather + tables + dumps
        return dict(rows=rows, screen=screen, gather=gather, tables=tables, dumps=dumps, total=total, ratio=total / max(1, rows))

    def write_rows(self, field: int, local_row0: int, rows) -> None:
        a = _Arg(rows, np.float32, self.device)
        shape = tuple(a.keep.shape)
        if len(shape) != 2 or shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}], got {shape}")
        _native.check(_native.lib().mfar_index_write_rows(self._h, int(field), int(local_row0), shape[0], a.ptr,
                                                           int(a.on_device), _current_stream(self.device, a.on_device)))

    def read_rows(self, field: int, local_row0: int = 0, n: Optional[int] = None, out=None):
        n = self.n_rows - local_row0 if n is None else n
        if out is None:
            out = np.empty((n, self.dim), dtype=np.float32)
        a = _Arg(out, np.float32, self.device)
        _native.check(_native.lib().mfar_index_read_rows(self._h, int(field), int(local_row0), int(n), a.ptr,
                                                          int(a.on_device), _current_stream(self.device, a.on_device)))
        return out

    # ---- stage 1: DenseFlatIndex.retrieve_batch for all fields (index.py:181-222; contrastive.py:672-674) ----
    def retrieve_fields(self, q, top_k: int = 100, sentinel: bool = True):
        qa = _Arg(q, np.float32, self.device)
        Q = qa.keep.shape[0]
        ids = _empty_like_side(qa.on_device, self.device, (Q, self.n_fields, top_k), np.int64)
        sc = _empty_like_side(qa.on_device, self.device, (Q, self.n_fields, top_k), np.float32)
        ia, sa = _Arg(ids, np.int64, self.device), _Arg(sc, np.float32, self.device)
        _native.check(_native.lib().mfar_retrieve_fields(self._h, qa.ptr, Q, int(top_k), int(bool(sentinel)), ia.ptr, sa.ptr,
                                                          int(qa.on_device), _current_stream(self.device, qa.on_device)))
        return ids, sc

    def retrieve_field(self, field: int, q, top_k: int = 100, sentinel: bool = True):
        """Stage 1 of ONE field (one `retrieve_batch` call of the reference's per-field loop): -> ids [Q, k], scores [Q, k]."""
        qa = _Arg(q, np.float32, self.device)
        Q = qa.keep.shape[0]
        ids = _empty_like_side(qa.on_device, self.device, (Q, top_k), np.int64)
        sc = _empty_like_side(qa.on_device, self.device, (Q, top_k), np.float32)
        ia, sa = _Arg(ids, np.int64, self.device), _Arg(sc, np.float32, self.device)
        _native.check(_native.lib().mfar_retrieve_field(self._h, int(field), qa.ptr, Q, int(top_k), int(bool(sentinel)), ia.ptr, sa.ptr,
                                                         int(qa.on_device), _current_stream(self.device, qa.on_device)))
        return ids, sc

    # ---- stage 2: DenseFlatIndex.score_batch for all fields (index.py:227-232) ----
    def score_candidates(self, q, cand):
        qa, ca = _Arg(q, np.float32, self.device), _Arg(cand, np.int64, self.device)
        on_dev = _same_side([qa, ca])
        Q, C = ca.keep.shape
        out = _empty_like_side(on_dev, self.device, (Q, C, self.n_fields), np.float32)
        oa = _Arg(out, np.float32, self.device)
        _native.check(_native.lib().mfar_score_candidates(self._h, qa.ptr, Q, ca.ptr, C, oa.ptr, int(on_dev),
                                                           _current_stream(self.device, on_dev)))
        return out

    # ---- the whole trec_eval_step scorer (contrastive.py:669-704) ----
    def search(self, q, W, mask=None, k1: int = 100, k2: int = 100, sentinel: bool = True, query_cond: bool = True,
               return_fields: bool = False, out=None):
        """Returns dict(ids [Q,k2] int64, scores [Q,k2] f32, n_valid [Q] int32 [, field_ids, field_scores, n_cand]).
        `out` may carry preallocated ids/scores/n_valid arrays (device path: no allocation in steady state)."""
        qa, Wa = _Arg(q, np.float32, self.device), _Arg(W, np.float32, self.device)
        ma = _Arg(mask, np.float32, self.device, allow_none=True)
        on_dev = _same_side([qa, Wa, ma])
        Q = qa.keep.shape[0]
        out = dict(out) if out else {}
        ids = out.get("ids") if out.get("ids") is not None else _empty_like_side(on_dev, self.device, (Q, k2), np.int64)
        sc = out.get("scores") if out.get("scores") is not None else _empty_like_side(on_dev, self.device, (Q, k2), np.float32)
        nv = out.get("n_valid") if out.get("n_valid") is not None else _empty_like_side(on_dev, self.device, (Q,), np.int32)
        ia, sa, na = _Arg(ids, np.int64, self.device), _Arg(sc, np.float32, self.device), _Arg(nv, np.int32, self.device)
        fid = fsc = nc = None
        fa = fs = nca = _Arg(None, np.int64, self.device, allow_none=True)
        if return_fields:
            fid = _empty_like_side(on_dev, self.device, (Q, self.n_fields, k1), np.int64)
            fsc = _empty_like_side(on_dev, self.device, (Q, self.n_fields, k1), np.float32)
            nc = _empty_like_side(on_dev, self.device, (Q,), np.int32)
            fa, fs, nca = _Arg(fid, np.int64, self.device), _Arg(fsc, np.float32, self.device), _Arg(nc, np.int32, self.device)
        _native.check(_native.lib().mfar_search_two_stage(
            self._h, qa.ptr, Q, Wa.ptr, int(bool(query_cond)), ma.ptr, int(k1), int(k2), int(bool(sentinel)),
            ia.ptr, sa.ptr, na.ptr, fa.ptr, fs.ptr, nca.ptr, int(on_dev), _current_stream(self.device, on_dev)))
        res = dict(ids=ids, scores=sc, n_valid=nv)
        if return_fields:
            res.update(field_ids=fid, field_scores=fsc, n_cand=nc)
        return res

    # ---- fused mode: exhaustive top-k of the gate-folded inner product (include/mfar_hip.h) ----
    def search_fused(self, q, W, mask=None, k: int = 100, query_cond: bool = True):
        """-> dict(ids [Q,k] int64, scores [Q,k] f32): top-k over ALL rows of sum_f softmax_f(q W) mask_f <q, d_f> (the
        reference mixes only the union of the per-field lists: Recall parity, not id parity)."""
        qa, Wa = _Arg(q, np.float32, self.device), _Arg(W, np.float32, self.device)
        ma = _Arg(mask, np.float32, self.device, allow_none=True)
        on_dev = _same_side([qa, Wa, ma])
        Q = qa.keep.shape[0]
        ids = _empty_like_side(on_dev, self.device, (Q, k), np.int64)
        sc = _empty_like_side(on_dev, self.device, (Q, k), np.float32)
        ia, sa = _Arg(ids, np.int64, self.device), _Arg(sc, np.float32, self.device)
        _native.check(_native.lib().mfar_search_fused(self._h, qa.ptr, Q, Wa.ptr, int(bool(query_cond)), ma.ptr, int(k), ia.ptr, sa.ptr,
                                                      int(on_dev), _current_stream(self.device, on_dev)))
        return dict(ids=ids, scores=sc)

    # ---- multi-GPU: local half + merge of all-gathered payloads ----
    def payload_bytes(self, Q: int, k1: int = 100) -> int:
        return int(_native.lib().mfar_payload_bytes(int(Q), self.n_fields, int(k1)))

    def search_stage2(self, q, W, field_ids, mask=None, k1: int = 100, k2: int = 100, query_cond: bool = True, slot: int = 0,
                      out=None, field_scores=None, sentinel: bool = True):
        """Second half of `search` given the stage-1 lists (device tensors only, asynchronous on the current stream).
        `field_scores` (the lists' exact scores, with their padding convention `sentinel`): when given, stage 2 does not gather a
        candidate's row again for the field whose list it came from."""
        qa, Wa = _Arg(q, np.float32, self.device), _Arg(W, np.float32, self.device)
        ma = _Arg(mask, np.float32, self.device, allow_none=True)
        fa = _Arg(field_ids, np.int64, self.device)
        if not _same_side([qa, Wa, ma, fa]):
            raise ValueError("search_stage2 needs CUDA tensors")
        Q = qa.keep.shape[0]
        out = dict(out) if out else {}
        ids = out.get("ids") if out.get("ids") is not None else _empty_like_side(True, self.device, (Q, k2), np.int64)
        sc = out.get("scores") if out.get("scores") is not None else _empty_like_side(True, self.device, (Q, k2), np.float32)
        nv = out.get("n_valid") if out.get("n_valid") is not None else _empty_like_side(True, self.device, (Q,), np.int32)
        ia, sa, na = _Arg(ids, np.int64, self.device), _Arg(sc, np.float32, self.device), _Arg(nv, np.int32, self.device)
        fsa = _Arg(field_scores, np.float32, self.device, allow_none=True)
        _native.check(_native.lib().mfar_search_stage2(
            self._h, qa.ptr, Q, Wa.ptr, int(bool(query_cond)), ma.ptr, int(k1), int(k2), fa.ptr, fsa.ptr, int(bool(sentinel)), int(slot),
            ia.ptr, sa.ptr, na.ptr, None, _current_stream(self.device, True)))
        return dict(ids=ids, scores=sc, n_valid=nv)

    def search_stage2_masks(self, q, W, field_ids, masks, k1: int = 100, k2: int = 100, query_cond: bool = True, slot: int = 0,
                            out=None, field_scores=None, sentinel: bool = True):
        """`search_stage2` for a sweep of field masks [M, F]: candidate union and stage 2 once, the mixer once per mask
        (include/mfar_hip.h mfar_search_stage2_masks).  -> ids / scores [M, Q, k2], n_valid [M, Q]."""
        qa, Wa = _Arg(q, np.float32, self.device), _Arg(W, np.float32, self.device)
        ma, fa = _Arg(masks, np.float32, self.device), _Arg(field_ids, np.int64, self.device)
        if not _same_side([qa, Wa, ma, fa]):
            raise ValueError("search_stage2_masks needs CUDA tensors")
        Q, M = qa.keep.shape[0], ma.keep.shape[0]
        if ma.keep.dim() != 2 or ma.keep.shape[1] != self.n_fields:
            raise ValueError("masks must be [M, n_fields]")
        out = dict(out) if out else {}
        ids = out.get("ids") if out.get("ids") is not None else _empty_like_side(True, self.device, (M, Q, k2), np.int64)
        sc = out.get("scores") if out.get("scores") is not None else _empty_like_side(True, self.device, (M, Q, k2), np.float32)
        nv = out.get("n_valid") if out.get("n_valid") is not None else _empty_like_side(True, self.device, (M, Q), np.int32)
        ia, sa, na = _Arg(ids, np.int64, self.device), _Arg(sc, np.float32, self.device), _Arg(nv, np.int32, self.device)
        fsa = _Arg(field_scores, np.float32, self.device, allow_none=True)
        _native.check(_native.lib().mfar_search_stage2_masks(
            self._h, qa.ptr, Q, Wa.ptr, int(bool(query_cond)), ma.ptr, int(M), int(k1), int(k2), fa.ptr, fsa.ptr, int(bool(sentinel)), int(slot),
            ia.ptr, sa.ptr, na.ptr, None, _current_stream(self.device, True)))
        return dict(ids=ids, scores=sc, n_valid=nv)

    def search_local(self, q, k1: int = 100, sentinel: bool = True, payload=None, phases: int = 3):
        qa = _Arg(q, np.float32, self.device)
        Q = qa.keep.shape[0]
        nbytes = self.payload_bytes(Q, k1)
        if payload is None:
            payload = _empty_like_side(qa.on_device, self.device, (nbytes,), np.uint8)
        pa = _Arg(payload, np.uint8, self.device)
        have = payload.numel() if _is_torch(payload) else pa.keep.size
        if have < nbytes:
            raise ValueError("payload buffer too small")
        _native.check(_native.lib().mfar_search_local(self._h, qa.ptr, Q, int(k1), int(bool(sentinel)), pa.ptr, int(phases),
                                                       int(qa.on_device), _current_stream(self.device, qa.on_device)))
        return payload

    # ---- lists-first exchange (two small collectives; see include/mfar_hip.h) ----
    def lists_bytes(self, Q: int, k1: int = 100) -> int:
        return int(_native.lib().mfar_lists_bytes(int(Q), self.n_fields, int(k1)))

    @staticmethod
    def topk_bytes(Q: int, k2: int = 100) -> int:
        return int(_native.lib().mfar_topk_bytes(int(Q), int(k2)))

    def retrieve_lists(self, q, lists, k1: int = 100, sentinel: bool = True):
        """Stage 1 into a flat uint8 CUDA buffer of lists_bytes() (the unit of the first all-gather)."""
        qa, la = _Arg(q, np.float32, self.device), _Arg(lists, np.uint8, self.device)
        if not _same_side([qa, la]):
            raise ValueError("retrieve_lists needs CUDA tensors")
        _native.check(_native.lib().mfar_retrieve_lists(self._h, qa.ptr, qa.keep.shape[0], int(k1), int(bool(sentinel)), la.ptr,
                                                         _current_stream(self.device, True)))
        return lists

    # ---- split-phase stage 1 (two slots; see include/mfar_hip.h).  q is a CUDA tensor, field_ids / field_scores are CUDA
    # tensors or raw device addresses (the lists-first exchange passes offsets into its flat payload buffer)
    def stage1_begin(self, q, slot: int, field_ids, field_scores, k1: int = 100, sentinel: bool = True):
        qa = _Arg(q, np.float32, self.device)
        if not qa.on_device:
            raise ValueError("stage1_begin needs CUDA tensors")
        _native.check(_native.lib().mfar_stage1_begin(self._h, qa.ptr, qa.keep.shape[0], int(k1), int(bool(sentinel)), int(slot),
                                                       _dev_ptr(field_ids), _dev_ptr(field_scores), _current_stream(self.device, True)))

    def stage1_finish(self, q, slot: int, field_ids, field_scores, k1: int = 100, sentinel: bool = True, any_fail=None):
        qa = _Arg(q, np.float32, self.device)
        if not qa.on_device:
            raise ValueError("stage1_finish needs CUDA tensors")
        _native.check(_native.lib().mfar_stage1_finish(self._h, qa.ptr, qa.keep.shape[0], int(k1), int(bool(sentinel)), int(slot),
                                                        _dev_ptr(field_ids), _dev_ptr(field_scores),
                                                        None if any_fail is None else _dev_ptr(any_fail),
                                                        _current_stream(self.device, True)))

    def search_owned(self, gathered_lists, n_shards: int, q, W, topk, mask=None, k1: int = 100, k2: int = 100, sentinel: bool = True,
                     query_cond: bool = True, slot: int = 0, any_fail=None):
        """Merge the gathered lists, score + mix the candidates this shard owns, write the local top-k2 payload (which
        also carries `any_fail`, this rank's certificate flag of the batch: a CUDA int32 tensor or None)."""
        qa, Wa = _Arg(q, np.float32, self.device), _Arg(W, np.float32, self.device)
        ma = _Arg(mask, np.float32, self.device, allow_none=True)
        ga, ta = _Arg(gathered_lists, np.uint8, self.device), _Arg(topk, np.uint8, self.device)
        if not _same_side([qa, Wa, ma, ga, ta]):
            raise ValueError("search_owned needs CUDA tensors")
        _native.check(_native.lib().mfar_search_owned(self._h, ga.ptr, int(n_shards), qa.ptr, qa.keep.shape[0], Wa.ptr,
                                                       int(bool(query_cond)), ma.ptr, int(k1), int(k2), int(bool(sentinel)), int(slot),
                                                       None if any_fail is None else _dev_ptr(any_fail), ta.ptr,
                                                       _current_stream(self.device, True)))
        return topk

    def search_owned_masks(self, gathered_lists, n_shards: int, q, W, topk, masks, k1: int = 100, k2: int = 100, sentinel: bool = True,
                           query_cond: bool = True, slot: int = 0, any_fail=None):
        """`search_owned` for a sweep of field masks [M, F]: `topk` receives M payloads of topk_bytes() each, back to back."""
        qa, Wa = _Arg(q, np.float32, self.device), _Arg(W, np.float32, self.device)
        ma = _Arg(masks, np.float32, self.device)
        ga, ta = _Arg(gathered_lists, np.uint8, self.device), _Arg(topk, np.uint8, self.device)
        if not _same_side([qa, Wa, ma, ga, ta]):
            raise ValueError("search_owned_masks needs CUDA tensors")
        _native.check(_native.lib().mfar_search_owned_masks(self._h, ga.ptr, int(n_shards), qa.ptr, qa.keep.shape[0], Wa.ptr,
                                                             int(bool(query_cond)), ma.ptr, int(ma.keep.shape[0]), int(k1), int(k2),
                                                             int(bool(sentinel)), int(slot),
                                                             None if any_fail is None else _dev_ptr(any_fail), ta.ptr,
                                                             _current_stream(self.device, True)))
        return topk

    # instrumentation for bench.py
    def set_timing(self, enable: bool):
        _native.check(_native.lib().mfar_set_timing(self._h, int(bool(enable))))

    def stage1_timing(self):
        tot, n = ctypes.c_double(), ctypes.c_int()
        _native.check(_native.lib().mfar_stage1_timing(self._h, ctypes.byref(tot), ctypes.byref(n)))
        return tot.value, n.value

    def last_stage1_kernel(self) -> str:
        """Name of the scan kernel the most recent timed stage-1 launch ran (include/mfar_hip.h)."""
        return _native.lib().mfar_last_stage1_kernel(self._h).decode()

    def set_wgs_per_cu(self, n: int):
        _native.check(_native.lib().mfar_set_wgs_per_cu(self._h, int(n)))

    def max_split_batch(self, k1: int = 100) -> int:
        """Queries one stage1_begin / stage1_finish batch may hold: 128 when the wide screened pass is available for this index
        (builds the screen slab if it is not current), else 64 (include/mfar_hip.h)."""
        return int(_native.lib().mfar_max_split_batch(self._h, int(k1)))

    def set_wide(self, enable: bool = True):
        """Blocks of 65 .. 128 queries use the wide (one fp16 term, 128 columns) screened pass; False: always 64 per pass."""
        _native.check(_native.lib().mfar_set_wide(self._h, int(bool(enable))))

    def set_repair_mode(self, thorough: bool = True):
        """Repairs of failed certificates on the device: thorough = with their own sample pass (faster when many fields fail,
        two more launches when none does) -- for callers that repair only after a reported failure (include/mfar_hip.h)."""
        _native.check(_native.lib().mfar_set_repair_mode(self._h, int(bool(thorough))))

    def set_screen(self, mode: int = 1, eps_mult: float = 1.0):
        """Certified fp16 screening of an fp32 index (include/mfar_hip.h): 0 off, 1 auto, 2 whenever possible.
        Outputs are bit-identical in every mode; `eps_mult` is a test knob (1 = rigorous proof)."""
        _native.check(_native.lib().mfar_set_screen(self._h, int(mode), float(eps_mult)))

    def set_stage2_mode(self, mode: int = 1):
        """Stage 2 of an fp32 index (include/mfar_hip.h): 0 = gather every (candidate, field) row from the fp32 slab, 1 = the
        certified two-level stage 2 whenever its gather slab is current (calls with at most two masks), 2 = also for mask sweeps
        of any size.  Outputs are bit-identical in every mode."""
        _native.check(_native.lib().mfar_set_stage2_mode(self._h, int(mode)))

    def set_row_mode(self, mode: int = 1):
        """ROW MODE of the certified screen (include/mfar_hip.h): heavy-tailed fields rank rows by a per-row upper bound.  0 never, 1 auto
        (activated after a failed certificate: `activate_row_mode`, called by PipelinedSearcher), 2 always.  Same output bits."""
        _native.check(_native.lib().mfar_set_row_mode(self._h, int(mode)))

    def activate_row_mode(self):
        _native.check(_native.lib().mfar_row_mode_activate(self._h))

    def row_mode_info(self) -> dict:
        e, a = ctypes.c_uint32(), ctypes.c_uint32()
        _native.check(_native.lib().mfar_row_mode_info(self._h, ctypes.byref(e), ctypes.byref(a)))
        return dict(eligible=[f for f in range(self.n_fields) if (e.value >> f) & 1], active=[f for f in range(self.n_fields) if (a.value >> f) & 1])

    def set_auto_off(self, mode: int = 1, off_fails: int = 0, probe_every: int = 0):
        """AUTO-OFF of the certified screen (include/mfar_hip.h): a field whose certificates failed in `off_fails` of its last 16 screened
        launches is scanned by the exact pass only, re-probed every `probe_every` launches (0 = keep the current value).  Same bits."""
        _native.check(_native.lib().mfar_set_auto_off(self._h, int(mode), int(off_fails), int(probe_every)))

    def auto_off_info(self) -> dict:
        m, a, b, c, r = ctypes.c_uint32(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _native.check(_native.lib().mfar_auto_off_info(self._h, ctypes.byref(m), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(r)))
        return dict(off=[f for f in range(self.n_fields) if (m.value >> f) & 1], n_switched_off=a.value, n_switched_on=b.value,
                    n_probes=c.value, inline_repair=bool(r.value))

    def set_stage2_kernels(self, family: int = 1):
        """Kernel family of the tail (include/mfar_hip.h): 1 = round 6 (default), 0 = rounds 3-5.  Identical results."""
        _native.check(_native.lib().mfar_set_stage2_kernels(self._h, int(family)))

    def set_tier2(self, mode: int = 1):
        """TIER 2 of the certified screen, the threshold rescan (include/mfar_hip.h): 0 never, 1 auto (armed by failed certificates),
        2 always; + 4 (5, 6): every failed list takes the rescan instead of trusting the launch's own chunk lists (diagnostic).  Outputs
        are bit-identical in every mode."""
        _native.check(_native.lib().mfar_set_tier2(self._h, int(mode)))

    def set_deep_scan(self, mode: int = 1):
        """DEEP SCAN of fields whose first certificates keep failing (include/mfar_hip.h): 0 never, 1 auto, 2 every field always."""
        _native.check(_native.lib().mfar_set_deep_scan(self._h, int(mode)))

    def deep_scan_info(self) -> dict:
        m, n = ctypes.c_uint32(), ctypes.c_int64()
        _native.check(_native.lib().mfar_deep_scan_info(self._h, ctypes.byref(m), ctypes.byref(n)))
        return dict(fields=[f for f in range(self.n_fields) if (m.value >> f) & 1], n_switched=n.value)

    def tier2_stats(self) -> dict:
        a, n, b = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64()
        c = (ctypes.c_int64 * 4)()
        _native.check(_native.lib().mfar_tier2_stats(self._h, ctypes.byref(a), ctypes.byref(n), ctypes.byref(b), c))
        s, r = ctypes.c_int64(), ctypes.c_int64()
        _native.check(_native.lib().mfar_tier2_rescan_stats(self._h, ctypes.byref(s), ctypes.byref(r)))
        return dict(armed=bool(a.value), lists=n.value, passed_on_to_exact=b.value,
                    passed_on_because=dict(chunk_list_full=c[0], too_many_rows_above_threshold=c[1], too_many_candidates_in_band=c[2], ties_at_cut=c[3]),
                    candidates_from_the_launch_scan=s.value, lists_rescanned=r.value)

    def set_stage2_dump(self, mode: int = 1):
        """Score dump of the wide screened pass as the approximate level of stage 2 (include/mfar_hip.h): 0 never, 1 when it moves
        less than a third of the row gathers' bytes, 2 whenever possible.  Outputs are bit-identical in every mode."""
        _native.check(_native.lib().mfar_set_stage2_dump(self._h, int(mode)))

    def stage2_dump_info(self, k1: int = 100) -> dict:
        w, b, n = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64()
        _native.check(_native.lib().mfar_stage2_dump_info(self._h, int(k1), ctypes.byref(w), ctypes.byref(b), ctypes.byref(n)))
        return dict(wanted=bool(w.value), bytes_per_launch=b.value, n_launches=n.value)

    def stage2_stats(self) -> dict:
        ok, nbytes, nc, ns = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _native.check(_native.lib().mfar_stage2_stats(self._h, ctypes.byref(ok), ctypes.byref(nbytes), ctypes.byref(nc), ctypes.byref(ns)))
        return dict(two_level=bool(ok.value), gather_slab_bytes=nbytes.value, n_candidates=nc.value, n_survivors=ns.value)

    def screen_field_info(self, field: int):
        """(distinct vectors of the field = rows the screened pass scans, size of its largest group of identical rows);
        (-1, -1) while no screen is current (include/mfar_hip.h)."""
        nu, big = ctypes.c_int64(), ctypes.c_int64()
        _native.check(_native.lib().mfar_screen_field_info(self._h, int(field), ctypes.byref(nu), ctypes.byref(big)))
        return nu.value, big.value

    @property
    def screen_setting(self):
        """(mode, eps_mult) currently in force."""
        mode, mult = ctypes.c_int(), ctypes.c_float()
        _native.check(_native.lib().mfar_get_screen(self._h, ctypes.byref(mode), ctypes.byref(mult)))
        return mode.value, mult.value

    def screen_stats(self) -> dict:
        built, nbytes, chk, bad = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _native.check(_native.lib().mfar_screen_stats(self._h, ctypes.byref(built), ctypes.byref(nbytes), ctypes.byref(chk),
                                                      ctypes.byref(bad)))
        st = dict(built=bool(built.value), screen_bytes=nbytes.value, n_checked=chk.value, n_failed=bad.value)
        if st["built"]:
            st["unique_rows"] = [self.screen_field_info(f)[0] for f in range(self.n_fields)]
            # rows one screened scan reads: the unique rows of the fp16 screen slab (fp32 index) / every document of the bf16
            # slab itself (bf16 index: no second copy; duplicates are scanned and masked)
            st["scan_rows"] = sum(st["unique_rows"]) if self.dtype == "f32" else self.n_rows * self.n_fields
        return st


def merge_workspace_bytes(Q: int, n_fields: int, k1: int = 100) -> int:
    return int(_native.lib().mfar_merge_workspace_bytes(int(Q), int(n_fields), int(k1)))


def merge_payloads(payloads, n_shards: int, q, W, mask=None, n_fields: int = None, k1: int = 100, k2: int = 100,
                   sentinel: bool = True, query_cond: bool = True, device: int = 0, workspace=None, out=None):
    """Merge the all-gathered per-shard payloads (concatenated, shard-major) into the final top-k2.
    Replaces the per-rank .qres files + rank-0 merge of the reference (contrastive.py:519-536, 566-581)."""
    qa, Wa = _Arg(q, np.float32, device), _Arg(W, np.float32, device)
    ma = _Arg(mask, np.float32, device, allow_none=True)
    pa = _Arg(payloads, np.uint8, device)
    on_dev = _same_side([qa, Wa, ma, pa])
    Q, E = qa.keep.shape
    if n_fields is None:
        n_fields = Wa.keep.shape[-1] if query_cond else int(np.prod(Wa.keep.shape))
    out = dict(out) if out else {}
    ids = out.get("ids") if out.get("ids") is not None else _empty_like_side(on_dev, device, (Q, k2), np.int64)
    sc = out.get("scores") if out.get("scores") is not None else _empty_like_side(on_dev, device, (Q, k2), np.float32)
    nv = out.get("n_valid") if out.get("n_valid") is not None else _empty_like_side(on_dev, device, (Q,), np.int32)
    ia, sa, na = _Arg(ids, np.int64, device), _Arg(sc, np.float32, device), _Arg(nv, np.int32, device)
    wa = _Arg(workspace, np.uint8, device, allow_none=True)
    wbytes = int(workspace.numel()) if workspace is not None else 0
    _native.check(_native.lib().mfar_merge_payloads(
        int(device), pa.ptr, int(n_shards), qa.ptr, Q, E, Wa.ptr, int(bool(query_cond)), ma.ptr, int(n_fields), int(k1), int(k2),
        int(bool(sentinel)), ia.ptr, sa.ptr, na.ptr, wa.ptr, wbytes, int(on_dev), _current_stream(device, on_dev)))
    return dict(ids=ids, scores=sc, n_valid=nv)


def merge_topk(gathered_topk, n_shards: int, Q: int, k2: int = 100, device: int = 0, out=None, any_fail=None):
    """Final merge of the all-gathered local top-k payloads (lists-first exchange); `any_fail` (CUDA int32 tensor or None)
    receives the OR of the ranks' certificate flags that travelled in the payloads."""
    ga = _Arg(gathered_topk, np.uint8, device)
    if not ga.on_device:
        raise ValueError("merge_topk needs CUDA tensors")
    out = dict(out) if out else {}
    ids = out.get("ids") if out.get("ids") is not None else _empty_like_side(True, device, (Q, k2), np.int64)
    sc = out.get("scores") if out.get("scores") is not None else _empty_like_side(True, device, (Q, k2), np.float32)
    nv = out.get("n_valid") if out.get("n_valid") is not None else _empty_like_side(True, device, (Q,), np.int32)
    ia, sa, na = _Arg(ids, np.int64, device), _Arg(sc, np.float32, device), _Arg(nv, np.int32, device)
    _native.check(_native.lib().mfar_merge_topk(int(device), ga.ptr, int(n_shards), int(Q), int(k2), ia.ptr, sa.ptr, na.ptr,
                                                None if any_fail is None else _dev_ptr(any_fail), _current_stream(device, True)))
    return dict(ids=ids, scores=sc, n_valid=nv)


def mix_topk(cand_scores, cand_ids, q, W, mask=None, n_cand=None, k: int = 100, query_cond: bool = True, device: int = 0):
    """mask * scores (contrastive.py:686) -> LinearWeights.forward (weighting.py:17-29) -> topk (contrastive.py:696)."""
    xa, ca = _Arg(cand_scores, np.float32, device), _Arg(cand_ids, np.int64, device)
    qa = _Arg(q, np.float32, device, allow_none=not query_cond)
    Wa = _Arg(W, np.float32, device)
    ma = _Arg(mask, np.float32, device, allow_none=True)
    na = _Arg(n_cand, np.int32, device, allow_none=True)
    on_dev = _same_side([xa, ca, qa, Wa, ma, na])
    Q, C, F = xa.keep.shape
    E = qa.keep.shape[1] if qa.keep is not None else 0
    ids = _empty_like_side(on_dev, device, (Q, k), np.int64)
    sc = _empty_like_side(on_dev, device, (Q, k), np.float32)
    nv = _empty_like_side(on_dev, device, (Q,), np.int32)
    ia, sa, nva = _Arg(ids, np.int64, device), _Arg(sc, np.float32, device), _Arg(nv, np.int32, device)
    _native.check(_native.lib().mfar_mix_topk(int(device), xa.ptr, ca.ptr, na.ptr, qa.ptr, Wa.ptr, int(bool(query_cond)), ma.ptr,
                                              Q, C, F, E, int(k), ia.ptr, sa.ptr, nva.ptr, int(on_dev),
                                              _current_stream(device, on_dev)))
    return dict(ids=ids, scores=sc, n_valid=nv)


class BM25sSparseIndex(Index[str, str]):
    """A sparse field index (reference index.py:39-157): same constructor, methods and return shapes.  The reference
    wraps the third-party `bm25s` package; here the BM25 arithmetic is the restatement in mfar/data/bm25.py (host CPU,
    parity unpinned -- see that module).  Ties follow the canonical order (score desc, doc number asc)."""

    def __init__(self, keys: List[str], index, stemmer=None, index_limit: int = 5000, safe_docs=None):
        self.keys = keys
        self.key_to_id = {key: i for i, key in enumerate(keys)}
        self.index = index
        self.stemmer = stemmer
        self.index_limit = index_limit
        self.safe_docs = safe_docs if safe_docs is not None else {}
        self.name = None
        # the reference memoises get_scores per query (lru_cache 2**15 entries, index.py:86); a dense [D] fp32 array per entry is
        # 4 MB at 1 M documents, so the memo here is an LRU bounded by BYTES (MFAR_BM25_MEMO_MB, default 256 MB per index)
        from collections import OrderedDict
        self._score_memo: "OrderedDict[str, np.ndarray]" = OrderedDict()
        self._memo_bytes = 0
        self._memo_cap = int(float(__import__("os").environ.get("MFAR_BM25_MEMO_MB", "256")) * (1 << 20))

    def set_safe_docs(self, safe_docs):
        self.safe_docs = safe_docs

    @staticmethod
    def tokenize_single(query: str, stopwords: str, stemmer=None, return_ids: bool = False):
        if return_ids:
            raise NotImplementedError("token-id output (bm25s `Tokenized`) is not part of the restated surface")
        from mfar.data import bm25
        return bm25.tokenize(query, stopwords=stopwords, stemmer=stemmer)[0]

    def tokenize(self, queries, stopwords, stemmer, return_ids=False):
        if isinstance(queries, str):
            return BM25sSparseIndex.tokenize_single(queries, stopwords, stemmer, return_ids)
        return [BM25sSparseIndex.tokenize_single(q, stopwords, stemmer, return_ids) for q in queries]

    def get_scores(self, query: str) -> np.ndarray:        # [D]
        s = self._score_memo.get(query)
        if s is not None:
            self._score_memo.move_to_end(query)
            return s
        s = self.index.get_scores(self.tokenize(query, stopwords="en", stemmer=self.stemmer))
        if s.nbytes <= self._memo_cap:
            self._score_memo[query] = s
            self._memo_bytes += s.nbytes
            while self._memo_bytes > self._memo_cap or len(self._score_memo) > 2 ** 15:      # evict the least recently used
                _, old = self._score_memo.popitem(last=False)
                self._memo_bytes -= old.nbytes
        return s

    def get_scores_sparse(self, query: str) -> Dict[int, float]:
        dense = self.index.get_scores(self.tokenize(query, stopwords="en", stemmer=self.stemmer))
        return {int(i): dense[i] for i in np.nonzero(dense)[0] if int(i) in self.safe_docs}

    def retrieve(self, query: str, top_k: int) -> Sequence[Tuple[str, float]]:
        return self.retrieve_batch([query], top_k)[0]

    def retrieve_batch(self, queries: Sequence[str], top_k: int) -> Sequence[Sequence[Tuple[str, float]]]:
        toks = self.tokenize(list(queries), stopwords="en", stemmer=self.stemmer)
        ids, scores = self.index.retrieve(toks, k=top_k)
        return [[(self.keys[ids[i, j]], scores[i, j]) for j in range(ids.shape[1])] for i in range(ids.shape[0])]

    def score(self, query: str, keys: Sequence[str]) -> np.ndarray:       # [Cand]; KeyError on unknown keys like the reference
        doc_ids = np.array([self.key_to_id[key] for key in keys], dtype=np.int64)
        return self.get_scores(query)[doc_ids]

    def score_batch(self, queries: Sequence[str], keys: Sequence[str]):   # -> tensor [Query, Cand]; unknown keys score 0
        import torch
        doc_ids = np.array([self.key_to_id.get(key, -1) for key in keys], dtype=np.int64)
        all_scores = np.stack([self.get_scores(q) for q in queries], axis=0) if len(queries) else np.zeros((0, len(self.keys)), np.float32)
        out = all_scores[:, np.maximum(doc_ids, 0)] if doc_ids.size else np.zeros((len(queries), 0), np.float32)
        out[:, doc_ids < 0] = 0
        return torch.tensor(out)

    def score_batch_with_cache(self, query_ids, keys: Sequence[str], sparse_scores: Dict):
        import torch
        doc_ids = [self.key_to_id[key] for key in keys]
        rows = [sparse_scores.get(qid, {}) for qid in query_ids]
        return torch.tensor([[row.get(d, 0) for d in doc_ids] for row in rows])

    @classmethod
    def create(cls, corpus, stemmer=None, dataset_name: Optional[str] = ""):
        from mfar.data import bm25
        keys = list(corpus.keys())
        texts = [d.text for d in corpus.docs]
        index = bm25.BM25(method="lucene", k1=1.2, b=0.75).index(bm25.tokenize(texts, stopwords="en", stemmer=stemmer))
        return cls(keys, index, stemmer, 5000 if dataset_name == "amazon" else 12000)

    def save(self, path: str):
        import json
        self.index.save(f"{path}/index")
        with open(f"{path}/keys.json", "w") as f:
            json.dump(self.keys, f)

    @classmethod
    def load(cls, path: str, stemmer=None):
        import json
        from mfar.data import bm25
        with open(f"{path}/keys.json", "r") as f:
            keys = json.load(f)
        return cls(keys, bm25.BM25.load(f"{path}/index", mmap=True), stemmer)


class DenseFlatIndex(Index[str, str]):
    """Drop-in for the reference's DenseFlatIndex (index.py:160-232): same constructor arguments, `.vectors` may be
    re-assigned from outside after the corpus encode (contrastive.py:494), `retrieve_batch` accepts an ndarray of
    query embeddings or query texts, `score_batch` returns a tensor [len(queries), len(keys)] and raises KeyError
    for unknown keys.  The arithmetic runs on the GPU: this object is a one-field view of a `MultiFieldIndex`
    (pass `slab=` and `field_index=` to share one slab between the fields, as read_and_create_indices does here).

    Two documented differences from the reference: ties are broken (score desc, doc index asc) instead of
    unspecified, and `vector_batch_size` is accepted but irrelevant (there is no chunked CPU matmul).
    """

    def __init__(self, model, vectors, numeric_ids_to_keys: Sequence[str], keys_to_numeric_ids: Dict[str, int],
                 device=None, vector_batch_size: int = 1048576, slab: Optional[MultiFieldIndex] = None, field_index: int = 0):
        self.model = model
        self.numeric_ids_to_key = numeric_ids_to_keys
        self.key_to_numeric_ids = keys_to_numeric_ids
        self.vector_batch_size = vector_batch_size
        self.field_index = field_index
        dev_index = 0
        if device is not None and getattr(device, "type", "cpu") == "cuda" and device.index is not None:
            dev_index = device.index
        self.device = device
        self._slab = slab
        self._own_slab = slab is None
        self._dev_index = slab.device if slab is not None else dev_index
        self._vectors = None
        if vectors is not None:
            self.vectors = vectors

    @property
    def slab(self) -> MultiFieldIndex:
        if self._slab is None:
            raise RuntimeError("DenseFlatIndex has no vectors yet")
        return self._slab

    @property
    def vectors(self):
        return self._vectors

    @vectors.setter
    def vectors(self, v):
        """(Re-)load this field's [D, E] matrix into the slab -- the reference re-assigns `.vectors` with the reopened
        memmap after every corpus encode (contrastive.py:492-494)."""
        self._vectors = v
        shape = tuple(v.shape)
        if len(shape) != 2:
            raise ValueError("vectors must be [D, E]")
        if self._slab is None:
            self._slab = MultiFieldIndex(shape[0], 1, shape[1], device=self._dev_index)
            self.field_index = 0
        if shape[0] != self._slab.n_rows or shape[1] != self._slab.dim:
            raise ValueError(f"vectors {shape} do not match the slab [{self._slab.n_rows}, {self._slab.dim}]")
        step = max(1, (256 << 20) // (shape[1] * 4))
        for r0 in range(0, shape[0], step):
            self._slab.write_rows(self.field_index, r0, np.asarray(v[r0:r0 + step], dtype=np.float32)
                                  if not _is_torch(v) else v[r0:r0 + step].contiguous())

    def _encode(self, queries):
        if isinstance(queries, np.ndarray):
            return np.ascontiguousarray(queries, dtype=np.float32)          # index.py:184-185
        enc = self.model.encode(list(queries), convert_to_tensor=True)       # index.py:187
        return enc.detach().to("cpu").float().numpy()

    def retrieve(self, query, top_k: int):
        return self.retrieve_batch([query], top_k)[0]

    def retrieve_batch(self, queries: Union[np.ndarray, Sequence[str]], top_k: int):
        qe = self._encode(queries)
        ids, sc = self.slab.retrieve_field(self.field_index, qe, top_k, sentinel=True)      # scans this field only
        ids_l, sc_l = ids.tolist(), sc.tolist()
        # ids are global row numbers == positions in numeric_ids_to_key (the corpus line order)
        return [list(zip([self.numeric_ids_to_key[j] for j in ids_l[i]], sc_l[i])) for i in range(len(queries))]

    def score(self, query, keys: Sequence[str]):
        return self.score_batch([query], keys)[0]

    def score_batch(self, queries: Sequence[str], keys: Sequence[str]):
        import torch
        qe = self._encode(queries)
        rows = np.asarray([self.key_to_numeric_ids[k] for k in keys], dtype=np.int64)   # KeyError like index.py:229
        # a row-sharded slab holds only [row_offset, row_offset + n_rows): a valid key owned by another rank cannot be
        # scored here (the reference keeps the full memmap on every rank).  Fail loudly instead of returning NaN.
        lo, hi = self.slab.row_offset, self.slab.row_offset + self.slab.n_rows
        off = rows[(rows < lo) | (rows >= hi)]
        if off.size:
            raise KeyError(f"{keys[int(np.nonzero((rows < lo) | (rows >= hi))[0][0])]!r} (row {int(off[0])}) is outside this "
                           f"rank's row shard [{lo}, {hi})")
        cand = np.broadcast_to(rows, (qe.shape[0], rows.size)).copy()                 # global row numbers
        x = self.slab.score_candidates(qe, cand)
        return torch.from_numpy(np.ascontiguousarray(x[:, :, self.field_index]))


def candidate_encoding_stream(encoder, corpus: Iterable[Tuple[str, str]], batch_size: int = 64, multiprocess: bool = True,
                              show_progress: bool = True, as_tensor: bool = False) -> Iterable[Tuple[str, np.ndarray]]:
    """(id, text) pairs -> (id, embedding[E]) pairs in chunks of `batch_size`, in input order (reference index.py:234-258).
    `multiprocess` is accepted for signature compatibility; one process drives one GPU here (the reference's eval
    path always passes multiprocess=False, contrastive.py:487).  `as_tensor=True` (extension, used by `on_eval_start`)
    yields the rows as tensors on the encoder's device instead of numpy arrays, so the corpus encode writes into the
    HBM slab without a host round trip."""
    it = corpus
    if show_progress:
        try:
            from tqdm import tqdm
            it = tqdm(corpus)
        except ImportError:
            pass
    batch = []
    for item in it:
        batch.append(item)
        if len(batch) == batch_size:
            yield from _encode_batch(encoder, batch, batch_size, as_tensor)
            batch = []
    if batch:
        yield from _encode_batch(encoder, batch, batch_size, as_tensor)


def _encode_batch(encoder, batch, batch_size, as_tensor=False):
    ids = [i for i, _ in batch]
    texts = [t for _, t in batch]
    if as_tensor:
        embs = encoder.encode(texts, batch_size=batch_size, convert_to_tensor=True)
    else:
        embs = encoder.encode(texts, batch_size=batch_size, convert_to_numpy=True)
    return zip(ids, embs)
