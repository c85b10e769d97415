"""Vector-store helpers (reference mfar/data/util.py).

* `MemoryMapDict` -- the reference's key -> row view over a RAW (headerless despite the .npy suffix) float32
  `np.memmap` (data/util.py:28-59).  Kept with identical behaviour as the host-side import/export format.
* `HbmFieldVectors` -- the same mapping interface over one field of the on-HBM slab: what `read_and_create_indices`
  hands out as `vectors_dict[field]` here, so `vectors_dict[field][key] = vec` (contrastive.py:490) lands in HBM.
"""
import os
from enum import Enum
from typing import Any, Iterable, MutableMapping, Tuple

import numpy as np


class SpecialToken(Enum):
    query_start = "<QRY>"
    doc_start = "<DOC>"
    id_start = "<ID>"

    def __str__(self):
        return self.value


class MemoryMapDict(MutableMapping):
    def __init__(self, path: str, keys: Iterable[str], shape: Tuple[int, ...], mode: str = "r+", dtype=np.float32):
        self._keys = {k: i for i, k in enumerate(keys)}
        self._path, self._shape, self._dtype = path, shape, dtype
        self.file = np.memmap(path, dtype=dtype, mode=mode, shape=shape)

    def __getitem__(self, key):
        return self.file[self._keys[key], :]

    def __setitem__(self, key, value):
        self.file[self._keys[key], :] = value

    def __delitem__(self, key):
        raise NotImplementedError

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return self._shape[0]

    def __contains__(self, key):
        return key in self._keys

    def close(self):
        self.file.flush()

    def reopen(self):
        self.file = np.memmap(self._path, dtype=self._dtype, mode="r+", shape=self._shape)


class HbmFieldVectors(MutableMapping):
    """key -> vector over field `field_index` of a `MultiFieldIndex` (rows of THIS rank's shard)."""

    def __init__(self, slab, field_index: int, keys: Iterable[str], path: str = None):
        self.slab, self.field_index = slab, field_index
        self._keys = {k: i for i, k in enumerate(keys)}      # global row numbers
        self._path = path

    def _local(self, key):
        r = self._keys[key] - self.slab.row_offset
        if not 0 <= r < self.slab.n_rows:
            raise KeyError(f"{key} is not in this rank's row shard")
        return r

    def __getitem__(self, key):
        return self.slab.read_rows(self.field_index, self._local(key), 1)[0]

    def __setitem__(self, key, value):
        self.slab.write_rows(self.field_index, self._local(key), np.asarray(value, dtype=np.float32).reshape(1, -1))

    def write_block(self, first_key, rows):
        """Consecutive rows starting at `first_key` (what the batched corpus encode uses; rows: numpy or CUDA tensor)."""
        self.slab.write_rows(self.field_index, self._local(first_key), rows)

    def __delitem__(self, key):
        raise NotImplementedError

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __contains__(self, key):
        return key in self._keys

    @property
    def file(self):
        """This shard's [n_rows, E] float32 matrix (a host copy), the counterpart of MemoryMapDict.file."""
        return self.slab.read_rows(self.field_index)

    def close(self):
        pass

    def reopen(self):
        pass

    # raw float32 [D, E] files == the reference's {temp_dir}/{field}.npy layout (data/util.py:35)
    def export_memmap(self, path: str, n_total_rows: int = None):
        """Write this shard's rows into the raw [n_total, E] float32 file.  With several ranks, rank 0 creates and sizes
        the file, everybody waits at a barrier, then every rank opens it "r+" and writes only its own row range (a second
        rank opening with "w+" would zero rows already written)."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        n_total = n_total_rows or (self.slab.row_offset + self.slab.n_rows)
        nbytes = n_total * self.slab.dim * 4
        if not multi or dist.get_rank() == 0:
            with open(path, "ab"):
                pass
            os.truncate(path, nbytes)          # sparse resize; existing rows of a right-sized file are kept
        if multi:
            dist.barrier()
        mm = np.memmap(path, dtype=np.float32, mode="r+", shape=(n_total, self.slab.dim))
        mm[self.slab.row_offset:self.slab.row_offset + self.slab.n_rows] = self.file
        mm.flush()
        del mm
        if multi:
            dist.barrier()

    def import_memmap(self, path: str, n_total_rows: int):
        mm = np.memmap(path, dtype=np.float32, mode="r", shape=(n_total_rows, self.slab.dim))
        lo = self.slab.row_offset
        self.slab.write_rows(self.field_index, 0, np.ascontiguousarray(mm[lo:lo + self.slab.n_rows]))


def remove_irregularities(obj: Any) -> Any:
    """Make a STaRK record JSON/TSV safe (data/util.py:62-75): control separators inside strings become blanks."""
    if isinstance(obj, str):
        for code in (10, 9, 13, 31):
            obj = obj.replace(chr(code), " ")
        return obj.strip()
    if isinstance(obj, list):
        return [remove_irregularities(x) for x in obj]
    if isinstance(obj, dict):
        return {k: remove_irregularities(v) for k, v in obj.items()}
    if isinstance(obj, np.bool_):
        return obj.item()
    if obj is None or isinstance(obj, (int, float, bool)):
        return obj
    raise ValueError(f"Unexpected type {type(obj)}")
