"""Per-bag row segments for the attention kernels (include/mil_hip.h, K2): which query rows attend to which
key rows, plus the tile/block maps the kernels iterate over.  Built on the host from the bag lengths, cached."""
from __future__ import annotations

from typing import List, Sequence

from collections import OrderedDict

import numpy as np
import torch

from . import lifetime

POOL_KEYS_PER_TILE = 64
ROWS_PER_BLOCK = 32
MAX_SMALL = 16


def _offsets(lengths: np.ndarray) -> np.ndarray:
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    off[1:] = np.cumsum(lengths)
    return off


def _tiles(lengths: np.ndarray, off: np.ndarray, tile: int):
    n = (lengths + tile - 1) // tile
    toff = _offsets(n)
    T = int(toff[-1])
    tm = np.zeros((T, 3), dtype=np.int32)
    if T:
        bag = np.repeat(np.arange(len(lengths)), n)
        local = np.arange(T) - toff[bag]
        r0 = off[bag] + local * tile
        tm[:, 0], tm[:, 1], tm[:, 2] = bag, r0, np.minimum(tile, off[bag + 1] - r0)
    return tm, toff.astype(np.int32)


class AttnSegs:
    """Queries of bag b (q_lengths[b] rows) attend to the keys of bag b (k_lengths[b] rows)."""
    _cache: "OrderedDict[tuple, AttnSegs]" = OrderedDict()
    CACHE_ENTRIES = 512

    def __init__(self, q_lengths: Sequence[int], k_lengths: Sequence[int], device):
        ql = np.asarray(q_lengths, dtype=np.int64)
        kl = np.asarray(k_lengths, dtype=np.int64)
        assert len(ql) == len(kl)
        self.B = len(ql)
        self.q_lengths, self.k_lengths = [int(v) for v in ql], [int(v) for v in kl]
        qo, ko = _offsets(ql), _offsets(kl)
        self.Tq, self.Tk = int(qo[-1]), int(ko[-1])
        self.Tq_max, self.Tk_max = int(ql.max(initial=0)), int(kl.max(initial=0))
        dev = device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(dev)  # noqa: E731
        self.q_off, self.k_off = t(qo), t(ko)
        self.q_bag = t(np.repeat(np.arange(self.B), ql))
        self.k_bag = t(np.repeat(np.arange(self.B), kl))
        tm, toff = _tiles(kl, ko, POOL_KEYS_PER_TILE)          # pool form: tiles over the keys
        self.ntiles, self.tile_map, self.bag_tile_off = int(tm.shape[0]), t(tm), t(toff)
        bm, boff = _tiles(ql, qo, ROWS_PER_BLOCK)              # rows-form backward: blocks over the queries
        self.nblk, self.blk_map, self.bag_blk_off = int(bm.shape[0]), t(bm), t(boff)

    @classmethod
    def make(cls, q_lengths: Sequence[int], k_lengths: Sequence[int], device) -> "AttnSegs":
        key = (tuple(int(v) for v in q_lengths), tuple(int(v) for v in k_lengths), str(device))
        hit = cls._cache.get(key)
        if hit is None:
            hit = cls._cache[key] = cls(q_lengths, k_lengths, device)
            while len(cls._cache) > cls.CACHE_ENTRIES:          # LRU; captured graphs keep their own references
                cls._cache.popitem(last=False)
        else:
            cls._cache.move_to_end(key)
        return lifetime.note(hit)
