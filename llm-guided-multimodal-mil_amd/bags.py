"""Ragged bag batches: bags concatenated along the patch axis plus the tile map the
attention-pool kernels iterate over (include/mil_hip.h, "tile map")."""
from __future__ import annotations

from dataclasses import dataclass
from typing import ClassVar, List, Optional, Sequence

from collections import OrderedDict

import numpy as np
import torch

from . import lifetime

POOL_TILE = 32


def build_tile_map(lengths: Sequence[int], tile: int = POOL_TILE):
    """(tile_map int32 [T, 4] = {bag, row0, nrows, 0}, bag_tile_off int32 [B+1], bag_off int32 [B+1])."""
    lengths = np.asarray(lengths, dtype=np.int64)
    B = len(lengths)
    bag_off = np.zeros(B + 1, dtype=np.int64)
    bag_off[1:] = np.cumsum(lengths)
    ntiles = (lengths + tile - 1) // tile
    bag_tile_off = np.zeros(B + 1, dtype=np.int64)
    bag_tile_off[1:] = np.cumsum(ntiles)
    T = int(bag_tile_off[-1])
    tm = np.zeros((T, 4), dtype=np.int32)
    if T:
        bag = np.repeat(np.arange(B), ntiles)
        local = np.arange(T) - bag_tile_off[bag]
        row0 = bag_off[bag] + local * tile
        tm[:, 0] = bag
        tm[:, 1] = row0
        tm[:, 2] = np.minimum(tile, bag_off[bag + 1] - row0)
    return tm, bag_tile_off.astype(np.int32), bag_off.astype(np.int32)


def _np_offsets(lengths) -> np.ndarray:
    off = np.zeros(len(lengths) + 1, dtype=np.int32)
    off[1:] = np.cumsum(lengths)
    return off


@dataclass
class BagLayout:
    """Device-resident description of how R rows split into B bags."""
    lengths: List[int]
    R: int
    B: int
    T: int
    tile_map: torch.Tensor       # int32 [T, 4]
    bag_tile_off: torch.Tensor   # int32 [B+1]
    bag_off: torch.Tensor        # int32 [B+1]
    _row_bag: Optional[torch.Tensor] = None
    aligned32: bool = False      # single segment, every bag length a multiple of 32: tile t = rows 32 t .. 32 t + 31

    def row_bag(self) -> torch.Tensor:
        """int32 [R]: the bag of every row (built from the tile map on first use; mil_gate_bwd_input_pool reads it)."""
        if self._row_bag is None:
            tm = self.tile_map.cpu().numpy()
            rb = np.zeros(self.R, dtype=np.int32)
            for bag, row0, nrows, _ in tm:
                rb[row0:row0 + nrows] = bag
            self._row_bag = torch.from_numpy(rb).to(self.tile_map.device)
        return self._row_bag

    _cache: ClassVar["OrderedDict[tuple, BagLayout]"] = OrderedDict()
    CACHE_ENTRIES: ClassVar[int] = 256

    @classmethod
    def _get(cls, key):
        hit = cls._cache.get(key)
        if hit is not None:
            cls._cache.move_to_end(key)
            lifetime.note(hit)           # a graph being captured keeps the layout it points at (lifetime.py)
        return hit

    @classmethod
    def _put(cls, key, lay):
        # LRU, bounded: one ragged bag per step means a new key almost every step.  Eviction drops only the cache's
        # reference; captured graphs hold their own.
        cls._cache[key] = lay
        while len(cls._cache) > cls.CACHE_ENTRIES:
            cls._cache.popitem(last=False)
        return lifetime.note(lay)

    @classmethod
    def make(cls, lengths: Sequence[int], device) -> "BagLayout":
        key = (tuple(int(v) for v in lengths), str(device))
        hit = cls._get(key)
        if hit is not None:
            return hit
        tm, bto, bo = build_tile_map(lengths)
        lay = cls(lengths=list(key[0]), R=int(bo[-1]), B=len(key[0]), T=int(tm.shape[0]),
                  tile_map=torch.from_numpy(tm).to(device), bag_tile_off=torch.from_numpy(bto).to(device),
                  bag_off=torch.from_numpy(bo).to(device), aligned32=all(v % 32 == 0 for v in key[0]))
        return cls._put(key, lay)

    @classmethod
    def uniform(cls, B: int, N: int, device) -> "BagLayout":
        return cls.make([N] * B, device)

    @classmethod
    def multi_segment(cls, seg_lengths: Sequence[Sequence[int]], device) -> "BagLayout":
        """Rows laid out as [segment 0 of all bags | segment 1 of all bags | ...]; bag b owns its range of every segment.
        The 4-segment multi-modal bag of model/aggregator.py:173 (text-from-CT tokens, CT tokens, text-from-pathology
        tokens, patch tokens) without materialising the per-bag concatenation: the pool kernels only follow the tile map
        and attention pooling does not depend on the order of a bag's rows."""
        key = ("mseg", tuple(tuple(int(v) for v in seg) for seg in seg_lengths), str(device))
        hit = cls._get(key)
        if hit is not None:
            return hit
        segs = [np.asarray(k, dtype=np.int64) for k in key[1]]
        B = len(segs[0])
        maps, base = [], 0
        for seg in segs:
            tm, bto, bo = build_tile_map(seg)
            tm = tm.copy()
            tm[:, 1] += base
            maps.append((tm, bto))
            base += int(bo[-1])
        tiles = []
        for b in range(B):
            for tm, bto in maps:
                tiles.append(tm[bto[b]:bto[b + 1]])
        tm = np.concatenate(tiles, 0) if tiles else np.zeros((0, 4), np.int32)
        bto = sum(m[1].astype(np.int64) for m in maps).astype(np.int32)
        lay = cls(lengths=[int(sum(seg[b] for seg in segs)) for b in range(B)], R=base, B=B, T=int(tm.shape[0]),
                  tile_map=torch.from_numpy(np.ascontiguousarray(tm)).to(device),
                  bag_tile_off=torch.from_numpy(bto).to(device),
                  bag_off=torch.from_numpy(_np_offsets(segs[0])).to(device))
        return cls._put(key, lay)

    @classmethod
    def two_segment(cls, n_lengths: Sequence[int], t_lengths: Sequence[int], device) -> "BagLayout":
        """Rows laid out as [all patch rows of all bags | all token rows of all bags]; bag b owns its patch range
        and its token range.  The pool kernels only follow the tile map, so a bag need not be contiguous - this is
        how the fused model pools over (text tokens + patches) without materialising the per-bag concatenation
        of model/aggregator.py:192."""
        key = ("2seg", tuple(int(v) for v in n_lengths), tuple(int(v) for v in t_lengths), str(device))
        hit = cls._get(key)
        if hit is not None:
            return hit
        n = np.asarray(key[1], dtype=np.int64)
        t = np.asarray(key[2], dtype=np.int64)
        B = len(n)
        tm_n, bto_n, bo_n = build_tile_map(n)
        tm_t, bto_t, bo_t = build_tile_map(t)
        tm_t = tm_t.copy()
        tm_t[:, 1] += int(bo_n[-1])                      # token rows start after every patch row
        tiles = []
        for b in range(B):
            tiles.append(tm_n[bto_n[b]:bto_n[b + 1]])
            tiles.append(tm_t[bto_t[b]:bto_t[b + 1]])
        tm = np.concatenate(tiles, 0) if tiles else np.zeros((0, 4), np.int32)
        bto = (bto_n.astype(np.int64) + bto_t.astype(np.int64)).astype(np.int32)
        lay = cls(lengths=[int(a + c) for a, c in zip(n, t)], R=int(bo_n[-1] + bo_t[-1]), B=B, T=int(tm.shape[0]),
                  tile_map=torch.from_numpy(np.ascontiguousarray(tm)).to(device),
                  bag_tile_off=torch.from_numpy(bto).to(device), bag_off=torch.from_numpy(bo_n).to(device))
        return cls._put(key, lay)


def bucket_rows(n: int, floor: int = 256) -> int:
    """Capacity bucket for n rows: the smallest of {2^k, 1.5 * 2^k} (multiples of 256) that holds n - at most a third
    of a bucket is padding, and bags of 2 000 .. 15 592 patches (dataset.py:383-391) fall into seven buckets."""
    n = max(int(n), 1)
    b = floor
    while True:
        if b >= n:
            return b
        if b + b // 2 >= n and (b + b // 2) % 256 == 0:
            return b + b // 2
        b *= 2


def upload_lengths(dst: torch.Tensor, lengths) -> None:
    """dst (int32, on the device) <- lengths on the current stream.  Up to 8 values go as kernel arguments of one tiny launch
    (mil_set_i32); a copy from pageable host memory cost a blit kernel behind a ~9 us gap in front of every replayed step."""
    n = len(lengths)
    if n <= 8 and dst.is_cuda:
        import ctypes
        from . import _lib
        arr = (ctypes.c_int32 * n)(*lengths)
        rc = _lib.lib().mil_set_i32(ctypes.c_void_p(dst.data_ptr()), arr, n,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "mil_set_i32")
    else:
        dst.copy_(torch.tensor(lengths, dtype=torch.int32), non_blocking=True)


class DeviceBagLayout:
    """Bag layout whose lengths live ON THE DEVICE: the step kernels rebuild the tile map from `bag_len_dev` every pass
    (mil_build_tile_map) and mask the rows beyond the true total, so one set of launch parameters - one captured hipGraph -
    serves every batch that fits the capacity (R rows, B bags).  The authors' regime is one ragged bag per GPU whose
    length changes every step (run_train.sh:81; dataset.py:374-381 drops a random 10-20 % of the patches per epoch):
    host-built maps keyed by the exact lengths (BagLayout) never repeat there."""

    def __init__(self, capacity_rows: int, B: int, device):
        self.R, self.B = int(capacity_rows), int(B)
        self.T = self.R // POOL_TILE + self.B                  # every bag may end in a partial tile
        self.tile_map = torch.zeros((self.T, 4), device=device, dtype=torch.int32)
        self.bag_tile_off = torch.zeros(self.B + 1, device=device, dtype=torch.int32)
        self.bag_len_dev = torch.zeros(self.B, device=device, dtype=torch.int32)
        self.rows_dev = torch.zeros(1, device=device, dtype=torch.int32)
        self.lengths = None

    def set_lengths(self, lengths):
        """Upload this step's lengths (a tiny async copy; the tile map itself is rebuilt by the step on the device)."""
        lengths = [int(v) for v in lengths]
        if len(lengths) != self.B or sum(lengths) > self.R:
            raise ValueError(f"{len(lengths)} bags / {sum(lengths)} rows do not fit a layout of {self.B} bags / {self.R} rows")
        self.lengths = lengths
        upload_lengths(self.bag_len_dev, lengths)
        return self

    def note_lengths(self, lengths):
        """The lengths are ALREADY in bag_len_dev (cohort.DeviceCohort.feed wrote them in its gather launch): host
        bookkeeping only, no upload."""
        lengths = [int(v) for v in lengths]
        if len(lengths) != self.B or sum(lengths) > self.R:
            raise ValueError(f"{len(lengths)} bags / {sum(lengths)} rows do not fit a layout of {self.B} bags / {self.R} rows")
        self.lengths = lengths
        return self
