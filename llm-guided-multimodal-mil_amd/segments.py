"""Per-bag row segments for the attention kernels (include/mil_hip.h, K2): which query rows attend to which
key rows, plus the tile/block maps the kernels iterate over.  Built on the host from the bag lengths, cached."""
from __future__ import annotations

from typing import List, Sequence

import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import lifetime
from .bags import upload_lengths

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


class _SegView:
    """One AttnSegs-shaped face of a FusionBucket (the attribute names the attention wrappers in ops.py read)."""
    device_lengths = True

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def blk_map(self):
        raise NotImplementedError("rows-form attention backward is not built for device-side bag lengths (one text token per "
                                  "bag takes the absorbed one-token path)")


class FusionBucket:
    """Device-side segments of the fusion step for a capacity bucket: `cap` patch rows shared by B bags, P text tokens per
    bag (1: one clinical note - the absorbed one-token kernels; 2 .. 12: the prompts of `CI_prompt_version='devided'` or of
    the learnable-prompt branch - the multi-token grouped products).  The reference trains one ragged bag per GPU with a fresh patch drop every epoch (dataset.py:366-393,
    run_train.sh:81), so host-built segment maps keyed by the exact lengths (AttnSegs, BagLayout) never repeat and a captured
    step never replays.  Here the lengths live in `len_dev`; `refresh()` - the first launch of the step, inside the captured
    graph - rebuilds every map from them (mil_build_fusion_segs), launch grids and buffers depend on (cap, B) only, padding
    rows carry zero softmax weight and receive exactly zero gradient.  Faces:
        s_tt, s_ti, s_it   AttnSegs-shaped (token-token, token-image, image-token)
        layout             BagLayout-shaped two-segment multi-modal bag: rows [cap patch rows | B token rows]."""

    def __init__(self, capacity_rows: int, B: int, device, P: int = 1, tail=None):
        """tail: per-bag row counts of the static segments behind the patch rows, in row order.  Default [P] - the bag of
        aggregator.py:192, [patch rows | P text tokens].  The CT + pathology bag of aggregator.py:173 is [P, D, P]: patch rows,
        then P text-from-CT tokens, D CT tokens, P text-from-pathology tokens per bag."""
        from . import _lib
        from .bags import POOL_TILE
        cap, B, P = int(capacity_rows), int(B), int(P)
        if cap % 256 or cap <= 0 or not (0 < B <= 1024) or not (1 <= P <= 12):
            raise ValueError("FusionBucket: capacity must be a positive multiple of 256 rows, 1 <= B <= 1024, 1 <= P <= 12")
        self.P = P
        self.tail = [int(v) for v in (tail if tail is not None else [P])]
        if not (1 <= len(self.tail) <= 4) or min(self.tail) <= 0:
            raise ValueError("FusionBucket: 1 .. 4 tail segments of >= 1 row per bag")
        tail_rows = sum(self.tail)
        tail_tiles = sum((v + POOL_TILE - 1) // POOL_TILE for v in self.tail)
        self.cap, self.B, self.device = cap, B, device
        i32 = lambda *shape: torch.zeros(shape, device=device, dtype=torch.int32)      # noqa: E731
        self.len_dev, self.rows_dev = i32(B), i32(1)
        self.k_off, self.k_bag = i32(B + 1), i32(cap)
        self.T64, self.T32 = cap // POOL_KEYS_PER_TILE + B + 2, cap // POOL_TILE + B * (1 + tail_tiles)
        self.tile64, self.bag_tile64_off = i32(self.T64, 3), i32(B + 1)
        self.tile32, self.bag_tile32_off = i32(self.T32, 4), i32(B + 1)
        self.row_bag_dev = i32(cap + B * tail_rows)
        self.tail_rows = B * tail_rows
        self._tail_host = (ctypes.c_int32 * len(self.tail))(*self.tail)
        # the ABMIL pool's score gradient: real rows are written by the fused tail, padding rows zeroed by refresh()
        self.ds = torch.zeros(cap + B * tail_rows, device=device, dtype=torch.float32)
        self.lengths = self._uploaded = None
        tok = AttnSegs.make([P] * B, [P] * B, device)
        self.s_tt = tok
        ones = [P] * B
        # token -> image: queries = the B tokens (static), keys = the patch rows (device lengths).  Tk_max is the capacity.
        self.s_ti = _SegView(B=B, q_lengths=ones, k_lengths=None, Tq=B * P, Tk=cap, Tq_max=P, Tk_max=cap, q_off=tok.q_off,
                             q_bag=tok.q_bag, k_off=self.k_off, k_bag=self.k_bag, ntiles=self.T64, tile_map=self.tile64,
                             bag_tile_off=self.bag_tile64_off, pad_tiles=True)
        # image -> token: queries = the patch rows, keys = the tokens
        self.s_it = _SegView(B=B, q_lengths=None, k_lengths=ones, Tq=cap, Tk=B * P, Tq_max=cap, Tk_max=P, q_off=self.k_off,
                             q_bag=self.k_bag, k_off=tok.k_off, k_bag=tok.k_bag)
        self.layout = _SegView(B=B, R=cap + B * tail_rows, T=self.T32, tile_map=self.tile32, bag_tile_off=self.bag_tile32_off,
                               bag_off=self.k_off, lengths=None, aligned32=False, row_bag=lambda: self.row_bag_dev,
                               ds_buffer=self.ds)
        self._lib = _lib
        self._min_rows = _lib.lib().mil_layernorm_bagrow_rows_per_block(cap) if B > 1 else 1

    def fits(self, lengths) -> bool:
        """Side-effect-free: can this bucket's kernels take these bags (count, capacity, shortest bag)?"""
        lengths = [int(v) for v in lengths]
        return len(lengths) == self.B and sum(lengths) <= self.cap and min(lengths) >= self._min_rows

    def set_lengths(self, lengths, on_device: bool = False):
        """Upload this step's true patch counts (one tiny launch; skipped when the same lengths are already there or when
        `on_device` says the feed launch wrote them - cohort.DeviceCohort.feed); the maps follow on the device at refresh()."""
        lengths = [int(v) for v in lengths]
        if not self.fits(lengths):
            raise ValueError(f"FusionBucket: lengths {lengths} do not fit {self.B} bags / {self.cap} rows "
                             f"(every bag needs >= {self._min_rows} rows)")
        if not on_device and lengths != self._uploaded:
            upload_lengths(self.len_dev, lengths)
        self.lengths = self._uploaded = lengths
        return self

    def refresh(self):
        """Rebuild every map from len_dev on the current stream (capture-safe: one launch, no host sync)."""
        p = lambda t: t.data_ptr()      # noqa: E731
        rc = self._lib.lib().mil_build_fusion_segs_tail(p(self.len_dev), self.B, len(self.tail), self._tail_host, self.cap,
                                                        p(self.k_off), p(self.k_bag), p(self.tile64), p(self.bag_tile64_off),
                                                        self.T64, p(self.tile32), p(self.bag_tile32_off), self.T32,
                                                        p(self.row_bag_dev), p(self.rows_dev), p(self.ds),
                                                        torch.cuda.current_stream().cuda_stream)
        self._lib.check(rc, "mil_build_fusion_segs_tail")
        return lifetime.note(self)
