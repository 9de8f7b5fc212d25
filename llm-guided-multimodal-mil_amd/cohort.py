"""The cohort resident in HBM: the input side of the authors' training regime, MI355X-first.

Reference pipeline (dataset.py:366-393 + train_ddp.py:193,274-293): per step, DataLoader workers np.load one bag
[n, 768] fp32 (2 000 .. 15 592 patches, 6 - 48 MB), keep a sorted random 90 % / 80 % of its rows, pin it, and the loop copies it
to the GPU.  Here every bag is loaded ONCE into one flat device buffer (288 GB of HBM: 1 000 bags x <= 48 MB fit with room to
spare), the per-epoch drop is drawn on the device for the whole cohort in one launch (`mil_patch_drop_select`), and a step's
input is ONE gather launch (`mil_cohort_feed`) that writes the kept rows of the step's bags straight into the capacity bucket's
static input buffer together with the bag lengths, labels and notes (token ids or cached text embeddings) - no host copy,
no zero-pad to [B, maxN, 768], no D2D hop.  The lengths of a step are known on the host without a sync: k = int(n * keep),
the reference's own expression.

`HostFeed` is the fallback for a cohort that does not fit: a background thread fills two pinned staging buffers, the copies
run on their own stream and the feed kernel picks the rows up from a device-side double buffer."""
from __future__ import annotations

import ctypes
import threading
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib


def keep_count(n: int, keep: float) -> int:
    """dataset.py:376,379: int(feat.shape[0] * 0.9) for biopsies, * 0.8 for resections."""
    return int(n * keep)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class DeviceCohort:
    """bags: sequence of [n_j, F] fp32 arrays/tensors (or a callable j -> array with `lengths` given); labels [nb, C] fp32;
    ids [nb, P, ctx] int64 (optional); keep: per-bag keep fraction of the train-time patch drop (1.0 = no drop)."""

    def __init__(self, bags, labels: torch.Tensor, device, ids: Optional[torch.Tensor] = None,
                 keep: Optional[Sequence[float]] = None, seed: int = 1234, lengths: Optional[Sequence[int]] = None):
        self.device = device
        self.seed = int(seed)
        nb = len(lengths) if lengths is not None else len(bags)
        get = bags if callable(bags) else (lambda j: bags[j])
        first = get(0)
        self.F = int(first.shape[1])
        if self.F % 4:
            raise ValueError("DeviceCohort: the feature width must be a multiple of 4 floats")
        self.n = [int(v) for v in lengths] if lengths is not None else [int(get(j).shape[0]) for j in range(nb)]
        off = np.zeros(nb + 1, dtype=np.int64)
        off[1:] = np.cumsum(self.n)
        if off[-1] >= 2 ** 31:
            raise ValueError("DeviceCohort: more than 2^31 rows")
        self.row_off = [int(v) for v in off]
        self.x = torch.empty((int(off[-1]), self.F), device=device, dtype=torch.float32)
        for j in range(nb):                       # one H2D copy per bag, once per run
            a = first if j == 0 else get(j)
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
            if t.shape != (self.n[j], self.F):
                raise ValueError(f"DeviceCohort: bag {j} has shape {tuple(t.shape)}, expected {(self.n[j], self.F)}")
            self.x[self.row_off[j]:self.row_off[j + 1]].copy_(t)
        self.labels = labels.to(device=device, dtype=torch.float32).contiguous()
        self.ids = None if ids is None else ids.to(device=device, dtype=torch.int64).contiguous()
        self.text: Optional[torch.Tensor] = None          # [nb, P, E] cached frozen-tower embeddings (set_text)
        keep = [1.0] * nb if keep is None else [float(v) for v in keep]
        self.keep_frac = keep
        self.k_train = [keep_count(n, f) for n, f in zip(self.n, keep)]
        koff = np.zeros(nb + 1, dtype=np.int64)
        koff[1:] = np.cumsum(self.k_train)
        self.sel_off = [int(v) for v in koff]
        i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=device)      # noqa: E731
        self.row_off_dev, self.keep_dev, self.sel_off_dev = i32(self.row_off), i32(self.k_train), i32(self.sel_off)
        self.sel = torch.zeros(max(1, int(koff[-1])), device=device, dtype=torch.int32)
        self.epoch: Optional[int] = None                  # epoch of the drawn selection; None = identity (no drop)
        self.nb = nb

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def bytes_needed(lengths: Sequence[int], F: int) -> int:
        return int(sum(lengths)) * int(F) * 4

    @staticmethod
    def fits(nbytes: int, device, reserve: int = 24 << 30) -> bool:
        """True when the cohort plus `reserve` bytes of working set fit the device's free memory."""
        free, _ = torch.cuda.mem_get_info(device)
        return nbytes + reserve <= free

    @classmethod
    def from_dataset(cls, ds, device, seed: int = 1234, augmentation: bool = True) -> "DeviceCohort":
        """From the entry points' datasets (dataset.NpyBagDataset: the `.npy` files are read here, un-dropped;
        dataset.SyntheticBags: generated)."""
        import os
        nb = len(ds)
        if hasattr(ds, "root"):                  # NpyBagDataset
            paths = [os.path.join(ds.root, k + ".npy") for k in ds.keys]
            lengths = [int(np.load(p, mmap_mode="r").shape[0]) for p in paths]
            get = lambda j: np.load(paths[j])            # noqa: E731   allow_pickle stays False
            metas = [ds.index[k] for k in ds.keys]
            labels = torch.nn.functional.one_hot(torch.tensor([int(m["label"]) for m in metas]), ds.C).float()
            ids = torch.tensor([m.get("ids", [[0] * 77]) for m in metas], dtype=torch.int64)
            train = ds.mode == "train" and ds.aug and augmentation
            keep = [(0.9 if m.get("kind", "Biopsy") == "Biopsy" else 0.8) if train else 1.0 for m in metas]
        else:                                    # SyntheticBags
            lengths = list(ds.lengths)
            get = lambda j: ds[j]["pathology"]           # noqa: E731
            labels, ids = ds.labels, ds.ids
            keep = [getattr(ds, "keep", 1.0) if augmentation else 1.0] * nb
        return cls(get, labels, device, ids=ids, keep=keep, seed=seed, lengths=lengths)

    def set_text(self, text: torch.Tensor):
        """Cached embeddings of the frozen text tower, one row per bag ([nb, P, E]): `--cache_text 1` as a table."""
        self.text = text.to(device=self.device, dtype=torch.float32).contiguous()
        return self

    # ------------------------------------------------------------------ per epoch
    def draw_epoch(self, epoch: int, augment: bool = True):
        """The epoch's patch drop for every bag, on the current stream (one launch).  augment=False: identity."""
        if not augment or all(k == n for k, n in zip(self.k_train, self.n)):
            self.epoch = None
            return self
        rc = _lib.lib().mil_patch_drop_select(_p(self.row_off_dev), _p(self.keep_dev), _p(self.sel_off_dev), self.nb, 0,
                                              ctypes.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF), ctypes.c_uint64(int(epoch)),
                                              _p(self.sel), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "mil_patch_drop_select")
        self.epoch = int(epoch)
        return self

    def lengths(self, idxs: Sequence[int]) -> List[int]:
        """Rows each bag contributes this epoch - host arithmetic, no sync."""
        return [self.k_train[j] if self.epoch is not None else self.n[j] for j in idxs]

    # ------------------------------------------------------------------ per step
    def feed(self, idxs: Sequence[int], x_dst: torch.Tensor, len_dev: Optional[torch.Tensor] = None,
             y_dst: Optional[torch.Tensor] = None, ids_dst: Optional[torch.Tensor] = None,
             text_dst: Optional[torch.Tensor] = None) -> List[int]:
        """Gather the kept rows of bags `idxs` back to back into x_dst (rows [0, sum k)), write their lengths to len_dev and
        their labels / token ids / cached text embeddings to the given destinations: ONE launch per <= 8 bags on the current
        stream.  Returns the lengths."""
        idxs = [int(j) for j in idxs]
        ks = self.lengths(idxs)
        if sum(ks) > x_dst.shape[0] or x_dst.shape[1] != self.F or not x_dst.is_contiguous():
            raise ValueError(f"DeviceCohort.feed: {sum(ks)} rows x {self.F} do not fit the destination {tuple(x_dst.shape)}")
        aux = []
        if y_dst is not None:
            aux.append((self.labels, y_dst))
        if ids_dst is not None:
            if self.ids is None:
                raise ValueError("DeviceCohort.feed: the cohort holds no token ids")
            aux.append((self.ids, ids_dst))
        if text_dst is not None:
            if self.text is None:
                raise ValueError("DeviceCohort.feed: set_text() first")
            aux.append((self.text, text_dst))
        for tab, dst in aux:
            if tab[0].numel() * tab.element_size() != dst[0].numel() * dst.element_size() or not dst.is_contiguous():
                raise ValueError("DeviceCohort.feed: a side table's row does not match its destination row")
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        sel = _p(self.sel) if self.epoch is not None else None
        row0 = 0
        for s in range(0, len(idxs), _lib.FEED_MAX_BAGS):
            part, kp = idxs[s:s + _lib.FEED_MAX_BAGS], ks[s:s + _lib.FEED_MAX_BAGS]
            d = _lib.CohortFeedDesc()
            d.struct_bytes = ctypes.sizeof(_lib.CohortFeedDesc)
            d.nb, d.L, d.dst_row0, d.dst_bag0, d.naux = len(part), self.F, row0, s, len(aux)
            for b, (j, k) in enumerate(zip(part, kp)):
                d.sel_off[b], d.src_row0[b], d.rows[b], d.bag_id[b] = self.sel_off[j], self.row_off[j], k, j
            for a, (tab, dst) in enumerate(aux):
                d.aux_table[a], d.aux_dst[a] = tab.data_ptr(), dst.data_ptr()
                d.aux_words[a] = tab[0].numel() * tab.element_size() // 4
            rc = _lib.lib().mil_cohort_feed(_p(self.x), sel, ctypes.byref(d), _p(x_dst), _p(len_dev), stream)
            _lib.check(rc, "mil_cohort_feed")
            row0 += sum(kp)
        return ks


class HostFeed:
    """Fallback when the cohort does not fit the device: the reference's pipeline shape (train_ddp.py:193: worker + pinned
    memory + non-blocking copy) without its per-step costs.  A background thread fetches bag t+1 (un-dropped) while step t
    runs: a bag that `load` returns as a PINNED host tensor (the cohort kept in pinned host memory: the intended form - an H2D
    copy of a 45 MB bag takes ~1 ms, the time of a fusion step, and overlaps it) is copied from where it lies; anything else
    (np.load from disk, pageable arrays) goes through one of two pinned staging buffers first, and then the host memcpy sets
    the pace.  `next()` issues the H2D copy on a copy stream into one of two device staging buffers and makes the compute
    stream wait for it by an event; the patch drop and the placement into the bucket then run on the device exactly as for the
    resident cohort (mil_patch_drop_select on one bag + mil_cohort_feed), so the step sees the same rows either way."""

    def __init__(self, load, lengths: Sequence[int], F: int, labels: torch.Tensor, device, ids: Optional[torch.Tensor] = None,
                 keep: Optional[Sequence[float]] = None, seed: int = 1234):
        self.load, self.n, self.F, self.device, self.seed = load, [int(v) for v in lengths], int(F), device, int(seed)
        nmax = max(self.n)
        self.pinned = [torch.empty((nmax, F), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.stage = [torch.empty((nmax, F), device=device, dtype=torch.float32) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device)
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self.labels = labels.to(device=device, dtype=torch.float32).contiguous()
        self.ids = None if ids is None else ids.to(device=device, dtype=torch.int64).contiguous()
        keep = [1.0] * len(self.n) if keep is None else [float(v) for v in keep]
        self.k_train = [keep_count(n, f) for n, f in zip(self.n, keep)]
        i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=device)      # noqa: E731
        # one-bag tables for the select launch: [0, n], [k], [0, k] per staging buffer
        self.tab = [dict(row_off=i32([0, 0]), keep=i32([0]), out_off=i32([0, 0]),
                         sel=torch.zeros(nmax, device=device, dtype=torch.int32)) for _ in range(2)]
        self._thread: Optional[threading.Thread] = None
        self._slot = 0
        self._pending = None          # (bag index, staging slot) being loaded by the thread
        self._used = [False, False]

    def _load_into(self, j: int, s: int):
        a = self.load(j)
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        if t.is_pinned() and t.is_contiguous() and t.dtype == torch.float32:
            src = t                                     # already pinned: the H2D copy reads it where it lies
        else:
            self.pinned[s][:self.n[j]].copy_(t)
            src = self.pinned[s]
        # the H2D copy is issued HERE, from the worker, on the copy stream: it runs while the compute stream is busy with the
        # current step; next() only makes the compute stream wait for its event
        n = self.n[j]
        with torch.cuda.stream(self.copy_stream):
            if self._used[s]:
                self.copy_stream.wait_event(self.consumed[s])       # the feed launch that last read stage[s]
            self.stage[s][:n].copy_(src[:n], non_blocking=True)
            self.copied[s].record(self.copy_stream)

    def prefetch(self, j: int):
        """Start loading bag j on the background thread (host side only; returns at once)."""
        s = self._slot
        self._slot ^= 1
        if self._used[s]:
            self.consumed[s].synchronize()          # the copy that last read pinned[s] / wrote stage[s] has been consumed
        th = threading.Thread(target=self._load_into, args=(j, s), daemon=True)
        th.start()
        self._thread, self._pending = th, (j, s)

    def next(self, x_dst: torch.Tensor, len_dev: Optional[torch.Tensor], y_dst: Optional[torch.Tensor], epoch: Optional[int],
             ids_dst: Optional[torch.Tensor] = None) -> int:
        """Bag prefetched last -> rows [0, k) of x_dst (+ length, label, ids), on the current stream.  Returns k."""
        j, s = self._pending
        self._thread.join()
        n = self.n[j]
        cur = torch.cuda.current_stream()
        cur.wait_event(self.copied[s])
        k = self.k_train[j] if epoch is not None else n
        stream = ctypes.c_void_p(cur.cuda_stream)
        tab = self.tab[s]
        sel = None
        if epoch is not None and k < n:
            vals = (ctypes.c_int32 * 2)(0, n)
            _lib.check(_lib.lib().mil_set_i32(_p(tab["row_off"]), vals, 2, stream), "mil_set_i32")
            vals = (ctypes.c_int32 * 1)(k)
            _lib.check(_lib.lib().mil_set_i32(_p(tab["keep"]), vals, 1, stream), "mil_set_i32")
            # a one-bag table for cohort bag j (bag0 = j): the subset the resident cohort would draw for it
            rc = _lib.lib().mil_patch_drop_select(_p(tab["row_off"]), _p(tab["keep"]), _p(tab["out_off"]), 1, j,
                                                  ctypes.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF),
                                                  ctypes.c_uint64(int(epoch)), _p(tab["sel"]), stream)
            _lib.check(rc, "mil_patch_drop_select")
            sel = _p(tab["sel"])
        d = _lib.CohortFeedDesc()
        d.struct_bytes = ctypes.sizeof(_lib.CohortFeedDesc)
        d.nb, d.L, d.dst_row0, d.dst_bag0, d.naux = 1, self.F, 0, 0, 0
        d.sel_off[0], d.src_row0[0], d.rows[0], d.bag_id[0] = 0, 0, k, j
        aux = [(self.labels, y_dst)] if y_dst is not None else []
        if ids_dst is not None and self.ids is not None:
            aux.append((self.ids, ids_dst))
        d.naux = len(aux)
        for a, (t_, dst) in enumerate(aux):
            d.aux_table[a], d.aux_dst[a], d.aux_words[a] = t_.data_ptr(), dst.data_ptr(), t_[0].numel() * t_.element_size() // 4
        _lib.check(_lib.lib().mil_cohort_feed(_p(self.stage[s]), sel, ctypes.byref(d), _p(x_dst), _p(len_dev), stream),
                   "mil_cohort_feed")
        self.consumed[s].record(cur)
        self._used[s] = True
        return k
