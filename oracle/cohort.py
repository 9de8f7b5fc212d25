"""CPU restatement of the cohort feed (csrc/cohort.hip): the per-epoch patch drop of dataset.py:374-381 drawn from
Philox4x32-10, and the row gather that follows it.  TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else.

Reference semantics (dataset.py:374-381): `feat[sorted(random.sample(range(n), int(n * keep))), :]` with keep = 0.9 for
biopsies and 0.8 for resections - a uniformly random k-subset of the bag's rows, kept in ascending order.  The reference
draws it from Python's Mersenne Twister inside DataLoader workers (an unseeded, per-worker stream: not reproducible even
on the reference); the device draws the SAME distribution from Philox: the k rows with the smallest 32-bit keys, ties by
row number.  `keep_count` is the reference's own expression; `patch_drop_select` pins the device kernel bit for bit."""
import numpy as np

from .philox import philox4x32_10

COHORT_KEY_XOR = 0x70617463685F6472


def keep_count(n: int, keep: float) -> int:
    """dataset.py:376,379: int(feat.shape[0] * 0.9) / int(feat.shape[0] * 0.8)."""
    return int(n * keep)


def row_keys(n: int, bag: int, seed: int, epoch: int) -> np.ndarray:
    """uint32 [n]: key of row i = word (i & 3) of Philox(counter = (i >> 2, bag, epoch_lo, epoch_hi), key = seed ^ XOR)."""
    key = (seed ^ COHORT_KEY_XOR) & 0xFFFFFFFFFFFFFFFF
    q = np.arange((n + 3) // 4, dtype=np.uint32)
    r = philox4x32_10(q, np.uint32(bag), np.uint32(epoch & 0xFFFFFFFF), np.uint32((epoch >> 32) & 0xFFFFFFFF),
                      key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)
    return np.stack(r, 1).reshape(-1)[:n].astype(np.uint32)


def patch_drop_select(n: int, k: int, bag: int, seed: int, epoch: int) -> np.ndarray:
    """int64 [k]: rows (bag-local, ascending) holding the k smallest keys, ties by row number."""
    k = max(0, min(int(k), int(n)))
    if k == n:
        return np.arange(n, dtype=np.int64)
    keys = row_keys(n, bag, seed, epoch)
    order = np.argsort(keys, kind="stable")          # stable: equal keys stay in row order
    return np.sort(order[:k]).astype(np.int64)


def select_epoch(row_off, keeps, seed: int, epoch: int) -> np.ndarray:
    """The whole `sel` array of mil_patch_drop_select: absolute cohort rows, bag after bag."""
    out = []
    for j, k in enumerate(keeps):
        n = int(row_off[j + 1] - row_off[j])
        out.append(int(row_off[j]) + patch_drop_select(n, k, j, seed, epoch))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)
