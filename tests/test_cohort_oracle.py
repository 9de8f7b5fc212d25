"""CPU: the numpy restatement of the on-device patch drop (oracle/cohort.py) against the reference's semantics
(dataset.py:374-381: `sorted(random.sample(range(n), int(n * keep)))`): count, order, distinctness, tie rule, and the
inclusion statistics of a uniformly random k-subset."""
import random

import numpy as np
import pytest

from oracle import cohort as oc


@pytest.mark.parametrize("n,keep", [(1, 0.9), (3, 0.8), (10, 0.9), (41, 0.9), (97, 0.8), (1000, 0.9), (15592, 0.8)])
def test_selection_is_a_sorted_k_subset_with_the_reference_count(n, keep):
    k = oc.keep_count(n, keep)
    assert k == int(n * keep)                                   # dataset.py:376,379
    sel = oc.patch_drop_select(n, k, bag=5, seed=1234, epoch=3)
    assert sel.shape == (k,)
    assert (np.diff(sel) > 0).all()                             # ascending and distinct (sorted(random.sample(...)))
    assert k == 0 or (0 <= sel[0] and sel[-1] < n)
    # the rule itself, brute force: the k smallest (key, row) pairs
    keys = oc.row_keys(n, 5, 1234, 3)
    brute = sorted(sorted(range(n), key=lambda i: (int(keys[i]), i))[:k])
    assert sel.tolist() == brute


def test_no_drop_is_the_identity_and_epochs_and_bags_differ():
    assert oc.patch_drop_select(50, 50, 0, 1, 0).tolist() == list(range(50))
    a = oc.patch_drop_select(500, 450, 0, 1234, 0)
    assert not np.array_equal(a, oc.patch_drop_select(500, 450, 0, 1234, 1))      # a fresh drop every epoch
    assert not np.array_equal(a, oc.patch_drop_select(500, 450, 1, 1234, 0))      # and per bag
    assert not np.array_equal(a, oc.patch_drop_select(500, 450, 0, 1235, 0))
    assert np.array_equal(a, oc.patch_drop_select(500, 450, 0, 1234, 0))          # reproducible


def test_ties_go_to_the_lower_row(monkeypatch):
    keys = np.array([7, 3, 7, 3, 3, 9, 1], dtype=np.uint32)
    monkeypatch.setattr(oc, "row_keys", lambda n, bag, seed, epoch: keys[:n])
    assert oc.patch_drop_select(7, 3, 0, 0, 0).tolist() == [1, 3, 6]              # keys 1, 3, 3: rows 6, 1, 3
    assert oc.patch_drop_select(7, 5, 0, 0, 0).tolist() == [0, 1, 3, 4, 6]        # the first of the two 7s


def test_inclusion_frequency_matches_random_sample():
    """Every row of a uniformly random k-subset is kept with probability k / n - for the Philox draw as for the
    reference's random.sample (both checked against the binomial band, 5 sigma)."""
    n, keep, epochs = 200, 0.8, 400
    k = oc.keep_count(n, keep)
    cnt = np.zeros(n)
    for e in range(epochs):
        cnt[oc.patch_drop_select(n, k, 2, 99, e)] += 1
    rng = random.Random(0)
    ref = np.zeros(n)
    for _ in range(epochs):
        ref[sorted(rng.sample(range(n), k))] += 1
    p = k / n
    band = 5 * np.sqrt(epochs * p * (1 - p))
    assert np.abs(cnt - epochs * p).max() <= band and np.abs(ref - epochs * p).max() <= band
    assert abs(cnt.mean() - ref.mean()) < 1e-9                  # exactly k rows per draw, both


def test_select_epoch_lays_the_bags_out_back_to_back():
    ns, keeps = [10, 0, 33, 7], [0.9, 0.9, 0.8, 1.0]
    off = np.concatenate([[0], np.cumsum(ns)])
    ks = [oc.keep_count(n, f) for n, f in zip(ns, keeps)]
    sel = oc.select_epoch(off, ks, 11, 2)
    assert sel.shape == (sum(ks),)
    p = 0
    for j, k in enumerate(ks):
        part = sel[p:p + k]
        assert ((part >= off[j]) & (part < off[j + 1])).all()
        assert np.array_equal(part - off[j], oc.patch_drop_select(ns[j], k, j, 11, 2))
        p += k
