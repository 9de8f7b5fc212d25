"""CPU: the Philox4x32-10 restatement (oracle/philox.py) against Random123's known-answer vectors, and the keep-bit
layout the HIP dropout kernels share with it (csrc/dropout.hip)."""
import numpy as np

from oracle import philox as P


def test_known_answer_vectors():
    for ctr, key, want in P.KAT:
        got = P.philox4x32_10(*[np.uint32(v) for v in ctr], *key)
        assert [int(v) for v in got] == list(want)


def test_keep_bits_modes_and_layout():
    for p, lo, hi in ((0.5, 0.49, 0.51), (0.25, 0.74, 0.76), (0.1, 0.89, 0.91)):
        bits = P.keep_bits(64, 512, p, seed=1234, offset=3)
        assert bits.shape == (64, 16) and bits.dtype == np.uint32
        keep = P.unpack_bits(bits, 512)
        assert lo < keep.mean() < hi
        # bit (col & 31) of word (col >> 5)
        assert keep[5, 37] == float((int(bits[5, 1]) >> 5) & 1)
    a = P.keep_bits(8, 64, 0.5, 1, 0)
    assert not np.array_equal(a, P.keep_bits(8, 64, 0.5, 1, 1))      # the offset moves the stream
    assert not np.array_equal(a, P.keep_bits(8, 64, 0.5, 2, 0))      # so does the seed
    assert np.array_equal(a, P.keep_bits(8, 64, 0.5, 1, 0))


def test_p_half_word_is_a_raw_philox_word():
    bits = P.keep_bits(1, 256, 0.5, seed=(7 << 32) | 9, offset=(1 << 32) | 5)
    r = P.philox4x32_10(np.uint32(1), np.uint32(0), np.uint32(5), np.uint32(1), 9, 7)
    assert [int(v) for v in bits[0, 4:8]] == [int(v) for v in r]
