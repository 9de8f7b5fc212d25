"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in numpy, and the
dropout keep-bit tensors the HIP kernels draw from it (csrc/dropout.hip).  TEST INFRASTRUCTURE ONLY: imported by tests/
and nothing else.

The reference draws its dropout masks (model/dim1/ABMIL.py:26,49; model/aggregator.py:129) from torch's CUDA Philox
stream; the element -> stream-position map is ATen-internal, so parity under dropout is defined with the mask as an
explicit input: the GPU kernels write the keep bits they used, the oracle multiplies by exactly those.  This module
pins the generator itself: known-answer vectors of Random123 (`kat_vectors`, philox4x32 10 rounds) and the bit layout."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)

# (counter, key, expected) from Random123's kat_vectors for philox4x32, 10 rounds
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (broadcast); returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def keep_bits(rows: int, cols: int, p_drop: float, seed: int, offset: int) -> np.ndarray:
    """uint32 [rows, cols // 32]: the tensor mil_dropout_keep_bits writes (same modes, same layout)."""
    assert cols % 32 == 0
    nwords = rows * (cols // 32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    o_lo, o_hi = offset & 0xFFFFFFFF, (offset >> 32) & 0xFFFFFFFF
    if p_drop == 0.5:
        nb = (nwords + 3) // 4
        blk = np.arange(nb, dtype=np.uint64)
        r = philox4x32_10((blk & MASK32).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32), o_lo, o_hi, k0, k1)
        words = np.stack(r, 1).reshape(-1)[:nwords]
    elif p_drop == 0.25:
        nb = (nwords + 1) // 2
        blk = np.arange(nb, dtype=np.uint64)
        r = philox4x32_10((blk & MASK32).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32), o_lo, o_hi, k0, k1)
        words = np.stack([~(r[0] & r[1]), ~(r[2] & r[3])], 1).reshape(-1)[:nwords]
    else:
        t = p_drop * 4294967296.0
        thr = np.uint32(0xFFFFFFFF if t >= 4294967295.0 else int(t))
        blk = np.arange(nwords * 8, dtype=np.uint64)
        r = philox4x32_10((blk & MASK32).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32), o_lo, o_hi, k0, k1)
        keep = (np.stack(r, 1).reshape(nwords, 32) >= thr).astype(np.uint32)
        words = (keep << np.arange(32, dtype=np.uint32)).sum(1, dtype=np.uint64).astype(np.uint32)
    return words.astype(np.uint32).reshape(rows, cols // 32)


def unpack_bits(bits: np.ndarray, cols: int) -> np.ndarray:
    """uint32 [rows, cols // 32] -> float32 0/1 [rows, cols]."""
    b = np.asarray(bits, dtype=np.uint32)
    out = ((b[:, :, None] >> np.arange(32, dtype=np.uint32)) & np.uint32(1)).reshape(b.shape[0], -1)
    return out[:, :cols].astype(np.float32)
