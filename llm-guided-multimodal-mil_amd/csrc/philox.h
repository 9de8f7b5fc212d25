// Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants) and the word layout of the keep-bit tensors
// (see dropout.hip): shared by the stand-alone generator and by k_gate_fwd2, which can draw its own workgroup's words.
#pragma once
#include "mil_common.h"

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

struct philox4 { uint32_t v[4]; };

__device__ __forceinline__ philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(PHILOX_M0, c0), lo0 = PHILOX_M0 * c0;
        const uint32_t hi1 = __umulhi(PHILOX_M1, c2), lo1 = PHILOX_M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return philox4{{c0, c1, c2, c3}};
}

// The 0.5-drop words 4 blk .. 4 blk + 3 of a bit tensor: one Philox block (1 random bit per element).
__device__ __forceinline__ uint4 philox_keep_words_half(uint64_t blk, uint64_t offset, uint32_t seed_lo, uint32_t seed_hi) {
    const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), seed_lo, seed_hi);
    return make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
}
// The 0.25-drop words 2 blk, 2 blk + 1: dropped iff two random bits are set.
__device__ __forceinline__ uint2 philox_keep_words_quarter(uint64_t blk, uint64_t offset, uint32_t seed_lo, uint32_t seed_hi) {
    const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), seed_lo, seed_hi);
    return make_uint2(~(r.v[0] & r.v[1]), ~(r.v[2] & r.v[3]));
}
