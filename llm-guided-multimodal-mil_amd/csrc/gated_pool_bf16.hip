// K1, bf16-storage variant (BASELINE config 5: large bags, N=4096, D=1024): x and the gate weights are held
// in bf16, every accumulation (gate pre-activations, scores, softmax, pooled sum, gradients) stays fp32.
// Same arithmetic as gated_pool.hip (reference model/dim1/ABMIL.py:47-59) on rounded inputs; parity with the
// fp32 oracle is REPORTED (max |dlogit|), the 1e-3 bar applies to the fp32 path.
//
//   k_gate_fwd_bf16   v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate, so this kernel sits at the HBM/L2 ridge
//   k_pool_*_bf16     the HBM-bound pool stages read half the bytes
//   k_gate_bwd_dw_x16 weight gradient with x read as bf16 and widened in staging (fp32 MFMA; a bf16-MFMA
//                     version of the transposed product is the next step)
#include <stdlib.h>

#include "mil_common.h"
#include "gate_reduce.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float(((unsigned)v) << 16); }

// Train mode (ABMIL.py:49): zero the dropped elements of 8 consecutive bf16 values; bits8 = their 8 keep bits (bit e =
// element e).  Two forms, chosen per kernel by measurement (32 x 4096 x 1024, one box, A/B libraries):
//   keep_bf16x8   per dword (two values): y = bits b0 -> bit 0, b1 -> bit 16; y * 0xffff = the 16-bit lane masks.  The
//                 gate forward's hand-scheduled loop wants this one (137 us; the bit-field form: 196 us).
//   keep_bf16x8_b two one-bit signed fields (all ones / zero) merged into the dword's two lanes (v_bfe_i32 x 2, v_bfi,
//                 v_and: 4 instead of 6 VALU per dword).  Pool pass 72 -> 69 us, weight gradient 192 -> 186 us.
__device__ __forceinline__ u16x8 keep_bf16x8(const u16x8 v, unsigned bits8) {
    typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
    u32x4k d = __builtin_bit_cast(u32x4k, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned m2 = (bits8 >> (2 * e)) & 3u;
        const unsigned y = (m2 | (m2 << 15)) & 0x00010001u;
        d[e] &= (y << 16) - y;
    }
    return __builtin_bit_cast(u16x8, d);
}
__device__ __forceinline__ u16x8 keep_bf16x8_b(const u16x8 v, unsigned bits8) {
    typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
    u32x4k d = __builtin_bit_cast(u32x4k, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)bits8, 2 * e, 1);
        const unsigned hi = (unsigned)__builtin_amdgcn_sbfe((int)bits8, 2 * e + 1, 1);
        d[e] &= (lo & 0x0000ffffu) | (hi & 0xffff0000u);
    }
    return __builtin_bit_cast(u16x8, d);
}

// ---------------------------------------------------------------------------------------------------- cast
__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ src, u16* __restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
        ushort4 o;
        o.x = __builtin_bit_cast(u16, (__bf16)v[0]);      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
        o.y = __builtin_bit_cast(u16, (__bf16)v[1]);
        o.z = __builtin_bit_cast(u16, (__bf16)v[2]);
        o.w = __builtin_bit_cast(u16, (__bf16)v[3]);
        *reinterpret_cast<ushort4*>(dst + i) = o;
    } else {
        for (size_t j = i; j < n; ++j) dst[j] = __builtin_bit_cast(u16, (__bf16)src[j]);
    }
}

// Saved gates in bf16 (round 2).  Round 1 kept {V | U} as fp32 [R, 384]: at config 5 that is 201 MB written by the
// forward (4-byte scattered stores in the epilogue, nothing to overlap with) and 201 MB re-read by every column tile of
// the weight-gradient kernel - more traffic than the bf16 x itself (268 MB).  The bf16-MFMA weight gradient rounds dPre
// to bf16 anyway, so V and U are stored rounded to bf16: half the bytes, and written 16 bytes per lane: a wave converts
// its 32 x 192 tile, transposes it through its own 12 KB of LDS and stores whole 16-byte chunks.
//   tile_lds: this wave's [32][192] u16 scratch; val(c, u, i) = gate value of d-chunk c, V/U u, accumulator register i
//   row_base: first global row of the tile; dcol0: first d of the wave (96 wc)
template <typename F>
__device__ __forceinline__ void store_gates16_tile(u16* tile_lds, u16* __restrict__ gates16, int row_base, int R, int dcol0,
                                                   int lane, F val) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                tile_lds[mfma32_row(i, h) * 192 + u * 96 + 32 * c + r] = __builtin_bit_cast(u16, (__bf16)val(c, u, i));
    // same wave wrote what it reads: LDS operations of a wave complete in order
#pragma unroll
    for (int n = 0; n < 12; ++n) {
        const int k = lane + 64 * n, row = k / 24, ch = k % 24;        // 24 chunks of 8 columns per row
        const int u = ch / 12, d = dcol0 + 8 * (ch % 12);
        const u16x8 v = *reinterpret_cast<const u16x8*>(tile_lds + row * 192 + 8 * ch);
        const int gr = row_base + row;
        if (gr < R) *reinterpret_cast<u16x8*>(gates16 + (size_t)gr * 384 + u * 192 + d) = v;
    }
}

// ---------------------------------------------------------------------------------------------------- gate forward
// Same decomposition as k_gate_fwd: 512 threads, 128 rows x 384 gate columns, wave (wr, wc) = 32 rows x 3 d-chunks
// x {V, U}.  K-slices of 64 bf16 = 128 B per row: byte-for-byte the fp32 kernel's LDS geometry, so the same LDS-DMA
// staging (global_load_lds_dwordx4) and the same chunk swizzle c ^ ((row >> 1) & 7) apply; lane (r, h) reads the
// 16-byte chunk 2 ks + h of its row = the 8 k's 16 ks + 8h .. +7 = exactly one 32x32x16 operand fragment.
#define HB_TM 128
#define HB_BK 64
#define HB_RS 64         // row stride in bf16 elements (128 B, unpadded)
#define HB_NG 384
typedef __attribute__((address_space(3))) void hb_lds_void;

__global__ __launch_bounds__(512) void k_gate_fwd_bf16(const u16* __restrict__ x, const u16* __restrict__ Wv,
                                                       const float* __restrict__ bv, const u16* __restrict__ Wu,
                                                       const float* __restrict__ bu, const float* __restrict__ wvec,
                                                       const float* __restrict__ battn, float* __restrict__ scores,
                                                       float* __restrict__ gates, int R, int L, u16* __restrict__ gates16,
                                                       const uint32_t* __restrict__ xbits, float xscale) {
    // xbits (train mode, ABMIL.py:49): keep bits of the patch dropout [R][L/32]; a 64-wide K slice is two words per row,
    // loaded one slice ahead; the A fragment (8 k-values of the lane's row) is masked after its LDS read, the 1/(1-p)
    // multiplies the finished pre-activation
    __shared__ __attribute__((aligned(16))) u16 smem[2 * (HB_TM + HB_NG) * HB_RS];
    u16* xs = smem;                        // [2][128][64]
    u16* ws = smem + 2 * HB_TM * HB_RS;    // [2][384][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * HB_TM;
    const int prow = lane >> 3, pch = lane & 7;
    const u16* gsrc[8];
    int ldst[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < 2) {
            const int lr = (2 * wave + i) * 8 + prow;
            const int gr = min(row0 + lr, R - 1);
            gsrc[i] = x + (size_t)gr * L + 8 * (pch ^ ((lr >> 1) & 7));
            ldst[i] = (2 * wave + i) * 8 * HB_RS;
        } else {
            const int wrow = (6 * wave + (i - 2)) * 8 + prow;
            gsrc[i] = (wrow < 192 ? Wv + (size_t)wrow * L : Wu + (size_t)(wrow - 192) * L) + 8 * (pch ^ ((wrow >> 1) & 7));
            ldst[i] = (6 * wave + (i - 2)) * 8 * HB_RS;
        }
    }
    auto dma_piece = [&](int i, int buf, int k0) {
        u16* dst = (i < 2 ? xs + buf * HB_TM * HB_RS : ws + buf * HB_NG * HB_RS) + ldst[i];
        __builtin_amdgcn_global_load_lds(gsrc[i] + k0, (hb_lds_void*)dst, 16, 0, 0);
    };

    f32x16 acc[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][u][i] = 0.f;

    const int nslice = L / HB_BK;
    const bool drop = xbits != nullptr;
    const uint32_t* mrow = drop ? xbits + (size_t)min(row0 + 32 * wr + r, R - 1) * (L >> 5) : nullptr;
    uint2 mnext = make_uint2(0u, 0u);
    if (drop) mnext = *reinterpret_cast<const uint2*>(mrow);
    // (starting each workgroup's K loop at a different slice, to de-correlate the L2 requests for the shared gate
    //  weights, measured no gain: 107.4 vs 106.2 us fp32, 172 vs 171 us bf16 - the slices stay in natural order)
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(i, 0, 0);
    __syncthreads();
    const int fx = (r >> 1) & 7;
    for (int s = 0; s < nslice; ++s) {
        const int buf = s & 1;
        const int k1 = min(s + 1, nslice - 1) * HB_BK;
        const uint2 mcur = mnext;
        if (drop) mnext = *reinterpret_cast<const uint2*>(mrow + 2 * min(s + 1, nslice - 1));
        const u16* xa = xs + (buf * HB_TM + 32 * wr + r) * HB_RS;
        const u16* wb = ws + (buf * HB_NG + 32 * 3 * wc + r) * HB_RS;
        u16x8 a[2], b[2][3][2];
        auto frag_piece = [&](int ks, int q, int p) {
            const int ch = 8 * ((2 * ks + h) ^ fx);
            if (p == 0) {
                a[q] = *reinterpret_cast<const u16x8*>(xa + ch);
            } else {
                const int c = (p - 1) >> 1, u = (p - 1) & 1;
                b[q][c][u] = *reinterpret_cast<const u16x8*>(wb + (u * 192 + 32 * c) * HB_RS + ch);
            }
        };
#pragma unroll
        for (int p = 0; p < 7; ++p) frag_piece(0, 0, p);
        if (drop) a[0] = keep_bf16x8(a[0], mcur.x >> (8 * h));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks & 1;
            dma_piece(2 * ks, buf ^ 1, k1);            // two DMA pieces of the next slice per k-step
            dma_piece(2 * ks + 1, buf ^ 1, k1);
            if (ks < 3) {
#pragma unroll
                for (int p = 0; p < 7; ++p) frag_piece(ks + 1, q ^ 1, p);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[c][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q]),
                                                                        __builtin_bit_cast(bf16x8, b[q][c][u]), acc[c][u],
                                                                        0, 0, 0);
            if (drop && ks < 3) {
                // keep bits of k-step ks + 1 (k = 16 (ks + 1) + 8 h .. + 7 of the slice) on the fragment just read
                const unsigned wd_ = (ks + 1) < 2 ? mcur.x : mcur.y;
                a[q ^ 1] = keep_bf16x8(a[q ^ 1], wd_ >> (16 * ((ks + 1) & 1) + 8 * h));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    float part[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) part[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int d = 32 * (3 * wc + c) + r;
        const float bvd = bv[d], bud = bu[d], wd = wvec[d];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = fast_tanh(fmaf(acc[c][0][i], xscale, bvd));
            const float u = fast_sigmoid(fmaf(acc[c][1][i], xscale, bud));
            part[i] += wd * v * u;
            if (gates != nullptr) {
                const int gr = row0 + 32 * wr + mfma32_row(i, h);
                if (gr < R) {
                    gates[(size_t)gr * HB_NG + d] = v;
                    gates[(size_t)gr * HB_NG + 192 + d] = u;
                }
            }
        }
    }
    float* sred = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = half_sum_lane31(part[i]);
        if (r == 31) sred[wc * HB_TM + 32 * wr + mfma32_row(i, h)] = v;
    }
    __syncthreads();
    if (tid < HB_TM) {
        const int gr = row0 + tid;
        if (gr < R) scores[gr] = sred[tid] + sred[HB_TM + tid] + battn[0];
    }
    if (gates16 != nullptr) {
        __syncthreads();                   // sred is read; the staging area is free again
        store_gates16_tile(smem + wave * (32 * 192), gates16, row0 + 32 * wr, R, 96 * wc, lane, [&](int c, int u, int i) {
            const int d = 32 * (3 * wc + c) + r;
            return u == 0 ? fast_tanh(fmaf(acc[c][0][i], xscale, bv[d])) : fast_sigmoid(fmaf(acc[c][1][i], xscale, bu[d]));
        });
    }
}

// ---------------------------------------------------------------------------------------------------- gate forward, deep pipeline
// Large-R form (config 5: 131 072 rows x 1024): 256 rows x 384 gate columns per workgroup, K-slices of 32 bf16
// (64-byte rows).  Two LDS rings filled by LDS-DMA: the x stream (HBM, long latency) runs FOUR slices ahead in a
// five-slot ring, the gate weights (L2-resident) one slice ahead in a two-slot ring.  Per slice a wave issues its
// three weight pieces first and its two x pieces last, so one counted s_waitcnt vmcnt(2) at the end of the slice
// publishes "weights of s+1 and every x slice up to s+3" while the newest x pieces stay in flight across the barrier.
// Against the 128-row kernel above the gate weights are re-streamed from L2 half as often.
// Measured (round 1, ablations of this loop at R = 131 072, L = 1024): with the MFMAs removed the two streams alone
// take 87 us (x only 60 us = 4.4 TB/s, weights only 54 us), i.e. ~0.27 us per 1 KiB LDS-DMA piece per wave: the
// LDS-DMA issue path (~30 GB/s per CU), not HBM or the matrix pipe, bounds this kernel, and the two waves of a SIMD
// run their DMA and MFMA phases in lockstep, so the full kernel is close to the SUM (140-150 us) rather than the max.
// Also tried (round 1): a one-wave-per-SIMD layout (256 threads, wave = 96 x 192 = 18 accumulator tiles, register-staged
// global_load_dwordx4 + ds_write_b128 with asm loads and counted vmcnt): 240-250 us.  288 accumulator registers do
// not fit the 256 AGPRs and hipcc then shuttles tiles between the two register files around every MFMA
// (hundreds of v_accvgpr_read/write per slice); 16 tiles per wave is the limit, which N = 384 does not tile evenly.
// And the 128-row kernel with register staging (asm global_load_dwordx4 two slices ahead + ds_write_b128) instead of
// LDS-DMA: 170 us, bit-identical results - the same as with DMA (171), so at 128 rows the staging method is not the
// limiter either (7 LDS fragment reads per 6 MFMAs: 224 KB of reads + 64 KB of writes per slice against 1536 MFMA cycles).
// Wave (wr, wc) = 64 rows x 3 d-chunks x {V, U}: 12 accumulator tiles (192 registers); B fragments rotate through
// three register slots read two ahead of use (counted lgkmcnt waits).
// LDS image: row = 4 chunks of 16 B, chunk c of row `row` stored at c ^ ((row >> 2) & 3): conflict-free for the
// 16-lane groups of ds_read_b128, applied on the SOURCE address of the DMA (LDS side is lane-linear).
//
// LDS-DMA is issued through inline asm: the compiler models global_load_lds as a FLAT access that may return out of
// order with DS reads, so with one in flight it turns EVERY LDS wait into lgkmcnt(0) and a read-ahead fragment
// schedule collapses.  Issued here it is invisible to that bookkeeping; the vmcnt side is handled by hand anyway.
__device__ __forceinline__ void dma16_raw(const void* gptr, unsigned lds_byte_addr_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte_addr_uniform)
                 : "memory");
}
#define HC_TM 256
#define HC_BK 32
#define HC_XS (HC_TM * HC_BK)        // u16 per x slot  (16 KB)
#define HC_WS (HB_NG * HC_BK)        // u16 per weight slot (24 KB)
#define HC_NX 4          // x ring: slices s .. s + 3
#define HC_NW 3          // weight ring: slices s .. s + 2

// DROP (train mode, ABMIL.py:49): the keep words of the patch dropout ride the same LDS-DMA stream as x - one 1 KiB piece
// = the 4 words (4 K slices) of 64 rows, issued by waves 0-3 every fourth slice, TWO four-slice groups ahead into a
// three-group ring (12 KB) - so no ordinary load enters the kernel's hand-counted vmcnt accounting (the piece is the
// oldest operation of its slice: that slice's wait becomes vmcnt(8), and by the next slice's vmcnt(7) it has landed);
// the A fragments are masked after their LDS read (keep_bf16x8), the 1 / (1 - p) multiplies the finished pre-activation.
//
// PQ > 0 (= L / 512; round 4): the attention-pool PARTIAL PASS of the workgroup's eight 32-row tiles runs in the epilogue,
// as k_gate_fwd2<.., PQ> does for fp32.  Wave w takes tile w (rows 32 w .. 32 w + 31 of the workgroup) once the 256 scores
// are final and plays k_pool_partial_bf16's four waves one after the other - virtual wave v = rows v, v + 4, ..., the same
// per-element accumulation order, the same 16-value reduction for the head-projection by-product, the four partial sums
// folded in the same order - so partials / hrow equal the stand-alone kernel's bit for bit.  The rows come back from the
// Infinity Cache (the main loop streamed the workgroup's 512 KB moments ago; one round of the grid is 128 MiB of the 256):
// the stand-alone pass's 257 MiB HBM read and its launch are gone.  Virtual wave 0's rows are requested between the two row
// halves of the activation epilogue (the first half's 96 accumulator registers are free by then), every further virtual
// wave's rows while its predecessor is being accumulated.  Only for batches whose tiles are all full and aligned.
struct GateFwdPool16 {
    const int32_t* tile_map;    // [T][4] = {bag, row0, nrows, 0}
    float* partials;            // [T][L] then [T][2]
    int T;
    const float* Wf;            // [2][L] head rows (C == 2)
    float* hrow;                // [R][2]
    const uint32_t* mbits;      // [B][L/32] keep words of the head's dropout (train mode), else NULL
    float mscale;
};

template <int GMODE, bool DROP, int PQ = 0>      // saved gates: 0 none, 1 fp32 [R, 384], 2 bf16 [R, 384] (branch-free epilogue per mode)
__global__ __launch_bounds__(512) void k_gate_fwd_bf16_deep(const u16* __restrict__ x, const u16* __restrict__ Wv,
                                                            const float* __restrict__ bv, const u16* __restrict__ Wu,
                                                            const float* __restrict__ bu, const float* __restrict__ wvec,
                                                            const float* __restrict__ battn, float* __restrict__ scores,
                                                            float* __restrict__ gates, int R, int L,
                                                            u16* __restrict__ gates16, const uint32_t* __restrict__ xbits,
                                                            float xscale, GateFwdPool16 pool = GateFwdPool16{}) {
    __shared__ __attribute__((aligned(16))) u16 smem[HC_NX * HC_XS + HC_NW * HC_WS];      // 64 + 72 KB
    __shared__ __attribute__((aligned(16))) uint32_t mring[DROP ? 3 * HC_TM * 4 : 4];    // keep words [3 groups][256 rows][4 slices]
    u16* xring = smem;
    u16* wring = smem + HC_NX * HC_XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * HC_TM;
    // DMA pieces: 16 rows x 64 B each.  x: 16 pieces per slice, wave takes w and w + 8; weights: 24 pieces, wave takes
    // w, w + 8, w + 16.
    const int prow = lane >> 2, pch = lane & 3;
    const u16* xsrc[2];
    const u16* wsrc[3];
    unsigned xdst[2], wdst[3];          // byte address inside slot 0 of the ring, wave-uniform
    const unsigned lds0 = (unsigned)(uintptr_t)(hb_lds_void*)smem;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wave + 8 * i, lr = 16 * p + prow;
        const int gr = min(row0 + lr, R - 1);
        xsrc[i] = x + (size_t)gr * L + 8 * (pch ^ ((lr >> 2) & 3));
        xdst[i] = __builtin_amdgcn_readfirstlane(lds0 + 2u * (unsigned)(16 * p * HC_BK));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = wave + 8 * i, wrow = 16 * p + prow;
        wsrc[i] = (wrow < 192 ? Wv + (size_t)wrow * L : Wu + (size_t)(wrow - 192) * L) + 8 * (pch ^ ((wrow >> 2) & 3));
        wdst[i] = __builtin_amdgcn_readfirstlane(lds0 + 2u * (unsigned)(HC_NX * HC_XS + 16 * p * HC_BK));
    }
    auto dma_x = [&](int i, int slot, int k0) { dma16_raw(xsrc[i] + k0, xdst[i] + 2u * (unsigned)(slot * HC_XS)); };
    auto dma_w = [&](int i, int slot, int k0) { dma16_raw(wsrc[i] + k0, wdst[i] + 2u * (unsigned)(slot * HC_WS)); };
    // keep words: wave w < 4 moves rows 64 w .. 64 w + 63 (lane = row), the 4 words of slice group `grp`, into ring slot grp % 3
    const int ngrp = (L / HC_BK) / 4;                                 // L % 128 == 0 (host)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);         // scalar copy: branch conditions and LDS bases of the keep words
    const uint32_t* msrc = DROP ? xbits + (size_t)min(row0 + 64 * (wave_u & 3) + lane, R - 1) * (L >> 5) : nullptr;
    const unsigned mdst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(hb_lds_void*)mring + (unsigned)(64 * (wave_u & 3) * 16));
    auto dma_m = [&](int grp) {
        dma16_raw(msrc + 4 * min(grp, ngrp - 1), __builtin_amdgcn_readfirstlane(mdst + (unsigned)((grp % 3) * HC_TM * 16)));
    };

    f32x16 acc[2][3][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][c][u][i] = 0.f;

    const int nslice = L / HC_BK;
    // prologue: x slices 0 .. 2 and weight slices 0, 1
#pragma unroll
    for (int q = 0; q < HC_NX - 1; ++q) {
        const int k0 = min(q, nslice - 1) * HC_BK;
        dma_x(0, q, k0);
        dma_x(1, q, k0);
    }
#pragma unroll
    for (int q = 0; q < HC_NW - 1; ++q) {
        const int k0 = min(q, nslice - 1) * HC_BK;
#pragma unroll
        for (int i = 0; i < 3; ++i) dma_w(i, q, k0);
    }
    if (DROP && wave_u < 4) { dma_m(0); dma_m(1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

#if defined(HC_STAMP)
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int fx = (r >> 2) & 3;       // rows 64 wr + 32 q + r and 32 (..) + r: (row >> 2) & 3 == (r >> 2) & 3
    // Per slice: 12 B fragments j = 6 ks + 2 c + u, each feeding two MFMAs (row tiles q = 0, 1).  B fragments rotate
    // through three register slots and are read two ahead of their use; the A pair of k-step 1 is read during
    // k-step 0; the DMA pieces are spread over the fragment steps.  sched_barrier pins the order.
    // Both streams run TWO slices ahead of their use (round 2; round 1 kept the weights one slice ahead in a two-slot
    // ring: every slice then had to wait out one full issue-to-landed latency of its weight pieces, ~1.1 us against
    // 0.85 us of MFMA time per slice - the "weights only: 54 us" of the round-1 ablation was that latency x 64 slices,
    // not bandwidth).  Issue order inside a slice: the three weight pieces of slice s + 2 FIRST, the two x pieces of slice
    // s + 3 LAST; the queue retires in order, so one counted s_waitcnt vmcnt(7) at the end of slice s (= all but the x
    // pieces of s + 2, the weight pieces of s + 2 and the x pieces of s + 3) publishes "weights of s + 1, x of s + 1"
    // and leaves every younger piece in flight across the barrier.
    int xs = 0, wsl = 0;                // slots of slice s in the x ring / the weight ring
    for (int s = 0; s < nslice; ++s) {
        const int kw = min(s + HC_NW - 1, nslice - 1) * HC_BK, kx = min(s + HC_NX - 1, nslice - 1) * HC_BK;
        const int xnew = xs == 0 ? HC_NX - 1 : xs - 1;          // slot of slice s + 3 = the one slice s - 1 used
        const int wnew = wsl == 0 ? HC_NW - 1 : wsl - 1;        // slot of slice s + 2 = the one slice s - 1 used
        const u16* xa = xring + xs * HC_XS + (64 * wr + r) * HC_BK;
        const u16* wb = wring + wsl * HC_WS + (96 * wc + r) * HC_BK;
        const bool mask_slice = DROP && wave_u < 4 && (s & 3) == 0;         // scalar
        if (mask_slice) dma_m((s >> 2) + 2);                                  // first operation of the slice
        unsigned mw[2] = {0u, 0u};
        if (DROP) {
            const uint32_t* mp = mring + ((s >> 2) % 3) * (HC_TM * 4) + (64 * wr + r) * 4 + (s & 3);
            mw[0] = mp[0];
            mw[1] = mp[32 * 4];
        }
#if !defined(HC_PHASED_LOOP)
        u16x8 a[2][2], bs[3];
        auto read_a = [&](int ks) {
            const int ch = 8 * ((2 * ks + h) ^ fx);
            a[ks][0] = *reinterpret_cast<const u16x8*>(xa + ch);
            a[ks][1] = *reinterpret_cast<const u16x8*>(xa + 32 * HC_BK + ch);
            if (DROP) {
                a[ks][0] = keep_bf16x8(a[ks][0], mw[0] >> (16 * ks + 8 * h));
                a[ks][1] = keep_bf16x8(a[ks][1], mw[1] >> (16 * ks + 8 * h));
            }
        };
        auto read_b = [&](int j) {
            const int ks = j / 6, c = (j % 6) >> 1, u = j & 1;
            const int ch = 8 * ((2 * ks + h) ^ fx);
            bs[j % 3] = *reinterpret_cast<const u16x8*>(wb + (u * 192 + 32 * c) * HC_BK + ch);
        };
        read_a(0);
        read_b(0);
        read_b(1);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (j + 2 < 12) read_b(j + 2);
            if (j == 2) read_a(1);
            if (j == 1) dma_w(0, wnew, kw);
            if (j == 3) dma_w(1, wnew, kw);
            if (j == 5) dma_w(2, wnew, kw);
            if (j == 7) dma_x(0, xnew, kx);
            if (j == 9) dma_x(1, xnew, kx);
            const int ks = j / 6, c = (j % 6) >> 1, u = j & 1;
            const bf16x8 bf = __builtin_bit_cast(bf16x8, bs[j % 3]);
            acc[0][c][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][0]), bf, acc[0][c][u], 0, 0, 0);
            acc[1][c][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][1]), bf, acc[1][c][u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        // Phased k-step (round 2): ALL eight fragments of a k-step are read as one block, then its twelve MFMAs run as one
        // uninterrupted cluster at raised priority.  With the fragments trickling in one per MFMA pair (round 1) the two
        // waves of a SIMD stayed in lockstep - both waiting on LDS, then both multiplying: MFMA busy 0.37.  A contiguous
        // read block and a contiguous 384-cycle cluster let them fall into alternation on their own: the matrix pipe
        // serialises the two clusters once, and from then on one wave reads while the other multiplies.
        u16x8 a[2], bq[6];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = 8 * ((2 * ks + h) ^ fx);
            a[0] = *reinterpret_cast<const u16x8*>(xa + ch);
            a[1] = *reinterpret_cast<const u16x8*>(xa + 32 * HC_BK + ch);
#pragma unroll
            for (int j = 0; j < 6; ++j) bq[j] = *reinterpret_cast<const u16x8*>(wb + ((j & 1) * 192 + 32 * (j >> 1)) * HC_BK + ch);
#if !defined(HC_ABL_NOW)
            if (ks == 0) { dma_w(0, wnew, kw); dma_w(1, wnew, kw); dma_w(2, wnew, kw); }
#endif
#if !defined(HC_ABL_NOX)
            if (ks == 1) { dma_x(0, xnew, kx); dma_x(1, xnew, kx); }
#endif
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const bf16x8 bf = __builtin_bit_cast(bf16x8, bq[j]);
#if !defined(HC_ABL_NOMFMA)
                acc[0][j >> 1][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), bf, acc[0][j >> 1][j & 1], 0, 0, 0);
                acc[1][j >> 1][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]), bf, acc[1][j >> 1][j & 1], 0, 0, 0);
#else
                asm volatile("" ::"v"(bf), "v"(a[0]), "v"(a[1]));
#endif
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#if defined(HC_ABL_NOW) && defined(HC_ABL_NOX)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif defined(HC_ABL_NOW)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // ablation: x pieces only (2 per slice): x of s + 1 landed
#elif defined(HC_ABL_NOX)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");     // ablation: weight pieces only (3 per slice)
#else
        // everything issued before the previous slice's x pieces has landed (a slice that opened with a keep-word piece has
        // one more operation behind them)
        if (mask_slice) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        xs = xs == HC_NX - 1 ? 0 : xs + 1;
        wsl = wsl == HC_NW - 1 ? 0 : wsl + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail pieces must not land on the scratch below
    __builtin_amdgcn_s_barrier();

#if defined(HC_STAMP)
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
#endif
    float* sred = reinterpret_cast<float*>(smem) + 8 * (32 * 192) / 2;       // [2][256], behind the eight gate tiles
    u16* tile_lds = smem + wave * (32 * 192);                               // this wave's [32][192] bf16 tile (12 KB)
    // fused pool pass: this wave's tile, its row registers and the loader.  Buffer q2 holds the 512-column block q2 of the
    // eight rows of ONE virtual wave (32 registers); it is refilled with the next virtual wave's rows as soon as the current
    // ones are accumulated, so every load has the other block's arithmetic to land under
    constexpr int PNQ = PQ > 0 ? PQ : 1;
    constexpr int PROWS = PQ > 0 ? MIL_POOL_TILE / 4 : 1;
    const int ptrow0 = row0 + 32 * wave_u;                                   // wave-uniform
    const bool plive = PQ > 0 && ptrow0 < R;                                 // R % 32 == 0 (host): a tile is whole or absent
    u16x8 pv[PNQ][PROWS];
    unsigned pmw[PNQ][PROWS];
    auto pool_load = [&](int q2, int vw) {                                   // q2 compile-time, vw wave-uniform
#pragma unroll
        for (int i = 0; i < PROWS; ++i) {
            const int row = ptrow0 + vw + 4 * i;
            pv[q2][i] = *reinterpret_cast<const u16x8*>(x + (size_t)row * L + 8 * lane + 512 * q2);
            if (DROP) pmw[q2][i] = xbits[(size_t)row * (L >> 5) + (lane >> 2) + 16 * q2];
        }
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (PQ > 0 && q == 1 && plive) {                       // under the second half of the activation epilogue
#pragma unroll
            for (int q2 = 0; q2 < PNQ; ++q2) pool_load(q2, 0);
        }
        float part[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) part[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int d = 32 * (3 * wc + c) + r;
            const float bvd = bv[d], bud = bu[d], wd = wvec[d];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = fast_tanh(DROP ? fmaf(acc[q][c][0][i], xscale, bvd) : acc[q][c][0][i] + bvd);
                const float u = fast_sigmoid(DROP ? fmaf(acc[q][c][1][i], xscale, bud) : acc[q][c][1][i] + bud);
                part[i] += wd * v * u;
                if (GMODE == 2) {
                    u16* t = tile_lds + mfma32_row(i, h) * 192 + 32 * c + r;
                    t[0] = __builtin_bit_cast(u16, (__bf16)v);
                    t[96] = __builtin_bit_cast(u16, (__bf16)u);
                } else if (GMODE == 1) {
                    const int gr = row0 + 64 * wr + 32 * q + mfma32_row(i, h);
                    if (gr < R) {
                        gates[(size_t)gr * HB_NG + d] = v;
                        gates[(size_t)gr * HB_NG + 192 + d] = u;
                    }
                }
            }
        }
        if (GMODE == 2) {
            // bf16 gates: the wave's 32 x 192 tile leaves through its own LDS scratch as whole 16-byte chunks (same wave
            // wrote what it reads: the LDS operations of a wave complete in order)
            const int row_base = row0 + 64 * wr + 32 * q;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                const int k = lane + 64 * n, row = k / 24, ch = k % 24;          // 24 chunks of 8 columns per row
                const u16x8 vv = *reinterpret_cast<const u16x8*>(tile_lds + row * 192 + 8 * ch);
                if (row_base + row < R)
                    *reinterpret_cast<u16x8*>(gates16 + (size_t)(row_base + row) * HB_NG + (ch / 12) * 192 + 96 * wc + 8 * (ch % 12)) = vv;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = half_sum_lane31(part[i]);
            if (r == 31) sred[wc * HC_TM + 64 * wr + 32 * q + mfma32_row(i, h)] = v;
        }
    }
    __syncthreads();
    float* sc_lds = sred + 2 * HC_TM;                  // [256] final scores (fused pool pass)
    if (tid < HC_TM) {
        const int gr = row0 + tid;
        const float sc = sred[tid] + sred[HC_TM + tid] + battn[0];
        if (gr < R) scores[gr] = sc;
        if (PQ > 0) sc_lds[tid] = gr < R ? sc : -INFINITY;
    }
    if (PQ > 0) {
        __syncthreads();
        // tile weights, as wave 0 of k_pool_partial_bf16 forms them
        const float s_ = lane < 32 ? sc_lds[32 * wave_u + lane] : -INFINITY;
        const float m_ = wave_allmax(s_);
        const float p_ = lane < 32 ? expf(s_ - m_) : 0.f;
        const float l_ = wave_allsum(p_);
        if (plive) {
            const int t = ptrow0 >> 5;
            const int bag_ = pool.tile_map[4 * t];
            // head rows as this tile's bag sees them (k_pool_partial_bf16's head_row), one 512-column block at a time
            auto head_rows = [&](int q2, float (*wf)[8]) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(pool.Wf + (size_t)c * L + 512 * q2 + 8 * lane);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(pool.Wf + (size_t)c * L + 512 * q2 + 8 * lane + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { wf[c][e] = DROP ? w0[e] * xscale : w0[e]; wf[c][4 + e] = DROP ? w1[e] * xscale : w1[e]; }
                    if (DROP) {
                        const unsigned mm = pool.mbits[(size_t)bag_ * (L >> 5) + 16 * q2 + (lane >> 2)] >> (8 * (lane & 3));
#pragma unroll
                        for (int e = 0; e < 8; ++e) wf[c][e] = ((mm >> e) & 1u) ? wf[c][e] * pool.mscale : 0.f;
                    }
                }
            };
            float tot[PNQ][8];
#pragma unroll
            for (int q2 = 0; q2 < PNQ; ++q2)
#pragma unroll
                for (int e = 0; e < 8; ++e) tot[q2][e] = 0.f;
#pragma unroll 1
            for (int vw = 0; vw < 4; ++vw) {
                float d16[16];
#pragma unroll
                for (int q2 = 0; q2 < PNQ; ++q2) {
                    float wf[2][8];
                    head_rows(q2, wf);
                    if (DROP) {
#pragma unroll
                        for (int i = 0; i < PROWS; ++i) pv[q2][i] = keep_bf16x8_b(pv[q2][i], pmw[q2][i] >> (8 * (lane & 3)));
                    }
                    float pacc[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) pacc[e] = 0.f;
#pragma unroll
                    for (int i = 0; i < PROWS; ++i) {
                        const float pl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_), vw + 4 * i));
                        const float p = DROP ? pl * xscale : pl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pacc[e] += p * bf16_to_f32(pv[q2][i][e]);
                    }
                    // by-product: d16[2 i + c] = sum over the row's columns, block q2 = 0 first (the stand-alone kernel's order)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int i = 0; i < PROWS; ++i) {
                            float d = q2 == 0 ? 0.f : d16[2 * i + c];
#pragma unroll
                            for (int e = 0; e < 8; ++e) d += bf16_to_f32(pv[q2][i][e]) * wf[c][e];
                            d16[2 * i + c] = d;
                        }
                    // this block's registers are free: the next virtual wave's rows of the same block
                    __builtin_amdgcn_sched_barrier(0);
                    if (vw + 1 < 4) pool_load(q2, vw + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    // the four virtual waves' sums in the stand-alone kernel's fold order: ((a0 + a1) + a2) + a3
#pragma unroll
                    for (int e = 0; e < 8; ++e) tot[q2][e] += pacc[e];
                }
                const float toth = wave_reduce16(d16, lane);
                const int k = wave_reduce16_index(lane), rr = vw + 4 * (k >> 1);
                if ((lane & 3) == 0) pool.hrow[(size_t)(ptrow0 + rr) * 2 + (k & 1)] = toth;
            }
            float* out = pool.partials + (size_t)t * L;
#pragma unroll
            for (int q2 = 0; q2 < PNQ; ++q2) {
                *reinterpret_cast<f32x4*>(out + 512 * q2 + 8 * lane) = f32x4{tot[q2][0], tot[q2][1], tot[q2][2], tot[q2][3]};
                *reinterpret_cast<f32x4*>(out + 512 * q2 + 8 * lane + 4) = f32x4{tot[q2][4], tot[q2][5], tot[q2][6], tot[q2][7]};
            }
            if (lane == 0) {
                float* ml = pool.partials + (size_t)pool.T * L + 2 * t;
                ml[0] = m_;
                ml[1] = l_;
            }
        }
    }
#if defined(HC_STAMP)
    if (tid == 0 && GMODE == 2) {      // diagnostic build: (loop cycles, loop 100 MHz ticks, epilogue cycles, start tick) per workgroup
        const unsigned long long st2 = __builtin_amdgcn_s_memtime();
        float* dbg = reinterpret_cast<float*>(gates16) + 4 * blockIdx.x;
        dbg[0] = (float)(st1 - st0); dbg[1] = (float)(sr1 - sr0); dbg[2] = (float)(st2 - st1); dbg[3] = (float)(sr0 & 0xffffff);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------- pool stages, bf16 x
// Lane l owns the 8 columns 8l + 512q of a row (16-byte loads); NQ = L / 512.
template <int NQ, bool DROP, bool NT>     // DROP: train mode (keep-bit tensors); the eval instantiation carries none of it.  NT: as k_pool_partial
__global__ __launch_bounds__(256) void k_pool_partial_bf16(const u16* __restrict__ x, const float* __restrict__ scores,
                                                           const int32_t* __restrict__ tile_map,
                                                           float* __restrict__ partials, int L,
                                                           const float* __restrict__ Wf, int C, float* __restrict__ hrow,
                                                           const uint32_t* __restrict__ xbits, float xscale,
                                                           const uint32_t* __restrict__ mbits, float mscale) {
    // (Round 4, measured and dropped: taking the tiles LAST FIRST so that a cache-sized x - 256 MiB at config 5, just streamed
    // front to back by the gate forward - is met where the Infinity Cache still holds it: 58.8 us against 51.0 us for the
    // front-to-back nontemporal sweep, and the tail launch behind it 18.9 against 14.2 us.)
    // train mode: xbits [R][L/32] keep bits of the patch dropout - the DROPPED x is what gets pooled (ABMIL.py:49,59):
    // dropped elements are zeroed right after the load, the 1/(1-p) goes into the tile weights; mbits [B][L/32] = the
    // head's Dropout(.25) folded into the head rows of the by-product (as k_pool_partial)
    __shared__ float p_lds[MIL_POOL_TILE];
    __shared__ float ml_lds[2];
    __shared__ __attribute__((aligned(16))) float red[3 * NQ * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int row0 = tile_map[4 * t + 1], nrows = tile_map[4 * t + 2];
    // the tile's rows are requested FIRST: they depend on nothing, and their trip from HBM then runs under the softmax of
    // the tile's scores (wave 0) and the barrier behind it
    u16x8 v[MIL_POOL_TILE / 4][NQ];
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const int rr = max(min(wave + 4 * i, nrows - 1), 0);
        const u16* xr = x + (size_t)(row0 + rr) * L + 8 * lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            v[i][q] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(xr + 512 * q)) : *reinterpret_cast<const u16x8*>(xr + 512 * q);
    }
    // the keep words of those rows travel with them (behind the barrier they were a second, exposed round trip per tile:
    // 51 -> 75 us for the pass at 32 x 4096 x 1024)
    unsigned mw[MIL_POOL_TILE / 4][NQ];
    if (DROP && xbits != nullptr) {
#pragma unroll
        for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
            const int rr = max(min(wave + 4 * i, nrows - 1), 0);
            const uint32_t* mr = xbits + (size_t)(row0 + rr) * (L >> 5) + (lane >> 2);
#pragma unroll
            for (int q = 0; q < NQ; ++q) mw[i][q] = mr[16 * q];
        }
    }
    if (wave == 0) {
        const float s = lane < nrows ? scores[row0 + lane] : -INFINITY;
        const float m = wave_allmax(s);
        const float p = lane < nrows ? expf(s - m) : 0.f;
        const float l = wave_allsum(p);
        if (lane < MIL_POOL_TILE) p_lds[lane] = p;
        if (lane == 0) { ml_lds[0] = m; ml_lds[1] = l; }
    }
    __syncthreads();
    float acc[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
    if (DROP && xbits != nullptr) {
#pragma unroll
        for (int i = 0; i < MIL_POOL_TILE / 4; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[i][q] = keep_bf16x8_b(v[i][q], mw[i][q] >> (8 * (lane & 3)));
    }
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const float p = DROP ? p_lds[wave + 4 * i] * xscale : p_lds[wave + 4 * i];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[q][e] += p * bf16_to_f32(v[i][q][e]);
    }
    const int bag_ = tile_map[4 * t];
    auto head_row = [&](int c, float (*wf)[8]) {          // Wf[c] (x keep_M x mscale x xscale) of this lane's columns
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wf + (size_t)c * L + 512 * q + 8 * lane);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wf + (size_t)c * L + 512 * q + 8 * lane + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { wf[q][e] = DROP ? w0[e] * xscale : w0[e]; wf[q][4 + e] = DROP ? w1[e] * xscale : w1[e]; }
            if (DROP && mbits != nullptr) {
                const unsigned mm = mbits[(size_t)bag_ * (L >> 5) + 16 * q + (lane >> 2)] >> (8 * (lane & 3));
#pragma unroll
                for (int e = 0; e < 8; ++e) wf[q][e] = ((mm >> e) & 1u) ? wf[q][e] * mscale : 0.f;
            }
        }
    };
    if (Wf != nullptr && C == 2) {
        float d16[16];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float wf[NQ][8];
            head_row(c, wf);
#pragma unroll
            for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += bf16_to_f32(v[i][q][e]) * wf[q][e];
                d16[2 * i + c] = d;
            }
        }
        const float tot = wave_reduce16(d16, lane);
        const int k = wave_reduce16_index(lane), rr = wave + 4 * (k >> 1);
        if ((lane & 3) == 0 && rr < nrows) hrow[(size_t)(row0 + rr) * 2 + (k & 1)] = tot;
    } else if (Wf != nullptr) {
        // by-product (as in k_pool_partial): h[row][c] = x_row . Wf[c], so the backward needs no second pass over x
        const int nrows_ = nrows;
        for (int c = 0; c < C; ++c) {
            float wf[NQ][8];
            head_row(c, wf);
#pragma unroll
            for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += bf16_to_f32(v[i][q][e]) * wf[q][e];
                d = wave_allsum(d);
                const int rr = wave + 4 * i;
                if (lane == 0 && rr < nrows_) hrow[(size_t)(row0 + rr) * C + c] = d;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[((wave - 1) * NQ + q) * 512 + 8 * lane + e] = acc[q][e];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partials + (size_t)t * L;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = acc[q][e];
#pragma unroll
                for (int w = 0; w < 3; ++w) s += red[(w * NQ + q) * 512 + 8 * lane + e];
                out[512 * q + 8 * lane + e] = s;
            }
        if (lane == 0) {
            float* ml = partials + (size_t)gridDim.x * L + 2 * t;
            ml[0] = ml_lds[0];
            ml[1] = ml_lds[1];
        }
    }
}

template <int NQ>
__global__ __launch_bounds__(256) void k_pool_bwd_ds_bf16(const u16* __restrict__ x, const float* __restrict__ scores,
                                                          const float* __restrict__ lse, const float* __restrict__ dM,
                                                          const float* __restrict__ cdot,
                                                          const int32_t* __restrict__ tile_map, float* __restrict__ ds,
                                                          int L, const uint32_t* __restrict__ xbits, float xscale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int bag = tile_map[4 * t], row0 = tile_map[4 * t + 1], nrows = tile_map[4 * t + 2];
    float g[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) g[q][e] = dM[(size_t)bag * L + 512 * q + 8 * lane + e];
    const float lse_b = lse[bag], c_b = cdot[bag];
    u16x8 v[MIL_POOL_TILE / 4][NQ];
    float sc[MIL_POOL_TILE / 4];
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const size_t row = (size_t)(row0 + min(wave + 4 * i, nrows - 1));
        const u16* xr = x + row * L + 8 * lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[i][q] = *reinterpret_cast<const u16x8*>(xr + 512 * q);
        if (xbits != nullptr) {
            const uint32_t* mr = xbits + row * (L >> 5) + (lane >> 2);
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[i][q] = keep_bf16x8(v[i][q], mr[16 * q] >> (8 * (lane & 3)));
        }
        sc[i] = scores[row];
    }
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const int rr = wave + 4 * i;
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) dot += bf16_to_f32(v[i][q][e]) * g[q][e];
        dot = wave_allsum(dot);
        if (rr < nrows && lane == 0) ds[row0 + rr] = expf(sc[i] - lse_b) * (dot * xscale - c_b);
    }
}

// ---------------------------------------------------------------------------------------------------- gate dW, bf16 MFMA
// dW[gi][j] = sum_rows dPre[row][gi] x[row][j] with both operands rounded to bf16 and fp32 accumulation
// (v_mfma_f32_32x32x16_bf16).  Same split-K / permuted-gate-index scheme and the same fp32 partial buffers and
// reduce kernel as k_gate_bwd_dw.  Both operands are k-major in memory ([row][gi], [row][j]), so their LDS images stay
// row-major and the fragments come from the hardware transpose read ds_read_b64_tr_b16: the lane with column r and half
// h gets 4 consecutive rows of its column per read (probed on gfx950: lane l <- column (l & 31), rows q = 0..3 of the
// block whose row q / columns 4p..4p+3 were addressed by lane 4q + p of its 16-lane group).
// Workgroup 256 threads, output tile 128 (gi) x 256 (j): wave (wi, wj) = 64 x 128 = 2 x 4 MFMA tiles, so one staged
// (and VALU-built) dPre slice feeds twice the MFMAs of a 128-wide tile.  Row stride 160 bf16 = 320 B: the four rows of
// a transpose read start 80 words apart -> banks 0/16/32/48, conflict-free.
// waves_per_eu(2, 2): left alone hipcc takes 134 VGPRs + 128 AGPRs = one wave per SIMD, i.e. ONE workgroup per CU and
// two rounds for the 504-workgroup grid; capped at 256 registers it allocates without spills and both rounds are resident.
// Measured at config 5 (R = 131 072, L = 1024; tools/kbench_dw16.py): 230 us with the reduce.  Ablations: MFMAs removed
// 192 us, loads removed 125 us, both register-staged operands prefetched two slices ahead (needs the bias sums dropped to
// avoid spills) 225 us / 159 us without MFMAs.  PMC: 535 MB of HBM reads (minimum 470), 13.2 M L2 requests = 1.7 GB.  So
// neither HBM nor prefetch depth: per slice the CU moves 96 KB of transpose reads + 48 KB of staging writes through LDS
// (~1400 clk at the LDS rates) against 1024 MFMA clk, with the fragment reads of each k-step exposed (no register room
// to double-buffer 6 fragments next to 128 accumulators at two waves per SIMD).
#define WB_S 160
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ u16x8 tr_frag(const u16* img, int row, int col) {
    // 8 consecutive rows [row, row + 8) of column (col + lane column), as one 32x32x16 operand fragment
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + row * WB_S + col));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (row + 4) * WB_S + col));
    u16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__device__ __forceinline__ ushort4 pack_bf16x4(const f32x4 v) {
    ushort4 o;
    o.x = __builtin_bit_cast(u16, (__bf16)v[0]);
    o.y = __builtin_bit_cast(u16, (__bf16)v[1]);
    o.z = __builtin_bit_cast(u16, (__bf16)v[2]);
    o.w = __builtin_bit_cast(u16, (__bf16)v[3]);
    return o;
}

// Round 2 structure: ONE 256-thread workgroup per CU (252 of them: 21 row chunks x 12 output tiles), i.e. one wave per
// SIMD with the whole 512-entry register file, and
//   * 64-row slices (four k-steps of 16 rows) - half the barriers, and half as many post-barrier bubbles in which a lone
//     wave has no partner to cover the LDS latency of its first fragments;
//   * fragments double-buffered in registers: the twelve transpose reads of k-step ks + 1 are issued before the eight
//     MFMAs of k-step ks (round 1 read, waited, multiplied);
//   * the staging of the next slice (8 x pieces, 4 gate pieces per thread) cut into twelve parts laid between MFMA pairs
//     (the bf16 MFMA hides VALU and LDS-write issue, unlike the f32 one).
// Round 1 ran 2 x 256 threads per CU on 32-row slices: 200 us + reduce at config 5.
#define WB_BKR 64
#if !defined(WB_SETS)
#define WB_SETS 1      // register sets of the staging prefetch; 2 = loads two slices ahead: measured, no change (see the kernel)
#endif
template <bool DROP>      // DROP: train mode, x read through the keep bits (templated: the eval kernel is the one tuned above)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_gate_bwd_dw_bf16(const u16* __restrict__ x, const u16* __restrict__ gates,
                                                          const float* __restrict__ ds, const float* __restrict__ wvec,
                                                          float* __restrict__ part, float* __restrict__ pbias, int R, int L,
                                                          int KC, int NJ, const uint32_t* __restrict__ xbits) {
    // xbits (train mode): keep bits of the patch dropout; x enters the product keep-masked, the 1 / (1 - p) is applied by
    // the reduce (wscale)
    __shared__ __attribute__((aligned(16))) u16 smem[2 * (WB_BKR * WB_S + WB_BKR * 2 * WB_S)];      // 2 x 60 KB
    // per stage: A image [64][160] (128 gi used), B image [64][320] (256 j used, two 160-wide panels of 128 j)
    constexpr int ASZ = WB_BKR * WB_S, BSZ = WB_BKR * 2 * WB_S;
    u16* ab = smem;
    u16* xb = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int h = lane >> 5;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;     // transpose-read address roles
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int jt = bid % NJ, m = (bid / NJ) % 3, s = bid / (3 * NJ);
    const int j0 = jt * 256;
    const int rbeg = s * KC, rend = min(R, rbeg + KC);
    const int nslice = (rend - rbeg + WB_BKR - 1) / WB_BKR;

    // staging maps: x: 64 rows x 256 cols bf16 = 2048 16-byte chunks, 8 per thread: row (tid >> 5) + 8i, chunk tid & 31
    //               gates: rows (tid >> 4) + 16i (i < 4), d = 64m + 4 (tid & 15)
    const int xrow = tid >> 5, xc = tid & 31;
    const int arow = tid >> 4, ad4 = tid & 15;
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wvec + 64 * m + 4 * ad4);
    // WB_SETS register sets: slice q lives in set q % WB_SETS, its global loads are issued WB_SETS slices before its staging
    // parts need them.  Round 4 tested the hypothesis that the kernel sits at "one memory latency per 64-row slice" (1.7 us per
    // slice against 0.43 us of MFMA time) because with one set every load has exactly one slice time to land: with two sets
    // (234 VGPRs + 128 AGPRs, no spills, bit-identical results) the weight gradient of config 5 takes 168.7 us against
    // 169 - 172 us - no change.  It is not load latency; what the lone wave of a SIMD cannot hide is the ISSUE of its own
    // staging work (~240 VALU + 16 LDS writes + 20 loads per slice next to 32 MFMAs), as the ablations of round 2 said.
    u16x8 rx[WB_SETS][8];
    ushort4 hv[WB_SETS][4], hu[WB_SETS][4];
    float rds[WB_SETS][4], rmask[WB_SETS][4];
    f32x4 acc_bv = {0, 0, 0, 0}, acc_bu = {0, 0, 0, 0}, acc_w = {0, 0, 0, 0};
    float acc_ds = 0.f;

    unsigned rxm[WB_SETS][DROP ? 8 : 1];
    auto xload = [&](int z, int i, int rs) {
        const int gr = min(rs + xrow + 8 * i, rend - 1);
#if !defined(WB_ABL_NOX)       // ablation builds (tools/build_variants.sh + tools/kbench_dw16.py): times only, results meaningless
        rx[z][i] = *reinterpret_cast<const u16x8*>(x + (size_t)gr * L + j0 + 8 * xc);
#else
        asm volatile("" : "+v"(rx[z][i]) : "v"(gr));
#endif
        if (DROP) rxm[z][i] = xbits[(size_t)gr * (L >> 5) + ((j0 + 8 * xc) >> 5)];
    };
    auto xwrite = [&](int z, int i, int buf) {
        // columns 0..127 -> panel 0, 128..255 -> panel 1 (each panel is its own 160-stride image)
        u16* dst = xb + buf * BSZ + (xc >> 4) * ASZ + (xrow + 8 * i) * WB_S + 8 * (xc & 15);
#if !defined(WB_ABL_NOXW)
        *reinterpret_cast<u16x8*>(dst) = DROP ? keep_bf16x8_b(rx[z][i], rxm[z][i] >> (8 * (xc & 3))) : rx[z][i];
#else
        asm volatile("" ::"v"(rx[z][i]), "v"(dst));
#endif
    };
    auto aload = [&](int z, int i, int rs, bool live) {
        const int gr = rs + arow + 16 * i;
        const int gc = min(gr, rend - 1);
        const u16* gp = gates + (size_t)gc * HB_NG + 64 * m + 4 * ad4;
#if !defined(WB_ABL_NOG)
        hv[z][i] = *reinterpret_cast<const ushort4*>(gp);
        hu[z][i] = *reinterpret_cast<const ushort4*>(gp + 192);
        rds[z][i] = ds[gc];
#else
        asm volatile("" : "+v"(hv[z][i]), "+v"(hu[z][i]), "+v"(rds[z][i]) : "v"(gp));
#endif
        rmask[z][i] = (live && gr < rend) ? 1.f : 0.f;
    };
    auto awrite = [&](int z, int i, int buf) {
#if defined(WB_ABL_NOAW)
        asm volatile("" ::"v"(hv[z][i]), "v"(hu[z][i]), "v"(rds[z][i]), "v"(buf));
        return;
#endif
        const f32x4 v = {bf16_to_f32(hv[z][i].x), bf16_to_f32(hv[z][i].y), bf16_to_f32(hv[z][i].z), bf16_to_f32(hv[z][i].w)};
        const f32x4 u = {bf16_to_f32(hu[z][i].x), bf16_to_f32(hu[z][i].y), bf16_to_f32(hu[z][i].z), bf16_to_f32(hu[z][i].w)};
        const float dsv = rds[z][i] * rmask[z][i];
        const f32x4 a = (dsv * w4) * u;
        const f32x4 t = a * v;
        const f32x4 pv = a - t * v;               // ds w U (1 - V^2)
        const f32x4 pu = t - t * u;               // ds w V U (1 - U)
        u16* dst = ab + buf * ASZ + (arow + 16 * i) * WB_S + 4 * ad4;
        *reinterpret_cast<ushort4*>(dst) = pack_bf16x4(pv);
        *reinterpret_cast<ushort4*>(dst + 64) = pack_bf16x4(pu);
        acc_bv += pv;
        acc_bu += pu;
        acc_w += (dsv * v) * u;
        if (ad4 == 0) acc_ds += dsv;
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // slice q of the chunk: rows rbeg + 64 q ..; beyond the last slice the x rows are clamped and the dPre rows masked to zero
    auto slice_rows = [&](int q) { return rbeg + min(q, max(nslice - 1, 0)) * WB_BKR; };
    auto load_all = [&](int z, int q) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xload(z, i, slice_rows(q));
#pragma unroll
        for (int i = 0; i < 4; ++i) aload(z, i, slice_rows(q), q < nslice);
    };
    if (nslice > 0) {
        load_all(0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) xwrite(0, i, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) awrite(0, i, 0);
        // slice 1 -> set 1 % WB_SETS, slice 2 -> set 2 % WB_SETS (one set: slice 1 only, as before)
        load_all(1 % WB_SETS, 1);
        if (WB_SETS > 1) load_all(0, 2);
    }
    __syncthreads();
    // fragment column offsets inside an image: wave tile + 16-lane group + 4 p
    const int acol = 64 * wi + 16 * tg + 4 * tp;                 // + 32 a
    const int bcol = 16 * tg + 4 * tp;                           // + 32 b inside panel wj
    // one slice: multiply stage `buf` (slice sl) while slice sl + 1 goes from register set ZW to the other stage and that
    // set is reloaded with slice sl + 1 + WB_SETS
    auto do_slice = [&](int sl, auto zw_c) {
        constexpr int ZW = decltype(zw_c)::value;
        const int buf = sl & 1;
        const int qn = sl + 1 + WB_SETS;                         // the slice the freed registers are reloaded with
        const bool liven = qn < nslice;
        const int rsn = slice_rows(qn);
        const u16* ai = ab + buf * ASZ;
        const u16* bi = xb + buf * BSZ + wj * ASZ;
        u16x8 fa[2][2], fb[2][4];                                // [register set][tile]
        auto frags = [&](int ks, int q) {
            const int row = 16 * ks + 8 * h + tq;
#if !defined(WB_ABL_NOFRAG)
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[q][a] = tr_frag(ai, row, acol + 32 * a);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[q][b] = tr_frag(bi, row, bcol + 32 * b);
#else
#pragma unroll
            for (int a = 0; a < 2; ++a) asm volatile("" : "+v"(fa[q][a]) : "v"(row), "v"(ai));
#pragma unroll
            for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(fb[q][b]) : "v"(bi));
#endif
        };
        // twelve staging parts of the next slice, three per k-step
        auto stage = [&](int p) {
            if (p < 8) { xwrite(ZW, p, buf ^ 1); xload(ZW, p, rsn); }
            else { awrite(ZW, p - 8, buf ^ 1); aload(ZW, p - 8, rsn, liven); }
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < WB_BKR / 16; ++ks) {
            const int q = ks & 1;
            if (ks + 1 < WB_BKR / 16) frags(ks + 1, q ^ 1);      // next k-step's fragments fly under this k-step's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#if !defined(WB_ABL_NOMFMA)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[q][a]),
                                                                        __builtin_bit_cast(bf16x8, fb[q][b]), acc[a][b], 0, 0, 0);
#else
                    asm volatile("" : "+v"(acc[a][b]) : "v"(fa[q][a]), "v"(fb[q][b]));
#endif
                    const int g = a * 4 + b;                     // 8 MFMAs per k-step: a staging part behind #1, #3, #5
                    if (g == 1 || g == 3 || g == 5) {
                        __builtin_amdgcn_sched_barrier(0);
                        stage(3 * ks + (g >> 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        // (round 4: __syncthreads() here compiles to s_waitcnt lgkmcnt(0) + s_barrier on gfx950 - it does NOT drain vmcnt, so the
        // prefetched loads stay in flight across it; a hand-written LDS-only barrier produced the same code)
        __syncthreads();
    };
    {
        using Z0 = std::integral_constant<int, 0>;
        using Z1 = std::integral_constant<int, 1 % WB_SETS>;
        int sl = 0;
        for (; sl + 1 < nslice; sl += 2) {                       // slice sl + 1 sits in set (sl + 1) % WB_SETS
            do_slice(sl, Z1{});
            do_slice(sl + 1, Z0{});
        }
        if (sl < nslice) do_slice(sl, Z1{});
    }

    // partial tile -> part[s][128m + 64wi + 32a + row][j0 + 128wj + 32b + r]
    const int r = lane & 31;
    float* pt = part + ((size_t)s * HB_NG + 128 * m + 64 * wi) * L + j0 + 128 * wj + r;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) pt[(size_t)(32 * a + mfma32_row(i, h)) * L + 32 * b] = acc[a][b][i];

    if (jt == 0) {
        float* redf = reinterpret_cast<float*>(smem);   // [16 row groups][3][64]
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            redf[(arow * 3 + 0) * 64 + 4 * ad4 + e] = acc_bv[e];
            redf[(arow * 3 + 1) * 64 + 4 * ad4 + e] = acc_bu[e];
            redf[(arow * 3 + 2) * 64 + 4 * ad4 + e] = acc_w[e];
        }
        __syncthreads();
        if (tid < 192) {
            const int which = tid / 64, d = tid % 64;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) v += redf[(g * 3 + which) * 64 + d];
            pbias[((size_t)s * 4 + which) * 192 + 64 * m + d] = v;
        }
        if (m == 0) {
            __syncthreads();
            redf[tid] = acc_ds;
            __syncthreads();
            if (tid == 0) {
                float v = 0.f;
                for (int g = 0; g < 256; g += 16) v += redf[g];
                pbias[((size_t)s * 4 + 3) * 192] = v;
            }
        }
    }
}



// Round 4: the same product on EIGHT waves (two per SIMD) and a 128 (gi) x 512 (j) tile, 32-row slices.  What bounds the
// four-wave kernel above is neither HBM nor load latency (two-set prefetch: no change) nor the matrix pipe (MFMA-busy 0.29) but
// the lone in-order wave of a SIMD: every LDS / memory wait of its staging work and every fragment-read latency is exposed,
// the measured slice takes 2 900 cycles for 1 024 cycles of MFMAs.  Here a SIMD carries two waves that cover each other's
// stalls (the forward's structure), each wave keeps the 2 x 4 tile (128 accumulators, 0.75 fragment reads per MFMA), and the
// doubled j width halves the gate re-reads (2 j tiles instead of 4) and the dPre arithmetic per flop of a workgroup:
// per 32-row slice a thread stages 4 x 16 bytes of x and ONE (V, U) float4 pair.  Price: 3 x 2 output tiles -> 42 row chunks
// instead of 21, i.e. twice the split-K partials (66 MB at config 5) for the fold to read.
#define W8_BKR 32
template <bool DROP>
__global__ __launch_bounds__(512) void k_gate_bwd_dw_bf16_w8(const u16* __restrict__ x, const u16* __restrict__ gates,
                                                             const float* __restrict__ ds, const float* __restrict__ wvec,
                                                             float* __restrict__ part, float* __restrict__ pbias, int R, int L,
                                                             int KC, int NJ, const uint32_t* __restrict__ xbits) {
    constexpr int ASZ = W8_BKR * WB_S, BSZ = 4 * ASZ;            // per stage: A image [32][160], B image 4 panels of [32][160]
    __shared__ __attribute__((aligned(16))) u16 smem[2 * (ASZ + BSZ)];      // 100 KB
    u16* ab = smem;
    u16* xb = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3;
    const int h = lane >> 5;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int jt = bid % NJ, m = (bid / NJ) % 3, s = bid / (3 * NJ);
    const int j0 = jt * 512;
    const int rbeg = s * KC, rend = min(R, rbeg + KC);
    const int nslice = (rend - rbeg + W8_BKR - 1) / W8_BKR;

    // staging maps: x: 32 rows x 512 cols bf16 = 2048 16-byte chunks, 4 per thread: row (tid >> 6) + 8 i, chunk tid & 63
    //               gates: row tid >> 4 (0 .. 31), d = 64 m + 4 (tid & 15): one (V, U) quad pair per thread and slice
    const int xrow = tid >> 6, xc = tid & 63;
    const int arow = tid >> 4, ad4 = tid & 15;
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wvec + 64 * m + 4 * ad4);
    u16x8 rx[4];
    unsigned rxm[DROP ? 4 : 1];
    ushort4 hv, hu;
    float rds, rmask;
    f32x4 acc_bv = {0, 0, 0, 0}, acc_bu = {0, 0, 0, 0}, acc_w = {0, 0, 0, 0};
    float acc_ds = 0.f;
    auto slice_rows = [&](int q) { return rbeg + min(q, max(nslice - 1, 0)) * W8_BKR; };
    auto xload = [&](int i, int rs) {
        const int gr = min(rs + xrow + 8 * i, rend - 1);
        rx[i] = *reinterpret_cast<const u16x8*>(x + (size_t)gr * L + j0 + 8 * xc);
        if (DROP) rxm[i] = xbits[(size_t)gr * (L >> 5) + ((j0 + 8 * xc) >> 5)];
    };
    auto xwrite = [&](int i, int buf) {          // columns 128 p .. 128 p + 127 -> panel p (each panel its own 160-stride image)
        u16* dst = xb + buf * BSZ + (xc >> 4) * ASZ + (xrow + 8 * i) * WB_S + 8 * (xc & 15);
        *reinterpret_cast<u16x8*>(dst) = DROP ? keep_bf16x8_b(rx[i], rxm[i] >> (8 * (xc & 3))) : rx[i];
    };
    auto aload = [&](int rs, bool live) {
        const int gr = rs + arow;
        const int gc = min(gr, rend - 1);
        const u16* gp = gates + (size_t)gc * HB_NG + 64 * m + 4 * ad4;
        hv = *reinterpret_cast<const ushort4*>(gp);
        hu = *reinterpret_cast<const ushort4*>(gp + 192);
        rds = ds[gc];
        rmask = (live && gr < rend) ? 1.f : 0.f;
    };
    auto awrite = [&](int buf) {
        const f32x4 v = {bf16_to_f32(hv.x), bf16_to_f32(hv.y), bf16_to_f32(hv.z), bf16_to_f32(hv.w)};
        const f32x4 u = {bf16_to_f32(hu.x), bf16_to_f32(hu.y), bf16_to_f32(hu.z), bf16_to_f32(hu.w)};
        const float dsv = rds * rmask;
        const f32x4 a = (dsv * w4) * u;
        const f32x4 t = a * v;
        const f32x4 pv = a - t * v;               // ds w U (1 - V^2)
        const f32x4 pu = t - t * u;               // ds w V U (1 - U)
        u16* dst = ab + buf * ASZ + arow * WB_S + 4 * ad4;
        *reinterpret_cast<ushort4*>(dst) = pack_bf16x4(pv);
        *reinterpret_cast<ushort4*>(dst + 64) = pack_bf16x4(pu);
        acc_bv += pv;
        acc_bu += pu;
        acc_w += (dsv * v) * u;
        if (ad4 == 0) acc_ds += dsv;
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    if (nslice > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xload(i, slice_rows(0));
        aload(slice_rows(0), true);
#pragma unroll
        for (int i = 0; i < 4; ++i) xwrite(i, 0);
        awrite(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) xload(i, slice_rows(1));
        aload(slice_rows(1), nslice > 1);
    }
    __syncthreads();
    const int acol = 64 * wi + 16 * tg + 4 * tp;                 // + 32 a
    const int bcol = 16 * tg + 4 * tp;                           // + 32 b inside panel wj
    for (int sl = 0; sl < nslice; ++sl) {
        const int buf = sl & 1;
        const bool live2 = sl + 2 < nslice;
        const int rs2 = slice_rows(sl + 2);
        const u16* ai = ab + buf * ASZ;
        const u16* bi = xb + buf * BSZ + wj * ASZ;
        u16x8 fa[2][2], fb[2][4];                                // [register set][tile]
        auto frags = [&](int ks, int q) {
            const int row = 16 * ks + 8 * h + tq;
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[q][a] = tr_frag(ai, row, acol + 32 * a);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[q][b] = tr_frag(bi, row, bcol + 32 * b);
        };
        // five staging parts of the next slice (four x chunks, one gate pair) between the sixteen MFMAs of this one
        auto stage = [&](int p) {
            if (p < 4) { xwrite(p, buf ^ 1); xload(p, rs2); }
            else { awrite(buf ^ 1); aload(rs2, live2); }
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < W8_BKR / 16; ++ks) {
            const int q = ks & 1;
            if (ks + 1 < W8_BKR / 16) frags(ks + 1, q ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[q][a]),
                                                                        __builtin_bit_cast(bf16x8, fb[q][b]), acc[a][b], 0, 0, 0);
                    const int g = a * 4 + b;
                    const int p = ks == 0 ? (g == 1 ? 0 : g == 3 ? 1 : g == 5 ? 2 : -1) : (g == 1 ? 3 : g == 3 ? 4 : -1);
                    if (p >= 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        stage(p);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        __syncthreads();
    }

    // partial tile -> part[s][128m + 64wi + 32a + row][j0 + 128wj + 32b + r]
    const int r = lane & 31;
    float* pt = part + ((size_t)s * HB_NG + 128 * m + 64 * wi) * L + j0 + 128 * wj + r;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) pt[(size_t)(32 * a + mfma32_row(i, h)) * L + 32 * b] = acc[a][b][i];

    if (jt == 0) {
        float* redf = reinterpret_cast<float*>(smem);   // [32 row groups][3][64]
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            redf[(arow * 3 + 0) * 64 + 4 * ad4 + e] = acc_bv[e];
            redf[(arow * 3 + 1) * 64 + 4 * ad4 + e] = acc_bu[e];
            redf[(arow * 3 + 2) * 64 + 4 * ad4 + e] = acc_w[e];
        }
        __syncthreads();
        if (tid < 192) {
            const int which = tid / 64, d = tid % 64;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 32; ++g) v += redf[(g * 3 + which) * 64 + d];
            pbias[((size_t)s * 4 + which) * 192 + 64 * m + d] = v;
        }
        if (m == 0) {
            __syncthreads();
            redf[tid] = acc_ds;
            __syncthreads();
            if (tid == 0) {
                float v = 0.f;
                for (int g = 0; g < 512; g += 16) v += redf[g];
                pbias[((size_t)s * 4 + 3) * 192] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- host entry points
extern "C" int mil_cast_bf16(const float* src, uint16_t* dst, size_t n, void* stream) {
    if (!src || !dst) return MIL_EINVAL;
    if (n == 0) return MIL_OK;
    hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gate_scores_fwd_bf16(const uint16_t* x, const uint16_t* Wv, const float* bv, const uint16_t* Wu,
                                        const float* bu, const float* w, const float* b, float* scores, float* gates,
                                        int R, int L, int D, uint16_t* gates16, const uint32_t* xbits, float xscale,
                                        void* stream) {
    if (!x || !Wv || !bv || !Wu || !bu || !w || !b || !scores) return MIL_EINVAL;
    if (D != MIL_GATE_D || L <= 0 || (L % HB_BK) != 0 || R < 0) return MIL_EINVAL;
    if (R == 0) return MIL_OK;
    // 256-row tiles with the three-stage pipeline once they fill the chip (train mode too: the keep words ride the kernel's
    // own LDS-DMA stream); the 128-row kernel for small R
    if (R >= HC_TM * MIL_NUM_CU && (xbits == nullptr || (L % 128) == 0)) {
        const dim3 grid((R + HC_TM - 1) / HC_TM);
        const hipStream_t st_ = (hipStream_t)stream;
#define HC_LAUNCH(G, D) hipLaunchKernelGGL((k_gate_fwd_bf16_deep<G, D>), grid, dim3(512), 0, st_, x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, gates16, xbits, xbits ? xscale : 1.0f)
        if (xbits) {
            if (gates16) HC_LAUNCH(2, true); else if (gates) HC_LAUNCH(1, true); else HC_LAUNCH(0, true);
        } else {
            if (gates16) HC_LAUNCH(2, false); else if (gates) HC_LAUNCH(1, false); else HC_LAUNCH(0, false);
        }
#undef HC_LAUNCH
    }
    else
        hipLaunchKernelGGL(k_gate_fwd_bf16, dim3((R + HB_TM - 1) / HB_TM), dim3(512), 0, (hipStream_t)stream, x, Wv, bv, Wu,
                           bu, w, b, scores, gates, R, L, gates16, xbits, xbits ? xscale : 1.0f);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// Gate forward of the bf16 step with the pool partial pass in its epilogue (csrc/step.hip, MIL_STAGE_POOL_FUSED): *fused = 1
// when the deep kernel with PQ took it, else the plain forward ran and the caller launches the stand-alone pass.
int gate_fwd_bf16_with_pool(const uint16_t* x, const uint16_t* Wv, const float* bv, const uint16_t* Wu, const float* bu,
                            const float* w, const float* b, float* scores, uint16_t* gates16, int R, int L, const uint32_t* xbits,
                            float xscale, const int32_t* tile_map, int T, float* partials, const float* Wf, float* hrow,
                            const uint32_t* mbits, float mscale, int* fused, void* stream) {
    *fused = 0;
    const bool ok = x && gates16 && tile_map && partials && Wf && hrow && (L == 512 || L == 1024) && R >= HC_TM * MIL_NUM_CU &&
                    (R % 32) == 0 && (long)T * MIL_POOL_TILE == R && ((xbits != nullptr) == (mbits != nullptr));
    if (!ok)
        return mil_gate_scores_fwd_bf16(x, Wv, bv, Wu, bu, w, b, scores, nullptr, R, L, MIL_GATE_D, gates16, xbits, xscale, stream);
    if (!Wv || !bv || !Wu || !bu || !w || !b || !scores) return MIL_EINVAL;
    const GateFwdPool16 pool{tile_map, partials, T, Wf, hrow, mbits, mbits ? mscale : 1.0f};
    const dim3 grid((R + HC_TM - 1) / HC_TM);
    const hipStream_t st_ = (hipStream_t)stream;
#define HCP_LAUNCH(D, Q) hipLaunchKernelGGL((k_gate_fwd_bf16_deep<2, D, Q>), grid, dim3(512), 0, st_, x, Wv, bv, Wu, bu, w, b, scores, (float*)nullptr, R, L, gates16, xbits, xbits ? xscale : 1.0f, pool)
    if (xbits) { if (L == 512) HCP_LAUNCH(true, 1); else HCP_LAUNCH(true, 2); }
    else { if (L == 512) HCP_LAUNCH(false, 1); else HCP_LAUNCH(false, 2); }
#undef HCP_LAUNCH
    MIL_CHECK_LAUNCH();
    *fused = 1;
    return MIL_OK;
}

// eval / train instantiations: the eval kernels carry no trace of the keep-bit handling (templated, not branched: the
// branched form cost the eval-mode pool pass 16 us at config 5)
static void launch_pool_partial_bf16(const uint16_t* x, const float* scores, const int32_t* tile_map, int T, int L,
                                     float* partials, const float* Wf, int C, float* hrow, const uint32_t* xbits, float xscale,
                                     const uint32_t* mbits, float mscale, hipStream_t st) {
    const bool drop = xbits != nullptr || mbits != nullptr;
    const float xs = xbits ? xscale : 1.0f, ms = mbits ? mscale : 1.0f;
    const bool nt = (size_t)T * MIL_POOL_TILE * L * sizeof(uint16_t) > MIL_STREAM_BYTES;
#define POOL16(NQ_, D_, N_) hipLaunchKernelGGL((k_pool_partial_bf16<NQ_, D_, N_>), dim3(T), dim3(256), 0, st, x, scores, tile_map, partials, L, Wf, C, hrow, xbits, (D_) ? xs : 1.0f, mbits, (D_) ? ms : 1.0f)
    if (L == 512) {
        if (drop) { if (nt) POOL16(1, true, true); else POOL16(1, true, false); }
        else { if (nt) POOL16(1, false, true); else POOL16(1, false, false); }
    } else {
        if (drop) { if (nt) POOL16(2, true, true); else POOL16(2, true, false); }
        else { if (nt) POOL16(2, false, true); else POOL16(2, false, false); }
    }
#undef POOL16
}

extern "C" int mil_attn_pool_partial_bf16(const uint16_t* x, const float* scores, const int32_t* tile_map, int T, int L,
                                          float* partials, const uint32_t* xbits, float xscale, void* stream) {
    if (!x || !scores || !tile_map || !partials) return MIL_EINVAL;
    if (L <= 0 || (L % 512) != 0 || L > 1024 || T < 0) return MIL_EINVAL;
    if (T == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    launch_pool_partial_bf16(x, scores, tile_map, T, L, partials, nullptr, 0, nullptr, xbits, xscale, nullptr, 1.0f, st);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_attn_pool_partial_h_bf16(const uint16_t* x, const float* scores, const int32_t* tile_map, int T, int L,
                                            float* partials, const float* Wf, int C, float* hrow, const uint32_t* xbits,
                                            float xscale, const uint32_t* mbits, float mscale, void* stream) {
    if (!x || !scores || !tile_map || !partials || !Wf || !hrow) return MIL_EINVAL;
    if (L <= 0 || (L % 512) != 0 || L > 1024 || T < 0 || C <= 0 || C > 4) return MIL_EINVAL;
    if (T == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    launch_pool_partial_bf16(x, scores, tile_map, T, L, partials, Wf, C, hrow, xbits, xscale, mbits, mscale, st);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_attn_pool_bwd_bf16(const uint16_t* x, const float* scores, const float* lse, const float* dM,
                                      const float* cdot, const int32_t* tile_map, int T, int L, float* ds,
                                      const uint32_t* xbits, float xscale, void* stream) {
    if (!x || !scores || !lse || !dM || !cdot || !tile_map || !ds) return MIL_EINVAL;
    if (L <= 0 || (L % 512) != 0 || L > 1024 || T < 0) return MIL_EINVAL;
    if (T == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (L == 512) hipLaunchKernelGGL(k_pool_bwd_ds_bf16<1>, dim3(T), dim3(256), 0, st, x, scores, lse, dM, cdot, tile_map, ds, L, xbits, xbits ? xscale : 1.0f);
    else hipLaunchKernelGGL(k_pool_bwd_ds_bf16<2>, dim3(T), dim3(256), 0, st, x, scores, lse, dM, cdot, tile_map, ds, L, xbits, xbits ? xscale : 1.0f);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

static inline int split_plan_bf16(int R, int L, int* KC_out) {
    const int NJ = L / 256;
    int smax = MIL_NUM_CU / (3 * NJ);          // one 256-thread workgroup per CU (128 accumulator registers per lane)
    if (smax < 1) smax = 1;
    int kc = ((R + smax - 1) / smax + WB_BKR - 1) / WB_BKR * WB_BKR;
    if (kc < WB_BKR) kc = WB_BKR;
    *KC_out = kc;
    return (R + kc - 1) / kc;
}

// eight-wave kernel (128 x 512 tiles; L a multiple of 512): OPT-IN, MIL_DW16_W8=1.  Measured at config 5 on one box
// (tools/cfg5_quick.py, weight gradient + fold): eval 177.4 us against 170.9 us for the four-wave kernel, train 194.8 against
// 201.6 us - a tie, although every thread stages half as much and two waves per SIMD cover each other's stalls.  What the
// two kernels share is the wave tile (2 x 4: 0.75 transposing fragment reads per MFMA) - i.e. the LDS traffic per flop; that,
// not issue, latency or HBM, is what binds them (96 KB of ds_read_b64_tr_b16 + 48 KB of staging writes per 4.2 MFLOP).
static inline bool dw16_use_w8(int R, int L) {
    const char* e = getenv("MIL_DW16_W8");               // read per call: the A/B tests toggle it inside one process
    (void)R;
    return (L % 512) == 0 && e != nullptr && atoi(e) != 0;
}
static inline int split_plan_bf16_w8(int R, int L, int* KC_out) {
    const int NJ = L / 512;
    int smax = MIL_NUM_CU / (3 * NJ);
    if (smax < 1) smax = 1;
    int kc = ((R + smax - 1) / smax + 63) / 64 * 64;
    if (kc < 64) kc = 64;
    *KC_out = kc;
    return (R + kc - 1) / kc;
}
// weight-gradient launch of either form; returns the number of row chunks S (partials part[S][384][L], pbias[S][4][192])
static int launch_gate_bwd_dw16(const uint16_t* x, const uint16_t* gates, const float* ds, const float* w, float* workspace,
                                size_t workspace_floats, int R, int L, const uint32_t* xbits, hipStream_t st, int* S_out,
                                float** pbias_out) {
    int kc;
    const bool w8 = dw16_use_w8(R, L);
    const int S = w8 ? split_plan_bf16_w8(R, L, &kc) : split_plan_bf16(R, L, &kc);
    const size_t need = (size_t)S * HB_NG * L + (size_t)S * 4 * 192;
    if (workspace_floats < need) return MIL_ENOSPC;
    float* part = workspace;
    float* pbias = workspace + (size_t)S * HB_NG * L;
    if (w8) {
        const int NJ = L / 512;
        if (xbits != nullptr)
            hipLaunchKernelGGL(k_gate_bwd_dw_bf16_w8<true>, dim3(S * 3 * NJ), dim3(512), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits);
        else
            hipLaunchKernelGGL(k_gate_bwd_dw_bf16_w8<false>, dim3(S * 3 * NJ), dim3(512), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits);
    } else {
        const int NJ = L / 256;
        if (xbits != nullptr)
            hipLaunchKernelGGL(k_gate_bwd_dw_bf16<true>, dim3(S * 3 * NJ), dim3(256), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits);
        else
            hipLaunchKernelGGL(k_gate_bwd_dw_bf16<false>, dim3(S * 3 * NJ), dim3(256), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits);
    }
    *S_out = S;
    *pbias_out = pbias;
    return MIL_OK;
}

extern "C" size_t mil_gate_bwd_workspace_floats_bf16(int R, int L) {
    if (R <= 0 || L <= 0 || (L % 256) != 0) return 0;
    int kc;
    int S = split_plan_bf16(R, L, &kc);
    if ((L % 512) == 0) S = max(S, split_plan_bf16_w8(R, L, &kc));      // either kernel may run (dw16_use_w8)
    return (size_t)S * HB_NG * L + (size_t)S * 4 * 192;
}

// The bf16-storage step's weight gradient with its whole tail in the fold launch (internal: step.hip): the split-K fold also
// forms the head's parameter gradients and the loss (appended workgroups, as mil_gate_bwd_reduce_head) and, when the
// optimizer stage runs in the same call (param_flat != NULL: world size 1, nothing between gradient and update), applies
// Adam on the spot and refreshes the bf16 shadows Wv16 / Wu16 of the gate weights - round 2 ran k_head_bwd_params, k_adam
// and two k_cast_bf16 launches behind the fold (~23 us of a 0.38 ms step).
int gate_bwd_params_bf16_tail(const uint16_t* x, const uint16_t* gates, const float* ds, const float* w, int R, int L,
                              float* workspace, size_t workspace_floats, float* dWv, float* dbv, float* dWu, float* dbu,
                              float* dw, float* db, int accumulate, const uint32_t* xbits, float xscale, const float* dz,
                              const float* M, float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out,
                              float* param_flat, const float* grad_flat, size_t n_param, float* exp_avg, float* exp_avg_sq,
                              int step, const int* step_dev, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, uint16_t* Wv16, uint16_t* Wu16, void* stream, int* done) {
    if (!x || !gates || !ds || !w || !workspace || !dWv || !dbv || !dWu || !dbu || !dw || !db) return MIL_EINVAL;
    if (!dz || !M || !dWf || !dbf || B <= 0 || C <= 0 || C > 32 || (loss_bag && !loss_out)) return MIL_EINVAL;
    if (L <= 0 || (L % 256) != 0 || R <= 0) return MIL_EINVAL;
    AdamFuse ad{};
    if (param_flat != nullptr) {
        if (!grad_flat || !exp_avg || !exp_avg_sq || !Wv16 || !Wu16 || (step_dev == nullptr && step < 1)) return MIL_EINVAL;
        const float* outs[8] = {dWv, dbv, dWu, dbu, dw, db, dWf, dbf};
        const size_t lens[8] = {(size_t)192 * L, 192, (size_t)192 * L, 192, 192, 1, (size_t)C * L, (size_t)C};
        for (int i = 0; i < 8; ++i)
            if (outs[i] < grad_flat || outs[i] + lens[i] > grad_flat + n_param) return MIL_EINVAL;
        if (((dWv - grad_flat) | (dWu - grad_flat)) & 3) return MIL_EINVAL;
        if ((reinterpret_cast<uintptr_t>(param_flat) | reinterpret_cast<uintptr_t>(grad_flat) |
             reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
            return MIL_EINVAL;
        if (step_dev != nullptr) step = 1;
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        const double bc2 = 1.0 - pow((double)beta2, (double)step);
        ad = AdamFuse{param_flat, grad_flat, exp_avg, exp_avg_sq, (float)bc1, beta1, beta2, eps, weight_decay, grad_scale,
                      (float)sqrt(bc2), step_dev, lr, lr_dev, step_dev ? done : nullptr};
    }
    hipStream_t st = (hipStream_t)stream;
    int S;
    float* pbias;
    float* part = workspace;
    const int rc_dw = launch_gate_bwd_dw16(x, gates, ds, w, workspace, workspace_floats, R, L, xbits, st, &S, &pbias);
    if (rc_dw != MIL_OK) return rc_dw;
    MIL_CHECK_LAUNCH();
    const int nthreads = HB_NG * (L / 4) + GR_NB * (3 * 192 + 1);
    const int nred = (nthreads + 255) / 256, nhead = C * ((L + 63) / 64) + 1;
    const HeadBwdArgs head{dz, M, dWf, dbf, loss_bag, loss_out, B, L, C, accumulate};
    hipLaunchKernelGGL(k_gate_bwd_reduce, dim3(nred + nhead), dim3(256), 0, st, part, pbias, S, S, L, dWv, dbv, dWu, dbu, dw, db,
                       accumulate, xbits ? xscale : 1.0f, nred, head, ad, param_flat ? Wv16 : nullptr, param_flat ? Wu16 : nullptr);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gate_bwd_params_bf16(const uint16_t* x, const uint16_t* gates, const float* ds, const float* w, int R,
                                        int L, int D, float* workspace, size_t workspace_floats, float* dWv, float* dbv,
                                        float* dWu, float* dbu, float* dw, float* db, int accumulate,
                                        const uint32_t* xbits, float xscale, void* stream) {
    if (!x || !gates || !ds || !w || !workspace || !dWv || !dbv || !dWu || !dbu || !dw || !db) return MIL_EINVAL;
    if (D != MIL_GATE_D || L <= 0 || (L % 256) != 0 || R <= 0) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int S;
    float* pbias;
    float* part = workspace;
    const int rc_dw = launch_gate_bwd_dw16(x, gates, ds, w, workspace, workspace_floats, R, L, xbits, st, &S, &pbias);
    if (rc_dw != MIL_OK) return rc_dw;
    MIL_CHECK_LAUNCH();
    const int nthreads = HB_NG * (L / 4) + GR_NB * (3 * 192 + 1);
    hipLaunchKernelGGL(k_gate_bwd_reduce, dim3((nthreads + 255) / 256), dim3(256), 0, st, part, pbias, S, S, L, dWv, dbv, dWu,
                       dbu, dw, db, accumulate, xbits ? xscale : 1.0f);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
