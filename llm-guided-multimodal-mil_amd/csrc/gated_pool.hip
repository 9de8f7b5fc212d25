// K1: gated-attention MIL pooling (ABMIL) forward and backward for gfx950.
//
// Reference arithmetic: model/dim1/ABMIL.py:47-59 (forward), torch autograd of the same
// ops (backward).  Two regimes:
//   * gate GEMMs  x[R,L] . [Wv;Wu]^T -> [R,384] and its transpose-product in the backward are
//     dense fp32 contractions: v_mfma_f32_32x32x2_f32 (exact f32, 157 TFLOP/s roof).
//   * scores -> softmax over the bag -> A.x (and ds in the backward) stream x once: HBM-bound.
#include "mil_common.h"
#include "philox.h"
#include <type_traits>

// ================================================================================ K1a gate forward
// Workgroup: 512 threads = 8 waves, tile = 128 rows x all 384 gate columns, K-slices of 32, double-buffered in LDS.
//   wave (wr, wc): rows 32*wr..+31, d-chunks {3wc, 3wc+1, 3wc+2} for both V and U  -> 6 accumulators, so V_d and
//   U_d of a row land in the same lane and the gate product / score reduction never leave registers.
// k permutation: lane (r, h) reads 4 consecutive k = 8t+4h..+3 (one ds_read_b128) and feeds element j to the j-th of
// 4 MFMAs, so MFMA (t, j) contracts k in {8t+j, 8t+4+j}: A and B use the same map, the sum over a slice is complete.
// Staging: global -> LDS directly (global_load_lds_dwordx4: no staging VGPRs, no ds_write).  An LDS-DMA
// wave-instruction writes 64 x 16 B contiguously (8 rows of 128 B), so the image cannot be padded: rows are 32 words
// and the 16-byte chunk c of row `row` sits at chunk c ^ ((row >> 1) & 7) (applied to the per-lane SOURCE address,
// the same XOR on the fragment reads).  Over any 16-lane ds_read_b128 group (row >> 1) & 7 takes all 8 values for
// both row parities -> conflict-free (SQ_LDS_BANK_CONFLICT = 0).
// Schedule: the 8 DMA pieces of slice s+1 and the fragment reads of the next k-group are pinned between the MFMA
// groups of slice s (sched_barrier); one barrier per slice, in front of which hipcc drains the DMA (vmcnt(0)).
// Measured at 32 x 1024 x 512 (us): register-staged + padded image 109.4, this form 105.8, bit-identical results;
// main loop alone 92.5 (bare MFMA stream), +staging, +6 for the epilogue (fast activations; ocml tanhf/expf +4).
#define GF_TM 128
#define GF_BK 32
#define GF_NG 384

typedef __attribute__((address_space(3))) void lds_void;
// all-ones if bit `b` of m is set, else 0 (v_bfe_i32: a one-bit signed field)
__device__ __forceinline__ unsigned bitmask1(unsigned m, int b) { return (unsigned)__builtin_amdgcn_sbfe((int)m, b, 1); }
__device__ __forceinline__ float keep_if(float v, unsigned m, int b) {
    return __uint_as_float(__float_as_uint(v) & bitmask1(m, b));
}
// DROP: train mode (ABMIL.py:49).  xbits [R][L/32] keep bits (csrc/dropout.hip); a K-slice is 32 columns = ONE word per
// row, loaded one slice ahead next to the DMA pieces; the A fragment of lane (r, h) holds k = 8t + 4h + j, so the word is
// shifted by 4h once and element (t, j) is kept by bit 8t + j (two VALU per element, under the MFMAs).  The survivors'
// scale 1/(1-p) is applied to the accumulators in the epilogue.
template <bool DROP>
__global__ __launch_bounds__(512) void k_gate_fwd(const float* __restrict__ x, const float* __restrict__ Wv,
                                                       const float* __restrict__ bv, const float* __restrict__ Wu,
                                                       const float* __restrict__ bu, const float* __restrict__ wvec,
                                                       const float* __restrict__ battn, float* __restrict__ scores,
                                                       float* __restrict__ gates, int R, int L,
                                                       const uint32_t* __restrict__ xbits, float xscale) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (GF_TM + GF_NG) * 32];
    float* xs = smem;                      // [2][128][32]
    float* ws = smem + 2 * GF_TM * 32;     // [2][384][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * GF_TM;
    // DMA pieces of this wave: 2 x-pieces (8 rows each) and 6 W-pieces; lane -> (row in piece, physical chunk)
    const int prow = lane >> 3, pch = lane & 7;
    const float* gsrc[8];
    int ldst[8];                           // LDS float offset of the piece base (wave-uniform) inside one buffer
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < 2) {
            const int lr = (2 * wave + i) * 8 + prow;                         // row inside the 128-row tile
            const int gr = min(row0 + lr, R - 1);
            gsrc[i] = x + (size_t)gr * L + 4 * (pch ^ ((lr >> 1) & 7));
            ldst[i] = (2 * wave + i) * 8 * 32;
        } else {
            const int wrow = (6 * wave + (i - 2)) * 8 + prow;                 // 0..383
            gsrc[i] = (wrow < 192 ? Wv + (size_t)wrow * L : Wu + (size_t)(wrow - 192) * L) + 4 * (pch ^ ((wrow >> 1) & 7));
            ldst[i] = (6 * wave + (i - 2)) * 8 * 32;
        }
    }
    auto dma_piece = [&](int i, int buf, int k0) {
        float* dst = (i < 2 ? xs + buf * GF_TM * 32 : ws + buf * GF_NG * 32) + ldst[i];
        __builtin_amdgcn_global_load_lds(gsrc[i] + k0, (lds_void*)dst, 16, 0, 0);
    };
    f32x16 acc[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][u][i] = 0.f;

    const int nslice = L / GF_BK;
    // (starting each workgroup's K loop at a different slice, to de-correlate the L2 requests for the shared gate
    //  weights, measured no gain: 107.4 vs 106.2 us fp32, 172 vs 171 us bf16 - the slices stay in natural order)
    const uint32_t* mrow = nullptr;                    // keep bits of this lane's fragment row
    unsigned mnext = 0;
    if (DROP) {
        mrow = xbits + (size_t)min(row0 + 32 * wr + r, R - 1) * nslice;
        mnext = mrow[0];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(i, 0, 0);
    __syncthreads();                                   // hipcc drains the DMA (vmcnt(0)) in front of the barrier
    const int fx = (r >> 1) & 7;                       // swizzle term of this lane's fragment rows (row % 32 == r)
    for (int s = 0; s < nslice; ++s) {
        const int buf = s & 1;
        const int k1 = min(s + 1, nslice - 1) * GF_BK;
        unsigned mcur = 0;
        if (DROP) {
            mcur = mnext >> (4 * h);
            mnext = mrow[min(s + 1, nslice - 1)];
        }
        const float* xa = xs + (buf * GF_TM + 32 * wr + r) * 32;
        const float* wb = ws + (buf * GF_NG + 32 * 3 * wc + r) * 32;
        f32x4 a[2], b[2][3][2];
        auto frag_piece = [&](int t, int q, int p) {
            const int ch = 4 * ((2 * t + h) ^ fx);
            if (p == 0) {
                a[q] = *reinterpret_cast<const f32x4*>(xa + ch);
            } else {
                const int c = (p - 1) >> 1, u = (p - 1) & 1;
                b[q][c][u] = *reinterpret_cast<const f32x4*>(wb + (u * 192 + 32 * c) * 32 + ch);
            }
        };
#pragma unroll
        for (int p = 0; p < 7; ++p) frag_piece(0, 0, p);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = 4 * t + j;
                if (g < 8) dma_piece(g, buf ^ 1, k1);     // next slice, one DMA piece per MFMA group
                if (t < 3) {
                    frag_piece(t + 1, q ^ 1, 2 * j);
                    if (2 * j + 1 < 7) frag_piece(t + 1, q ^ 1, 2 * j + 1);
                }
                const float av = DROP ? keep_if(a[q][j], mcur, 8 * t + j) : a[q][j];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[c][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[q][c][u][j], acc[c][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
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
            const float v = fast_tanh(DROP ? fmaf(acc[c][0][i], xscale, bvd) : acc[c][0][i] + bvd);
            const float u = fast_sigmoid(DROP ? fmaf(acc[c][1][i], xscale, bud) : acc[c][1][i] + bud);
            part[i] += wd * v * u;
            if (gates != nullptr) {
                const int gr = row0 + 32 * wr + mfma32_row(i, h);
                if (gr < R) {
                    gates[(size_t)gr * GF_NG + d] = v;
                    gates[(size_t)gr * GF_NG + 192 + d] = u;
                }
            }
        }
    }
    float* sred = smem;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = half_sum_lane31(part[i]);
        if (r == 31) sred[wc * GF_TM + 32 * wr + mfma32_row(i, h)] = v;
    }
    __syncthreads();
    if (tid < GF_TM) {
        const int gr = row0 + tid;
        if (gr < R) scores[gr] = sred[tid] + sred[GF_TM + tid] + battn[0];
    }
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define MIL_SRD_FLAGS 0x00020000      /* raw buffer resource word 3 as hipcc's examples build it for gfx950 */

// ================================================================================ K1a gate forward, "VALU diet" form
// Same tiling, k order and results (bit for bit) as k_gate_fwd; the main loop sheds the vector-ALU instructions that the
// f32 MFMA cannot hide (see k_gate_bwd_dw2 / tools/mfma_valu_mix.hip):
//   * LDS-DMA through buffer resources (buffer_load_dwordx4 ... lds): the per-lane source offset is a loop invariant, the
//     K offset an SGPR, the LDS destination goes to M0 by scalar ALU (no 64-bit pointer adds, no v_readfirstlane);
//     rows of the last tile beyond R read as zeros by the hardware range check;
//   * fragment reads are ds_read_b128 with immediate offsets from twelve precomputed per-lane LDS addresses (the buffer
//     index is a compile-time constant: the slice loop is unrolled by two);
//   * in train mode the keep-mask word of the next slice comes through a buffer resource as well.
// Left in the loop per 32-column slice and wave: the mask itself (2 VALU per A element, train mode only).
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
#define GF2_XS_BYTES (GF_TM * 32 * 4)          /* one x buffer  (16 KB) */
#define GF2_WS_BYTES (GF_NG * 32 * 4)          /* one W buffer  (48 KB) */

// GEN (train mode, L <= 1024): the workgroup DRAWS the keep words of its 128 rows itself - the same Philox blocks
// k_dropout_keep_bits would produce (philox.h) - keeps them in LDS for its own A fragments and writes them to xbits for
// the later consumers (pool, weight gradient); workgroup 0 also draws the head's [B, L/32] words.  One launch less per
// step, and the mask words of the loop come from LDS instead of a buffer load.
struct GateFwdGen {
    uint32_t* xbits_out;        // [R, L/32]
    uint32_t* mbits_out;        // [B, L/32] or NULL
    int B;
    uint32_t seed_lo, seed_hi, mseed_lo, mseed_hi;
    uint64_t offset;
    const int32_t* offset_dev;
};

// PQ > 0 (= L / 256): the attention-pool PARTIAL PASS of the workgroup's four 32-row tiles runs in the epilogue, once the
// 128 scores are final - k_pool_partial's arithmetic, operation for operation (tile weights by the same wave reductions,
// virtual wave v = rows v, v + 4, ..., the four virtual waves of a tile folded in the same order, the head-projection
// by-product through the same 16-value reduction), so partials / hrow equal the stand-alone kernel's.  The rows come back
// from L2 / Infinity Cache (the DMA stream of the main loop read them moments ago): one launch and one cold start less
// per step - the stand-alone pass costs 14 us at 32 x 1024 x 512 although its bytes take 7.  Only for batches whose
// tiles are all full and aligned (T * 32 == R, i.e. tile t = rows 32 t ..), which the host checks.
struct GateFwdPool {
    const int32_t* tile_map;    // [T][4] = {bag, row0, nrows, 0}
    float* partials;            // [T][L] then [T][2]
    int T;
    const float* Wf;            // [2][L] head rows (C == 2)
    float* hrow;                // [R][2]
    const uint32_t* mbits;      // [B][L/32] keep words of the head's dropout (train mode without GEN), else NULL
    float mscale;
};

template <bool DROP, bool GEN, int PQ>
__global__ __launch_bounds__(512) void k_gate_fwd2(const float* __restrict__ x, const float* __restrict__ Wv,
                                                   const float* __restrict__ bv, const float* __restrict__ Wu,
                                                   const float* __restrict__ bu, const float* __restrict__ wvec,
                                                   const float* __restrict__ battn, float* __restrict__ scores,
                                                   float* __restrict__ gates, int R, int L,
                                                   const uint32_t* __restrict__ xbits, float xscale, GateFwdGen gen,
                                                   GateFwdPool pool) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (GF_TM + GF_NG) * 32];      // [2] x buffers, then [2] W buffers
    __shared__ __attribute__((aligned(16))) uint32_t mlds[GEN ? GF_TM * 32 : 4];       // keep words [128][nslice <= 32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * GF_TM;
    const int nslice = L / GF_BK;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    // ---- DMA pieces of this wave: 2 x pieces (8 rows each), 6 W pieces; lane -> (row in piece, physical 16-byte chunk)
    const int prow = lane >> 3, pch = lane & 7;
    const int rows_here = min(GF_TM, R - row0);
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)row0 * L), 0, rows_here * L * 4, MIL_SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc((void*)Wv, 0, MIL_GATE_D * L * 4, MIL_SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t srd_u = __builtin_amdgcn_make_buffer_rsrc((void*)Wu, 0, MIL_GATE_D * L * 4, MIL_SRD_FLAGS);
    int vsrc[8];                 // per-lane byte offset of the piece's source (slice 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < 2) {
            const int lr = (2 * wave + i) * 8 + prow;
            vsrc[i] = (lr * L + 4 * (pch ^ ((lr >> 1) & 7))) * 4;
        } else {
            const int wrow = (6 * wave + (i - 2)) * 8 + prow;                 // 0..383: pieces never straddle Wv / Wu
            vsrc[i] = ((wrow % MIL_GATE_D) * L + 4 * (pch ^ ((wrow >> 1) & 7))) * 4;
        }
    }
    auto dma_piece = [&](int i, int buf, int kbytes) {            // i, buf compile-time after unrolling; kbytes scalar
        if (i < 2) {
            const unsigned dst = lds0 + (unsigned)(buf * GF2_XS_BYTES + (2 * wave + i) * 8 * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_x, (lds_void*)(uintptr_t)dst, 16, vsrc[i], kbytes, 0, 0);
        } else {
            const int p = 6 * wave + (i - 2);
            const unsigned dst = lds0 + (unsigned)(2 * GF2_XS_BYTES + buf * GF2_WS_BYTES + p * 8 * 128);
            if (p < 24) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_void*)(uintptr_t)dst, 16, vsrc[i], kbytes, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_u, (lds_void*)(uintptr_t)dst, 16, vsrc[i], kbytes, 0, 0);
        }
    };
    // ---- fragment addresses (bytes): row (32 wr + r) of the x image / row (96 wc + r) of the W image, swizzled chunk of k-group t
    const int fx = (r >> 1) & 7;
    unsigned fa_addr[4], fb_addr[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned ch = 16u * (unsigned)((2 * t + h) ^ fx);
        fa_addr[t] = lds0 + (unsigned)((32 * wr + r) * 128) + ch;
#pragma unroll
        for (int b = 0; b < 2; ++b)
            fb_addr[b][t] = lds0 + (unsigned)(2 * GF2_XS_BYTES + b * GF2_WS_BYTES + (96 * wc + r) * 128) + ch;
    }
    // ---- keep bits of this lane's fragment row (train mode)
    __amdgpu_buffer_rsrc_t srd_m = srd_x;
    int vmask = 0;
    unsigned mnext = 0;
    if (DROP && !GEN) {
        srd_m = __builtin_amdgcn_make_buffer_rsrc((void*)(xbits + (size_t)row0 * nslice), 0, rows_here * nslice * 4, MIL_SRD_FLAGS);
        vmask = (32 * wr + r) * nslice * 4;
        mnext = __builtin_amdgcn_raw_buffer_load_b32(srd_m, vmask, 0, 0);
    }
    uint64_t off = 0;
    if (GEN) {
        off = gen.offset;
        if (gen.offset_dev != nullptr) off += (uint64_t)(uint32_t)gen.offset_dev[0];
        const size_t blk0 = (size_t)row0 * nslice / 4;                  // 128 nslice words per workgroup: a multiple of 4
        const int nblk = rows_here * nslice / 4;                        // nslice % 4 == 0 (host)
        for (int q = tid; q < nblk; q += 512) {
            const uint4 wds = philox_keep_words_half(blk0 + q, off, gen.seed_lo, gen.seed_hi);
            *reinterpret_cast<uint4*>(gen.xbits_out + 4 * (blk0 + q)) = wds;
            *reinterpret_cast<uint4*>(mlds + 4 * q) = wds;
        }
        if (blockIdx.x == 0 && gen.mbits_out != nullptr) {
            const int nw = gen.B * (L >> 5);                            // even (L % 64 == 0)
            for (int q = tid; 2 * q < nw; q += 512) {
                const uint2 wds = philox_keep_words_quarter((uint64_t)q, off, gen.mseed_lo, gen.mseed_hi);
                gen.mbits_out[2 * q] = wds.x;
                gen.mbits_out[2 * q + 1] = wds.y;
            }
        }
        vmask = (32 * wr + r) * nslice;                                 // word index into mlds
    }

    f32x16 acc[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][u][i] = 0.f;

#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(i, 0, 0);
    __syncthreads();                                   // hipcc drains the DMA (vmcnt(0)) in front of the barrier
    if (GEN) mnext = mlds[vmask];

    auto slice = [&](int s, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const int s1 = min(s + 1, nslice - 1);
        const int k1bytes = s1 * GF_BK * 4;
        unsigned mcur = 0;
        if (DROP) {
            mcur = mnext >> (4 * h);
            if (GEN) mnext = mlds[vmask + s1];
            else mnext = __builtin_amdgcn_raw_buffer_load_b32(srd_m, vmask, s1 * 4, 0);
        }
        f32x4 a[2], b[2][3][2];
        auto frag_piece = [&](int t, int q, int p) {
            if (p == 0) {
                a[q] = *(lds_cf4*)(uintptr_t)(fa_addr[t] + (unsigned)(buf * GF2_XS_BYTES));
            } else {
                const int c = (p - 1) >> 1, u = (p - 1) & 1;
                b[q][c][u] = *(lds_cf4*)(uintptr_t)(fb_addr[buf][t] + (unsigned)((u * 192 + 32 * c) * 128));
            }
        };
#pragma unroll
        for (int p = 0; p < 7; ++p) frag_piece(0, 0, p);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = 4 * t + j;
#if defined(GF2_ABL_NODMA)
                (void)k1bytes;
#elif defined(GF2_ABL_XONLY)
                if (g < 2) dma_piece(g, buf ^ 1, k1bytes);
#elif defined(GF2_ABL_WONLY)
                if (g >= 2 && g < 8) dma_piece(g, buf ^ 1, k1bytes);
#else
                if (g < 8) dma_piece(g, buf ^ 1, k1bytes);     // next slice, one DMA piece per MFMA group
#endif
                if (t < 3) {
                    frag_piece(t + 1, q ^ 1, 2 * j);
                    if (2 * j + 1 < 7) frag_piece(t + 1, q ^ 1, 2 * j + 1);
                }
                const float av = DROP ? keep_if(a[q][j], mcur, 8 * t + j) : a[q][j];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[c][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[q][c][u][j], acc[c][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    int s = 0;
    for (; s + 1 < nslice; s += 2) {
        slice(s, std::integral_constant<int, 0>{});
        __syncthreads();
        slice(s + 1, std::integral_constant<int, 1>{});
        __syncthreads();
    }
    if (s < nslice) {
        slice(s, std::integral_constant<int, 0>{});
        __syncthreads();
    }

    // fused pool pass: the x rows of this wave's tile are requested NOW (they depend on nothing the epilogue computes), so
    // that their trip from L2 / Infinity Cache runs under the activation epilogue
    constexpr int PNQ = PQ > 0 ? PQ : 1;
    f32x4 pv[PQ > 0 ? 2 : 1][PQ > 0 ? MIL_POOL_TILE / 4 : 1][PNQ];
    if (PQ > 0) {
        const int trow0 = row0 + 32 * wr;
        if (trow0 < R) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                    const float* xr = x + (size_t)(trow0 + 2 * wc + j + 4 * i) * L + 4 * lane;
#pragma unroll
                    for (int q = 0; q < PNQ; ++q) pv[j][i][q] = *reinterpret_cast<const f32x4*>(xr + 256 * q);
                }
        }
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
            const float v = fast_tanh(DROP ? fmaf(acc[c][0][i], xscale, bvd) : acc[c][0][i] + bvd);
            const float u = fast_sigmoid(DROP ? fmaf(acc[c][1][i], xscale, bud) : acc[c][1][i] + bud);
            part[i] += wd * v * u;
            if (gates != nullptr) {
                const int gr = row0 + 32 * wr + mfma32_row(i, h);
                if (gr < R) {
                    gates[(size_t)gr * GF_NG + d] = v;
                    gates[(size_t)gr * GF_NG + 192 + d] = u;
                }
            }
        }
    }
    float* sred = smem;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = half_sum_lane31(part[i]);
        if (r == 31) sred[wc * GF_TM + 32 * wr + mfma32_row(i, h)] = v;
    }
    __syncthreads();
    float* sc_lds = smem + 2 * GF_TM;                  // [128] final scores (fused pool pass)
    if (tid < GF_TM) {
        const int gr = row0 + tid;
        const float sc = sred[tid] + sred[GF_TM + tid] + battn[0];
        if (gr < R) scores[gr] = sc;
        if (PQ > 0) sc_lds[tid] = gr < R ? sc : -INFINITY;
    }
    if (PQ > 0) {
        constexpr int NQ = PQ > 0 ? PQ : 1;
        float* pred = smem + 1024;                     // [4 tiles][2 virtual waves][NQ][256]
        __syncthreads();
        const int trow0 = row0 + 32 * wr;              // this wave's tile: rows trow0 .. trow0 + 31 (wave-uniform)
        const bool live = trow0 < R;                   // R % 32 == 0 (host): a tile is whole or absent
        const int t = trow0 >> 5;
        // tile weights, as wave 0 of k_pool_partial forms them
        const float s_ = lane < 32 ? sc_lds[32 * wr + lane] : -INFINITY;
        const float m_ = wave_allmax(s_);
        const float p_ = lane < 32 ? expf(s_ - m_) : 0.f;
        const float l_ = wave_allsum(p_);
        const float xs = DROP ? xscale : 1.0f;
        f32x4 pacc[2][NQ];
        if (live) {
            auto& v = pv;
            const int sh = 4 * (lane & 7);
            if (DROP) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                        const int rr = 2 * wc + j + 4 * i;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const unsigned wd = GEN ? mlds[(32 * wr + rr) * nslice + 8 * q + (lane >> 3)]
                                                    : xbits[(size_t)(trow0 + rr) * (L >> 5) + 8 * q + (lane >> 3)];
                            const unsigned mm = wd >> sh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[j][i][q][e] = keep_if(v[j][i][q][e], mm, e);
                        }
                    }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) pacc[j][q] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                    const int rr = 2 * wc + j + 4 * i;
                    const float pw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_), rr)) * xs;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) pacc[j][q] += pw * v[j][i][q];
                }
            }
            // head rows as this tile's bag sees them (k_pool_partial's head_row), then the by-product h[row][c]
            const int bag_ = pool.tile_map[4 * t];
            f32x4 wf[2][NQ];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f32x4 w = *reinterpret_cast<const f32x4*>(pool.Wf + (size_t)c * L + 256 * q + 4 * lane) * xs;
                    if (DROP) {
                        const int widx = bag_ * (L >> 5) + 8 * q + (lane >> 3);
                        unsigned wd;
                        if (GEN) {      // the word workgroup 0 writes to mbits_out in this same launch: drawn again here
                            const uint2 pr = philox_keep_words_quarter((uint64_t)(widx >> 1), off, gen.mseed_lo, gen.mseed_hi);
                            wd = (widx & 1) ? pr.y : pr.x;
                        } else {
                            wd = pool.mbits[widx];
                        }
                        const unsigned mm = wd >> sh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = keep_if(w[e], mm, e) * pool.mscale;
                    }
                    wf[c][q] = w;
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float d16[16];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                        float d = 0.f;
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            d += v[j][i][q][0] * wf[c][q][0] + v[j][i][q][1] * wf[c][q][1] + v[j][i][q][2] * wf[c][q][2] + v[j][i][q][3] * wf[c][q][3];
                        d16[2 * i + c] = d;
                    }
                const float tot = wave_reduce16(d16, lane);
                const int k = wave_reduce16_index(lane), rr = 2 * wc + j + 4 * (k >> 1);
                if ((lane & 3) == 0) pool.hrow[(size_t)(trow0 + rr) * 2 + (k & 1)] = tot;
            }
            if (wc == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        *reinterpret_cast<f32x4*>(pred + ((wr * 2 + j) * NQ + q) * 256 + 4 * lane) = pacc[j][q];
            }
        }
        __syncthreads();
        if (live && wc == 0) {
            float* out = pool.partials + (size_t)t * L;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 vv = pacc[0][q];
                vv += pacc[1][q];
                vv += *reinterpret_cast<const f32x4*>(pred + ((wr * 2 + 0) * NQ + q) * 256 + 4 * lane);
                vv += *reinterpret_cast<const f32x4*>(pred + ((wr * 2 + 1) * NQ + q) * 256 + 4 * lane);
                *reinterpret_cast<f32x4*>(out + 256 * q + 4 * lane) = vv;
            }
            if (lane == 0) {
                float* ml = pool.partials + (size_t)pool.T * L + 2 * t;
                ml[0] = m_;
                ml[1] = l_;
            }
        }
    }
}

// ================================================================================ K1b attention pool forward
// One workgroup (256 threads) per tile of <= 32 rows of one bag.  Wave w streams rows w, w+4, ...;
// lane l owns columns 4l + 256q (16-byte loads, 1 KiB per wave instruction).
// Output partial: acc[t][L] weighted row sum with weights exp(s_i - m_tile) in partials[0 .. T*L),
// then (m_tile, l_tile) pairs in partials[T*L + 2t ..].
// Optional by-product (Wf != NULL, C <= 4): h[row][c] = x_row . Wf[c], the head's projection of every patch.  The
// backward then needs no second pass over x: x_i . dM = sum_c dz[bag][c] h[i][c] because dM = dz Wf
// (k_pool_ds_from_h replaces the 64 MiB read of k_pool_bwd_ds).
// Train mode: xbits keeps (ABMIL.py:49: the DROPPED x is what gets pooled, :59) - lane l owns columns 4l + 256q, i.e. bits
// 4 (l & 7).. of word 8q + (l >> 3) of its row; the survivors' scale rides on the softmax weight.  mbits / mscale: the
// head's Dropout(.25) on the bag embedding (aggregator.py:129) folds into the head rows used for the by-product
// h[row][c] = x_row . (Wf[c] * keepM[bag] * mscale), so that x_i . dM = sum_c dz_c h[i][c] still holds in the backward.
// NT: x is larger than the Infinity Cache, so the pass is a pure HBM stream: nontemporal loads (tools/cu_load_bw.hip: a
// streaming read of 512 MB runs at 7.2 TB/s with the hint, 6.5 TB/s without).  Off when x fits the cache and the
// weight-gradient pass re-reads it from there.
template <int NQ, bool NT>
__global__ __launch_bounds__(256) void k_pool_partial(const float* __restrict__ x, const float* __restrict__ scores,
                                                      const int32_t* __restrict__ tile_map, float* __restrict__ partials,
                                                      int L, const float* __restrict__ Wf, int C,
                                                      float* __restrict__ hrow, const uint32_t* __restrict__ xbits,
                                                      float xscale, const uint32_t* __restrict__ mbits, float mscale) {
    __shared__ float p_lds[MIL_POOL_TILE];
    __shared__ float ml_lds[2];
    __shared__ __attribute__((aligned(16))) float red[3 * NQ * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int row0 = tile_map[4 * t + 1], nrows = tile_map[4 * t + 2];

    f32x4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0, 0, 0, 0};
    // Rows past the tile end are clamped to its last row and carry weight 0 (p_lds is 0 there): the
    // loop is branch-free, so all 8 x NQ 16-byte loads of a wave are in flight together - and they are issued BEFORE the
    // tile's softmax weights are formed (below), whose score load + two wave reductions then run under the x stream.
    f32x4 v[MIL_POOL_TILE / 4][NQ];
    unsigned mk[MIL_POOL_TILE / 4][NQ];
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const int rr = max(min(wave + 4 * i, nrows - 1), 0);      // nrows == 0: a padding tile of a device-built map
        const float* xr = x + (size_t)(row0 + rr) * L + 4 * lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            v[i][q] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 256 * q)) : *reinterpret_cast<const f32x4*>(xr + 256 * q);
        if (xbits != nullptr) {
            const uint32_t* mr = xbits + (size_t)(row0 + rr) * (L >> 5) + (lane >> 3);
#pragma unroll
            for (int q = 0; q < NQ; ++q) mk[i][q] = mr[8 * q];
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
    if (xbits != nullptr) {
        const int sh = 4 * (lane & 7);
#pragma unroll
        for (int i = 0; i < MIL_POOL_TILE / 4; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned m = mk[i][q] >> sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][q][e] = keep_if(v[i][q][e], m, e);
            }
    } else {
        xscale = 1.0f;
    }
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const float p = p_lds[wave + 4 * i] * xscale;
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] += p * v[i][q];
    }
    // head rows as this tile's bag sees them: x scale and the head's dropout mask folded in
    const int bag_ = tile_map[4 * t];
    auto head_row = [&](int c, int q) {
        f32x4 w = *reinterpret_cast<const f32x4*>(Wf + (size_t)c * L + 256 * q + 4 * lane) * xscale;
        if (mbits != nullptr) {
            const unsigned m = mbits[(size_t)bag_ * (L >> 5) + 8 * q + (lane >> 3)] >> (4 * (lane & 7));
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = keep_if(w[e], m, e) * mscale;
        }
        return w;
    };
    if (Wf != nullptr && C == 2) {
        // two classes (the usual head): the tile's 8 x 2 dot products of this wave go through ONE 16-value reduction
        float d16[16];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 wf[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) wf[q] = head_row(c, q);
#pragma unroll
            for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    d += v[i][q][0] * wf[q][0] + v[i][q][1] * wf[q][1] + v[i][q][2] * wf[q][2] + v[i][q][3] * wf[q][3];
                d16[2 * i + c] = d;
            }
        }
        const float tot = wave_reduce16(d16, lane);
        const int k = wave_reduce16_index(lane), rr = wave + 4 * (k >> 1);
        if ((lane & 3) == 0 && rr < nrows) hrow[(size_t)(row0 + rr) * 2 + (k & 1)] = tot;
    } else if (Wf != nullptr) {
        for (int c = 0; c < C; ++c) {
            f32x4 wf[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) wf[q] = head_row(c, q);
#pragma unroll
            for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    d += v[i][q][0] * wf[q][0] + v[i][q][1] * wf[q][1] + v[i][q][2] * wf[q][2] + v[i][q][3] * wf[q][3];
                d = wave_allsum(d);
                const int rr = wave + 4 * i;
                if (lane == 0 && rr < nrows) hrow[(size_t)(row0 + rr) * C + c] = d;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            *reinterpret_cast<f32x4*>(red + ((wave - 1) * NQ + q) * 256 + 4 * lane) = acc[q];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partials + (size_t)t * L;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 v = acc[q];
#pragma unroll
            for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4*>(red + (w * NQ + q) * 256 + 4 * lane);
            *reinterpret_cast<f32x4*>(out + 256 * q + 4 * lane) = v;
        }
        if (lane == 0) {
            float* ml = partials + (size_t)gridDim.x * L + 2 * t;
            ml[0] = ml_lds[0];
            ml[1] = ml_lds[1];
        }
    }
}

// Merge the tile partials of one bag: M = sum_t e^{m_t - m} acc_t / sum_t e^{m_t - m} l_t.
// grid = (B, L / 128): workgroup (b, cb) owns 128 columns of bag b; thread (g, c4): float4 column c4 < 32,
// tile group g < 8, tile loads unrolled 4 deep (the kernel is latency-bound, so many loads in flight
// and 4x more workgroups than bags).
__global__ __launch_bounds__(256) void k_pool_merge(const float* __restrict__ partials,
                                                    const int32_t* __restrict__ bag_tile_off, float* __restrict__ M,
                                                    float* __restrict__ lse, int L, int T) {
    __shared__ float red[4];
    __shared__ float scale_lds[1024];
    __shared__ __attribute__((aligned(16))) float part_lds[8 * 128];
    const int b = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x;
    const int t0 = bag_tile_off[b], t1 = bag_tile_off[b + 1], nt = t1 - t0;
    const float* ml = partials + (size_t)T * L;
    float m = -INFINITY;
    for (int t = t0 + tid; t < t1; t += 256) m = fmaxf(m, ml[2 * t]);
    m = wave_allmax(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float l = 0.f;
    for (int t = t0 + tid; t < t1; t += 256) l += ml[2 * t + 1] * expf(ml[2 * t] - m);
    l = wave_allsum(l);
    if ((tid & 63) == 0) red[tid >> 6] = l;
    __syncthreads();
    l = red[0] + red[1] + red[2] + red[3];
    const float inv = nt > 0 ? 1.0f / l : 0.f;

    const int c4 = tid & 31, g = tid >> 5;
    const float* base = partials + 128 * cb + 4 * c4;
    f32x4 acc = {0, 0, 0, 0};
    for (int tb = 0; tb < nt; tb += 1024) {
        __syncthreads();
        for (int k = tid; k < 1024; k += 256) scale_lds[k] = (tb + k < nt) ? expf(ml[2 * (t0 + tb + k)] - m) : 0.f;
        __syncthreads();
        const int cnt = min(1024, nt - tb);
        int k = g;
        for (; k + 24 < cnt; k += 32) {
            f32x4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const f32x4*>(base + (size_t)(t0 + tb + k + 8 * e) * L);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += scale_lds[k + 8 * e] * v[e];
        }
        for (; k < cnt; k += 8) acc += scale_lds[k] * *reinterpret_cast<const f32x4*>(base + (size_t)(t0 + tb + k) * L);
    }
    *reinterpret_cast<f32x4*>(part_lds + g * 128 + 4 * c4) = acc;
    __syncthreads();
    if (tid < 128) {
        float v = 0.f;
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) v += part_lds[gg * 128 + tid];
        M[(size_t)b * L + 128 * cb + tid] = v * inv;
    }
    if (tid == 0 && cb == 0) lse[b] = nt > 0 ? m + logf(l) : -INFINITY;
}

// ================================================================================ K1 backward: ds (HBM-bound)
// ds_i = A_i (x_i . dM - cdot),  A_i = exp(s_i - lse).  Optionally dx_i = A_i dM (the pool term).
template <int NQ>
__global__ __launch_bounds__(256) void k_pool_bwd_ds(const float* __restrict__ x, const float* __restrict__ scores,
                                                     const float* __restrict__ lse, const float* __restrict__ dM,
                                                     const float* __restrict__ cdot,
                                                     const int32_t* __restrict__ tile_map, float* __restrict__ ds,
                                                     float* __restrict__ dx, int L, const uint32_t* __restrict__ xbits,
                                                     float xscale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int bag = tile_map[4 * t], row0 = tile_map[4 * t + 1], nrows = tile_map[4 * t + 2];
    constexpr bool NT = false;
    if (xbits == nullptr) xscale = 1.0f;
    f32x4 g[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) g[q] = *reinterpret_cast<const f32x4*>(dM + (size_t)bag * L + 256 * q + 4 * lane);
    const float lse_b = lse[bag], c_b = cdot[bag];
    // branch-free loads (rows past the tile end clamp to its last row), guarded stores
    f32x4 v[MIL_POOL_TILE / 4][NQ];
    unsigned mk[MIL_POOL_TILE / 4][NQ];
    float sc[MIL_POOL_TILE / 4];
    const int sh = 4 * (lane & 7);
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const size_t row = (size_t)(row0 + max(min(wave + 4 * i, nrows - 1), 0));
        const float* xr = x + row * L + 4 * lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            v[i][q] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 256 * q)) : *reinterpret_cast<const f32x4*>(xr + 256 * q);
        if (xbits != nullptr) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mk[i][q] = xbits[row * (L >> 5) + 8 * q + (lane >> 3)] >> sh;
        }
        sc[i] = scores[row];
    }
#pragma unroll
    for (int i = 0; i < MIL_POOL_TILE / 4; ++i) {
        const int rr = wave + 4 * i;
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 xv = v[i][q];
            if (xbits != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = keep_if(xv[e], mk[i][q], e);
            }
            dot += xv[0] * g[q][0] + xv[1] * g[q][1] + xv[2] * g[q][2] + xv[3] * g[q][3];
        }
        dot = wave_allsum(dot) * xscale;
        const float a = expf(sc[i] - lse_b);
        if (rr < nrows) {
            const size_t row = (size_t)(row0 + rr);
            if (lane == 0) ds[row] = a * (dot - c_b);
            if (dx != nullptr) {
                // the pool term of the gradient of the (dropped) rows; the dropout's own backward (mask, scale) is applied
                // once, by the last writer of dx (k_gate_bwd_dx)
                float* dr = dx + row * L + 4 * lane;
#pragma unroll
                for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(dr + 256 * q) = a * g[q];
            }
        }
    }
}

// ds_i = A_i (sum_c dz[bag][c] h[i][c] - cdot[bag]) from the forward's head projections: no pass over x.
// 8 tiles per workgroup, 32 threads per tile.
__global__ __launch_bounds__(256) void k_pool_ds_from_h(const float* __restrict__ scores, const float* __restrict__ lse,
                                                        const float* __restrict__ hrow, const float* __restrict__ dz,
                                                        const float* __restrict__ cdot,
                                                        const int32_t* __restrict__ tile_map, int T, int C,
                                                        float* __restrict__ ds) {
    const int t = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (t >= T) return;
    const int bag = tile_map[4 * t], row0 = tile_map[4 * t + 1], nrows = tile_map[4 * t + 2];
    if (l >= nrows) return;
    const size_t row = (size_t)(row0 + l);
    float g = 0.f;
    for (int c = 0; c < C; ++c) g += dz[bag * C + c] * hrow[row * C + c];
    ds[row] = expf(scores[row] - lse[bag]) * (g - cdot[bag]);
}

// ================================================================================ K1 backward: gate dW (MFMA)
// dW[gi][j] = sum_rows dPre[row][gi] x[row][j], gi = permuted gate index: block m (0..2) holds
// d in [64m, 64m+64): local 0..63 -> dPreV_d, 64..127 -> dPreU_d, so one (V, U) load pair yields both.
// Workgroup 256 threads = 4 waves, output tile 128 (gi) x 128 (j), wave (wi, wj) owns 64 x 64.
// Split-K over row chunks of KC rows; partials [S][384][L] are summed by k_gate_bwd_reduce.
// Both operands are "k-major" in LDS ([row][i] and [row][j]): lane (i = l & 31, k = l >> 5) reads
// word row*128 + i: consecutive lanes -> consecutive banks, no conflicts, no padding.
#define GB_BKR 32
typedef unsigned short gb_u16x8 __attribute__((ext_vector_type(8)));

// XB16: x is stored as bf16 and widened to fp32 while it is staged (config-5 path; the product stays fp32 MFMA).
// KG: K groups per workgroup.  KG = 2 (tall inputs): 512 threads, the two halves of the workgroup run the same pipeline
// on the two halves of the row chunk (own LDS stages, common barriers) and fold their accumulators through LDS before the
// store, so a launch needs half as many row chunks for the same number of resident waves - half the partial tiles to
// write here and to read in k_gate_bwd_reduce.
template <bool XB16, int KG, bool DROP>
__global__ __launch_bounds__(256 * KG) void k_gate_bwd_dw(const void* __restrict__ xv, const float* __restrict__ gates,
                                                     const float* __restrict__ ds, const float* __restrict__ wvec,
                                                     float* __restrict__ part, float* __restrict__ pbias, int R, int L,
                                                     int KC, int NJ, const uint32_t* __restrict__ xbits,
                                                     const int32_t* __restrict__ rows_dev) {
    __shared__ __attribute__((aligned(16))) float smem_all[KG * 2 * 2 * GB_BKR * 128];
    if (rows_dev != nullptr) R = min(R, rows_dev[0]);      // bucketed batches: the true row count lives on the device
    const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);
    float* smem = smem_all + grp * (2 * 2 * GB_BKR * 128);      // this K group's stages
    float* ab = smem;                         // [2][32][128] dPre
    float* xb = smem + 2 * GB_BKR * 128;      // [2][32][128] x
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware order: hardware deals workgroup ids round-robin over the 8 XCDs (id % 8 shares an L2).
    // The 3*NJ workgroups of one row chunk re-read the same x / gate rows, so give them consecutive
    // slots of ONE XCD: logical = (id % 8) * ceil-share + id / 8 (bijective form for any grid size).
    int bid = blockIdx.x;
#if !defined(GB_NOXCD)
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
#endif
    const int jt = bid % NJ, m = (bid / NJ) % 3, s = bid / (3 * NJ);
    const int j0 = jt * 128;
    // row chunk of the workgroup, then this K group's part of it (whole slices to group 0 first); the slice loop runs
    // nloop times for everybody (common barriers): iterations past a group's own slices multiply dead copies (ds = 0)
    const int cbeg = min(s * KC, R), cend = min(R, cbeg + KC);
    const int half = KG == 1 ? cend - cbeg : ((cend - cbeg + 2 * GB_BKR - 1) / (2 * GB_BKR)) * GB_BKR;
    const int rbeg = min(cend, cbeg + grp * half), rend = KG == 1 ? cend : min(cend, rbeg + half);
    const int nslice = (rend - rbeg + GB_BKR - 1) / GB_BKR;
    const int nloop = (min(half, cend - cbeg) + GB_BKR - 1) / GB_BKR;

    const float* x = static_cast<const float*>(xv);
    const unsigned short* xh = static_cast<const unsigned short*>(xv);
    // staging maps
    const int xrow = tid >> 5, xc4 = tid & 31;    // x: rows xrow + 8i (i < 4), 16-byte chunk xc4
    const int hrow = tid >> 4, hc8 = tid & 15;    // bf16 x: rows hrow + 16i (i < 2), 16-byte chunk (8 columns) hc8
    gb_u16x8 rh[2];
    const int arow = tid >> 4, ad4 = tid & 15;    // gates: rows arow + 16i (i < 2), d = 64m + 4*ad4
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wvec + 64 * m + 4 * ad4);
    f32x4 rx[4], rv[2], ru[2];
    unsigned rm[4] = {0, 0, 0, 0};   // train mode: keep bits of the staged x chunks (the forward's mask, csrc/dropout.hip)
    float rds[2], rmask[2];      // raw ds value and its validity mask (applied at use, never at load)
    f32x4 acc_bv = {0, 0, 0, 0}, acc_bu = {0, 0, 0, 0}, acc_w = {0, 0, 0, 0};
    float acc_ds = 0.f;

    // Branch-free staging pieces (rows past the chunk end are clamped to its last row and get ds = 0, so they
    // add nothing): the loop body is one basic block and every piece sits between two MFMA groups.
    const int LW = L >> 5;
    auto xload = [&](int i, int rs) {
        if (XB16) {
            if (i < 2) {
                const int gr = max(min(rs + hrow + 16 * i, rend - 1), 0);
                rh[i] = *reinterpret_cast<const gb_u16x8*>(xh + (size_t)gr * L + j0 + 8 * hc8);
                if (DROP) rm[i] = xbits[(size_t)gr * LW + ((j0 + 8 * hc8) >> 5)];
            }
        } else {
            const int gr = max(min(rs + xrow + 8 * i, rend - 1), 0);
            rx[i] = *reinterpret_cast<const f32x4*>(x + (size_t)gr * L + j0 + 4 * xc4);
            if (DROP) rm[i] = xbits[(size_t)gr * LW + ((j0 + 4 * xc4) >> 5)];
        }
    };
    auto xwrite = [&](int i, int buf) {
        if (XB16) {
            if (i < 2) {
                float* dst = xb + (buf * GB_BKR + hrow + 16 * i) * 128 + 8 * hc8;
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(((unsigned)rh[i][e]) << 16);
                    hi[e] = __uint_as_float(((unsigned)rh[i][4 + e]) << 16);
                }
                if (DROP) {
                    const unsigned mm = rm[i] >> (8 * (hc8 & 3));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] = keep_if(lo[e], mm, e); hi[e] = keep_if(hi[e], mm, 4 + e); }
                }
                *reinterpret_cast<f32x4*>(dst) = lo;
                *reinterpret_cast<f32x4*>(dst + 4) = hi;
            }
        } else {
            f32x4 v = rx[i];
            if (DROP) {
                const unsigned mm = rm[i] >> (4 * (xc4 & 7));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = keep_if(v[e], mm, e);
            }
            *reinterpret_cast<f32x4*>(xb + (buf * GB_BKR + xrow + 8 * i) * 128 + 4 * xc4) = v;
        }
    };
    auto aload = [&](int i, int rs, bool live) {
        const int gr = rs + arow + 16 * i;
        const int gc = max(min(gr, rend - 1), 0);
        const float* gp = gates + (size_t)gc * GF_NG + 64 * m + 4 * ad4;
        rv[i] = *reinterpret_cast<const f32x4*>(gp);
        ru[i] = *reinterpret_cast<const f32x4*>(gp + 192);
        rds[i] = ds[gc];                                  // unconditional load: keeps the body branch-free
        rmask[i] = (live && gr < rend) ? 1.f : 0.f;
    };
#if defined(GB_OLD_DPRE)
    auto awrite_v = [&](int i, int buf) {           // dPreV = ds w U (1 - V^2)
        const f32x4 v = rv[i], u = ru[i];
        const f32x4 pv = (rds[i] * rmask[i] * w4) * u * (1.0f - v * v);
        *reinterpret_cast<f32x4*>(ab + (buf * GB_BKR + arow + 16 * i) * 128 + 4 * ad4) = pv;
        acc_bv += pv;
    };
    auto awrite_u = [&](int i, int buf) {           // dPreU = ds w V U (1 - U)
        const f32x4 v = rv[i], u = ru[i];
        const float dsv = rds[i] * rmask[i];
        const f32x4 pu = (dsv * w4) * v * u * (1.0f - u);
        *reinterpret_cast<f32x4*>(ab + (buf * GB_BKR + arow + 16 * i) * 128 + 64 + 4 * ad4) = pu;
        acc_bu += pu;
        acc_w += dsv * v * u;
        if (ad4 == 0) acc_ds += dsv;
    };
#else
    // dPreV = ds w U (1 - V^2) = a - (a V) V and dPreU = ds w V U (1 - U) = t - t U with a = ds w U, t = a V: four VALU
    // per (V, U) pair instead of seven.  (The bias / w / b sums stay unconditional: a wave-uniform `jt == 0` branch around
    // them splits the k-step into basic blocks and costs more than the six VALU it saves.)
    f32x4 rt[2];
    auto awrite_v = [&](int i, int buf) {
        const f32x4 v = rv[i];
        const f32x4 a = ((rds[i] * rmask[i]) * w4) * ru[i];
        const f32x4 t = a * v;
        const f32x4 pv = a - t * v;
        rt[i] = t;
        *reinterpret_cast<f32x4*>(ab + (buf * GB_BKR + arow + 16 * i) * 128 + 4 * ad4) = pv;
        acc_bv += pv;
    };
    auto awrite_u = [&](int i, int buf) {
        const f32x4 t = rt[i], u = ru[i];
        const f32x4 pu = t - t * u;
        *reinterpret_cast<f32x4*>(ab + (buf * GB_BKR + arow + 16 * i) * 128 + 64 + 4 * ad4) = pu;
        const float dsv = rds[i] * rmask[i];
        acc_bu += pu;
        acc_w += (dsv * rv[i]) * u;
        if (ad4 == 0) acc_ds += dsv;
    };
#endif

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    {
#pragma unroll
        for (int i = 0; i < 4; ++i) xload(i, rbeg);
#pragma unroll
        for (int i = 0; i < 2; ++i) aload(i, rbeg, nslice > 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) xwrite(i, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) { awrite_v(i, 0); awrite_u(i, 0); }
        const int rs1 = rbeg + max(min(1, nslice - 1), 0) * GB_BKR;
#pragma unroll
        for (int i = 0; i < 4; ++i) xload(i, rs1);
#pragma unroll
        for (int i = 0; i < 2; ++i) aload(i, rs1, nslice > 1);   // a single-slice chunk must not count slice 0 twice
    }
    __syncthreads();
#if defined(GB_STAMP)
    const uint64_t st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Issue order inside a k-step.  The two waves of a SIMD (w and w + 4) share its matrix pipe and the older one wins every
    // arbitration: measured with in-kernel stamps (tools/kbench_clock.py), waves 0-3 spent 29 % of the loop parked at the
    // slice barrier while waves 4-7, starved until then, finished the slice ALONE.  A wave on its own only keeps the pipe
    // busy if its non-matrix instructions sit in the shadow of its own MFMAs, so the k-step is laid out as
    //     MFMA . fragment reads . MFMA . staging part a . MFMA . staging part b . MFMA . staging part c
    // (one v_mfma_f32_32x32x2_f32 occupies the pipe for 64 cycles; a part is a handful of VALU / one LDS write / one or
    // two global loads), each boundary pinned with sched_barrier: hipcc otherwise gathers the staging in front of a
    // block of four MFMAs, behind which the wave sits blocked for 3 x 64 cycles with nothing else to issue.
    auto slice_body = [&](int sl) {
        const int buf = sl & 1;
        // registers hold slice sl+1 (or, past this group's last slice, a dead copy with ds forced to 0)
        const bool live2 = sl + 2 < nslice;
        const int rs2 = rbeg + max(min(sl + 2, nslice - 1), 0) * GB_BKR;
        const float* ap = ab + buf * GB_BKR * 128 + h * 128 + 64 * wi + r;
        const float* bp = xb + buf * GB_BKR * 128 + h * 128 + 64 * wj + r;
        float fa[2][2], fb[2][2];
        fa[0][0] = ap[0]; fa[0][1] = ap[32]; fb[0][0] = bp[0]; fb[0][1] = bp[32];
        // staging parts of k-step ks (p = 0, 1, 2): LDS image of slice sl+1 from the registers, registers reloaded with sl+2
        auto part = [&](int ks, int p) {
#if defined(GB_ABL_NOLD)
            if (ks >= 1 && ks <= 4 && p == 0) xwrite(ks - 1, buf ^ 1);
            if (ks == 5 && p == 0) awrite_v(0, buf ^ 1);
            if (ks == 6 && p == 0) awrite_u(0, buf ^ 1);
            if (ks == 7 && p == 0) awrite_v(1, buf ^ 1);
            if (ks == 8 && p == 0) awrite_u(1, buf ^ 1);
            (void)rs2; (void)live2;
#elif defined(GB_ABL_NOWR)
            if (ks >= 1 && ks <= 4 && p == 1) { asm volatile("" ::"v"(rx[ks - 1])); xload(ks - 1, rs2); }
            if (ks == 6 && p == 1) { asm volatile("" ::"v"(rv[0]), "v"(ru[0]), "v"(rds[0])); aload(0, rs2, live2); }
            if (ks == 8 && p == 1) { asm volatile("" ::"v"(rv[1]), "v"(ru[1]), "v"(rds[1])); aload(1, rs2, live2); }
#else
#if !defined(GB_ABL_NOX)
            if (ks >= 1 && ks <= 4) {
                if (p == 0) xwrite(ks - 1, buf ^ 1);
                if (p == 1) xload(ks - 1, rs2);
            }
#endif
#if !defined(GB_ABL_NOA)
            if (ks == 5 && p == 0) awrite_v(0, buf ^ 1);
            if (ks == 6 && p == 0) awrite_u(0, buf ^ 1);
            if (ks == 6 && p == 1) aload(0, rs2, live2);
            if (ks == 7 && p == 0) awrite_v(1, buf ^ 1);
            if (ks == 8 && p == 0) awrite_u(1, buf ^ 1);
            if (ks == 8 && p == 1) aload(1, rs2, live2);
#endif
#endif
        };
#pragma unroll
        for (int ks = 0; ks < GB_BKR / 2; ++ks) {
            const int q = ks & 1;
#if defined(GB_OLD_ORDER)
            if (ks + 1 < GB_BKR / 2) {
                fa[q ^ 1][0] = ap[(ks + 1) * 256]; fa[q ^ 1][1] = ap[(ks + 1) * 256 + 32];
                fb[q ^ 1][0] = bp[(ks + 1) * 256]; fb[q ^ 1][1] = bp[(ks + 1) * 256 + 32];
            }
            part(ks, 0); part(ks, 1); part(ks, 2);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#else
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < GB_BKR / 2) {
                fa[q ^ 1][0] = ap[(ks + 1) * 256]; fa[q ^ 1][1] = ap[(ks + 1) * 256 + 32];
                fb[q ^ 1][0] = bp[(ks + 1) * 256]; fb[q ^ 1][1] = bp[(ks + 1) * 256 + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][1], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            part(ks, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][0], acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            part(ks, 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            part(ks, 2);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
#if defined(GB_STAMP)
    uint64_t st_bar = 0;
#define GB_SYNC() do { const uint64_t b0_ = __builtin_amdgcn_s_memtime(); __syncthreads(); st_bar += __builtin_amdgcn_s_memtime() - b0_; } while (0)
#else
#define GB_SYNC() __syncthreads()
#endif
    for (int sl = 0; sl < nloop; ++sl) {
        slice_body(sl);
        GB_SYNC();
    }

#if defined(GB_STAMP)
    // diagnostic build only: shader-clock and 100 MHz real-time ticks across the main loop, into unused pbias slots
    if (threadIdx.x == 0) {
        const uint64_t st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        float* dbg = pbias + ((size_t)s * 4 + 3) * 192 + 8 + 2 * (m * NJ + jt);
        dbg[0] = (float)(st_t1 - st_t0);
        dbg[1] = (float)(st_r1 - st_r0);
    }
    if ((threadIdx.x & 63) == 0 && m == 0 && jt == 0)      // per-wave cycles spent inside the slice barriers
        pbias[((size_t)s * 4 + 3) * 192 + 64 + (threadIdx.x >> 6)] = (float)st_bar;
#endif
#if defined(GB_OLD_STORE)
    // KG = 2: group 1 hands its accumulators to group 0 through its own (now dead) staging area - same lane, same register
    // index, so no transpose - and group 0 alone stores the folded tile
    if (KG == 2) {
        __syncthreads();
        float* fold = smem_all + (2 * 2 * GB_BKR * 128) + wave * (64 * 64);      // group 1's 64 KB: 16 KB per wave
        if (grp == 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) fold[((a * 2 + b) * 16 + i) * 64 + lane] = acc[a][b][i];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] += fold[((a * 2 + b) * 16 + i) * 64 + lane];
        }
    }

    // partial tile -> part[s][128m + 64wi + row][j0 + 64wj + col].  The accumulators hold a column per lane
    // (16 rows each); going through LDS turns 64 four-byte stores per lane into 16 sixteen-byte ones
    // (each wave transposes its own 64 x 64 tile in its own 16 KB of the staging area; stride 64 is conflict-free both ways).
    {
        __syncthreads();                                  // the staging buffers are dead from here on
        if (grp == 0) {
            float* tw = smem + wave * (64 * 64);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) tw[(32 * a + mfma32_row(i, h)) * 64 + 32 * b + r] = acc[a][b][i];
            // same-wave write -> read: no barrier needed, the LDS ops of a wave complete in order (lgkmcnt)
            float* pt = part + ((size_t)s * GF_NG + 128 * m + 64 * wi) * L + j0 + 64 * wj;
            const int c4 = lane & 15, rr = lane >> 4;         // 16 float4 columns x 4 rows per pass
#pragma unroll
            for (int pass = 0; pass < 16; ++pass) {
                const int row = 4 * pass + rr;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tw + row * 64 + 4 * c4);
                *reinterpret_cast<f32x4*>(pt + (size_t)row * L + 4 * c4) = v;
            }
        }
    }
#else
    // partial tile -> part[s][128m + 64wi + row][j0 + 64wj + col].  The accumulators hold a column per lane (16 rows
    // each); going through LDS turns 64 four-byte stores per lane into sixteen-byte ones: every wave writes its 64 x 64
    // tile row-major into its own 16 KB of its group's (now dead) staging area (stride 64 is conflict-free both ways).
    // KG = 2: both K groups do that, then the two waves that own the same tile (one per group) each fold and store HALF
    // of its rows - one LDS round trip and all eight waves storing, instead of fold -> transpose -> store by four.
    {
        __syncthreads();                                  // the staging buffers are dead from here on
        float* tw = smem + wave * (64 * 64);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) tw[(32 * a + mfma32_row(i, h)) * 64 + 32 * b + r] = acc[a][b][i];
        if (KG == 2) __syncthreads();                     // KG = 1: same-wave write -> read, ordered by lgkmcnt
        float* pt = part + ((size_t)s * GF_NG + 128 * m + 64 * wi) * L + j0 + 64 * wj;
        const int c4 = lane & 15, rr = lane >> 4;         // 16 float4 columns x 4 rows per pass
        const float* t0 = smem_all + wave * (64 * 64);
        const float* t1 = smem_all + (2 * 2 * GB_BKR * 128) + wave * (64 * 64);
        constexpr int NP = 16 / KG;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int row = 4 * (pass + NP * grp) + rr;
            f32x4 v = *reinterpret_cast<const f32x4*>(t0 + row * 64 + 4 * c4);
            if (KG == 2) v += *reinterpret_cast<const f32x4*>(t1 + row * 64 + 4 * c4);
            *reinterpret_cast<f32x4*>(pt + (size_t)row * L + 4 * c4) = v;
        }
    }
#endif

    // bias / w partials (only the j-tile-0 workgroups publish them); both K groups contribute their row groups
    if (jt == 0) {
        float* redf = smem_all;   // [16 KG row groups][3][64]
        const int arow_all = arow + 16 * grp;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            redf[(arow_all * 3 + 0) * 64 + 4 * ad4 + e] = acc_bv[e];
            redf[(arow_all * 3 + 1) * 64 + 4 * ad4 + e] = acc_bu[e];
            redf[(arow_all * 3 + 2) * 64 + 4 * ad4 + e] = acc_w[e];
        }
        __syncthreads();
        if (grp == 0 && tid < 192) {
            const int which = tid / 64, d = tid % 64;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 16 * KG; ++g) v += redf[(g * 3 + which) * 64 + d];
            pbias[((size_t)s * 4 + which) * 192 + 64 * m + d] = v;
        }
        if (m == 0) {
            __syncthreads();
            redf[threadIdx.x] = acc_ds;
            __syncthreads();
            if (threadIdx.x == 0) {
                float v = 0.f;
                for (int g = 0; g < 256 * KG; g += 16) v += redf[g];
                pbias[((size_t)s * 4 + 3) * 192] = v;
            }
        }
    }
}


// ================================================================================ K1 backward: gate dW, "VALU diet" form
// Same product, tiling, split-K layout, k order and outputs (bit for bit) as k_gate_bwd_dw<false, 2, *>; what changes is
// how little VECTOR-ALU work the main loop carries.  Measured on MI355X (tools/mfma_valu_mix.hip): unlike the bf16 MFMAs,
// v_mfma_f32_32x32x2_f32 does NOT hide VALU instructions issued around it - it runs at the f32 VALU rate and every
// v_fma / v_and between two of them costs 3-5 cycles of matrix time, at one or two waves per SIMD alike (4 fillers per
// MFMA: 64 -> 84 cycles; LDS reads, scalar ALU and s_nop fillers are free).  The first kernel spent ~230 VALU
// instructions per wave and 32-row slice (64 MFMAs) on 64-bit address arithmetic, row clamps, LDS addresses, masks:
// 27 % of the loop.  Here
//   * global operands come through buffer resources whose base / size live in SGPRs and advance by scalar ALU: the
//     per-lane offset is a loop invariant, rows beyond the K group's end read as ZERO by the hardware range check (no
//     clamps, no validity masks - a zero ds row contributes nothing);
//   * both LDS images store their four 32-column blocks in the order {0, 2, 1, 3}, so the two operand values a lane
//     needs per k-step are 64 dwords apart and ONE ds_read2st64_b32 with immediate offsets fetches them (no address
//     VALU; the buffer index is a compile-time constant: the slice loop is unrolled by two);
//   * the bias / w / b sums are spread over the NJ column-tile workgroups of a row chunk (slice sl is summed by the
//     workgroup with jt == sl % NJ, scalar branch) instead of being summed by all of them and published by one.
// What is left is the arithmetic itself: dPre (20 VALU per (V, U) float4 pair) and, in train mode, the keep mask of x.

template <bool DROP>
__global__ __launch_bounds__(512) void k_gate_bwd_dw2(const float* __restrict__ x, const float* __restrict__ gates,
                                                      const float* __restrict__ ds, const float* __restrict__ wvec,
                                                      float* __restrict__ part, float* __restrict__ pbias, int R, int L,
                                                      int KC, int NJ, const uint32_t* __restrict__ xbits,
                                                      const int32_t* __restrict__ rows_dev) {
    __shared__ __attribute__((aligned(16))) float smem_all[2 * 2 * 2 * GB_BKR * 128];
    if (rows_dev != nullptr) R = min(R, __builtin_amdgcn_readfirstlane(rows_dev[0]));      // bucketed batches: true row count on the device
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    float* smem = smem_all + grp * (2 * 2 * GB_BKR * 128);      // this K group's stages: [2][32][128] dPre, [2][32][128] x
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int jt = bid % NJ, m = (bid / NJ) % 3, s = bid / (3 * NJ);
    const int j0 = jt * 128;
    const int cbeg = min(s * KC, R), cend = min(R, cbeg + KC);
    const int half = ((cend - cbeg + 2 * GB_BKR - 1) / (2 * GB_BKR)) * GB_BKR;
    const int rbeg = min(cend, cbeg + grp * half), rend = min(cend, rbeg + half);
    const int nloop = (min(half, cend - cbeg) + GB_BKR - 1) / GB_BKR;      // common to both groups (shared barriers)

    // per-lane byte offsets inside a slice (loop invariants)
    const int xrow = tid >> 5, xc4 = tid & 31;    // x: rows xrow + 8i (i < 4), 16-byte chunk xc4 of the 128-column tile
    const int arow = tid >> 4, ad4 = tid & 15;    // gates: rows arow + 16i (i < 2), d = 64m + 4 ad4
    const int LW = L >> 5;
    int vx[4], vm[4], vg[2], vd[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        vx[i] = ((xrow + 8 * i) * L + j0 + 4 * xc4) * 4;
        vm[i] = ((xrow + 8 * i) * LW + ((j0 + 4 * xc4) >> 5)) * 4;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        vg[i] = ((arow + 16 * i) * GF_NG + 64 * m + 4 * ad4) * 4;
        vd[i] = (arow + 16 * i) * 4;
    }
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wvec + 64 * m + 4 * ad4);
    // LDS positions: 32-column blocks stored in the order {0, 2, 1, 3}
    auto blkpos = [](int c) { return (c & 31) | ((c & 32) << 1) | ((c & 64) >> 1); };
    float* const ab = smem;                       // [2][32][128]
    float* const xb = smem + 2 * GB_BKR * 128;    // [2][32][128]
    float* const xw = xb + xrow * 128 + blkpos(4 * xc4);              // + (buf * 32 + 8 i) * 128
    float* const aw = ab + arow * 128 + blkpos(4 * ad4);              // V block; U block: blkpos(64 + 4 ad4) = + 32
    const float* const ap = ab + h * 128 + 32 * wi + r;               // + buf * 4096 + ks * 256 (+ 64)
    const float* const bp = xb + h * 128 + 32 * wj + r;

    u32x4_t rx[4], rv[2], ru[2];
    unsigned rm[4] = {0, 0, 0, 0};
    float rds[2];
    f32x4 rt[2];
    f32x4 acc_bv = {0, 0, 0, 0}, acc_bu = {0, 0, 0, 0}, acc_w = {0, 0, 0, 0};
    float acc_ds = 0.f;

    // scalar: resources of slice `row0`.  Rows >= rend are out of range -> the loads return zeros.
    auto x_srd = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)row0 * L), 0, max(rend - row0, 0) * L * 4, MIL_SRD_FLAGS);
    };
    auto g_srd = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(gates + (size_t)row0 * GF_NG), 0, max(rend - row0, 0) * GF_NG * 4,
                                                 MIL_SRD_FLAGS);
    };
    auto d_srd = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(ds + row0), 0, max(rend - row0, 0) * 4, MIL_SRD_FLAGS);
    };
    auto m_srd = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(xbits + (size_t)row0 * LW), 0, max(rend - row0, 0) * LW * 4,
                                                 MIL_SRD_FLAGS);
    };
    auto xload = [&](int i, int row0) {
        rx[i] = __builtin_amdgcn_raw_buffer_load_b128(x_srd(row0), vx[i], 0, 0);
        if (DROP) rm[i] = __builtin_amdgcn_raw_buffer_load_b32(m_srd(row0), vm[i], 0, 0);
    };
    auto aload = [&](int i, int row0) {
        const __amdgpu_buffer_rsrc_t g = g_srd(row0);
        rv[i] = __builtin_amdgcn_raw_buffer_load_b128(g, vg[i], 0, 0);
        ru[i] = __builtin_amdgcn_raw_buffer_load_b128(g, vg[i] + 192 * 4, 0, 0);
        rds[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(d_srd(row0), vd[i], 0, 0));
    };
    auto xwrite = [&](int i, int buf) {
        f32x4 v = __builtin_bit_cast(f32x4, rx[i]);
        if (DROP) {
            const unsigned mm = rm[i] >> (4 * (xc4 & 7));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = keep_if(v[e], mm, e);
        }
        *reinterpret_cast<f32x4*>(xw + (buf * GB_BKR + 8 * i) * 128) = v;
    };
    auto awrite_v = [&](int i, int buf) {           // dPreV = a - (a V) V,  a = ds w U
        const f32x4 v = __builtin_bit_cast(f32x4, rv[i]);
        const f32x4 a = (rds[i] * w4) * __builtin_bit_cast(f32x4, ru[i]);
        const f32x4 t = a * v;
        rt[i] = t;
        *reinterpret_cast<f32x4*>(aw + (buf * GB_BKR + 16 * i) * 128) = a - t * v;
    };
    auto awrite_u = [&](int i, int buf, bool pub) { // dPreU = t - t U,  t = ds w U V
        const f32x4 t = rt[i], u = __builtin_bit_cast(f32x4, ru[i]);
        const f32x4 pu = t - t * u;
        *reinterpret_cast<f32x4*>(aw + (buf * GB_BKR + 16 * i) * 128 + 32) = pu;
        if (pub) {                                    // scalar branch: this slice's sums belong to this workgroup
            const f32x4 v = __builtin_bit_cast(f32x4, rv[i]);
            acc_bv += (rds[i] * w4) * u - t * v;      // = dPreV again (3 VALU per element, only every NJ-th slice)
            acc_bu += pu;
            acc_w += (rds[i] * v) * u;
            if (ad4 == 0) acc_ds += rds[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // prologue: slice 0 -> LDS buffer 0, slice 1 -> registers
#pragma unroll
    for (int i = 0; i < 4; ++i) xload(i, rbeg);
#pragma unroll
    for (int i = 0; i < 2; ++i) aload(i, rbeg);
#pragma unroll
    for (int i = 0; i < 4; ++i) xwrite(i, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) { awrite_v(i, 0); awrite_u(i, 0, jt == 0); }
#pragma unroll
    for (int i = 0; i < 4; ++i) xload(i, rbeg + GB_BKR);
#pragma unroll
    for (int i = 0; i < 2; ++i) aload(i, rbeg + GB_BKR);
    __syncthreads();

    // one slice: 16 k-steps x 4 MFMAs; between the MFMAs the parts that stage slice sl + 1 and reload slice sl + 2
    auto slice = [&](int sl, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const int row2 = rbeg + (sl + 2) * GB_BKR;
        const bool pub = ((sl + 1) % NJ) == jt;                     // the slice being written now is sl + 1
        const float* apb = ap + buf * GB_BKR * 128;
        const float* bpb = bp + buf * GB_BKR * 128;
        float fa[2][2], fb[2][2];
        fa[0][0] = apb[0]; fa[0][1] = apb[64]; fb[0][0] = bpb[0]; fb[0][1] = bpb[64];
#pragma unroll
        for (int ks = 0; ks < GB_BKR / 2; ++ks) {
            const int q = ks & 1;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < GB_BKR / 2) {
                fa[q ^ 1][0] = apb[(ks + 1) * 256]; fa[q ^ 1][1] = apb[(ks + 1) * 256 + 64];
                fb[q ^ 1][0] = bpb[(ks + 1) * 256]; fb[q ^ 1][1] = bpb[(ks + 1) * 256 + 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][1], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks >= 1 && ks <= 4) xwrite(ks - 1, buf ^ 1);
            if (ks == 5) awrite_v(0, buf ^ 1);
            if (ks == 6) awrite_u(0, buf ^ 1, pub);
            if (ks == 7) awrite_v(1, buf ^ 1);
            if (ks == 8) awrite_u(1, buf ^ 1, pub);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][0], acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks >= 1 && ks <= 4) xload(ks - 1, row2);
            if (ks == 6) aload(0, row2);
            if (ks == 8) aload(1, row2);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int sl = 0;
    for (; sl + 1 < nloop; sl += 2) {
        slice(sl, std::integral_constant<int, 0>{});
        __syncthreads();
        slice(sl + 1, std::integral_constant<int, 1>{});
        __syncthreads();
    }
    if (sl < nloop) {
        slice(sl, std::integral_constant<int, 0>{});
        __syncthreads();
    }

    // partial tile -> part[s][128m + 64wi + row][j0 + 64wj + col]: every wave writes its 64 x 64 tile row-major into its own
    // 16 KB of its group's (now dead) staging area; the two waves that own the same tile (one per K group) each fold
    // and store half of its rows with 16-byte stores.
    {
        float* tw = smem + wave * (64 * 64);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) tw[(32 * a + mfma32_row(i, h)) * 64 + 32 * b + r] = acc[a][b][i];
        __syncthreads();
        float* pt = part + ((size_t)s * GF_NG + 128 * m + 64 * wi) * L + j0 + 64 * wj;
        const int c4 = lane & 15, rr = lane >> 4;
        const float* t0 = smem_all + wave * (64 * 64);
        const float* t1 = smem_all + (2 * 2 * GB_BKR * 128) + wave * (64 * 64);
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = 4 * (pass + 8 * grp) + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(t0 + row * 64 + 4 * c4) +
                            *reinterpret_cast<const f32x4*>(t1 + row * 64 + 4 * c4);
            *reinterpret_cast<f32x4*>(pt + (size_t)row * L + 4 * c4) = v;
        }
    }

    // bias / w / b partials: every (s, jt) workgroup publishes the sums of its share of the slices -> pbias[s][jt][4][192]
    {
        float* redf = smem_all;   // [32 row groups][3][64]
        const int arow_all = arow + 16 * grp;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            redf[(arow_all * 3 + 0) * 64 + 4 * ad4 + e] = acc_bv[e];
            redf[(arow_all * 3 + 1) * 64 + 4 * ad4 + e] = acc_bu[e];
            redf[(arow_all * 3 + 2) * 64 + 4 * ad4 + e] = acc_w[e];
        }
        __syncthreads();
        float* pb = pbias + ((size_t)s * NJ + jt) * 4 * 192;
        if (threadIdx.x < 192) {
            const int which = threadIdx.x / 64, d = threadIdx.x % 64;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 32; ++g) v += redf[(g * 3 + which) * 64 + d];
            pb[which * 192 + 64 * m + d] = v;
        }
        if (m == 0) {
            __syncthreads();
            redf[threadIdx.x] = acc_ds;
            __syncthreads();
            if (threadIdx.x == 0) {
                float v = 0.f;
                for (int g = 0; g < 512; g += 16) v += redf[g];
                pb[3 * 192] = v;
            }
        }
    }
}

#include "gate_reduce.h"

// ================================================================================ host entry points
// KG = 2 (two K groups per 512-thread workgroup, one workgroup per CU) once a row chunk is at least 512 rows deep
static inline int split_kg(int R, int L) {
#if defined(GB_NO_KG2)
    return 1;
#endif
    const int NJ = L / 128;
    const int smax = (MIL_NUM_CU) / (3 * NJ);
    return (smax >= 1 && R / smax >= 512) ? 2 : 1;
}
static inline int split_plan(int R, int L, int* KC_out) {
    const int NJ = L / 128;
    int smax = (2 * MIL_NUM_CU / split_kg(R, L)) / (3 * NJ);
    if (smax < 1) smax = 1;
    int kc = ((R + smax - 1) / smax + GB_BKR - 1) / GB_BKR * GB_BKR;
    if (kc < GB_BKR) kc = GB_BKR;
    *KC_out = kc;
    return (R + kc - 1) / kc;
}

extern "C" int mil_abi_version(void) { return 6; }

// Small batches (the authors train with ONE bag per GPU: R = 1 000 - 15 000 rows): 128-row tiles would leave most CUs
// idle (8 workgroups for 1024 patches, each walking all of K: the kernel takes its full ~100 us for 1/32 of the
// work).  This form uses 32-row tiles and 12 waves: wave (c, u) owns ONE accumulator tile - d-chunk c of V (u = 0) or
// of U (u = 1) - so the four SIMDs carry three waves each (six waves with two tiles would put two on some SIMDs
// and one on others); the U waves hand sigmoid(U) to their V partners through LDS for the gate product, and the
// scores still complete inside the workgroup.  Register-staged, two register sets: every load has two iterations to
// land (one L2 round trip is longer than one slice of MFMAs here).
#define GS_TM 32
#define GS_LS 36
#define GS_THREADS 768
// RT row tiles of 32 per workgroup (1 .. 3): the launch picks the smallest RT that covers the rows in ONE round of the
// grid - 384 workgroups of 32 rows are two rounds at one workgroup per CU (120 KB of LDS), 192 of 64 rows one round of
// twice the length: a bucket of 12 288 rows 57 -> 46 us.  Every wave then carries RT accumulator tiles against the same B
// fragments.
template <bool DROP, int RT>
__global__ __launch_bounds__(GS_THREADS) void k_gate_fwd_r32(const float* __restrict__ x, const float* __restrict__ Wv,
                                                             const float* __restrict__ bv, const float* __restrict__ Wu,
                                                             const float* __restrict__ bu, const float* __restrict__ wvec,
                                                             const float* __restrict__ battn, float* __restrict__ scores,
                                                             float* __restrict__ gates, int R, int L,
                                                             const uint32_t* __restrict__ xbits, float xscale,
                                                             const int32_t* __restrict__ rows_dev) {
    // bucketed batches (one ragged bag per step: R is the capacity the launch is sized for): tiles beyond the true row
    // count on the device have no reader - the tile map, the pool and the weight gradient all stop at that count
    constexpr int TM = GS_TM * RT;
    if (rows_dev != nullptr && (int)(blockIdx.x * TM) >= __builtin_amdgcn_readfirstlane(rows_dev[0])) return;
    __shared__ __attribute__((aligned(16))) float smem[2 * (TM + GF_NG) * GS_LS];
    float* xs = smem;                            // [2][32 RT][36]
    float* ws = smem + 2 * TM * GS_LS;           // [2][384][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = wave >> 1, isu = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * TM;
    const float* wsrc[4];
    int wdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + GS_THREADS * i, wrow = id >> 3, ch = id & 7;
        wsrc[i] = (wrow < 192 ? Wv + (size_t)wrow * L : Wu + (size_t)(wrow - 192) * L) + 4 * ch;
        wdst[i] = wrow * GS_LS + 4 * ch;
    }
    const int xrow = (tid % (256 * RT)) >> 3, xch = tid & 7;          // threads < 256 RT stage the x rows (768 = 3 x 256)
    const float* xsrc = x + (size_t)min(row0 + xrow, R - 1) * L + 4 * xch;
    const uint32_t* msrc = DROP ? xbits + (size_t)min(row0 + xrow, R - 1) * (L / 32) : nullptr;
    const int xdst = xrow * GS_LS + 4 * xch;
    f32x4 wreg[2][4], xreg[2];
    unsigned mreg[2] = {0, 0};
    auto gload = [&](int set, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[set][i] = *reinterpret_cast<const f32x4*>(wsrc[i] + k0);
        xreg[set] = *reinterpret_cast<const f32x4*>(xsrc + k0);
        if (DROP) mreg[set] = msrc[k0 >> 5];                     // a slice = 32 columns = one word of keep bits per row
    };
    auto swrite = [&](int set, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(ws + buf * GF_NG * GS_LS + wdst[i]) = wreg[set][i];
        if (tid < 256 * RT) {
            f32x4 v = xreg[set];
            if (DROP) {
                const unsigned m = mreg[set] >> (4 * xch);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = keep_if(v[e], m, e) * xscale;
            }
            *reinterpret_cast<f32x4*>(xs + buf * TM * GS_LS + xdst) = v;
        }
    };
    f32x16 acc[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    const int nslice = L / GF_BK;
    gload(0, 0);
    swrite(0, 0);
    gload(1, min(1, nslice - 1) * GF_BK);
    gload(0, min(2, nslice - 1) * GF_BK);
    __syncthreads();
    auto iteration = [&](int s, int set) {                       // set = (s + 1) & 1
        const int buf = s & 1;
        const float* xa = xs + buf * TM * GS_LS + r * GS_LS + 4 * h;
        const float* wb = ws + buf * GF_NG * GS_LS + (192 * isu + 32 * c + r) * GS_LS + 4 * h;
        f32x4 fa[RT][4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < RT; ++q) fa[q][t] = *reinterpret_cast<const f32x4*>(xa + q * GS_TM * GS_LS + 8 * t);
            fb[t] = *reinterpret_cast<const f32x4*>(wb + 8 * t);
        }
        swrite(set, buf ^ 1);                                    // slice s+1: registers -> the other buffer
        gload(set, min(s + 3, nslice - 1) * GF_BK);              // the same registers take slice s+3
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int q = 0; q < RT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][t][jj], fb[t][jj], acc[q], 0, 0, 0);
        __syncthreads();
    };
    for (int s = 0; s < nslice; s += 2) {
        iteration(s, 1);
        if (s + 1 < nslice) iteration(s + 1, 0);
    }
    const int d = 32 * c + r;
    float* uex = smem;                           // [6][32][33]: sigmoid(U) tiles for the V partners
    float* sred = smem + 6 * 32 * 33;            // [6][32]
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        float gv[16];
        if (q > 0) __syncthreads();              // the previous row tile's uex / sred have been read
        if (isu) {
            const float bud = bu[d];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                gv[i] = fast_sigmoid(acc[q][i] + bud);
                uex[(c * 32 + mfma32_row(i, h)) * 33 + r] = gv[i];
            }
        } else {
            const float bvd = bv[d];
#pragma unroll
            for (int i = 0; i < 16; ++i) gv[i] = fast_tanh(acc[q][i] + bvd);
        }
        if (gates != nullptr) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gr = row0 + GS_TM * q + mfma32_row(i, h);
                if (gr < R) gates[(size_t)gr * GF_NG + 192 * isu + d] = gv[i];
            }
        }
        __syncthreads();
        if (!isu) {
            const float wd = wvec[d];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lr = mfma32_row(i, h);
                const float part = half_allsum(wd * gv[i] * uex[(c * 32 + lr) * 33 + r]);
                if (r == 0) sred[c * GS_TM + lr] = part;
            }
        }
        __syncthreads();
        if (tid < GS_TM && row0 + GS_TM * q + tid < R) {
            float sc = battn[0];
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) sc += sred[cc * GS_TM + tid];
            scores[row0 + GS_TM * q + tid] = sc;
        }
    }
}

// Tile quantisation: with one 128-row workgroup per CU, R = k * 256 * 128 + (a few rows) costs a whole extra round of
// the grid for one workgroup (config 3: 32 bags x (1024 patches + 2 tokens) = 256.5 tiles -> 2x the kernel time).
// When the rows beyond a whole number of rounds fit the few-rows linear (<= 64), they take that path instead:
// V and U through mil_linear_small_fwd straight into the gates buffer, then one tiny scoring launch.
__global__ __launch_bounds__(64) void k_gate_tail_scores(const float* __restrict__ gates, const float* __restrict__ wvec,
                                                         const float* __restrict__ battn, float* __restrict__ scores) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const float* gr = gates + (size_t)row * GF_NG;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) v += wvec[64 * q + lane] * gr[64 * q + lane] * gr[192 + 64 * q + lane];
    v = wave_allsum(v);
    if (lane == 0) scores[row] = v + battn[0];
}

static inline int gate_tail_rows(int R, int tiles_per_round) {
    const int tail = R % GF_TM, full = R / GF_TM;
    return (tail >= 1 && tail <= MIL_SMALL_ROWS && full >= tiles_per_round && full % tiles_per_round == 0) ? tail : 0;
}

static int launch_gate_fwd_r32(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu,
                               const float* w, const float* b, float* scores, float* gates, int R, int L,
                               const uint32_t* xbits, float xscale, hipStream_t st, const int32_t* rows_dev = nullptr) {
    // row tiles per workgroup: the smallest count that covers R in one round of one-workgroup-per-CU launches
    const int tiles = (R + GS_TM - 1) / GS_TM;
    const int rt = tiles <= MIL_NUM_CU ? 1 : tiles <= 2 * MIL_NUM_CU ? 2 : 3;
    const dim3 grid((R + GS_TM * rt - 1) / (GS_TM * rt));
    const float xs = xbits ? xscale : 1.0f;
#define R32_LAUNCH(D_, RT_) hipLaunchKernelGGL((k_gate_fwd_r32<D_, RT_>), grid, dim3(GS_THREADS), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, xbits, xs, rows_dev)
    if (xbits) { if (rt == 1) R32_LAUNCH(true, 1); else if (rt == 2) R32_LAUNCH(true, 2); else R32_LAUNCH(true, 3); }
    else { if (rt == 1) R32_LAUNCH(false, 1); else if (rt == 2) R32_LAUNCH(false, 2); else R32_LAUNCH(false, 3); }
#undef R32_LAUNCH
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

int dropout_keep_bits_pair(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed, uint64_t offset,
                           const int32_t* offset_dev, void* stream);
int dropout_keep_bits_pair_tilemap(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed,
                                   uint64_t offset, const int32_t* offset_dev, const TileMapJob& tm, void* stream);
extern "C" int mil_build_tile_map(const int32_t* bag_len, int B, int32_t* tile_map, int32_t* bag_tile_off, int32_t* rows_out,
                                  int T_cap, void* stream);      // dropout.hip

static int gate_scores_fwd_impl(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu,
                                const float* w, const float* b, float* scores, float* gates, int R, int L, int D,
                                const uint32_t* xbits, float xscale, const GateFwdGen* gen, void* stream,
                                const GateFwdPool* pool = nullptr, int* fused = nullptr, const int32_t* rows_dev = nullptr,
                                const TileMapJob* tmap = nullptr) {
    if (!x || !Wv || !bv || !Wu || !bu || !w || !b || !scores) return MIL_EINVAL;
    if (D != MIL_GATE_D || L <= 0 || (L % GF_BK) != 0 || R < 0) return MIL_EINVAL;
    if (R == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    // tmap: the step's tile map is still to be built (it writes rows_dev, which this launch reads): it rides on the generator
    // launch where there is one, and is a launch of its own in front of the forward otherwise
    bool tmap_done = tmap == nullptr;
    auto tmap_alone = [&]() -> int {
        if (tmap_done) return MIL_OK;
        tmap_done = true;
        return mil_build_tile_map(tmap->bag_len, tmap->B, tmap->tile_map, tmap->bag_tile_off, tmap->rows_out, tmap->T_cap, stream);
    };
    const bool r32 = (R + GF_TM - 1) / GF_TM < (3 * MIL_NUM_CU) / 4;
    int tail = (!r32 && gates != nullptr) ? gate_tail_rows(R, MIL_NUM_CU) : 0;   // the tail path keeps V, U in `gates`
    if (!r32 && tail == 0) {
        // A last round that would hold only a few workgroups (T text tokens per bag appended to 32 x 1024 patches: 259
        // tiles on 256 CUs) costs a whole round: every row beyond the whole rounds, up to 1024 of them, goes through the
        // 32-row kernel instead (10 workgroups of a quarter of the time)
        const int per_round = GF_TM * MIL_NUM_CU;
        const int over = R % per_round;
        if (R >= per_round && over > 0 && over <= 1024) tail = -over;       // negative: "large tail", always the 32-row kernel
    }
    const bool big_tail = tail < 0;
    if (big_tail) tail = -tail;
#if defined(GF_NO_FWD2)
    const bool fwd2 = false;
#else
    const bool fwd2 = L <= 4096;                 // 128 rows x L floats must stay inside the 32-bit buffer offsets (and int math)
#endif
    if (fused != nullptr) *fused = 0;
    const GateFwdPool nopool{};
    // the pool partial pass rides in the forward's epilogue when every row goes through k_gate_fwd2 and every tile of the
    // map is a full, aligned 32-row tile (T * 32 == R: tile t = rows 32 t ..)
    const bool pool_in = pool != nullptr && fused != nullptr && !r32 && tail == 0 && fwd2 && L == 512 && (R % 32) == 0 &&
                         pool->T * MIL_POOL_TILE == R && pool->tile_map && pool->partials && pool->Wf && pool->hrow;
    if (gen != nullptr) {
        // the forward kernel can draw the keep bits itself when every row goes through k_gate_fwd2 and a workgroup's
        // [128][L/32] words fit its LDS slot; otherwise the stand-alone generator runs first
        const bool in_kernel = !r32 && tail == 0 && fwd2 && L <= 1024 && (L % 128) == 0;
        if (in_kernel) {
            { const int rc_ = tmap_alone(); if (rc_ != MIL_OK) return rc_; }
            const int grid = (R + GF_TM - 1) / GF_TM;
            if (pool_in && gen->mbits_out != nullptr) {
                hipLaunchKernelGGL((k_gate_fwd2<true, true, 2>), dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates,
                                   R, L, gen->xbits_out, xscale, *gen, *pool);
                *fused = 1;
            } else {
                hipLaunchKernelGGL((k_gate_fwd2<true, true, 0>), dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates,
                                   R, L, gen->xbits_out, xscale, *gen, nopool);
            }
            MIL_CHECK_LAUNCH();
            return MIL_OK;
        }
        int rc;
        if (gen->mbits_out != nullptr && !tmap_done) {
            rc = dropout_keep_bits_pair_tilemap(gen->xbits_out, R, gen->mbits_out, gen->B, L,
                                                ((uint64_t)gen->seed_hi << 32) | gen->seed_lo,
                                                ((uint64_t)gen->mseed_hi << 32) | gen->mseed_lo, gen->offset, gen->offset_dev, *tmap,
                                                stream);
            tmap_done = true;
        } else if (gen->mbits_out != nullptr)
            rc = dropout_keep_bits_pair(gen->xbits_out, R, gen->mbits_out, gen->B, L, ((uint64_t)gen->seed_hi << 32) | gen->seed_lo,
                                        ((uint64_t)gen->mseed_hi << 32) | gen->mseed_lo, gen->offset, gen->offset_dev, stream);
        else
            rc = mil_dropout_keep_bits(gen->xbits_out, R, L, 0.5f, ((uint64_t)gen->seed_hi << 32) | gen->seed_lo, gen->offset,
                                       gen->offset_dev, stream);
        if (rc != MIL_OK) return rc;
        xbits = gen->xbits_out;
    }
    { const int rc_ = tmap_alone(); if (rc_ != MIL_OK) return rc_; }
    if (r32) {
        // fewer 128-row tiles than 3/4 of the CUs: 32-row tiles (4x the workgroups, each a quarter of the time)
        return launch_gate_fwd_r32(x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, xbits, xscale, st, rows_dev);
    }
    const int Rm = R - tail;
    const int grid = (Rm + GF_TM - 1) / GF_TM;
    const GateFwdGen nogen{};
    if (pool_in && xbits && pool->mbits) {
        hipLaunchKernelGGL((k_gate_fwd2<true, false, 2>), dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates, Rm, L,
                           xbits, xscale, nogen, *pool);
        *fused = 1;
    } else if (pool_in && !xbits && !pool->mbits) {
        hipLaunchKernelGGL((k_gate_fwd2<false, false, 2>), dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates, Rm, L,
                           xbits, 1.0f, nogen, *pool);
        *fused = 1;
    } else if (fwd2 && xbits)
        hipLaunchKernelGGL((k_gate_fwd2<true, false, 0>), dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates, Rm, L,
                           xbits, xscale, nogen, nopool);
    else if (fwd2)
        hipLaunchKernelGGL((k_gate_fwd2<false, false, 0>), dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates, Rm, L,
                           xbits, 1.0f, nogen, nopool);
    else if (xbits)
        hipLaunchKernelGGL(k_gate_fwd<true>, dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates, Rm, L, xbits,
                           xscale);
    else
        hipLaunchKernelGGL(k_gate_fwd<false>, dim3(grid), dim3(512), 0, st, x, Wv, bv, Wu, bu, w, b, scores, gates, Rm, L, xbits,
                           1.0f);
    MIL_CHECK_LAUNCH();
    if (tail > 0 && (xbits || big_tail)) {
        // train mode: the few rows beyond whole rounds go through the 32-row kernel (it applies the keep bits while staging)
        return launch_gate_fwd_r32(x + (size_t)Rm * L, Wv, bv, Wu, bu, w, b, scores + Rm,
                                   gates ? gates + (size_t)Rm * GF_NG : nullptr, tail, L,
                                   xbits ? xbits + (size_t)Rm * (L / 32) : nullptr, xscale, st);
    }
    if (tail > 0) {
        const float* xt = x + (size_t)Rm * L;
        float* gt = gates + (size_t)Rm * GF_NG;
        int rc = mil_linear_small_fwd(xt, L, Wv, L, bv, 1 /* tanh */, nullptr, 0, gt, GF_NG, tail, MIL_GATE_D, L, stream);
        if (rc != MIL_OK) return rc;
        rc = mil_linear_small_fwd(xt, L, Wu, L, bu, 4 /* sigmoid */, nullptr, 0, gt + MIL_GATE_D, GF_NG, tail, MIL_GATE_D, L,
                                  stream);
        if (rc != MIL_OK) return rc;
        hipLaunchKernelGGL(k_gate_tail_scores, dim3(tail), dim3(64), 0, st, gt, w, b, scores + Rm);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

extern "C" int mil_gate_scores_fwd(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu,
                                   const float* w, const float* b, float* scores, float* gates, int R, int L, int D,
                                   const uint32_t* xbits, float xscale, void* stream) {
    return gate_scores_fwd_impl(x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, D, xbits, xscale, nullptr, stream);
}

extern "C" int mil_gate_scores_fwd_draw(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu,
                                        const float* w, const float* b, float* scores, float* gates, int R, int L, int D,
                                        uint32_t* xbits_out, float xscale, uint32_t* mbits_out, int B, uint64_t seed,
                                        uint64_t mseed, uint64_t offset, const int32_t* offset_dev, void* stream) {
    if (!xbits_out || (L % 64) != 0 || (mbits_out && B <= 0)) return MIL_EINVAL;
    GateFwdGen g{};
    g.xbits_out = xbits_out;
    g.mbits_out = mbits_out;
    g.B = B;
    g.seed_lo = (uint32_t)seed;
    g.seed_hi = (uint32_t)(seed >> 32);
    g.mseed_lo = (uint32_t)mseed;
    g.mseed_hi = (uint32_t)(mseed >> 32);
    g.offset = offset;
    g.offset_dev = offset_dev;
    return gate_scores_fwd_impl(x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, D, nullptr, xscale, &g, stream);
}

// Internal (step.hip): the gate forward of a bucketed batch (rows_dev = true row count on the device; R = capacity).
int gate_fwd_rows_dev(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu, const float* w,
                      const float* b, float* scores, float* gates, int R, int L, int draw, uint32_t* xbits, float xscale,
                      uint32_t* mbits, int B, uint64_t seed, uint64_t mseed, uint64_t offset, const int32_t* offset_dev,
                      const int32_t* rows_dev, void* stream, const TileMapJob* tmap) {
    if (draw) {
        if (!xbits || (L % 64) != 0 || (mbits && B <= 0)) return MIL_EINVAL;
        GateFwdGen g{};
        g.xbits_out = xbits;
        g.mbits_out = mbits;
        g.B = B;
        g.seed_lo = (uint32_t)seed;
        g.seed_hi = (uint32_t)(seed >> 32);
        g.mseed_lo = (uint32_t)mseed;
        g.mseed_hi = (uint32_t)(mseed >> 32);
        g.offset = offset;
        g.offset_dev = offset_dev;
        return gate_scores_fwd_impl(x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, MIL_GATE_D, nullptr, xscale, &g, stream, nullptr,
                                    nullptr, rows_dev, tmap);
    }
    return gate_scores_fwd_impl(x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, MIL_GATE_D, xbits, xscale, nullptr, stream, nullptr,
                                nullptr, rows_dev, tmap);
}

// Internal (step.hip): gate forward with the pool partial pass in its epilogue when the batch allows it; *fused says
// whether partials / hrow were produced (otherwise the caller runs the stand-alone pool pass).  seed / mseed / offset as
// mil_gate_scores_fwd_draw when draw != 0 (train mode, keep bits drawn by the kernel), else xbits / mbits are inputs.
int gate_fwd_with_pool(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu, const float* w,
                       const float* b, float* scores, float* gates, int R, int L, int draw, uint32_t* xbits, float xscale,
                       uint32_t* mbits, float mscale, int B, uint64_t seed, uint64_t mseed, uint64_t offset,
                       const int32_t* offset_dev, const int32_t* tile_map, int T, float* partials, const float* Wf, float* hrow,
                       int* fused, void* stream) {
    GateFwdPool pl{};
    pl.tile_map = tile_map;
    pl.partials = partials;
    pl.T = T;
    pl.Wf = Wf;
    pl.hrow = hrow;
    pl.mbits = draw ? nullptr : mbits;
    pl.mscale = mscale;
    if (draw) {
        if (!xbits || !mbits || (L % 64) != 0 || B <= 0) return MIL_EINVAL;
        GateFwdGen g{};
        g.xbits_out = xbits;
        g.mbits_out = mbits;
        g.B = B;
        g.seed_lo = (uint32_t)seed;
        g.seed_hi = (uint32_t)(seed >> 32);
        g.mseed_lo = (uint32_t)mseed;
        g.mseed_hi = (uint32_t)(mseed >> 32);
        g.offset = offset;
        g.offset_dev = offset_dev;
        return gate_scores_fwd_impl(x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, MIL_GATE_D, nullptr, xscale, &g, stream, &pl, fused);
    }
    return gate_scores_fwd_impl(x, Wv, bv, Wu, bu, w, b, scores, gates, R, L, MIL_GATE_D, xbits, xscale, nullptr, stream, &pl, fused);
}

static int launch_pool_partial(const float* x, const float* scores, const int32_t* tile_map, int T, int L,
                               float* partials, const float* Wf, int C, float* hrow, hipStream_t st,
                               const uint32_t* xbits = nullptr, float xscale = 1.0f, const uint32_t* mbits = nullptr,
                               float mscale = 1.0f) {
    if (T <= 0) return MIL_OK;
    const bool nt = (size_t)T * MIL_POOL_TILE * L * sizeof(float) > MIL_STREAM_BYTES;
#define POOL_LAUNCH(NQ_) do { \
        if (nt) hipLaunchKernelGGL((k_pool_partial<NQ_, true>), dim3(T), dim3(256), 0, st, x, scores, tile_map, partials, L, Wf, C, hrow, xbits, xscale, mbits, mscale); \
        else hipLaunchKernelGGL((k_pool_partial<NQ_, false>), dim3(T), dim3(256), 0, st, x, scores, tile_map, partials, L, Wf, C, hrow, xbits, xscale, mbits, mscale); } while (0)
    switch (L / 256) {
        case 1: POOL_LAUNCH(1); break;
        case 2: POOL_LAUNCH(2); break;
        case 3: POOL_LAUNCH(3); break;
        default: POOL_LAUNCH(4); break;
    }
#undef POOL_LAUNCH
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// Tile map on the device (ragged batches whose lengths change every step): one workgroup scans the B bag lengths and
// writes tile_map [T_cap][4], bag_tile_off [B + 1] and rows_out [1] = total rows.  Tiles beyond the last real one are
// padding: {0, 0, 0, 0} (the pool kernels emit a neutral partial for nrows == 0).  T_cap >= sum ceil(len / 32).
__global__ __launch_bounds__(256) void k_build_tile_map(const int32_t* __restrict__ bag_len, int B, int32_t* __restrict__ tile_map,
                                                        int32_t* __restrict__ bag_tile_off, int32_t* __restrict__ rows_out,
                                                        int T_cap) {
    build_tile_map_block(bag_len, B, tile_map, bag_tile_off, rows_out, T_cap);
}

extern "C" int mil_build_tile_map(const int32_t* bag_len, int B, int32_t* tile_map, int32_t* bag_tile_off, int32_t* rows_out,
                                  int T_cap, void* stream) {
    if (!bag_len || !tile_map || !bag_tile_off || !rows_out || B <= 0 || B > 1024 || T_cap < 0) return MIL_EINVAL;
    hipLaunchKernelGGL(k_build_tile_map, dim3(1), dim3(256), 0, (hipStream_t)stream, bag_len, B, tile_map, bag_tile_off, rows_out,
                       T_cap);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_attn_pool_partial_h(const float* x, const float* scores, const int32_t* tile_map, int T, int L,
                                       float* partials, const float* Wf, int C, float* hrow, const uint32_t* xbits,
                                       float xscale, const uint32_t* mbits, float mscale, void* stream) {
    if (!x || !scores || !tile_map || !partials || !Wf || !hrow) return MIL_EINVAL;
    if (L <= 0 || (L % 256) != 0 || L > 1024 || T < 0 || C <= 0 || C > 4) return MIL_EINVAL;
    return launch_pool_partial(x, scores, tile_map, T, L, partials, Wf, C, hrow, (hipStream_t)stream, xbits, xscale, mbits,
                               mscale);
}

extern "C" int mil_attn_pool_bwd_from_h(const float* scores, const float* lse, const float* hrow, const float* dz,
                                        const float* cdot, const int32_t* tile_map, int T, int C, float* ds, void* stream) {
    if (!scores || !lse || !hrow || !dz || !cdot || !tile_map || !ds || T < 0 || C <= 0 || C > 4) return MIL_EINVAL;
    if (T == 0) return MIL_OK;
    hipLaunchKernelGGL(k_pool_ds_from_h, dim3((T + 7) / 8), dim3(256), 0, (hipStream_t)stream, scores, lse, hrow, dz, cdot,
                       tile_map, T, C, ds);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_attn_pool_partial(const float* x, const float* scores, const int32_t* tile_map, int T, int L,
                                     float* partials, const uint32_t* xbits, float xscale, void* stream) {
    if (!x || !scores || !tile_map || !partials) return MIL_EINVAL;
    if (L <= 0 || (L % 256) != 0 || L > 1024 || T < 0) return MIL_EINVAL;
    return launch_pool_partial(x, scores, tile_map, T, L, partials, nullptr, 0, nullptr, (hipStream_t)stream, xbits, xscale);
}

extern "C" int mil_attn_pool_fwd(const float* x, const float* scores, const int32_t* tile_map,
                                 const int32_t* bag_tile_off, int T, int B, int L, float* partials, float* M,
                                 float* lse, const uint32_t* xbits, float xscale, void* stream) {
    if (!x || !scores || !tile_map || !bag_tile_off || !partials || !M || !lse) return MIL_EINVAL;
    if (L <= 0 || (L % 256) != 0 || L > 1024 || B < 0 || T < 0) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rc = launch_pool_partial(x, scores, tile_map, T, L, partials, nullptr, 0, nullptr, st, xbits, xscale);
    if (rc != MIL_OK) return rc;
    if (B > 0) {
        hipLaunchKernelGGL(k_pool_merge, dim3(B, L / 128), dim3(256), 0, st, partials, bag_tile_off, M, lse, L, T);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

extern "C" int mil_attn_pool_bwd(const float* x, const float* scores, const float* lse, const float* dM,
                                 const float* cdot, const int32_t* tile_map, int T, int L, float* ds, float* dx,
                                 const uint32_t* xbits, float xscale, void* stream) {
    if (!x || !scores || !lse || !dM || !cdot || !tile_map || !ds) return MIL_EINVAL;
    if (L <= 0 || (L % 256) != 0 || L > 1024 || T < 0) return MIL_EINVAL;
    if (T == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (L / 256) {
        case 1: hipLaunchKernelGGL(k_pool_bwd_ds<1>, dim3(T), dim3(256), 0, st, x, scores, lse, dM, cdot, tile_map, ds, dx, L, xbits, xscale); break;
        case 2: hipLaunchKernelGGL(k_pool_bwd_ds<2>, dim3(T), dim3(256), 0, st, x, scores, lse, dM, cdot, tile_map, ds, dx, L, xbits, xscale); break;
        case 3: hipLaunchKernelGGL(k_pool_bwd_ds<3>, dim3(T), dim3(256), 0, st, x, scores, lse, dM, cdot, tile_map, ds, dx, L, xbits, xscale); break;
        default: hipLaunchKernelGGL(k_pool_bwd_ds<4>, dim3(T), dim3(256), 0, st, x, scores, lse, dM, cdot, tile_map, ds, dx, L, xbits, xscale); break;
    }
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

template <bool XB16>
static void launch_gate_bwd_dw(const void* x, const float* gates, const float* ds, const float* w, float* part, float* pbias,
                               int R, int L, int kc, int NJ, int S, const uint32_t* xbits, hipStream_t st,
                               const int32_t* rows_dev = nullptr) {
    const dim3 grid(S * 3 * NJ);
    if (split_kg(R, L) == 2) {
        if (xbits) hipLaunchKernelGGL((k_gate_bwd_dw<XB16, 2, true>), grid, dim3(512), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits, rows_dev);
        else hipLaunchKernelGGL((k_gate_bwd_dw<XB16, 2, false>), grid, dim3(512), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits, rows_dev);
    } else {
        if (xbits) hipLaunchKernelGGL((k_gate_bwd_dw<XB16, 1, true>), grid, dim3(256), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits, rows_dev);
        else hipLaunchKernelGGL((k_gate_bwd_dw<XB16, 1, false>), grid, dim3(256), 0, st, x, gates, ds, w, part, pbias, R, L, kc, NJ, xbits, rows_dev);
    }
}

// fp32 x, two K groups per workgroup, offsets within 32 bits: the low-VALU kernel (k_gate_bwd_dw2)
static inline bool use_dw2(int R, int L) {
#if defined(GB_NO_DW2)
    return false;
#endif
    return split_kg(R, L) == 2 && (long long)R * L < (1ll << 29);
}
static inline size_t gate_bwd_ws_floats(int S, int L) { return (size_t)S * GF_NG * L + (size_t)S * (L / 128) * 4 * 192; }

extern "C" size_t mil_gate_bwd_workspace_floats(int R, int L) {
    if (R <= 0 || L <= 0 || (L % 128) != 0) return 0;
    int kc;
    const int S = split_plan(R, L, &kc);
    return gate_bwd_ws_floats(S, L);
}

// The two launches of mil_gate_bwd_params as separate entry points (bench.py times the MFMA kernel alone).
static int gate_bwd_partials_impl(const float* x, const float* gates, const float* ds, const float* w, int R, int L,
                                  int D, float* workspace, size_t workspace_floats, const uint32_t* xbits,
                                  const int32_t* rows_dev, void* stream) {
    if (!x || !gates || !ds || !w || !workspace) return MIL_EINVAL;
    if (D != MIL_GATE_D || L <= 0 || (L % 128) != 0 || R <= 0) return MIL_EINVAL;
    int kc;
    const int S = split_plan(R, L, &kc);
    if (workspace_floats < gate_bwd_ws_floats(S, L)) return MIL_ENOSPC;
    const int NJ = L / 128;
    if (use_dw2(R, L)) {
        float* pb = workspace + (size_t)S * GF_NG * L;
        if (xbits)
            hipLaunchKernelGGL(k_gate_bwd_dw2<true>, dim3(S * 3 * NJ), dim3(512), 0, (hipStream_t)stream, x, gates, ds, w, workspace, pb,
                               R, L, kc, NJ, xbits, rows_dev);
        else
            hipLaunchKernelGGL(k_gate_bwd_dw2<false>, dim3(S * 3 * NJ), dim3(512), 0, (hipStream_t)stream, x, gates, ds, w, workspace, pb,
                               R, L, kc, NJ, xbits, rows_dev);
        MIL_CHECK_LAUNCH();
        return MIL_OK;
    }
    launch_gate_bwd_dw<false>((const void*)x, gates, ds, w, workspace, workspace + (size_t)S * GF_NG * L, R, L, kc, NJ, S, xbits,
                              (hipStream_t)stream, rows_dev);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gate_bwd_partials(const float* x, const float* gates, const float* ds, const float* w, int R, int L,
                                     int D, float* workspace, size_t workspace_floats, const uint32_t* xbits,
                                     void* stream) {
    return gate_bwd_partials_impl(x, gates, ds, w, R, L, D, workspace, workspace_floats, xbits, nullptr, stream);
}
// rows_dev: device int32 holding the TRUE number of rows (<= R).  R then is the bucket the launch is sized for (grid,
// split-K plan, workspace); rows beyond the true count contribute nothing.  One launch configuration - one captured
// graph - serves every batch of the bucket.
extern "C" int mil_gate_bwd_partials_rows(const float* x, const float* gates, const float* ds, const float* w, int R, int L,
                                          int D, float* workspace, size_t workspace_floats, const uint32_t* xbits,
                                          const int32_t* rows_dev, void* stream) {
    return gate_bwd_partials_impl(x, gates, ds, w, R, L, D, workspace, workspace_floats, xbits, rows_dev, stream);
}

extern "C" int mil_gate_bwd_reduce(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                                   float* dw, float* db, int accumulate, float xscale, void* stream) {
    if (!workspace || !dWv || !dbv || !dWu || !dbu || !dw || !db) return MIL_EINVAL;
    if (L <= 0 || (L % 128) != 0 || R <= 0) return MIL_EINVAL;
    int kc;
    const int S = split_plan(R, L, &kc);
    const int nthreads = GF_NG * (L / 4) + GR_NB * (3 * 192 + 1);
    hipLaunchKernelGGL(k_gate_bwd_reduce, dim3((nthreads + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace,
                       workspace + (size_t)S * GF_NG * L, S, use_dw2(R, L) ? S * (L / 128) : S, L, dWv, dbv, dWu, dbu, dw, db,
                       accumulate, xscale);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gate_bwd_params_head(const float* x, const float* gates, const float* ds, const float* w, int R, int L,
                                        int D, float* workspace, size_t workspace_floats, float* dWv, float* dbv,
                                        float* dWu, float* dbu, float* dw, float* db, int accumulate, const float* dz,
                                        const float* M, float* dWf, float* dbf, int B, int C, const float* loss_bag,
                                        float* loss_out, const uint32_t* xbits, float xscale, void* stream) {
    if (!dWv || !dbv || !dWu || !dbu || !dw || !db || !dz || !M || !dWf || !dbf) return MIL_EINVAL;
    if (B <= 0 || C <= 0 || C > 32 || (loss_bag && !loss_out)) return MIL_EINVAL;
    const int rc = mil_gate_bwd_partials(x, gates, ds, w, R, L, D, workspace, workspace_floats, xbits, stream);
    if (rc != MIL_OK) return rc;
    int kc;
    const int S = split_plan(R, L, &kc);
    const int nthreads = GF_NG * (L / 4) + GR_NB * (3 * 192 + 1);
    const int nred = (nthreads + 255) / 256, nhead = C * ((L + 63) / 64) + 1;
    const HeadBwdArgs head{dz, M, dWf, dbf, loss_bag, loss_out, B, L, C, accumulate};
    hipLaunchKernelGGL(k_gate_bwd_reduce, dim3(nred + nhead), dim3(256), 0, (hipStream_t)stream, workspace,
                       workspace + (size_t)S * GF_NG * L, S, use_dw2(R, L) ? S * (L / 128) : S, L, dWv, dbv, dWu, dbu, dw, db,
                       accumulate, xbits ? xscale : 1.0f, nred, head);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// The reduce launch of mil_gate_bwd_params_head alone (split-K fold + the head's parameter gradients as appended
// workgroups), for a caller that issued mil_gate_bwd_partials itself (csrc/step.hip).
extern "C" int mil_gate_bwd_reduce_head(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                                        float* dw, float* db, int accumulate, float xscale, const float* dz, const float* M,
                                        float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out,
                                        void* stream) {
    if (!workspace || !dWv || !dbv || !dWu || !dbu || !dw || !db || !dz || !M || !dWf || !dbf) return MIL_EINVAL;
    if (L <= 0 || (L % 128) != 0 || R <= 0 || B <= 0 || C <= 0 || C > 32 || (loss_bag && !loss_out)) return MIL_EINVAL;
    int kc;
    const int S = split_plan(R, L, &kc);
    const int nthreads = GF_NG * (L / 4) + GR_NB * (3 * 192 + 1);
    const int nred = (nthreads + 255) / 256, nhead = C * ((L + 63) / 64) + 1;
    const HeadBwdArgs head{dz, M, dWf, dbf, loss_bag, loss_out, B, L, C, accumulate};
    hipLaunchKernelGGL(k_gate_bwd_reduce, dim3(nred + nhead), dim3(256), 0, (hipStream_t)stream, workspace,
                       workspace + (size_t)S * GF_NG * L, S, use_dw2(R, L) ? S * (L / 128) : S, L, dWv, dbv, dWu, dbu, dw, db,
                       accumulate, xscale, nred, head);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

int gate_bwd_reduce_head_adam_impl(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                                   float* dw, float* db, int accumulate, float xscale, const float* dz, const float* M,
                                   float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out,
                                   float* param_flat, const float* grad_flat, size_t n_param, float* exp_avg,
                                   float* exp_avg_sq, int step, const int* step_dev, float lr, const float* lr_dev, float beta1,
                                   float beta2, float eps, float weight_decay, float grad_scale, void* stream, int* done = nullptr);

// mil_gate_bwd_reduce_head with Adam applied by the threads that produce the final gradients (world size 1: nothing sits
// between the gradient and the update): param_flat / exp_avg / exp_avg_sq are indexed like grad_flat, in which dWv .. dbf all
// lie; `step` >= 1 is the update's number (bias corrections on the host).  Saves the Adam launch of the image-only step.
extern "C" int mil_gate_bwd_reduce_head_adam(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu,
                                             float* dbu, float* dw, float* db, int accumulate, float xscale, const float* dz,
                                             const float* M, float* dWf, float* dbf, int B, int C, const float* loss_bag,
                                             float* loss_out, float* param_flat, const float* grad_flat, size_t n_param,
                                             float* exp_avg, float* exp_avg_sq, int step, float lr, float beta1, float beta2,
                                             float eps, float weight_decay, float grad_scale, void* stream) {
    return gate_bwd_reduce_head_adam_impl(workspace, R, L, dWv, dbv, dWu, dbu, dw, db, accumulate, xscale, dz, M, dWf, dbf, B, C,
                                          loss_bag, loss_out, param_flat, grad_flat, n_param, exp_avg, exp_avg_sq, step, nullptr,
                                          lr, nullptr, beta1, beta2, eps, weight_decay, grad_scale, stream);
}

// step_dev != NULL: the update's number is (*step_dev + 1), read on the device (hipGraph replay); the counter is advanced by
// this launch itself when `done` (a zeroed sign-off word) is given, else by the caller afterwards.  Internal (step.hip).
int gate_bwd_reduce_head_adam_impl(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                                   float* dw, float* db, int accumulate, float xscale, const float* dz, const float* M,
                                   float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out,
                                   float* param_flat, const float* grad_flat, size_t n_param, float* exp_avg,
                                   float* exp_avg_sq, int step, const int* step_dev, float lr, const float* lr_dev, float beta1,
                                   float beta2, float eps, float weight_decay, float grad_scale, void* stream, int* done) {
    if (!workspace || !dWv || !dbv || !dWu || !dbu || !dw || !db || !dz || !M || !dWf || !dbf) return MIL_EINVAL;
    if (!param_flat || !grad_flat || !exp_avg || !exp_avg_sq || (step_dev == nullptr && step < 1)) return MIL_EINVAL;
    if (step_dev != nullptr) step = 1;
    if (L <= 0 || (L % 128) != 0 || R <= 0 || B <= 0 || C <= 0 || C > 32 || (loss_bag && !loss_out)) return MIL_EINVAL;
    // every gradient this launch produces must lie inside the flat buffer (16-byte aligned where it is stored as float4)
    const float* outs[8] = {dWv, dbv, dWu, dbu, dw, db, dWf, dbf};
    const size_t lens[8] = {(size_t)192 * L, 192, (size_t)192 * L, 192, 192, 1, (size_t)C * L, (size_t)C};
    for (int i = 0; i < 8; ++i)
        if (outs[i] < grad_flat || outs[i] + lens[i] > grad_flat + n_param) return MIL_EINVAL;
    if (((dWv - grad_flat) | (dWu - grad_flat)) & 3) return MIL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(param_flat) | reinterpret_cast<uintptr_t>(grad_flat) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return MIL_EINVAL;
    int kc;
    const int S = split_plan(R, L, &kc);
    const int nthreads = GF_NG * (L / 4) + GR_NB * (3 * 192 + 1);
    const int nred = (nthreads + 255) / 256, nhead = C * ((L + 63) / 64) + 1;
    const HeadBwdArgs head{dz, M, dWf, dbf, loss_bag, loss_out, B, L, C, accumulate};
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    // the kernel forms the same single-precision quotient k_adam forms (lr / (float)bc1): the two routes stay bit-identical
    const AdamFuse ad{param_flat, grad_flat, exp_avg, exp_avg_sq, (float)bc1, beta1, beta2, eps, weight_decay, grad_scale,
                      (float)sqrt(bc2), step_dev, lr, lr_dev, step_dev ? done : nullptr};
    hipLaunchKernelGGL(k_gate_bwd_reduce, dim3(nred + nhead), dim3(256), 0, (hipStream_t)stream, workspace,
                       workspace + (size_t)S * GF_NG * L, S, use_dw2(R, L) ? S * (L / 128) : S, L, dWv, dbv, dWu, dbu, dw, db,
                       accumulate, xscale, nred, head, ad);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gate_bwd_params(const float* x, const float* gates, const float* ds, const float* w, int R, int L,
                                   int D, float* workspace, size_t workspace_floats, float* dWv, float* dbv,
                                   float* dWu, float* dbu, float* dw, float* db, int accumulate, const uint32_t* xbits,
                                   float xscale, void* stream) {
    if (!dWv || !dbv || !dWu || !dbu || !dw || !db) return MIL_EINVAL;
    const int rc = mil_gate_bwd_partials(x, gates, ds, w, R, L, D, workspace, workspace_floats, xbits, stream);
    if (rc != MIL_OK) return rc;
    return mil_gate_bwd_reduce(workspace, R, L, dWv, dbv, dWu, dbu, dw, db, accumulate, xbits ? xscale : 1.0f, stream);
}

extern "C" int mil_gate_bwd_params_x16(const uint16_t* x, const float* gates, const float* ds, const float* w, int R,
                                       int L, int D, float* workspace, size_t workspace_floats, float* dWv, float* dbv,
                                       float* dWu, float* dbu, float* dw, float* db, int accumulate, const uint32_t* xbits,
                                       float xscale, void* stream) {
    if (!x || !gates || !ds || !w || !workspace || !dWv || !dbv || !dWu || !dbu || !dw || !db) return MIL_EINVAL;
    if (D != MIL_GATE_D || L <= 0 || (L % 128) != 0 || R <= 0) return MIL_EINVAL;
    int kc;
    const int S = split_plan(R, L, &kc);
    const size_t need = gate_bwd_ws_floats(S, L);
    if (workspace_floats < need) return MIL_ENOSPC;
    float* part = workspace;
    float* pbias = workspace + (size_t)S * GF_NG * L;
    const int NJ = L / 128;
    hipStream_t st = (hipStream_t)stream;
    launch_gate_bwd_dw<true>((const void*)x, gates, ds, w, part, pbias, R, L, kc, NJ, S, xbits, st);
    MIL_CHECK_LAUNCH();
    const int nthreads = GF_NG * (L / 4) + GR_NB * (3 * 192 + 1);
    hipLaunchKernelGGL(k_gate_bwd_reduce, dim3((nthreads + 255) / 256), dim3(256), 0, st, part, pbias, S, S, L, dWv, dbv, dWu,
                       dbu, dw, db, accumulate, xbits ? xscale : 1.0f);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ================================================================================ K1 backward: gate dx (MFMA)
// dx[row][j] += sum_d dPreV[row][d] Wv[d][j] + dPreU[row][d] Wu[d][j]     (M = R, N = L, K = 384).
// K-slices of 32 = 16 d's x {V, U} so one (V, U) load pair yields both dPre terms.
// Workgroup 256 threads, tile 128 rows x 128 cols, wave (wi, wj) owns 64 x 64.
#define GX_KS 36       // k-contiguous A image row stride (words), as in k_gemm
// Same pipeline as k_gemm (csrc/linear.hip): 128 x 128 x 32 tiles, 2 x 2 waves of 64 x 64, the A image k-contiguous
// ([128][36], ds_read_b128 fragments: lane (r, h) takes k = 8t + 4h + jj), the weights k-major ([32][128]); registers
// carry the slice after next and the staging is issued in pieces between MFMA groups.  A slice of 32 k's is 16 gate units:
// local k 0..15 = dPreV_d, 16..31 = dPreU_d (d = 16 kk + k), built from (V, U, ds, w) when the slice is written to LDS.
// Pool term: with (scores, lse, row_bag, dM) given the kernel does not read dx at all - the attention pool's own input
// gradient, a_row dM[bag(row)] with a_row = exp(score_row - lse[bag]) (ABMIL.py:57-59), is formed in the epilogue from the
// [B, L] table dM and dx is written once (no pool-backward pass over [R, L], no read-modify-write here).
__global__ __launch_bounds__(256) void k_gate_bwd_dx(const float* __restrict__ gates, const float* __restrict__ ds,
                                                     const float* __restrict__ wvec, const float* __restrict__ Wv,
                                                     const float* __restrict__ Wu, float* __restrict__ dx, int R, int L,
                                                     const uint32_t* __restrict__ xbits, float xscale,
                                                     const float* __restrict__ scores, const float* __restrict__ lse,
                                                     const int32_t* __restrict__ row_bag, const float* __restrict__ dM) {
    constexpr int ASZ = 128 * GX_KS, BSZ = 32 * 128;
    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
    float* as = smem;
    float* bs = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int NJ = L / 128;
    // XCD-aware order (as the weight-gradient kernels): the NJ column tiles of a row tile read the same 128 x 384 gates; with
    // the hardware's round-robin of workgroup ids over the 8 XCDs they sat on NJ different L2s (PMC: 200 MiB fetched for 48 MiB
    // of gates).  logical = (id % 8) * share + id / 8 puts them on consecutive slots of ONE XCD.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int jt = bid % NJ, rt = bid / NJ;
    const int row0 = rt * 128, j0 = jt * 128;

    // A producer: thread -> unit quad dq = tid & 3 (d = 16 kk + 4 dq ..+3), rows (tid >> 2) + 64 i (i < 2)
    const int dq = tid & 3, arow = tid >> 2;
    const float* gsrc[2];
    float dsr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gr = row0 + arow + 64 * i;
        gsrc[i] = gates + (size_t)min(gr, R - 1) * GF_NG + 4 * dq;
        dsr[i] = gr < R ? ds[gr] : 0.f;                       // rows past the end contribute zeros
    }
    // B: k row (tid >> 5) + 8 i (i < 2: Wv rows, i >= 2: Wu rows), 16-byte chunk tid & 31
    const int bk = tid >> 5, bc4 = tid & 31;
    const float* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bsrc[i] = ((i < 2) ? Wv : Wu) + (size_t)(bk + 8 * (i & 1)) * L + j0 + 4 * bc4;
    f32x4 rv[2], ru[2], rw[2], rb[4];
    auto a_load = [&](int i, int kk) {
        rv[i] = *reinterpret_cast<const f32x4*>(gsrc[i] + 16 * kk);
        ru[i] = *reinterpret_cast<const f32x4*>(gsrc[i] + 192 + 16 * kk);
        rw[i] = *reinterpret_cast<const f32x4*>(wvec + 16 * kk + 4 * dq);
    };
    auto a_store = [&](int i, float* dst) {
        const f32x4 v = rv[i], u = ru[i];
        const f32x4 dsw = dsr[i] * rw[i];
        float* ad = dst + (arow + 64 * i) * GX_KS + 4 * dq;
        *reinterpret_cast<f32x4*>(ad) = dsw * u * (1.0f - v * v);            // dPreV
        *reinterpret_cast<f32x4*>(ad + 16) = dsw * v * u * (1.0f - u);       // dPreU
    };
    auto b_load = [&](int i, int kk) { rb[i] = *reinterpret_cast<const f32x4*>(bsrc[i] + (size_t)(16 * kk) * L); };
    auto b_store = [&](int i, float* dst) { *reinterpret_cast<f32x4*>(dst + (bk + 8 * i) * 128 + 4 * bc4) = rb[i]; };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    constexpr int nslice = MIL_GATE_D / 16;   // 12
#pragma unroll
    for (int i = 0; i < 2; ++i) a_load(i, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) b_load(i, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) a_store(i, as);
#pragma unroll
    for (int i = 0; i < 4; ++i) b_store(i, bs);
#pragma unroll
    for (int i = 0; i < 2; ++i) a_load(i, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) b_load(i, 1);
    __syncthreads();
    for (int s = 0; s < nslice; ++s) {
        const int buf = s & 1;
        const int k2 = min(s + 2, nslice - 1);
        const float* ab = as + buf * ASZ;
        const float* bb = bs + buf * BSZ;
        float* an = as + (buf ^ 1) * ASZ;
        float* bn = bs + (buf ^ 1) * BSZ;
        f32x4 fa[2][2], fb[2][2];     // [register set][tile]
        auto frag_a = [&](int t, int q, int a) {
            fa[q][a] = *reinterpret_cast<const f32x4*>(ab + (64 * wi + 32 * a + r) * GX_KS + 8 * t + 4 * h);
        };
        auto frag_b = [&](int t, int q, int b) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) fb[q][b][jj] = bb[(8 * t + 4 * h + jj) * 128 + 64 * wj + 32 * b + r];
        };
        frag_a(0, 0, 0); frag_a(0, 0, 1); frag_b(0, 0, 0); frag_b(0, 0, 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t & 1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int g = 4 * t + jj;
                // staging pieces between MFMA groups: the next slice into LDS, registers reloaded with the slice after
                if (g == 2 || g == 4) { const int i = (g - 2) >> 1; a_store(i, an); a_load(i, k2); }
                if (g >= 6 && g < 10) { const int i = g - 6; b_store(i, bn); b_load(i, k2); }
                if (t < 3) {
                    if (jj == 0) frag_a(t + 1, q ^ 1, 0);
                    if (jj == 1) frag_a(t + 1, q ^ 1, 1);
                    if (jj == 2) frag_b(t + 1, q ^ 1, 0);
                    if (jj == 3) frag_b(t + 1, q ^ 1, 1);
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0][jj], fb[q][0][jj], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0][jj], fb[q][1][jj], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1][jj], fb[q][0][jj], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1][jj], fb[q][1][jj], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    // pool term: a_row and the bag of each of the tile's 128 rows, once per workgroup (the staging area is free: the loop's
    // last barrier has passed)
    float* s_arow = smem;
    int* s_bag = reinterpret_cast<int*>(smem + 128);
    if (dM != nullptr) {
        if (tid < 128) {
            const int gr = min(row0 + tid, R - 1);
            const int bg = row_bag[gr];                  // < 0: a padding row of a capacity bucket - no pool weight
            s_bag[tid] = max(bg, 0);
            s_arow[tid] = bg < 0 ? 0.f : expf(scores[gr] - lse[bg]);
        }
        __syncthreads();
    }
    // dx += tile: the 16 old values of a tile column are loaded as one batch before the adds
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float* o = dx + j0 + 64 * wj + 32 * b + r;
            const int rbase = row0 + 64 * wi + 32 * a;
            float cv[16];
            if (dM != nullptr) {
                const int col = j0 + 64 * wj + 32 * b + r;
                const int lr0 = 64 * wi + 32 * a;
#pragma unroll
                for (int i = 0; i < 16; ++i) cv[i] = dM[(size_t)s_bag[lr0 + mfma32_row(i, h)] * L + col];
#pragma unroll
                for (int i = 0; i < 16; ++i) cv[i] *= s_arow[lr0 + mfma32_row(i, h)];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) cv[i] = o[(size_t)min(rbase + mfma32_row(i, h), R - 1) * L];
            }
            if (xbits != nullptr) {
                // this launch is the last writer of dx: the backward of the patch dropout (ABMIL.py:49) is applied here,
                // dx = keep ? (pool term + gate term) / (1 - p) : 0
                const int col = j0 + 64 * wj + 32 * b + r;
                unsigned mw[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) mw[i] = xbits[(size_t)min(rbase + mfma32_row(i, h), R - 1) * (L >> 5) + (col >> 5)];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gr = rbase + mfma32_row(i, h);
                    if (gr < R) o[(size_t)gr * L] = ((mw[i] >> (col & 31)) & 1u) ? (cv[i] + acc[a][b][i]) * xscale : 0.f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gr = rbase + mfma32_row(i, h);
                    if (gr < R) o[(size_t)gr * L] = cv[i] + acc[a][b][i];
                }
            }
        }
}

// The same tile quantisation for dx (two 128 x 128 workgroups per CU): the rows beyond whole rounds, one workgroup
// per row, thread = four columns, d looped (W is L2-resident).
// One 16 x 16 output tile per workgroup (rows = tail rows, columns of L), the 8 waves split the 384 gate units, operands
// straight from global memory (16x16x4 MFMA, all loads of a wave issued before its first MFMA), partial tiles folded
// through LDS: every workgroup reads 24 KB of the weights.  (One workgroup per ROW streamed all 768 KB of Wv, Wu through a
// single CU: 21-23 us for 32 rows.)
__global__ __launch_bounds__(512) void k_gate_bwd_dx_tail(const float* __restrict__ gates, const float* __restrict__ ds,
                                                          const float* __restrict__ wvec, const float* __restrict__ Wv,
                                                          const float* __restrict__ Wu, float* __restrict__ dx, int L, int rows,
                                                          const uint32_t* __restrict__ xbits, float xscale,
                                                          const float* __restrict__ scores, const float* __restrict__ lse,
                                                          const int32_t* __restrict__ row_bag, const float* __restrict__ dM) {
    __shared__ float red[8][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int mrow = min(m0 + r, rows - 1);
    const int jc = j0 + r;                                   // L % 16 == 0: always a valid column
    const float dsr = ds[mrow];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // wave w: gate units [24 w, 24 w + 24) of V and of U = 12 k-chunks of 4 units; A = dPre[row][unit], B = W[unit][column]
    f32x4 fa[12];
    float fb[12][4];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        const int isu = u >= 6;
        const int d = 24 * wave + 4 * (u % 6);                    // first gate unit of the chunk (chunks 0-5: V half, 6-11: U half)
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(gates + (size_t)mrow * GF_NG + d);
        const f32x4 u4 = *reinterpret_cast<const f32x4*>(gates + (size_t)mrow * GF_NG + 192 + d);
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wvec + d);
        const f32x4 dsw = dsr * w4;
        fa[u] = isu ? dsw * v4 * u4 * (1.0f - u4) : dsw * u4 * (1.0f - v4 * v4);
        const float* Wp = (isu ? Wu : Wv) + (size_t)d * L + jc;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) fb[u][jj] = Wp[(size_t)jj * L];
    }
    __builtin_amdgcn_sched_barrier(0);
    // 16x16x4: lane (r, kq) supplies A[m = r][k = kq] and B[k = kq][n = r]; the chunk's 4 units are its 4 k values, so the
    // A value of lane kq is component kq of the chunk's dPre vector and the B value is row kq of the chunk's weight rows
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        const float av = kq == 0 ? fa[u][0] : kq == 1 ? fa[u][1] : kq == 2 ? fa[u][2] : fa[u][3];
        const float bv_ = kq == 0 ? fb[u][0] : kq == 1 ? fb[u][1] : kq == 2 ? fb[u][2] : fb[u][3];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv_, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    if (tid < 256) {
        const int i = tid >> 6, l = tid & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w][i][l];
        const int row = m0 + 4 * (l >> 4) + i, col = j0 + (l & 15);
        if (row < rows) {
            float* o = dx + (size_t)row * L + col;
            float t;
            if (dM != nullptr) {
                const int bg = row_bag[row];
                t = bg < 0 ? v : expf(scores[row] - lse[bg]) * dM[(size_t)bg * L + col] + v;
            } else {
                t = *o + v;
            }
            if (xbits != nullptr) t = ((xbits[(size_t)row * (L >> 5) + (col >> 5)] >> (col & 31)) & 1u) ? t * xscale : 0.f;
            *o = t;
        }
    }
}

static int gate_bwd_input_impl(const float* gates, const float* ds, const float* w, const float* Wv, const float* Wu,
                               int R, int L, int D, float* dx, const uint32_t* xbits, float xscale, const float* scores,
                               const float* lse, const int32_t* row_bag, const float* dM, void* stream) {
    if (!gates || !ds || !w || !Wv || !Wu || !dx) return MIL_EINVAL;
    if (D != MIL_GATE_D || L <= 0 || (L % 128) != 0 || R < 0) return MIL_EINVAL;
    if (R == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    const int per_round = 2 * MIL_NUM_CU / (L / 128);          // row tiles per round of the grid
    const int tail = per_round > 0 ? gate_tail_rows(R, per_round) : 0;
    const int Rm = R - tail;
    const int grid = ((Rm + 127) / 128) * (L / 128);
    hipLaunchKernelGGL(k_gate_bwd_dx, dim3(grid), dim3(256), 0, st, gates, ds, w, Wv, Wu, dx, Rm, L, xbits, xscale, scores, lse,
                       row_bag, dM);
    MIL_CHECK_LAUNCH();
    if (tail > 0) {
        hipLaunchKernelGGL(k_gate_bwd_dx_tail, dim3(L / 16, (tail + 15) / 16), dim3(512), 0, st, gates + (size_t)Rm * GF_NG, ds + Rm, w,
                           Wv, Wu, dx + (size_t)Rm * L, L, tail, xbits ? xbits + (size_t)Rm * (L >> 5) : nullptr, xscale,
                           scores ? scores + Rm : nullptr, lse, row_bag ? row_bag + Rm : nullptr, dM);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

extern "C" int mil_gate_bwd_input(const float* gates, const float* ds, const float* w, const float* Wv, const float* Wu,
                                  int R, int L, int D, float* dx, const uint32_t* xbits, float xscale, void* stream) {
    return gate_bwd_input_impl(gates, ds, w, Wv, Wu, R, L, D, dx, xbits, xscale, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int mil_gate_bwd_input_pool(const float* gates, const float* ds, const float* w, const float* Wv,
                                       const float* Wu, int R, int L, int D, float* dx, const uint32_t* xbits, float xscale,
                                       const float* scores, const float* lse, const int32_t* row_bag, const float* dM,
                                       void* stream) {
    if (!scores || !lse || !row_bag || !dM) return MIL_EINVAL;
    return gate_bwd_input_impl(gates, ds, w, Wv, Wu, R, L, D, dx, xbits, xscale, scores, lse, row_bag, dM, stream);
}
