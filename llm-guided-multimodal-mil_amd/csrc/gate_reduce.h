// Split-K reduce of the gate weight gradient, shared by the fp32 and bf16 dW kernels (included in both
// translation units: device code is not linked across them).
#pragma once
#include "mil_common.h"
#define GR_NG 384
#define GR_NB 16          // lanes per bias / w / b output in the reduce

// Parameter gradients of the head (model/aggregator.py:128-131: z = M Wf^T + bf):  dWf[c][j] = sum_b dz[b][c] M[b][j],
// dbf[c] = sum_b dz[b][c], plus the step's loss = sum_b loss_bag[b].  A handful of workgroups of latency-bound work that
// depends only on the fused per-bag tail: it rides at the end of the gate reduce launch (blocks >= first) instead of
// being a launch of its own.  Same arithmetic as k_head_bwd_params (head_loss.hip).
struct HeadBwdArgs {
    const float* dz;            // [B, C] or NULL (no head work in this launch)
    const float* M;             // [B, L]
    float* dWf;                 // [C, L]
    float* dbf;                 // [C]
    const float* loss_bag;      // [B] or NULL
    float* loss_out;            // [1]
    int B, L, C;
    int accumulate;             // != 0: add to dWf / dbf / loss_out (gradient accumulation over micro-batches)
};
static __device__ __forceinline__ void head_bwd_params_block(const HeadBwdArgs& a, int blk, float (*red)[64],
                                                             const AdamFuse& ad) {
    const int nlb = (a.L + 63) / 64;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    if (blk < a.C * nlb) {
        const int c = blk / nlb, j = (blk % nlb) * 64 + lane;
        float v = 0.f;
        if (j < a.L)
            for (int b0 = g; b0 < a.B; b0 += 32) {                  // eight (dz, M) pairs in flight per thread, not one
                float dzv[8], mv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int b = min(b0 + 4 * e, a.B - 1);
                    dzv[e] = a.dz[b * a.C + c];
                    mv[e] = a.M[(size_t)b * a.L + j];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v = b0 + 4 * e < a.B ? fmaf(dzv[e], mv[e], v) : v;
            }
        red[g][lane] = v;
        __syncthreads();
        if (g == 0 && j < a.L) {
            float* o = a.dWf + (size_t)c * a.L + j;
            float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
            if (a.accumulate) t += *o;
            *o = t;
            adam_fused(ad, o, t);
        }
    } else {
        if ((int)threadIdx.x < a.C) {
            const int c = threadIdx.x;
            float v = 0.f;
            for (int b0 = 0; b0 < a.B; b0 += 16) {
                float u[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) u[e] = a.dz[min(b0 + e, a.B - 1) * a.C + c];
#pragma unroll
                for (int e = 0; e < 16; ++e) v += b0 + e < a.B ? u[e] : 0.f;
            }
            if (a.accumulate) v += a.dbf[c];
            a.dbf[c] = v;
            adam_fused(ad, a.dbf + c, v);
        }
        if (a.loss_bag != nullptr && g == 1) {
            float v = 0.f;
            for (int b = lane; b < a.B; b += 64) v += a.loss_bag[b];
            v = wave_allsum(v);
            if (lane == 0) a.loss_out[0] = a.accumulate ? a.loss_out[0] + v : v;
        }
    }
}

// Sum the split-K partials and un-permute the gate index.  One thread per output float4.
// One thread per output float4 (8 independent 16-byte loads in flight), un-permuting the gate index; the last
// 577 threads fold the bias / w / b partials.  (A 4-threads-per-output variant with 4x the workgroups measured
// slower: 14.3 vs 12.1 us for the 31 MB of partials at L = 512.)
// SB: number of bias slabs [4][192] in pbias (S for kernels whose jt == 0 workgroups publish a chunk's sums, S * NJ for
// k_gate_bwd_dw2, where every column-tile workgroup publishes its share).
static __global__ __launch_bounds__(256) void k_gate_bwd_reduce(const float* __restrict__ part, const float* __restrict__ pbias,
                                                                int S, int SB, int L, float* __restrict__ dWv, float* __restrict__ dbv,
                                                                float* __restrict__ dWu, float* __restrict__ dbu,
                                                                float* __restrict__ dw, float* __restrict__ db, int accumulate,
                                                                float wscale, int head_first = 1 << 30,
                                                                HeadBwdArgs head = HeadBwdArgs{}, AdamFuse ad_in = AdamFuse{},
                                                                unsigned short* __restrict__ Wv16 = nullptr,
                                                                unsigned short* __restrict__ Wu16 = nullptr) {
    // Wv16 / Wu16 (bf16-storage step with Adam applied here): the bf16 shadows of the gate weights the forward reads are
    // refreshed by the thread that has just updated their fp32 masters - no cast launches after the update
    __shared__ float bcs[2];
    const int L4 = L / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nW = GR_NG * L4;
    const bool wrole = (int)blockIdx.x < head_first && idx < nW;
    // Round 4: the launch is a chain of dependent round trips (step counter -> bias corrections -> learning rate -> partials 8
    // + 8 + 5 x ONE at a time -> p, m, v).  The first 24 partials of a weight thread (every split the step's shapes produce) and
    // its Adam state are now requested BEFORE the workgroup waits for the bias corrections, the tail of the partial list as
    // one masked batch: the loads of a thread are in flight together.
    constexpr int GR_PRE = 24;
    f32x4 t[GR_PRE], pm[3];
    const int gi = wrole ? idx / L4 : 0, c4 = wrole ? idx % L4 : 0;
    const float* src = part + (size_t)gi * L + 4 * c4;
    const size_t stride = (size_t)GR_NG * L;
    const int mm_ = gi >> 7, ii = gi & 127;
    const size_t woff = (size_t)(ii < 64 ? 64 * mm_ + ii : 64 * mm_ + ii - 64) * L + 4 * c4;
    float* dst = (ii < 64 ? dWv : dWu) + woff;
    if (wrole) {
#pragma unroll
        for (int e = 0; e < GR_PRE; ++e) t[e] = *reinterpret_cast<const f32x4*>(src + (size_t)min(e, S - 1) * stride);
        if (ad_in.param != nullptr) {
            const size_t i = (size_t)(dst - ad_in.grad_base);
            pm[0] = *reinterpret_cast<const f32x4*>(ad_in.param + i);
            pm[1] = *reinterpret_cast<const f32x4*>(ad_in.m + i);
            pm[2] = *reinterpret_cast<const f32x4*>(ad_in.v + i);
        }
    }
    const AdamFuse ad = adam_fuse_resolve(ad_in, bcs);
    if ((int)blockIdx.x >= head_first) {          // appended workgroups: the head's parameter gradients (uniform branch)
        __shared__ float hred[4][64];
        head_bwd_params_block(head, blockIdx.x - head_first, hred, ad);
        return;
    }
    if (idx < nW) {
        f32x4 v = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < GR_PRE; ++e) v += e < S ? t[e] : f32x4{0, 0, 0, 0};
        for (int s = GR_PRE; s < S; s += 8) {       // longer splits: masked batches of eight
            f32x4 u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = *reinterpret_cast<const f32x4*>(src + (size_t)min(s + e, S - 1) * stride);
#pragma unroll
            for (int e = 0; e < 8; ++e) v += s + e < S ? u[e] : f32x4{0, 0, 0, 0};
        }
        v *= wscale;             // train mode: the 1/(1-p) of the patch dropout (x entered the product as keep-masked x)
        if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
        f32x4 pnew = v;
        if (ad.param != nullptr) {                 // adam_fused4 on the prefetched state
            const size_t i = (size_t)(dst - ad.grad_base);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pm[0][e], me = pm[1][e], ve = pm[2][e];
                adam_one(pe, v[e], me, ve, ad.lr_bc1, ad.b1, ad.b2, ad.eps, ad.wd, ad.gscale, ad.bc2_sqrt);
                pm[0][e] = pe; pm[1][e] = me; pm[2][e] = ve;
            }
            *reinterpret_cast<f32x4*>(ad.param + i) = pm[0];
            *reinterpret_cast<f32x4*>(ad.m + i) = pm[1];
            *reinterpret_cast<f32x4*>(ad.v + i) = pm[2];
            pnew = pm[0];
        }
        if (Wv16 != nullptr && ad.param != nullptr) {
            ushort4 o;                                  // the conversion k_cast_bf16 uses (round to nearest even)
            o.x = __builtin_bit_cast(unsigned short, (__bf16)pnew[0]);
            o.y = __builtin_bit_cast(unsigned short, (__bf16)pnew[1]);
            o.z = __builtin_bit_cast(unsigned short, (__bf16)pnew[2]);
            o.w = __builtin_bit_cast(unsigned short, (__bf16)pnew[3]);
            *reinterpret_cast<ushort4*>((ii < 64 ? Wv16 : Wu16) + woff) = o;
        }
    } else if (idx < nW + GR_NB * (3 * 192 + 1)) {
        // bias / w / b: GR_NB lanes per output, each sums every GR_NB-th slab (loads in flight), then a fixed-order
        // shuffle fold.  (One thread per output walked the S * NJ slabs of the low-VALU kernel serially: 9.5 us.)
        const int k = (idx - nW) / GR_NB, sub = (idx - nW) % GR_NB;
        const int which = k / 192, d = k % 192;
        float v = 0.f;
        for (int s0 = sub; s0 < SB; s0 += 8 * GR_NB) {              // eight slabs in flight per lane (they were one at a time)
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = pbias[((size_t)min(s0 + e * GR_NB, SB - 1) * 4 + which) * 192 + d];
#pragma unroll
            for (int e = 0; e < 8; ++e) v += s0 + e * GR_NB < SB ? u[e] : 0.f;
        }
#pragma unroll
        for (int mm = GR_NB / 2; mm >= 1; mm >>= 1) v += __shfl_xor(v, mm);
        if (sub != 0) return;
        float* dst = which == 0 ? dbv + d : which == 1 ? dbu + d : which == 2 ? dw + d : db;
        if (accumulate) v += *dst;
        *dst = v;
        adam_fused(ad, dst, v);
    }
}
