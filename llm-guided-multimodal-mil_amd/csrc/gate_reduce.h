// Split-K reduce of the gate weight gradient, shared by the fp32 and bf16 dW kernels (included in both
// translation units: device code is not linked across them).
#pragma once
#include "mil_common.h"
#define GR_NG 384

// Sum the split-K partials and un-permute the gate index.  One thread per output float4.
// One thread per output float4 (8 independent 16-byte loads in flight), un-permuting the gate index; the last
// 577 threads fold the bias / w / b partials.  (A 4-threads-per-output variant with 4x the workgroups measured
// slower: 14.3 vs 12.1 us for the 31 MB of partials at L = 512.)
static __global__ __launch_bounds__(256) void k_gate_bwd_reduce(const float* __restrict__ part, const float* __restrict__ pbias,
                                                                int S, int L, float* __restrict__ dWv, float* __restrict__ dbv,
                                                                float* __restrict__ dWu, float* __restrict__ dbu,
                                                                float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    const int L4 = L / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nW = GR_NG * L4;
    if (idx < nW) {
        const int gi = idx / L4, c4 = idx % L4;
        f32x4 v = {0, 0, 0, 0};
        const float* src = part + (size_t)gi * L + 4 * c4;
        const size_t stride = (size_t)GR_NG * L;
        int s = 0;
        for (; s + 8 <= S; s += 8) {
            f32x4 t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + e) * stride);
#pragma unroll
            for (int e = 0; e < 8; ++e) v += t[e];
        }
        for (; s < S; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * stride);
        const int m = gi >> 7, ii = gi & 127;
        float* dst = (ii < 64 ? dWv + (size_t)(64 * m + ii) * L : dWu + (size_t)(64 * m + ii - 64) * L) + 4 * c4;
        if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
    } else if (idx < nW + 3 * 192 + 1) {
        const int k = idx - nW;
        const int which = k / 192, d = k % 192;
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += pbias[((size_t)s * 4 + which) * 192 + d];
        float* dst = which == 0 ? dbv + d : which == 1 ? dbu + d : which == 2 ? dw + d : db;
        if (accumulate) v += *dst;
        *dst = v;
    }
}
