// K3a, mid-size layers: nn.Linear on 65 .. ~1000 rows - T text tokens x 32 bags on the token side of the two-way
// transformer (model/sam/transformer.py:413-416, sam/common.py:21-26), the 770-row text tower of a one-bag step with
// learnable prompts (clip/model.py:171-178).  Such a product is a microsecond of MFMA work: on the 128 x 128 tile it is
// a dozen workgroups, so mil_gemm splits K eight ways and folds the partial sums in a second launch (15 + 6 us forward,
// 14 + 14 + 6 (+6) us backward), while the 16 x 16 tiles of small_linear.hip re-read the operands M / 16 times.
// Here a workgroup owns a 32 x 32 (or 64 x 64) output tile and contracts ALL of K itself: each wave holds one 32 x 32
// accumulator, the waves of a tile quadrant split K between them, operands go global -> registers (a whole trip of loads
// issued before its first MFMA, as in k_small_fwd), the partial tiles are folded through LDS in a fixed order and the
// epilogue runs in the same launch - one launch per product, no workspace.
//
//   C[M, N] = epilogue( A_op[M, K] . B_op[K, N] ),   AKM / BKM = operand is k-major in memory:
//     forward   y  = act(x W^T + b) + res      A = x   [M, K] k-contiguous,  B = W [N, K] k-contiguous
//     dx            = dpre W                   A = dpre [M, K = n_out],        B = W [K = n_out, N = k_in] k-major
//     dW (+ db)     = dpre^T x                 A = dpre [K = rows, M = n_out] k-major, B = x [K = rows, N = k_in] k-major
//   dpre = dy (.) act'(y) is formed on the loaded fragment when `aux` (the saved layer output) is given; with AKM the
//   column sums of A_op (the bias gradient) are produced by the workgroups of the first column tile.
// fp32 MFMA 32x32x2, lane (r, h) feeds k = 8g + 4h + jj of group g (same convention as linear.hip).
#include "mil_common.h"

enum { ML_NONE = 0, ML_TANH = 1, ML_RELU = 2, ML_QUICKGELU = 3 };

__device__ __forceinline__ float ml_dact(float g, float yv, int act) {
    if (act == ML_TANH) return g * (1.0f - yv * yv);
    if (act == ML_RELU) return yv > 0.f ? g : 0.f;
    return g;
}

#define ML_TRIP 8          // k-groups (of 8) per trip: 16 operand loads in flight per lane (k-contiguous operands)
#ifndef ML_T32_LIMIT
#define ML_T32_LIMIT (4 * MIL_NUM_CU)
#endif

// TQ x TQ quadrants of 32 x 32 per workgroup, KS waves per quadrant splitting K: TQ * TQ * KS waves
template <bool AKM, bool BKM, int TQ, int KS>
__global__ __launch_bounds__(64 * TQ * TQ * KS) void k_mid(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                            int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                            const float* __restrict__ bias, int act,
                                                            const float* __restrict__ residual, int ldr,
                                                            const float* __restrict__ aux, int ldaux, int a_act,
                                                            float* __restrict__ colsum) {
    constexpr int NQ = TQ * TQ, NWV = NQ * KS;
    __shared__ float red[NWV][16][64];                  // 32 KB at 8 waves
    __shared__ float cred[AKM ? NWV : 1][AKM ? 32 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q = wave % NQ, ks = wave / NQ;
    const int i0 = blockIdx.y * (32 * TQ) + 32 * (q / TQ), j0 = blockIdx.x * (32 * TQ) + 32 * (q % TQ);
    const int ngrp = (K + 7) >> 3, per = (ngrp + KS - 1) / KS;
    const int g0 = ks * per, g1 = min(ngrp, g0 + per);
    const int ic = min(i0 + r, M - 1), jc = min(j0 + r, N - 1);      // clamped operand row / column of this lane
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float csum = 0.f;

    for (int g = g0; g < g1; g += ML_TRIP) {
        f32x4 fa[ML_TRIP], fb[ML_TRIP], fx[ML_TRIP];
#pragma unroll
        for (int u = 0; u < ML_TRIP; ++u) {
            const int gg = min(g + u, g1 - 1);
            const int k0 = 8 * gg + 4 * h;
            if (AKM) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int kk = min(k0 + jj, K - 1);
                    fa[u][jj] = A[(size_t)kk * lda + ic];
                    if (aux != nullptr) fx[u][jj] = aux[(size_t)kk * ldaux + ic];
                }
            } else {
                fa[u] = *reinterpret_cast<const f32x4*>(A + (size_t)ic * lda + k0);
                if (aux != nullptr) fx[u] = *reinterpret_cast<const f32x4*>(aux + (size_t)ic * ldaux + k0);
            }
            if (BKM) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) fb[u][jj] = B[(size_t)min(k0 + jj, K - 1) * ldb + jc];
            } else {
                fb[u] = *reinterpret_cast<const f32x4*>(B + (size_t)jc * ldb + k0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // all loads of the trip ahead of its first MFMA
#pragma unroll
        for (int u = 0; u < ML_TRIP; ++u) {
            const int k0 = 8 * (g + u) + 4 * h;
            const bool live = g + u < g1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float a = fa[u][jj];
                if (aux != nullptr) a = ml_dact(a, fx[u][jj], a_act);
                if (!live || k0 + jj >= K) a = 0.f;     // clamped duplicates and the ragged end of K contribute nothing
                if (AKM) csum += a;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fb[u][jj], acc, 0, 0, 0);
            }
        }
    }

    // fold the KS partial tiles of each quadrant (fixed order) and apply the epilogue
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    if (AKM) {
        const float t = csum + __shfl_xor(csum, 32);
        if (h == 0) cred[wave][r] = t;
    }
    __syncthreads();
    for (int idx = tid; idx < NQ * 16 * 64; idx += 64 * NWV) {
        const int qq = idx / (16 * 64), i = (idx >> 6) & 15, l = idx & 63;
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) v += red[s * NQ + qq][i][l];
        const int row = blockIdx.y * (32 * TQ) + 32 * (qq / TQ) + mfma32_row(i, l >> 5);
        const int col = blockIdx.x * (32 * TQ) + 32 * (qq % TQ) + (l & 31);
        if (row < M && col < N) {
            if (bias != nullptr) v += bias[col];
            if (act == ML_TANH) v = tanhf(v);
            else if (act == ML_RELU) v = fmaxf(v, 0.f);
            else if (act == ML_QUICKGELU) v = v / (1.0f + expf(-1.702f * v));
            if (residual != nullptr) v += residual[(size_t)row * ldr + col];
            C[(size_t)row * ldc + col] = v;
        }
    }
    if (AKM) {
        // bias gradient: column sums of A_op over k = row sums of the output's M axis; first column tile only
        if (colsum != nullptr && blockIdx.x == 0 && tid < 32 * TQ) {
            const int qa = tid >> 5, rr = tid & 31;         // quadrant row block qa: its quadrants are qa * TQ + 0 (column block 0)
            float v = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) v += cred[s * NQ + qa * TQ][rr];
            const int row = blockIdx.y * (32 * TQ) + 32 * qa + rr;
            if (row < M) colsum[row] = v;
        }
    }
}

static inline bool ml_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// tile choice: 32 x 32 tiles (8 waves split K) while they stay within ~4 workgroups per CU, else 64 x 64 (2 x 2 quadrants x 2)
template <bool AKM, bool BKM>
static void ml_launch(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                      const float* bias, int act, const float* residual, int ldr, const float* aux, int ldaux, int a_act,
                      float* colsum, hipStream_t st) {
    const long t32 = (long)((M + 31) / 32) * ((N + 31) / 32);
    if (t32 <= ML_T32_LIMIT) {
        hipLaunchKernelGGL((k_mid<AKM, BKM, 1, 8>), dim3((N + 31) / 32, (M + 31) / 32), dim3(512), 0, st, A, lda, B, ldb, C, ldc,
                           M, N, K, bias, act, residual, ldr, aux, ldaux, a_act, colsum);
    } else {
        hipLaunchKernelGGL((k_mid<AKM, BKM, 2, 2>), dim3((N + 63) / 64, (M + 63) / 64), dim3(512), 0, st, A, lda, B, ldb, C, ldc,
                           M, N, K, bias, act, residual, ldr, aux, ldaux, a_act, colsum);
    }
}

extern "C" int mil_linear_mid_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias, int act,
                                  const float* residual, int ldr, float* y, int ldy, int M, int N, int K, void* stream) {
    if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0) return MIL_EINVAL;
    if ((K & 7) || (ldx & 3) || (ldw & 3) || act < 0 || act > 3 || !ml_al16(x) || !ml_al16(W)) return MIL_EINVAL;
    ml_launch<false, false>(x, ldx, W, ldw, y, ldy, M, N, K, bias, act, residual, ldr, nullptr, 0, 0, nullptr,
                            (hipStream_t)stream);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// Whole backward of the layer in two launches (one per product): dx = dpre W;  dW = dpre^T x with db from the same pass.
// dpre = dy (.) act'(yv) (act 0: dpre = dy, yv may be NULL).  Any of dx / dW may be NULL; db needs dW.
extern "C" int mil_linear_mid_bwd(const float* dy, int lddy, const float* yv, int ldyv, int act, const float* x, int ldx,
                                  const float* W, int ldw, float* dx, int lddx, float* dW, int lddw, float* db, int M,
                                  int N, int K, void* stream) {
    if (!dy || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return MIL_EINVAL;
    if (act != ML_NONE && !yv) return MIL_EINVAL;
    if ((lddy & 3) || (yv && (ldyv & 3)) || !ml_al16(dy) || (yv && !ml_al16(yv))) return MIL_EINVAL;
    if (db && !dW) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const float* aux = act != ML_NONE ? yv : nullptr;
    if (dx) {
        if (!W || (N & 7)) return MIL_EINVAL;                 // contraction over n_out in whole groups of 8
        ml_launch<false, true>(dy, lddy, W, ldw, dx, lddx, M, K, N, nullptr, 0, nullptr, 0, aux, ldyv, act, nullptr, st);
        MIL_CHECK_LAUNCH();
    }
    if (dW) {
        if (!x) return MIL_EINVAL;
        ml_launch<true, true>(dy, lddy, x, ldx, dW, lddw, N, K, M, nullptr, 0, nullptr, 0, aux, ldyv, act, db, st);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}
