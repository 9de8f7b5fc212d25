// K3a, token side: nn.Linear on a handful of rows (M <= 64: one text token per bag, model/aggregator.py:44-68,
// model/sam/transformer.py:413-416 and sam/common.py:21-26 applied to the [B*T, 512] query stream).
// These products are weight-bandwidth- and latency-bound (a 512 x 512 weight is read once for 32 rows), so the
// tiled 128 x 128 GEMM of linear.hip wastes its launch on 1-4 workgroups walking all of K.  Here:
//   k_small_fwd   one workgroup per 32 output columns; its 8 waves split K, operands go global -> registers
//                 (16-byte loads along k, the k-permutation feeding 4 MFMA 32x32x2 per load), partial tiles are
//                 folded through LDS and the bias / activation / residual epilogue is applied in the same launch.
//   k_small_bwd   ONE launch for the whole backward of the layer: workgroups [0, nW) form dW = dpre^T x (and the
//                 bias gradient), workgroups [nW, nW + nX) form dx = dpre W; dpre = dy * act'(y) is evaluated on
//                 the fly, so no activation-backward pass, no column-sum pass and no split-K reduce exist.
// fp32 MFMA rounds like an fmaf chain, so results match the tiled path to accumulation order.
#include "mil_common.h"

#define SL_WAVES 8
enum { SL_NONE = 0, SL_TANH = 1, SL_RELU = 2, SL_QUICKGELU = 3, SL_SIGMOID = 4 };

__device__ __forceinline__ float sl_act(float v, int act) {
    if (act == SL_TANH) return tanhf(v);
    if (act == SL_RELU) return fmaxf(v, 0.f);
    if (act == SL_QUICKGELU) return v / (1.0f + expf(-1.702f * v));
    if (act == SL_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
// dy * act'(.), from the activation OUTPUT for tanh / relu and from the PRE-activation for QuickGELU
__device__ __forceinline__ float sl_dact(float g, float yv, int act) {
    if (act == SL_TANH) return g * (1.0f - yv * yv);
    if (act == SL_RELU) return yv > 0.f ? g : 0.f;
    if (act == SL_QUICKGELU) {
        const float s = 1.0f / (1.0f + expf(-1.702f * yv));
        return g * s * (1.0f + 1.702f * yv * (1.0f - s));
    }
    if (act == SL_SIGMOID) return g * yv * (1.0f - yv);
    return g;
}

template <int RT>
__device__ __forceinline__ void sl_fold_store(float (*red)[16][64], const f32x16* acc, int tid, int M, int N, int n0,
                                              const float* __restrict__ bias, int act,
                                              const float* __restrict__ residual, int ldr, float* __restrict__ out,
                                              int ldo) {
    // red is [SL_WAVES * RT][16][64]
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave * RT + a][i][lane] = acc[a][i];
    __syncthreads();
    for (int idx = tid; idx < RT * 1024; idx += 64 * SL_WAVES) {
        const int a = idx >> 10, i = (idx >> 6) & 15, l = idx & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SL_WAVES; ++w) v += red[w * RT + a][i][l];
        const int row = 32 * a + mfma32_row(i, l >> 5), col = n0 + (l & 31);
        if (row < M && col < N) {
            if (bias != nullptr) v += bias[col];
            v = sl_act(v, act);
            if (residual != nullptr) v += residual[(size_t)row * ldr + col];
            out[(size_t)row * ldo + col] = v;
        }
    }
}

template <int RT>
__global__ __launch_bounds__(64 * SL_WAVES) void k_small_fwd(const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias, int act,
                                                             const float* __restrict__ residual, int ldr,
                                                             float* __restrict__ y, int ldy, int M, int N, int K) {
    __shared__ float red[SL_WAVES * RT][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int nk8 = K >> 3, per = (nk8 + SL_WAVES - 1) / SL_WAVES;
    const int t0 = wave * per, t1 = min(nk8, t0 + per);
    const float* wrow = W + (size_t)min(n0 + r, N - 1) * ldw + 4 * h;
    const float* xrow[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) xrow[a] = x + (size_t)min(32 * a + r, M - 1) * ldx + 4 * h;
    f32x16 acc[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    // 8 k-chunks per trip, every load issued before the first MFMA (the layer is one or two memory round trips
    // deep: its time is latency, not bandwidth); chunks past the wave's range are clamped and zeroed
    for (int t = t0; t < t1; t += 8) {
        f32x4 fb[8], fa[RT][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = min(t + u, t1 - 1);
            fb[u] = *reinterpret_cast<const f32x4*>(wrow + 8 * tt);
#pragma unroll
            for (int a = 0; a < RT; ++a) fa[a][u] = *reinterpret_cast<const f32x4*>(xrow[a] + 8 * tt);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t + u >= t1) fb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int a = 0; a < RT; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][u][jj], fb[u][jj], acc[a], 0, 0, 0);
        }
    }
    sl_fold_store<RT>(red, acc, tid, M, N, n0, bias, act, residual, ldr, y, ldy);
}

template <int RT>
__global__ __launch_bounds__(64 * SL_WAVES) void k_small_bwd(const float* __restrict__ dy, int lddy,
                                                             const float* __restrict__ yv, int ldyv, int act,
                                                             const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ W, int ldw,
                                                             float* __restrict__ dx, int lddx, float* __restrict__ dW,
                                                             int lddw, float* __restrict__ db, int M, int N, int K,
                                                             int nW, int nKt) {
    __shared__ float red[SL_WAVES * RT][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    if ((int)blockIdx.x < nW) {
        // ---- dW[n][k] = sum_m dpre[m][n] x[m][k]: 64 (n) x 128 (k) per workgroup, one 32 x 32 tile per wave
        if (wave >= 8) return;
        const int kt = blockIdx.x % nKt, nt = blockIdx.x / nKt;
        const int wn = wave >> 2, wk = wave & 3;
        const int n = 64 * nt + 32 * wn + r, k = 128 * kt + 32 * wk + r;
        const int nc = min(n, N - 1), kc = min(k, K - 1);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float bsum = 0.f;
        const int steps = (M + 1) >> 1;
        for (int s0 = 0; s0 < steps; s0 += 16) {
            float fa[16], fb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int m = 2 * (s0 + u) + h;
                const int mm = min(m, M - 1);
                float g = dy[(size_t)mm * lddy + nc];
                if (act != SL_NONE) g = sl_dact(g, yv[(size_t)mm * ldyv + nc], act);
                fa[u] = m < M ? g : 0.f;
                fb[u] = m < M ? x[(size_t)mm * ldx + kc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                bsum += fa[u];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u], fb[u], acc, 0, 0, 0);
            }
        }
        if (dW != nullptr && k < K) {
            const int nbase = 64 * nt + 32 * wn;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int nn = nbase + mfma32_row(i, h);
                if (nn < N) dW[(size_t)nn * lddw + k] = acc[i];
            }
        }
        if (db != nullptr && kt == 0 && wk == 0) {
            const float tot = bsum + __shfl_xor(bsum, 32);
            if (h == 0 && n < N) db[n] = tot;
        }
        return;
    }
    // ---- dx[m][k] = sum_n dpre[m][n] W[n][k]: 32 k-columns per workgroup, the 8 waves split n
    const int k0 = ((int)blockIdx.x - nW) * 32;
    const int nn8 = N >> 3, per = (nn8 + SL_WAVES - 1) / SL_WAVES;
    const int t0 = wave * per, t1 = min(nn8, t0 + per);
    const int kc = min(k0 + r, K - 1);
    int mrow[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) mrow[a] = min(32 * a + r, M - 1);
    f32x16 acc[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    for (int t = t0; t < t1; t += 4) {
        f32x4 fa[RT][4];
        float fb[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = min(t + u, t1 - 1);          // ragged tail: reload the last chunk, zeroed below
            const int nb = 8 * tt + 4 * h;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) fb[u][jj] = W[(size_t)(nb + jj) * ldw + kc];
#pragma unroll
            for (int a = 0; a < RT; ++a) {
                f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)mrow[a] * lddy + nb);
                if (act != SL_NONE) {
                    const f32x4 yy = *reinterpret_cast<const f32x4*>(yv + (size_t)mrow[a] * ldyv + nb);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) g[jj] = sl_dact(g[jj], yy[jj], act);
                }
                if (t + u >= t1) g = f32x4{0.f, 0.f, 0.f, 0.f};
                fa[a][u] = g;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int a = 0; a < RT; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][u][jj], fb[u][jj], acc[a], 0, 0, 0);
    }
    sl_fold_store<RT>(red, acc, tid, M, K, k0, nullptr, SL_NONE, nullptr, 0, dx, lddx);
}

static inline bool sl_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int mil_linear_small_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias, int act,
                                    const float* residual, int ldr, float* y, int ldy, int M, int N, int K,
                                    void* stream) {
    if (!x || !W || !y || M <= 0 || M > MIL_SMALL_ROWS || N <= 0 || K <= 0) return MIL_EINVAL;
    if ((K & 7) || (ldx & 3) || (ldw & 3) || act < 0 || act > 4) return MIL_EINVAL;
    if (!sl_aligned16(x) || !sl_aligned16(W)) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + 31) / 32), block(64 * SL_WAVES);
    if (M <= 32)
        hipLaunchKernelGGL((k_small_fwd<1>), grid, block, 0, st, x, ldx, W, ldw, bias, act, residual, ldr, y, ldy, M, N, K);
    else
        hipLaunchKernelGGL((k_small_fwd<2>), grid, block, 0, st, x, ldx, W, ldw, bias, act, residual, ldr, y, ldy, M, N, K);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_linear_small_bwd(const float* dy, int lddy, const float* y_or_pre, int ldyv, int act, const float* x,
                                    int ldx, const float* W, int ldw, float* dx, int lddx, float* dW, int lddw,
                                    float* db, int M, int N, int K, void* stream) {
    if (!dy || M <= 0 || M > MIL_SMALL_ROWS || N <= 0 || K <= 0 || act < 0 || act > 4) return MIL_EINVAL;
    if (act != SL_NONE && !y_or_pre) return MIL_EINVAL;
    if ((dW || db) && !x) return MIL_EINVAL;
    if (dx && !W) return MIL_EINVAL;
    if ((N & 7) || (lddy & 3) || (act != SL_NONE && (ldyv & 3)) || !sl_aligned16(dy) ||
        (act != SL_NONE && !sl_aligned16(y_or_pre)))
        return MIL_EINVAL;
    const int nKt = (K + 127) / 128;
    const int nW = (dW || db) ? ((N + 63) / 64) * nKt : 0;
    const int nX = dx ? (K + 31) / 32 : 0;
    if (nW + nX == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(nW + nX), block(64 * SL_WAVES);
    if (M <= 32)
        hipLaunchKernelGGL((k_small_bwd<1>), grid, block, 0, st, dy, lddy, y_or_pre, ldyv, act, x, ldx, W, ldw, dx, lddx,
                           dW, lddw, db, M, N, K, nW, nKt);
    else
        hipLaunchKernelGGL((k_small_bwd<2>), grid, block, 0, st, dy, lddy, y_or_pre, ldyv, act, x, ldx, W, ldw, dx, lddx,
                           dW, lddw, db, M, N, K, nW, nKt);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
