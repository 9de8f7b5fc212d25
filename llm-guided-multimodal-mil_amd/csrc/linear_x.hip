// K3a, split-bf16 form of the NT product for FROZEN weights (the CLIP text tower under learnable prompts:
// clip/model.py:171-199 applied to ~10-25 k token rows, forward and the dx half of the backward, 3.7 TFLOP per step).
//
//   C[M, N] = epilogue( A[M, K] . B[N, K]^T )        A fp32 activations, B frozen fp32 weights
//
// Every fp32 operand is written as a sum of NP bf16 pieces (a = a1 + a2 (+ a3), a1 = bf16(a), a2 = bf16(a - a1), ...)
// and the product keeps the pieces' cross terms down to the last kept piece: NP = 2 -> a1b1 + a1b2 + a2b1 (relative
// error ~2^-16), NP = 3 -> six terms (~2^-23: the rounding of an fp32 product).  All accumulation stays fp32
// (v_mfma_f32_32x32x16_bf16).  Six bf16 MFMAs move 16 k per 192 cycles where v_mfma_f32_32x32x2_f32 needs 512:
// 2.7x the fp32 matrix rate at fp32 accuracy.  The weights' pieces are formed once (they are frozen), the activations
// are split while they are staged.  This is an OPT-IN path (model flag clip_gemm_pieces): the default tower runs the
// fp32 MFMA GEMM of linear.hip.
//
// Workgroup 256 threads = 4 waves, tile 128 x 128 x 32, wave (wi, wj) owns 64 x 64 = 2 x 2 MFMA tiles.
// LDS: one stage, NP images of A and of B, [128][32] bf16 each (64-byte rows, 16-byte chunk c of row `row` stored at
// c ^ ((row >> 2) & 3): conflict-free ds_read_b128 fragments); the next slice waits in registers.
#include "mil_common.h"

typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short xu16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short xu16;

enum { XA_NONE = 0, XA_TANH = 1, XA_RELU = 2, XA_QUICKGELU = 3 };
enum { XAUX_NONE = 0, XAUX_STORE_PRE = 1, XAUX_MUL_DGELU = 2 };

__device__ __forceinline__ xu16 x_bf16_bits(float v) { return __builtin_bit_cast(xu16, (__bf16)v); }
__device__ __forceinline__ float x_bf16_val(xu16 b) { return __uint_as_float(((unsigned)b) << 16); }

// pieces of four floats: out[p] = 4 bf16 (8 bytes)
template <int NP>
__device__ __forceinline__ void x_split4(const f32x4 v, ushort4* out) {
    f32x4 r = v;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        ushort4 o;
        o.x = x_bf16_bits(r[0]); o.y = x_bf16_bits(r[1]); o.z = x_bf16_bits(r[2]); o.w = x_bf16_bits(r[3]);
        out[p] = o;
        r[0] -= x_bf16_val(o.x); r[1] -= x_bf16_val(o.y); r[2] -= x_bf16_val(o.z); r[3] -= x_bf16_val(o.w);
    }
}

// dst[p][i] = piece p of src[i]   (weights, once)
__global__ __launch_bounds__(256) void k_split_bf16(const float* __restrict__ src, xu16* __restrict__ dst, size_t n, int np) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r = src[i];
    for (int p = 0; p < np; ++p) {
        const xu16 b = x_bf16_bits(r);
        dst[(size_t)p * n + i] = b;
        r -= x_bf16_val(b);
    }
}

#define XG_BK 32
#define XG_IMG (128 * XG_BK)        // u16 per operand image (8 KB)

template <int NP>
__global__ __launch_bounds__(256) void k_gemm_x(const float* __restrict__ A, int lda, const xu16* __restrict__ Bp,
                                                size_t piece_stride, int ldb, float* __restrict__ C, int ldc, int M, int N,
                                                int K, const float* __restrict__ bias, int act,
                                                const float* __restrict__ residual, int ldr, float* __restrict__ aux,
                                                int ldaux, int aux_mode) {
    __shared__ __attribute__((aligned(16))) xu16 smem[2 * NP * XG_IMG];
    xu16* as = smem;                    // [NP][128][32]
    xu16* bs = smem + NP * XG_IMG;      // [NP][128][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;

    // A staging: four float4 per thread: row (tid >> 3) + 32 i, k = 4 (tid & 7)
    const int arow = tid >> 3, ach = tid & 7;
    const float* asrc[4];
    int adst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = arow + 32 * i;
        asrc[i] = A + (size_t)min(i0 + row, M - 1) * lda + 4 * ach;
        adst[i] = row * XG_BK + 8 * ((ach >> 1) ^ ((row >> 2) & 3)) + 4 * (ach & 1);
    }
    // B staging: per piece two 16-byte chunks per thread: chunk id tid + 256 i: row id >> 2, chunk id & 3
    const xu16* bsrc[2];
    int bdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i, row = id >> 2, c16 = id & 3;
        bsrc[i] = Bp + (size_t)min(j0 + row, N - 1) * ldb + 8 * c16;
        bdst[i] = row * XG_BK + 8 * (c16 ^ ((row >> 2) & 3));
    }
    f32x4 areg[4];
    xu16x8 breg[NP][2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) areg[i] = *reinterpret_cast<const f32x4*>(asrc[i] + k0);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) breg[p][i] = *reinterpret_cast<const xu16x8*>(bsrc[i] + p * piece_stride + k0);
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ushort4 pc[NP];
            x_split4<NP>(areg[i], pc);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<ushort4*>(as + p * XG_IMG + adst[i]) = pc[p];
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<xu16x8*>(bs + p * XG_IMG + bdst[i]) = breg[p][i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int nslice = K / XG_BK;
    gload(0);
    swrite();
    gload(min(1, nslice - 1) * XG_BK);
    __syncthreads();
    const int fx = (r >> 2) & 3;
    for (int s = 0; s < nslice; ++s) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = 8 * ((2 * ks + h) ^ fx);
            xu16x8 fa[NP][2], fb[NP][2];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[p][t] = *reinterpret_cast<const xu16x8*>(as + p * XG_IMG + (64 * wi + 32 * t + r) * XG_BK + ch);
                    fb[p][t] = *reinterpret_cast<const xu16x8*>(bs + p * XG_IMG + (64 * wj + 32 * t + r) * XG_BK + ch);
                }
            // cross terms, smallest first: piece indices (p, q) with p + q <= NP - 1 (0-based), in decreasing p + q
#pragma unroll
            for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
                for (int p = 0; p <= sum; ++p) {
                    const int q = sum - p;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8, fa[p][a]),
                                                                                __builtin_bit_cast(xbf16x8, fb[q][b]),
                                                                                acc[a][b], 0, 0, 0);
                }
        }
        __syncthreads();                       // everyone has read slice s
        if (s + 1 < nslice) {
            swrite();                          // slice s+1: registers -> LDS
            gload(min(s + 2, nslice - 1) * XG_BK);
        }
        __syncthreads();
    }

    // epilogue (same contract as k_gemm in linear.hip): lane holds column j, 16 rows per tile; operands of a tile are
    // loaded as one batch before the arithmetic
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + 64 * wj + 32 * b + r;
        if (j >= N) continue;
        const float bj = bias != nullptr ? bias[j] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int rbase = i0 + 64 * wi + 32 * a;
            float rv[16], pv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rc = min(rbase + mfma32_row(i, h), M - 1);
                rv[i] = residual != nullptr ? residual[(size_t)rc * ldr + j] : 0.f;
                pv[i] = aux_mode == XAUX_MUL_DGELU ? aux[(size_t)rc * ldaux + j] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rbase + mfma32_row(i, h);
                if (row >= M) continue;
                float v = acc[a][b][i] + bj;
                if (aux_mode == XAUX_STORE_PRE) {
                    aux[(size_t)row * ldaux + j] = v;
                } else if (aux_mode == XAUX_MUL_DGELU) {
                    const float sg = 1.0f / (1.0f + __expf(-1.702f * pv[i]));
                    v *= sg * (1.0f + 1.702f * pv[i] * (1.0f - sg));
                }
                if (act == XA_TANH) v = tanhf(v);
                else if (act == XA_RELU) v = fmaxf(v, 0.f);
                else if (act == XA_QUICKGELU) v = v / (1.0f + expf(-1.702f * v));
                C[(size_t)row * ldc + j] = v + rv[i];
            }
        }
    }
}

extern "C" int mil_split_bf16(const float* src, uint16_t* dst, size_t n, int pieces, void* stream) {
    if (!src || !dst || pieces < 1 || pieces > 3) return MIL_EINVAL;
    if (n == 0) return MIL_OK;
    hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n, pieces);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gemm_split(const float* A, int lda, const uint16_t* B_pieces, int pieces, int ldb, float* C, int ldc, int M,
                              int N, int K, const float* bias, int act, const float* residual, int ldr, float* aux,
                              int ldaux, int aux_mode, void* stream) {
    if (!A || !B_pieces || !C || M < 0 || N <= 0 || K <= 0) return MIL_EINVAL;
    if (M == 0) return MIL_OK;
    if ((pieces != 2 && pieces != 3) || (K % XG_BK) != 0 || (lda & 3) || (ldb & 7) || act < 0 || act > 3) return MIL_EINVAL;
    if (aux_mode < 0 || aux_mode > 2 || (aux_mode != XAUX_NONE && (!aux || ldaux < N))) return MIL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B_pieces)) & 15) return MIL_EINVAL;
    const dim3 grid((N + 127) / 128, (M + 127) / 128);
    const size_t ps = (size_t)N * ldb;
    hipStream_t st = (hipStream_t)stream;
    if (pieces == 2)
        hipLaunchKernelGGL((k_gemm_x<2>), grid, dim3(256), 0, st, A, lda, B_pieces, ps, ldb, C, ldc, M, N, K, bias, act, residual,
                           ldr, aux, ldaux, aux_mode);
    else
        hipLaunchKernelGGL((k_gemm_x<3>), grid, dim3(256), 0, st, A, lda, B_pieces, ps, ldb, C, ldc, M, N, K, bias, act, residual,
                           ldr, aux, ldaux, aux_mode);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
