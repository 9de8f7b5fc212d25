// K3a, tall NT products on the low-VALU pipeline of k_gate_fwd2 (gated_pool.hip):  C[M, N] = act(A[M, K] W[N, K]^T + bias)
// with both operands K-contiguous (nn.Linear on a tall activation: fc_pathology, model/aggregator.py:47, 141-149).
//
// The f32 MFMA does not hide vector-ALU instructions (tools/mfma_valu_mix.hip: every VALU instruction between two
// v_mfma_f32_32x32x2_f32 costs 3-5 cycles of matrix time), so the loop carries none: both operands arrive by LDS-DMA through
// buffer resources (SGPR base / extent advanced by scalar ALU, per-lane offset loop-invariant, rows beyond M read as zeros by
// the hardware range check), fragment reads are ds_read_b128 at precomputed addresses with immediate offsets (slice loop
// unrolled by two), one b128 feeds four MFMAs (k-permutation), DMA pieces and fragment reads are pinned between MFMA groups.
// Workgroup 512 threads = 256 rows x 256 columns: wave (wr, wc) owns 64 x 128 = 2 x 4 MFMA tiles (128 accumulators), so a
// k-group reads 2 + 4 fragments for 32 MFMAs and a 32-deep slice is 128 MFMAs per wave between two barriers; LDS 2 x (32 +
// 32) KB.  32 768 x 512 (fc_pathology) is exactly 256 workgroups = one round of the chip.
// Epilogue: bias, tanh / ReLU (v_rcp-based tanh as in the gate kernels), 128-byte row segments per store instruction.
#include "mil_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void nt2_lds_void;
typedef __attribute__((address_space(3))) const f32x4 nt2_lds_cf4;
#define NT2_TM 256
#define NT2_TN 256
#define NT2_BK 32
#define NT2_AS_BYTES (NT2_TM * NT2_BK * 4)      /* one A stage (32 KB) */
#define NT2_WS_BYTES (NT2_TN * NT2_BK * 4)      /* one W stage (32 KB) */
#define NT2_SRD_FLAGS 0x00020000
enum { NT2_ACT_NONE = 0, NT2_ACT_TANH = 1, NT2_ACT_RELU = 2 };

template <int ACT>
__global__ __launch_bounds__(512) void k_gemm_nt2(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                  float* __restrict__ C, int ldc, int M, int N, int K,
                                                  const float* __restrict__ bias) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (NT2_TM + NT2_TN) * NT2_BK];     // [2] A stages, then [2] W stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int nct = N / NT2_TN;
    const int ct = blockIdx.x % nct, rt = blockIdx.x / nct;
    const int row0 = rt * NT2_TM, col0 = ct * NT2_TN;
    const int nslice = K / NT2_BK;
    const unsigned lds0 = (unsigned)(uintptr_t)(nt2_lds_void*)smem;

    // ---- DMA pieces of this wave: 4 A pieces + 4 W pieces of 8 rows x 128 B; lane -> (row in piece, physical 16-byte chunk)
    const int prow = lane >> 3, pch = lane & 7;
    const int rows_here = min(NT2_TM, M - row0);
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)row0 * lda), 0, ((rows_here - 1) * lda + K) * 4, NT2_SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t srd_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)col0 * ldw), 0, ((NT2_TN - 1) * ldw + K) * 4, NT2_SRD_FLAGS);
    int vsrc[8];                 // per-lane byte offset of the piece's source (slice 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int lr = (4 * wave + (i & 3)) * 8 + prow;                           // 0..255
        const int ld_ = i < 4 ? lda : ldw;
        // rows of A beyond M: an offset past the resource's extent, so the range check returns zeros
        vsrc[i] = (i < 4 && lr >= rows_here) ? 0x7ffffff0 : (lr * ld_ + 4 * (pch ^ ((lr >> 1) & 7))) * 4;
    }
    auto dma_piece = [&](int i, int buf, int kbytes) {            // i, buf compile-time after unrolling; kbytes scalar
        const unsigned dst = lds0 + (unsigned)((i < 4 ? buf * NT2_AS_BYTES : 2 * NT2_AS_BYTES + buf * NT2_WS_BYTES) +
                                               (4 * wave + (i & 3)) * 8 * 128);
        if (i < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (nt2_lds_void*)(uintptr_t)dst, 16, vsrc[i], kbytes, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (nt2_lds_void*)(uintptr_t)dst, 16, vsrc[i], kbytes, 0, 0);
    };
    // ---- fragment addresses (bytes): row (64 wr + r) of the A image / row (128 wc + r) of the W image, swizzled chunk of k-group t
    const int fx = (r >> 1) & 7;
    unsigned fa_addr[4], fb_addr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned ch = 16u * (unsigned)((2 * t + h) ^ fx);
        fa_addr[t] = lds0 + (unsigned)((64 * wr + r) * 128) + ch;
        fb_addr[t] = lds0 + (unsigned)(2 * NT2_AS_BYTES + (128 * wc + r) * 128) + ch;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(i, 0, 0);
    __syncthreads();                                   // hipcc drains the DMA (vmcnt(0)) in front of the barrier

    auto slice = [&](int s, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const int s1 = min(s + 1, nslice - 1);
        const int k1bytes = s1 * NT2_BK * 4;
        f32x4 a[2][2], b[2][4];
        auto frag_piece = [&](int t, int q, int p) {                  // p: 0, 1 = A row tiles, 2..5 = W column tiles
            if (p < 2) a[q][p] = *(nt2_lds_cf4*)(uintptr_t)(fa_addr[t] + (unsigned)(buf * NT2_AS_BYTES + p * 32 * 128));
            else b[q][p - 2] = *(nt2_lds_cf4*)(uintptr_t)(fb_addr[t] + (unsigned)(buf * NT2_WS_BYTES + (p - 2) * 32 * 128));
        };
#pragma unroll
        for (int p = 0; p < 6; ++p) frag_piece(0, 0, p);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = 4 * t + j;
                if (g < 8) dma_piece(g, buf ^ 1, k1bytes);            // next slice, one DMA piece per MFMA group
                if (t < 3) {
                    frag_piece(t + 1, q ^ 1, j);
                    if (j < 2) frag_piece(t + 1, q ^ 1, 4 + j);
                }
#pragma unroll
                for (int ai = 0; ai < 2; ++ai)
#pragma unroll
                    for (int bi = 0; bi < 4; ++bi)
                        acc[ai][bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][ai][j], b[q][bi][j], acc[ai][bi], 0, 0, 0);
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

    // ---- epilogue: lane (r, h) holds rows mfma32_row(i, h) of column r of every tile
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        const int col = col0 + 128 * wc + 32 * bi + r;
        const float bb = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
        for (int ai = 0; ai < 2; ++ai) {
            float* o = C + (size_t)(row0 + 64 * wr + 32 * ai) * ldc + col;
            const int rbase = row0 + 64 * wr + 32 * ai;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[ai][bi][i] + bb;
                if (ACT == NT2_ACT_TANH) v = fast_tanh(v);
                else if (ACT == NT2_ACT_RELU) v = fmaxf(v, 0.f);
                const int rr = mfma32_row(i, h);
                if (rbase + rr < M) o[(size_t)rr * ldc] = v;
            }
        }
    }
}

// Shapes this kernel is built for: whole 256-column tiles, whole 32-deep slices, 16-byte aligned rows, offsets inside 31 bits.
extern "C" int mil_gemm_nt2_ok(int lda, int ldw, int M, int N, int K) {
    if (M <= 0 || N <= 0 || (N % NT2_TN) != 0 || K < 2 * NT2_BK || (K % NT2_BK) != 0 || (lda & 3) || (ldw & 3)) return 0;
    if ((long long)NT2_TM * lda * 4 >= 0x7fff0000ll || (long long)NT2_TN * ldw * 4 >= 0x7fff0000ll) return 0;
    return 1;
}

extern "C" int mil_gemm_nt2(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                            const float* bias, int act, void* stream) {
    if (!A || !W || !C || !mil_gemm_nt2_ok(lda, ldw, M, N, K) || ldc < N || lda < K || ldw < K) return MIL_EINVAL;
    if (act != NT2_ACT_NONE && act != NT2_ACT_TANH && act != NT2_ACT_RELU) return MIL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(((M + NT2_TM - 1) / NT2_TM) * (N / NT2_TN));
    switch (act) {
        case NT2_ACT_TANH: hipLaunchKernelGGL(k_gemm_nt2<NT2_ACT_TANH>, grid, dim3(512), 0, st, A, lda, W, ldw, C, ldc, M, N, K, bias); break;
        case NT2_ACT_RELU: hipLaunchKernelGGL(k_gemm_nt2<NT2_ACT_RELU>, grid, dim3(512), 0, st, A, lda, W, ldw, C, ldc, M, N, K, bias); break;
        default: hipLaunchKernelGGL(k_gemm_nt2<NT2_ACT_NONE>, grid, dim3(512), 0, st, A, lda, W, ldw, C, ldc, M, N, K, bias); break;
    }
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
