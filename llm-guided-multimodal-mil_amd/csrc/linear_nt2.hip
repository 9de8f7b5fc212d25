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
    // XCD-aware order: the nct column tiles of a row tile read the same 256 rows of A; the hardware deals workgroup ids
    // round-robin over the 8 XCDs, which put them on different L2s (PMC: 198 MiB fetched for 96 MiB of A at 32 768 x 768).
    // logical = (id % 8) * share + id / 8 (bijective for any grid size) keeps them on consecutive slots of one XCD.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int ct = bid % nct, rt = bid / nct;
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

// ================================================================================ tall TN products (weight gradients)
// dW[n][k] = sum_rows G[row][n] X[row][k],  G = dY (.) act'(Y),  plus db[n] = sum_rows G[row][n]: the parameter half of a
// Linear layer's backward on a tall activation (fc_pathology: 32 768 rows, 512 x 768 outputs), split over the rows.
// The low-VALU form of k_gate_bwd_dw2 (gated_pool.hip) with both operands staged alike: 512 threads = two K groups x four
// waves x (2 x 2) MFMA tiles on a 128 (n) x 128 (k) output tile; global operands through buffer resources whose base
// advances by scalar ALU (rows beyond the chunk read as zeros: no clamps, no masks), LDS images [32 rows][128] with their
// 32-column blocks in the order {0, 2, 1, 3} so that one ds_read2st64_b32 fetches both operand values of a k-step, staging
// pieces and reloads pinned between the MFMAs, the two groups' tiles folded through LDS before the 16-byte partial stores.
// partial [S][N][K] and colsum partial [S][N] are folded by k_splitk_reduce (linear.hip).
#define TN2_BKR 32
typedef unsigned int tn2_u32x4 __attribute__((ext_vector_type(4)));

template <int ACT>
__global__ __launch_bounds__(512) void k_gemm_tn2(const float* __restrict__ dY, int lddy, const float* __restrict__ Y, int ldy,
                                                  const float* __restrict__ X, int ldx, float* __restrict__ part,
                                                  float* __restrict__ cs_part, int rows, int N, int K, int KC, int NJ, int NM,
                                                  const int32_t* __restrict__ rows_dev) {
    if (rows_dev != nullptr) {
        // a capacity bucket: only the first rows_dev[0] rows carry gradients - split THOSE evenly over the launch's row chunks
        rows = min(rows, __builtin_amdgcn_readfirstlane(rows_dev[0]));
        const int S = (int)gridDim.x / (NM * NJ);
        KC = max(2 * TN2_BKR, ((rows + S - 1) / S + 2 * TN2_BKR - 1) / (2 * TN2_BKR) * (2 * TN2_BKR));
    }
    __shared__ __attribute__((aligned(16))) float smem_all[2 * 2 * 2 * TN2_BKR * 128];
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    float* smem = smem_all + grp * (2 * 2 * TN2_BKR * 128);     // this K group's stages: [2][32][128] G, [2][32][128] X
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int jt = bid % NJ, m = (bid / NJ) % NM, s = bid / (NM * NJ);
    const int j0 = jt * 128, n0 = m * 128;
    const int cbeg = min(s * KC, rows), cend = min(rows, cbeg + KC);
    const int half = ((cend - cbeg + 2 * TN2_BKR - 1) / (2 * TN2_BKR)) * TN2_BKR;
    const int rbeg = min(cend, cbeg + grp * half), rend = min(cend, rbeg + half);
    const int nloop = (min(half, cend - cbeg) + TN2_BKR - 1) / TN2_BKR;      // common to both groups (shared barriers)

    // per-lane byte offsets inside a slice (loop invariants): rows xrow + 8 i (i < 4), 16-byte chunk xc4 of the 128-column tile
    const int xrow = tid >> 5, xc4 = tid & 31;
    int vx[4], vg[4], vy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        vx[i] = ((xrow + 8 * i) * ldx + j0 + 4 * xc4) * 4;
        vg[i] = ((xrow + 8 * i) * lddy + n0 + 4 * xc4) * 4;
        vy[i] = ((xrow + 8 * i) * ldy + n0 + 4 * xc4) * 4;
    }
    auto blkpos = [](int c) { return (c & 31) | ((c & 32) << 1) | ((c & 64) >> 1); };
    float* const ab = smem;                       // [2][32][128]
    float* const xb = smem + 2 * TN2_BKR * 128;   // [2][32][128]
    float* const xw = xb + xrow * 128 + blkpos(4 * xc4);              // + (buf * 32 + 8 i) * 128
    float* const aw = ab + xrow * 128 + blkpos(4 * xc4);
    const float* const ap = ab + h * 128 + 32 * wi + r;               // + buf * 4096 + ks * 256 (+ 64)
    const float* const bp = xb + h * 128 + 32 * wj + r;

    tn2_u32x4 rx[4], rg[4], ry[4];
    f32x4 acc_b = {0, 0, 0, 0};
    const bool pub = jt == 0;                     // this workgroup's G sums are the chunk's bias partials (scalar)

    auto x_srd = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)row0 * ldx), 0, max(rend - row0, 0) * ldx * 4, NT2_SRD_FLAGS);
    };
    auto g_srd = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(dY + (size_t)row0 * lddy), 0, max(rend - row0, 0) * lddy * 4, NT2_SRD_FLAGS);
    };
    auto y_srd = [&](int row0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(Y + (size_t)row0 * ldy), 0, max(rend - row0, 0) * ldy * 4, NT2_SRD_FLAGS);
    };
    auto xload = [&](int i, int row0) { rx[i] = __builtin_amdgcn_raw_buffer_load_b128(x_srd(row0), vx[i], 0, 0); };
    auto aload = [&](int i, int row0) {
        rg[i] = __builtin_amdgcn_raw_buffer_load_b128(g_srd(row0), vg[i], 0, 0);
        if (ACT != NT2_ACT_NONE) ry[i] = __builtin_amdgcn_raw_buffer_load_b128(y_srd(row0), vy[i], 0, 0);
    };
    auto xwrite = [&](int i, int buf) {
        *reinterpret_cast<f32x4*>(xw + (buf * TN2_BKR + 8 * i) * 128) = __builtin_bit_cast(f32x4, rx[i]);
    };
    auto awrite = [&](int i, int buf) {
        f32x4 g = __builtin_bit_cast(f32x4, rg[i]);
        if (ACT == NT2_ACT_TANH) {
            const f32x4 y = __builtin_bit_cast(f32x4, ry[i]);
            g = g - (g * y) * y;                      // dY (1 - Y^2)
        } else if (ACT == NT2_ACT_RELU) {
            const f32x4 y = __builtin_bit_cast(f32x4, ry[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(aw + (buf * TN2_BKR + 8 * i) * 128) = g;
        if (pub) acc_b += g;
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
    for (int i = 0; i < 4; ++i) { xload(i, rbeg); aload(i, rbeg); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { xwrite(i, 0); awrite(i, 0); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { xload(i, rbeg + TN2_BKR); aload(i, rbeg + TN2_BKR); }
    __syncthreads();

    // one slice: 16 k-steps x 4 MFMAs; between the MFMAs the parts that stage slice sl + 1 and reload slice sl + 2
    auto slice = [&](int sl, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const int row2 = rbeg + (sl + 2) * TN2_BKR;
        const float* apb = ap + buf * TN2_BKR * 128;
        const float* bpb = bp + buf * TN2_BKR * 128;
        float fa[2][2], fb[2][2];
        fa[0][0] = apb[0]; fa[0][1] = apb[64]; fb[0][0] = bpb[0]; fb[0][1] = bpb[64];
#pragma unroll
        for (int ks = 0; ks < TN2_BKR / 2; ++ks) {
            const int q = ks & 1;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < TN2_BKR / 2) {
                fa[q ^ 1][0] = apb[(ks + 1) * 256]; fa[q ^ 1][1] = apb[(ks + 1) * 256 + 64];
                fb[q ^ 1][0] = bpb[(ks + 1) * 256]; fb[q ^ 1][1] = bpb[(ks + 1) * 256 + 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][1], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks >= 1 && ks <= 4) xwrite(ks - 1, buf ^ 1);
            if (ks >= 5 && ks <= 8) awrite(ks - 5, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][0], acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks >= 1 && ks <= 4) xload(ks - 1, row2);
            if (ks >= 5 && ks <= 8) aload(ks - 5, row2);
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

    // partial tile -> part[s][n0 + 64 wi + row][j0 + 64 wj + col]: every wave writes its 64 x 64 tile row-major into its own
    // 16 KB of its group's (now dead) staging area; the two waves that own the same tile (one per K group) each fold and
    // store half of its rows with 16-byte stores
    {
        float* tw = smem + wave * (64 * 64);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) tw[(32 * a + mfma32_row(i, h)) * 64 + 32 * b + r] = acc[a][b][i];
        __syncthreads();
        float* pt = part + ((size_t)s * N + n0 + 64 * wi) * K + j0 + 64 * wj;
        const int c4 = lane & 15, rr = lane >> 4;
        const float* t0 = smem_all + wave * (64 * 64);
        const float* t1 = smem_all + (2 * 2 * TN2_BKR * 128) + wave * (64 * 64);
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = 4 * (pass + 8 * grp) + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(t0 + row * 64 + 4 * c4) +
                            *reinterpret_cast<const f32x4*>(t1 + row * 64 + 4 * c4);
            *reinterpret_cast<f32x4*>(pt + (size_t)row * K + 4 * c4) = v;
        }
    }
    // bias partial of this chunk: cs_part[s][n0 .. n0 + 127] (workgroups with jt == 0)
    if (pub && cs_part != nullptr) {
        float* redf = smem_all;                    // [16 row groups][128]
        const int rowg = xrow + 8 * grp;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) redf[rowg * 128 + 4 * xc4 + e] = acc_b[e];
        __syncthreads();
        if (threadIdx.x < 128) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) v += redf[g * 128 + threadIdx.x];
            cs_part[(size_t)s * N + n0 + threadIdx.x] = v;
        }
    }
}

static inline int tn2_plan(int rows, int N, int K, int* KC_out) {
    const int tiles = (N / 128) * (K / 128);
    if (tiles <= 0 || rows <= 0) { *KC_out = 2 * TN2_BKR; return 0; }   // not a shape of this kernel
    int smax = MIL_NUM_CU / tiles;                 // one 512-thread workgroup per CU
    if (smax < 1) smax = 1;
    int kc = ((rows + smax - 1) / smax + 2 * TN2_BKR - 1) / (2 * TN2_BKR) * (2 * TN2_BKR);
    if (kc < 2 * TN2_BKR) kc = 2 * TN2_BKR;
    *KC_out = kc;
    return (rows + kc - 1) / kc;
}

// 1 when mil_linear_bwd_params should take this kernel: whole 128-tiles, a row chunk deep enough for the two K groups, offsets
// inside 31 bits, and at least 3/4 of the chip busy.
extern "C" int mil_gemm_tn2_ok(int lddy, int ldy, int ldx, int rows, int N, int K) {
    if (rows < 4096 || N <= 0 || K <= 0 || (N % 128) != 0 || (K % 128) != 0 || (lddy & 3) || (ldy & 3) || (ldx & 3)) return 0;
    int kc;
    const int S = tn2_plan(rows, N, K, &kc);
    if ((long long)kc * (lddy > ldx ? lddy : ldx) * 4 >= 0x7fff0000ll || (long long)kc * ldy * 4 >= 0x7fff0000ll) return 0;
    if ((long)S * (N / 128) * (K / 128) * 4 < 3 * MIL_NUM_CU || kc < 512) return 0;
    return 1;
}
extern "C" int mil_gemm_tn2_splits(int rows, int N, int K) {
    if (rows <= 0 || N < 128 || K < 128 || (N % 128) != 0 || (K % 128) != 0) return 0;
    int kc;
    return tn2_plan(rows, N, K, &kc);
}

int mil_gemm_tn2_rows(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx, int rows, int N, int K,
                      float* partial, float* cs_partial, const int32_t* rows_dev, void* stream) {
    if (!dY || !X || !partial || !mil_gemm_tn2_ok(lddy, ldy, ldx, rows, N, K)) return MIL_EINVAL;
    if (act != NT2_ACT_NONE && act != NT2_ACT_TANH && act != NT2_ACT_RELU) return MIL_EINVAL;
    if (act != NT2_ACT_NONE && !Y) return MIL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) return MIL_EINVAL;
    int kc;
    const int S = tn2_plan(rows, N, K, &kc);
    const int NM = N / 128, NJ = K / 128;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(S * NM * NJ);
    switch (act) {
        case NT2_ACT_TANH: hipLaunchKernelGGL(k_gemm_tn2<NT2_ACT_TANH>, grid, dim3(512), 0, st, dY, lddy, Y, ldy, X, ldx, partial, cs_partial, rows, N, K, kc, NJ, NM, rows_dev); break;
        case NT2_ACT_RELU: hipLaunchKernelGGL(k_gemm_tn2<NT2_ACT_RELU>, grid, dim3(512), 0, st, dY, lddy, Y, ldy, X, ldx, partial, cs_partial, rows, N, K, kc, NJ, NM, rows_dev); break;
        default: hipLaunchKernelGGL(k_gemm_tn2<NT2_ACT_NONE>, grid, dim3(512), 0, st, dY, lddy, Y ? Y : dY, Y ? ldy : lddy, X, ldx, partial, cs_partial, rows, N, K, kc, NJ, NM, rows_dev); break;
    }
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
extern "C" int mil_gemm_tn2(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx, int rows,
                            int N, int K, float* partial, float* cs_partial, void* stream) {
    return mil_gemm_tn2_rows(dY, lddy, Y, ldy, act, X, ldx, rows, N, K, partial, cs_partial, nullptr, stream);
}
