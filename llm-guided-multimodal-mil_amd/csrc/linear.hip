// K3a: generic fp32-MFMA GEMM with fused epilogue, used for every dense projection of the path:
//   fc_pathology / fc_CI2* (model/aggregator.py:44-68), the q/k/v/out projections and MLP of the SAM
//   two-way transformer (model/sam/transformer.py:413-416, common.py:21-26), the CLIP text blocks
//   (clip/model.py:171-178) and their backward products.
//
//   C[M,N] (+)= epilogue( A_op[M,K] . B_op[K,N] )
//     a_mode 0: A_op[i][k] = A[i*lda + k]   (row-major activations)
//     a_mode 1: A_op[i][k] = A[k*lda + i]   (transposed use: dW = dY^T X)
//     b_mode 0: B_op[k][j] = B[j*ldb + k]   (nn.Linear weight [N, K]:  y = x W^T)
//     b_mode 1: B_op[k][j] = B[k*ldb + j]   (dx = dy W,  dW = dY^T X)
//   instantiated: NT (0,0), NN (0,1), TN (1,1).
//
// Workgroup 256 threads = 4 waves, tile 128 x 128 x 32, wave (wi, wj) owns 64 x 64 = 2 x 2 MFMA
// 32x32x2 f32 tiles.  k-contiguous operands live in LDS as [128][36] and are read with ds_read_b128
// (lane (r, h) takes k = 8t+4h..+3, one element per MFMA of the group); k-major operands as [32][128]
// read with ds_read_b32 at row 8t+4h+jj: both use the same k order, so modes mix freely.
// Staging is branch-free (indices clamped, out-of-range k zeroed by a mask applied at LDS-write
// time) and issued as 8 pieces between MFMA groups; registers hold the slice after next.
#include "mil_common.h"

#define LG_BK 32
#define LG_KS 36       // k-contiguous image row stride (words)

enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2, ACT_QUICKGELU = 3 };

// Grouped launches (one group per bag): the skinny products of the absorbed multi-token attention, where every bag
// multiplies its own rows with its own small matrix.
enum { GRP_NONE = 0, GRP_ROWS = 1, GRP_CONTRACT = 2 };
struct GemmGroups {
    const int32_t* off;      // [G + 1] row offsets
    long strideB, strideC, strideBias;
    int mode, splits;
};

// Weight-gradient extras (a_mode 1 only, template flag AX): the A operand is dY (.) act'(Y) formed while it is staged
// (a_aux = the layer's saved output Y, a_act = ACT_TANH / ACT_RELU / ACT_NONE), and the column sums of that operand -
// the bias gradient - are accumulated by the workgroups of the first column tile into a_colsum[split][M].  One launch
// (+ the split-K fold) then yields dW and db without a stand-alone activation-backward or column-sum pass.
struct GemmExtra {
    const float* a_aux;
    float* a_colsum;
    int ld_aux, a_act;
};

// aux_mode 1: also store the pre-activation (bias added, before act) to aux - the backward of QuickGELU needs it;
// aux_mode 2: multiply by QuickGELU'(aux[row][j]) - the activation backward fused into the epilogue of the product
// that forms the gradient (clip/model.py:162-164 inside the MLP of :176-178).
enum { AUX_NONE = 0, AUX_STORE_PRE = 1, AUX_MUL_DGELU = 2 };
__device__ __forceinline__ float epilogue_aux(float v, float* __restrict__ aux, int ldaux, int aux_mode, int row, int j) {
    if (aux_mode == AUX_STORE_PRE) {
        aux[(size_t)row * ldaux + j] = v;
    } else if (aux_mode == AUX_MUL_DGELU) {
        const float p = aux[(size_t)row * ldaux + j];
        const float sg = 1.0f / (1.0f + expf(-1.702f * p));
        v *= sg * (1.0f + 1.702f * p * (1.0f - sg));
    }
    return v;
}

template <int MODE>
struct OperandTile {
    // MODE 0: [128 rows][32 k] from row-major src (k contiguous);  MODE 1: [32 k][128 cols] from k-major src.
    const float* src[4];
    int lds_off[4];
    float kmask_k[4];     // MODE 1: local k row of each piece (mask is recomputed per slice)
    f32x4 reg[4];
    float mask[4];
    int ld, K, tid;

    __device__ __forceinline__ void init(const float* base, int ld_, int origin, int extent, int K_, int tid_) {
        ld = ld_; K = K_; tid = tid_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) {
                const int row = (tid >> 3) + 32 * i, ch = tid & 7;
                const int gr = min(origin + row, extent - 1);
                src[i] = base + (size_t)gr * ld + 4 * ch;
                lds_off[i] = row * LG_KS + 4 * ch;
            } else {
                const int kr = (tid >> 5) + 8 * i, c4 = tid & 31;
                const int gc = min(origin + 4 * c4, max(extent - 4, 0));
                src[i] = base + gc;
                lds_off[i] = kr * 128 + 4 * c4;
                kmask_k[i] = (float)kr;
            }
            mask[i] = 1.f;
        }
    }
    __device__ __forceinline__ void load(int i, int k0) {
        if (MODE == 0) {
            reg[i] = *reinterpret_cast<const f32x4*>(src[i] + k0);   // K % 32 == 0 here: k0 <= K - 32
        } else {
            const int kr = k0 + (tid >> 5) + 8 * i;
            reg[i] = *reinterpret_cast<const f32x4*>(src[i] + (size_t)min(kr, K - 1) * ld);
            mask[i] = kr < K ? 1.f : 0.f;
        }
    }
    __device__ __forceinline__ void store(int i, float* lds) const {
        if (MODE == 0) *reinterpret_cast<f32x4*>(lds + lds_off[i]) = reg[i];
        else *reinterpret_cast<f32x4*>(lds + lds_off[i]) = reg[i] * mask[i];
    }
};

template <int AMODE, int BMODE, bool AX = false>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, int M, int N, int K, int kchunk,
                                              const float* __restrict__ bias, int act, const float* __restrict__ residual,
                                              int ldr, int accumulate, float* __restrict__ partial,
                                              float* __restrict__ aux, int ldaux, int aux_mode, GemmGroups gg,
                                              GemmExtra ex = GemmExtra{nullptr, nullptr, 0, 0}) {
    static_assert(!AX || AMODE == 1, "operand extras are defined for the k-major A operand only");
    constexpr int ASZ = AMODE == 0 ? 128 * LG_KS : LG_BK * 128;
    constexpr int BSZ = BMODE == 0 ? 128 * LG_KS : LG_BK * 128;
    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
    float* as = smem;
    float* bs = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own
    // L2), so the column tiles of one row tile - which read the same A rows - would land on different XCDs and every A
    // row would come from HBM once per column tile.  Logical id = (id % 8) * share + id / 8 (bijective for any grid)
    // keeps consecutive logical tiles on one XCD, dispatched within a few slots of each other.  (Times are unchanged on
    // MI355X - the Infinity Cache already absorbs the cross-XCD re-reads; the order only spares L2 -> fabric traffic.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#if !defined(LG_NOXCD)
    {
        const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        int lin = bx + gx * (by + gy * bz);
        const int q = nwg >> 3, rem = nwg & 7, xcd = lin & 7;
        lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (lin >> 3);
        bx = lin % gx;
        by = (lin / gx) % gy;
        bz = lin / (gx * gy);
    }
#endif
    int i0 = by * 128;
    const int j0 = bx * 128;
    int kbeg = bz * kchunk, kend = min(K, kbeg + kchunk);
    if (gg.mode != GRP_NONE) {
        // grouped launch: bz is the group (bag); no split-K
        const int g = bz, goff = gg.off[g], gn = gg.off[g + 1] - goff;
        if (gg.mode == GRP_ROWS) {                 // rows of A / C belong to groups, B (and bias) are per group
            M = gn;
            if (i0 >= M) return;
            A += (size_t)goff * lda;
            C += (size_t)goff * ldc;
            if (residual != nullptr) residual += (size_t)goff * ldr;
            B += (size_t)g * gg.strideB;
            if (bias != nullptr) bias += (size_t)g * gg.strideBias;
            kbeg = 0;
            kend = K;
        } else {                                   // GRP_CONTRACT: the contraction runs over the group's rows, C is per group
            K = gn;
            A += (size_t)goff * lda;
            B += (size_t)goff * ldb;
            kbeg = 0;
            kend = K;
            if (gg.splits > 1) {
                // M <= 128: by is a chunk of the group's rows instead of a row tile; partial C tiles go to
                // C[(g * splits + chunk)] (a workspace) and are summed by k_grouped_fold
                const int sidx = by;
                const int chunk = ((gn + gg.splits - 1) / gg.splits + LG_BK - 1) / LG_BK * LG_BK;
                kbeg = min(gn, sidx * chunk);
                kend = min(gn, kbeg + chunk);
                i0 = 0;
                C += ((size_t)g * gg.splits + sidx) * gg.strideC;
            } else {
                C += (size_t)g * gg.strideC;
            }
        }
    }
    const int nslice = (kend - kbeg + LG_BK - 1) / LG_BK;

    OperandTile<AMODE> ta;
    OperandTile<BMODE> tb;
    // the K seen by the loaders is this block's chunk end (so the zero mask also cuts the split-K chunk)
    ta.init(A, lda, i0, M, kend, tid);
    tb.init(B, ldb, j0, N, kend, tid);
    OperandTile<AMODE> tx;                          // AX: the saved layer output, same tile as A
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};              // AX: column sums of the staged A operand (this thread's 4 columns)
    const bool ax_act = AX && ex.a_aux != nullptr;
    if (ax_act) tx.init(ex.a_aux, ex.ld_aux, i0, M, kend, tid);
    auto a_load = [&](int i, int k0) {
        ta.load(i, k0);
        if (AX) { if (ax_act) tx.load(i, k0); }
    };
    auto a_store = [&](int i, float* lds) {
        if (AX) {
            f32x4 v = ta.reg[i] * ta.mask[i];
            if (ax_act) {
                const f32x4 y = tx.reg[i];
                if (ex.a_act == ACT_TANH) {
                    v = v * (1.0f - y * y);
                } else if (ex.a_act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = y[e] > 0.f ? v[e] : 0.f;
                }
            }
            *reinterpret_cast<f32x4*>(lds + ta.lds_off[i]) = v;
            csum += v;
        } else {
            ta.store(i, lds);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    if (nslice > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a_load(i, kbeg); tb.load(i, kbeg); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { a_store(i, as); tb.store(i, bs); }
        const int k1 = kbeg + min(1, nslice - 1) * LG_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a_load(i, k1); tb.load(i, k1); }
    }
    __syncthreads();
    for (int s = 0; s < nslice; ++s) {
        const int buf = s & 1;
        const int k2 = kbeg + min(s + 2, nslice - 1) * LG_BK;
        const float* ab = as + buf * ASZ;
        const float* bb = bs + buf * BSZ;
        float* an = as + (buf ^ 1) * ASZ;
        float* bn = bs + (buf ^ 1) * BSZ;
        f32x4 fa[2][2], fb[2][2];     // [register set][tile]
        auto frag_a = [&](int t, int q, int a) {
            if (AMODE == 0) {
                fa[q][a] = *reinterpret_cast<const f32x4*>(ab + (64 * wi + 32 * a + r) * LG_KS + 8 * t + 4 * h);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) fa[q][a][jj] = ab[(8 * t + 4 * h + jj) * 128 + 64 * wi + 32 * a + r];
            }
        };
        auto frag_b = [&](int t, int q, int b) {
            if (BMODE == 0) {
                fb[q][b] = *reinterpret_cast<const f32x4*>(bb + (64 * wj + 32 * b + r) * LG_KS + 8 * t + 4 * h);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) fb[q][b][jj] = bb[(8 * t + 4 * h + jj) * 128 + 64 * wj + 32 * b + r];
            }
        };
        frag_a(0, 0, 0); frag_a(0, 0, 1); frag_b(0, 0, 0); frag_b(0, 0, 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t & 1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int g = 4 * t + jj;
                if (g >= 2 && g < 10) {            // staging piece g-2: next slice into LDS, reload with the slice after
                    const int pc = g - 2;
                    if (pc < 4) {
                        // AX: the registers of the last iteration hold a second copy of the final slice - it must not
                        // enter the column sums (its LDS image is never read)
                        if (!AX || s + 1 < nslice) a_store(pc, an);
                        a_load(pc, k2);
                    }
                    else { tb.store(pc - 4, bn); tb.load(pc - 4, k2); }
                }
                if (t < 3) {                        // next k-group's fragments: one tile per MFMA group
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

    if (AX) {
        if (ex.a_colsum != nullptr && bx == 0) {
            // fold the 8 k-row groups of threads (tid >> 5) that share the column quad tid & 31
            __syncthreads();
            float* red = smem;                       // [8][128]
            *reinterpret_cast<f32x4*>(red + (tid >> 5) * 128 + 4 * (tid & 31)) = csum;
            __syncthreads();
            if (tid < 128 && i0 + tid < M) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) v += red[q * 128 + tid];
                ex.a_colsum[(size_t)bz * M + i0 + tid] = v;
            }
        }
    }

    // epilogue: lane holds column j, 16 rows per tile.  The residual / auxiliary / accumulate operands of a tile are
    // loaded as one batch of 16 (clamped rows, no branches) before any arithmetic, so their latency is paid once per
    // tile instead of once per element.
    const bool split = partial != nullptr;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + 64 * wj + 32 * b + r;
        if (j >= N) continue;
        const float bj = (!split && bias != nullptr) ? bias[j] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int rbase = i0 + 64 * wi + 32 * a;
            if (split) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = rbase + mfma32_row(i, h);
                    if (row < M) partial[((size_t)bz * M + row) * N + j] = acc[a][b][i];
                }
                continue;
            }
            float rv[16], pv[16], cv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rc = min(rbase + mfma32_row(i, h), M - 1);
                rv[i] = residual != nullptr ? residual[(size_t)rc * ldr + j] : 0.f;
                pv[i] = aux_mode == AUX_MUL_DGELU ? aux[(size_t)rc * ldaux + j] : 0.f;
                cv[i] = accumulate ? C[(size_t)rc * ldc + j] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rbase + mfma32_row(i, h);
                if (row >= M) continue;
                float v = acc[a][b][i] + bj;
                if (aux_mode == AUX_STORE_PRE) {
                    aux[(size_t)row * ldaux + j] = v;
                } else if (aux_mode == AUX_MUL_DGELU) {
                    const float sg = 1.0f / (1.0f + __expf(-1.702f * pv[i]));
                    v *= sg * (1.0f + 1.702f * pv[i] * (1.0f - sg));
                }
                if (act == ACT_TANH) v = tanhf(v);
                else if (act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (act == ACT_QUICKGELU) v = v / (1.0f + expf(-1.702f * v));
                C[(size_t)row * ldc + j] = v + rv[i] + cv[i];
            }
        }
    }
}

// split-K reduce with the full epilogue: C (+)= act(sum_s partial[s] + bias) + residual
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ partial, int S, float* __restrict__ C,
                                                       int ldc, int M, int N, const float* __restrict__ bias, int act,
                                                       const float* __restrict__ residual, int ldr, int accumulate,
                                                       float* __restrict__ aux, int ldaux, int aux_mode,
                                                       const float* __restrict__ cs_part = nullptr,
                                                       float* __restrict__ cs_out = nullptr, int cs_accumulate = 0) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (cs_part != nullptr && idx < (size_t)M) {         // bias gradient: fold the per-split column sums of the A operand
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += cs_part[(size_t)s * M + idx];
        cs_out[idx] = cs_accumulate ? cs_out[idx] + v : v;
    }
    if (idx >= (size_t)M * N) return;
    const int row = (int)(idx / N), j = (int)(idx % N);
    float v = 0.f;
    {
        const float* src = partial + (size_t)row * N + j;
        const size_t stride = (size_t)M * N;
        int s = 0;
        for (; s + 8 <= S; s += 8) {                      // eight partials in flight per thread
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(s + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; s < S; ++s) v += src[(size_t)s * stride];
    }
    if (bias != nullptr) v += bias[j];
    v = epilogue_aux(v, aux, ldaux, aux_mode, row, j);
    if (act == ACT_TANH) v = tanhf(v);
    else if (act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == ACT_QUICKGELU) v = v / (1.0f + expf(-1.702f * v));
    if (residual != nullptr) v += residual[(size_t)row * ldr + j];
    float* o = C + (size_t)row * ldc + j;
    if (accumulate) v += *o;
    *o = v;
}

// out[j] (+)= sum_i Y[i][j]   (bias gradients, LayerNorm parameter gradients).
// grid = (ceil(N / 64), nchunk): workgroup (cb, ch) sums rows [ch*rpc, (ch+1)*rpc) of 64 columns with 4 row lanes;
// with nchunk > 1 the result goes to part[ch][N] and a second launch (nchunk = 1) folds the chunks, so the sum
// order is fixed (no atomics) and a tall matrix still fills the chip.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ Y, int ldy, int M, int N, int rpc,
                                                float* __restrict__ out, int ldo, int accumulate) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rpc, r1 = min(M, r0 + rpc);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (c < N) {
        int i = r0 + g;
        for (; i + 12 < r1; i += 16) {
            v0 += Y[(size_t)i * ldy + c];
            v1 += Y[(size_t)(i + 4) * ldy + c];
            v2 += Y[(size_t)(i + 8) * ldy + c];
            v3 += Y[(size_t)(i + 12) * ldy + c];
        }
        for (; i < r1; i += 4) v0 += Y[(size_t)i * ldy + c];
    }
    red[g][threadIdx.x & 63] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (g == 0 && c < N) {
        float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        float* o = out + (size_t)blockIdx.y * ldo + c;
        if (accumulate) v += *o;
        *o = v;
    }
}

// dpre = dy * act'(y) elementwise (y = post-activation output).  n multiple of 4 not required.
__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                 float* __restrict__ dpre, size_t n, int act) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float yy = y[i], g = dy[i];
    dpre[i] = act == ACT_TANH ? g * (1.0f - yy * yy) : act == ACT_RELU ? (yy > 0.f ? g : 0.f) : g;
}

#include "gemm64.h"

// Split-K policy.  Weight gradients (a_mode 1: K = number of rows) always split to fill the chip.  The other forms
// split only when the output is a handful of tiles (token-side projections: M = bags x text tokens <= a few dozen
// rows), where a single 128 x 128 workgroup per tile would walk all of K alone: 30-60 us of latency for <0.1 GFLOP.
static int splitk_plan(int M, int N, int K, int a_mode, int* kchunk_out) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int S;
    if (a_mode == 1) {
        S = (2 * MIL_NUM_CU) / (tiles > 0 ? tiles : 1);
    } else {
        if (K < 256) { *kchunk_out = K; return 1; }
        const int slots = 2 * MIL_NUM_CU;
        if (tiles > 32) {
            // One partial round of the 512 resident workgroups (e.g. 10 K text-tower rows x N = 512 = 324 tiles): a round costs
            // its full K walk however empty it is, so splitting K into S makes it ceil(tiles S / slots) / S of that - if
            // the partial-sum traffic (M N floats written and read per split) is cheaper than the gain.  Times in us:
            // ~6 us per 32-deep slice pair of a full round, ~4 TB/s for the partials.
            if (tiles >= slots) { *kchunk_out = K; return 1; }
            const float t_round = 6.0f * (float)K / 32.0f;
            const float t_part = 8.0f * (float)M * (float)N / 4.0e6f;
            float best = t_round;
            S = 1;
            for (int c = 2; c <= 6 && c <= K / 128; ++c) {
                const float cost = (float)((tiles * c + slots - 1) / slots) / (float)c * t_round + (float)c * t_part;
                if (cost < 0.9f * best) { best = cost; S = c; }
            }
        } else {
            S = slots / tiles;
            if (S > K / 64) S = K / 64;                 // at least two 32-deep slices per split
        }
    }
    const int maxS = (K + LG_BK - 1) / LG_BK;
    if (S > maxS) S = maxS;
    if (S < 2) { *kchunk_out = K; return 1; }
    const int kchunk = ((K + S - 1) / S + LG_BK - 1) / LG_BK * LG_BK;
    S = (K + kchunk - 1) / kchunk;
    *kchunk_out = S > 1 ? kchunk : K;
    return S;
}

// Tile quantisation of a tall product (a_mode 0): with two 128 x 128 workgroups per CU a launch runs in rounds of 512
// tiles, and a last round that is mostly empty still costs a whole round (M = 24 640 text-tower rows x N = 512:
// 772 tiles = 1.5 rounds -> 2).  The rows of that last partial round are issued as a second launch with split-K
// chosen to fill the round (772 -> 512 + 260 x 2), its partials folded by k_splitk_reduce with the epilogue.
static bool tail_plan(int M, int N, int K, int a_mode, int* rows_main, int* S_out, int* kchunk_out) {
    if (a_mode != 0) return false;
    const int slots = 2 * MIL_NUM_CU;
    const int ct = (N + 127) / 128, rt = (M + 127) / 128;
    if (ct > slots || (long)rt * ct <= slots) return false;
    const int per_round = slots / ct;
    const int rem_rt = rt % per_round;
    if (rem_rt == 0) return false;
    const int tiles_rem = rem_rt * ct;
    if (10 * tiles_rem > 7 * slots) return false;          // the last round is already mostly full
    // split factor: the tail then takes ceil(tiles_rem S / slots) / S of a round, plus the partial-sum traffic
    // (charged 0.04 of a round per split: 17 MB written and read per split at M_tail = 8 K rows, N = 512)
    int S = 1;
    float best = 1.0f;
    for (int c = 2; c <= 8 && c <= K / 64; ++c) {
        const float cost = (float)((tiles_rem * c + slots - 1) / slots) / (float)c + 0.04f * (float)c;
        if (cost < best - 0.05f) { best = cost; S = c; }
    }
    if (S < 2) return false;
    const int kchunk = ((K + S - 1) / S + LG_BK - 1) / LG_BK * LG_BK;
    S = (K + kchunk - 1) / kchunk;
    if (S < 2) return false;
    *rows_main = (rt - rem_rt) * 128;
    *S_out = S;
    *kchunk_out = kchunk;
    return true;
}

extern "C" size_t mil_gemm_workspace_floats(int M, int N, int K, int a_mode) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int kchunk, rows_main, S_tail;
#if !defined(LG_NO_TILE64N)
    {
        int S64;
        if (small_tile_plan(M, N, K, a_mode, &S64, &kchunk)) return S64 > 1 ? (size_t)S64 * M * N : 0;
    }
#endif
    if (tail_plan(M, N, K, a_mode, &rows_main, &S_tail, &kchunk)) return (size_t)S_tail * (M - rows_main) * N;
    const int S = splitk_plan(M, N, K, a_mode, &kchunk);
    return S > 1 ? (size_t)S * M * N : 0;
}

static int gemm_impl(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc, int M,
                     int N, int K, const float* bias, int act, const float* residual, int ldr, int accumulate,
                     float* workspace, size_t workspace_floats, float* aux, int ldaux, int aux_mode, void* stream,
                     const int32_t* rows_dev = nullptr, int* rows_honoured = nullptr) {
    if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return MIL_EINVAL;
    if (M == 0 || N == 0) return MIL_OK;
    if ((lda & 3) || (ldb & 3) || act < 0 || act > 3) return MIL_EINVAL;
    if (a_mode == 0 && (K % LG_BK) != 0) return MIL_EINVAL;        // k-contiguous operands: whole slices only
    if (b_mode == 0 && (K % LG_BK) != 0) return MIL_EINVAL;
    if (a_mode == 1 && (M < 4 || (M & 3))) return MIL_EINVAL;      // k-major operands: 16-byte columns
    if (b_mode == 1 && (N < 4 || (N & 3))) return MIL_EINVAL;
    if (a_mode == 1 && b_mode == 0) return MIL_EINVAL;             // TT form is never needed
    hipStream_t st = (hipStream_t)stream;
#if !defined(LG_NO_NT2)
    if (a_mode == 0 && b_mode == 0 && residual == nullptr && !accumulate && aux_mode == AUX_NONE && act <= ACT_RELU &&
        ldc >= N && mil_gemm_nt2_ok(lda, ldb, M, N, K) && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0) {
        // tall NT product whose 256 x 256 tiles fill whole rounds of the chip (fc_pathology: 32 768 x 512 = 256 tiles): the
        // low-VALU LDS-DMA kernel (linear_nt2.hip)
        const long tiles = (long)((M + 255) / 256) * (N / 256);
        const long rounds = (tiles + MIL_NUM_CU - 1) / MIL_NUM_CU;
        if (rows_dev == nullptr && tiles >= (3 * MIL_NUM_CU) / 4 && 8 * tiles >= 7 * rounds * MIL_NUM_CU)
            return mil_gemm_nt2(A, lda, B, ldb, C, ldc, M, N, K, bias, act, stream);
    }
#endif
#if !defined(LG_NO_TILE64)
    if (a_mode == 0) {
        // Tall products take 64 x 128 tiles (gemm64.h: three workgroups per CU = 768 slots) as soon as those fill half
        // of the chip: finer rounds and a third resident workgroup beat the 128-row tile on every shape measured
        // (tools/kbench_gemm64.py, kbench_gemm_k.py: 10 300 x 2048 x 512 241 -> 199 us, x 512 x 512 85 -> 65, 32 768 x 512 x 384
        // 137 -> 126, x 1536 equal), and no product of this kind needs split-K or the split last round any more.
        const long ct = (N + 127) / 128, t64 = ((M + 63) / 64) * ct;
        if ((t64 >= 3 * MIL_NUM_CU / 2 || (rows_dev != nullptr && t64 >= MIL_NUM_CU / 2 && (N & 3) == 0)) && K >= 256) {
            const dim3 grid((unsigned)ct, (M + 63) / 64);
            if (b_mode == 0)
                hipLaunchKernelGGL(k_gemm64<0>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, accumulate, aux, ldaux, aux_mode, rows_dev);
            else
                hipLaunchKernelGGL(k_gemm64<1>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, accumulate, aux, ldaux, aux_mode, rows_dev);
            MIL_CHECK_LAUNCH();
            if (rows_honoured != nullptr) *rows_honoured = 1;
            return MIL_OK;
        }
    }
#endif
#if !defined(LG_NO_TILE64N)
    {
        // a few hundred rows: 64 x 64 tiles, K split over blockIdx.z when the tiles alone leave CUs idle (gemm64.h)
        int S64, kc64;
        if (small_tile_plan(M, N, K, a_mode, &S64, &kc64) && (S64 == 1 || (workspace != nullptr && workspace_floats >= (size_t)S64 * M * N))) {
            const dim3 grid((N + 63) / 64, (M + 63) / 64, S64);
            float* part = S64 > 1 ? workspace : nullptr;
            if (b_mode == 0)
                hipLaunchKernelGGL(k_gemm64n<0>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, kc64, part, bias, act, residual, ldr, accumulate, aux, ldaux, aux_mode);
            else
                hipLaunchKernelGGL(k_gemm64n<1>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, kc64, part, bias, act, residual, ldr, accumulate, aux, ldaux, aux_mode);
            MIL_CHECK_LAUNCH();
            if (S64 > 1) {
                const size_t n = (size_t)M * N;
                hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, S64, C, ldc, M, N, bias,
                                   act, residual, ldr, accumulate, aux, ldaux, aux_mode);
                MIL_CHECK_LAUNCH();
            }
            return MIL_OK;
        }
    }
#endif
    {
        int rows_main, S_tail, kc_tail;
        if (workspace != nullptr && tail_plan(M, N, K, a_mode, &rows_main, &S_tail, &kc_tail) &&
            workspace_floats >= (size_t)S_tail * (M - rows_main) * N) {
            // whole rounds without split-K, then the rows of the last partial round with split-K
            int rc = gemm_impl(A, lda, a_mode, B, ldb, b_mode, C, ldc, rows_main, N, K, bias, act, residual, ldr, accumulate,
                               nullptr, 0, aux, ldaux, aux_mode, stream);
            if (rc != MIL_OK) return rc;
            const int Mt = M - rows_main;
            const float* At = A + (size_t)rows_main * lda;
            float* Ct = C + (size_t)rows_main * ldc;
            const float* Rt = residual ? residual + (size_t)rows_main * ldr : nullptr;
            float* Xt = aux ? aux + (size_t)rows_main * ldaux : nullptr;
            const dim3 gridt((N + 127) / 128, (Mt + 127) / 128, S_tail);
            if (b_mode == 0)
                hipLaunchKernelGGL((k_gemm<0, 0>), gridt, dim3(256), 0, st, At, lda, B, ldb, Ct, ldc, Mt, N, K, kc_tail, bias, act, Rt, ldr, accumulate, workspace, Xt, ldaux, aux_mode, GemmGroups{nullptr, 0, 0, 0, GRP_NONE, 1});
            else
                hipLaunchKernelGGL((k_gemm<0, 1>), gridt, dim3(256), 0, st, At, lda, B, ldb, Ct, ldc, Mt, N, K, kc_tail, bias, act, Rt, ldr, accumulate, workspace, Xt, ldaux, aux_mode, GemmGroups{nullptr, 0, 0, 0, GRP_NONE, 1});
            MIL_CHECK_LAUNCH();
            const size_t n = (size_t)Mt * N;
            hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, S_tail, Ct, ldc,
                               Mt, N, bias, act, Rt, ldr, accumulate, Xt, ldaux, aux_mode);
            MIL_CHECK_LAUNCH();
            return MIL_OK;
        }
    }
    int S = 1, kchunk = K;
    float* partial = nullptr;
    if (workspace != nullptr) {
        int kc;
        const int want = splitk_plan(M, N, K, a_mode, &kc);
        if (want > 1 && workspace_floats >= (size_t)want * M * N) { S = want; kchunk = kc; partial = workspace; }
    }
    dim3 grid((N + 127) / 128, (M + 127) / 128, S);
    if (a_mode == 0 && b_mode == 0)
        hipLaunchKernelGGL((k_gemm<0, 0>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, kchunk, bias, act, residual, ldr, accumulate, partial, aux, ldaux, aux_mode, GemmGroups{nullptr, 0, 0, 0, GRP_NONE, 1});
    else if (a_mode == 0 && b_mode == 1)
        hipLaunchKernelGGL((k_gemm<0, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, kchunk, bias, act, residual, ldr, accumulate, partial, aux, ldaux, aux_mode, GemmGroups{nullptr, 0, 0, 0, GRP_NONE, 1});
    else
        hipLaunchKernelGGL((k_gemm<1, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, kchunk, bias, act, residual, ldr, accumulate, partial, aux, ldaux, aux_mode, GemmGroups{nullptr, 0, 0, 0, GRP_NONE, 1});
    MIL_CHECK_LAUNCH();
    if (partial != nullptr) {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, S, C, ldc, M, N,
                           bias, act, residual, ldr, accumulate, aux, ldaux, aux_mode);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

extern "C" int mil_gemm(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
                        int M, int N, int K, const float* bias, int act, const float* residual, int ldr, int accumulate,
                        float* workspace, size_t workspace_floats, void* stream) {
    return gemm_impl(A, lda, a_mode, B, ldb, b_mode, C, ldc, M, N, K, bias, act, residual, ldr, accumulate, workspace,
                     workspace_floats, nullptr, 0, AUX_NONE, stream);
}

// C rows [64 * ceil(rows_dev / 64), M) <- 0: what k_gemm64 writes for the tiles wholly behind the true row count, for the
// dispatches that do not read rows_dev themselves (small-tile plan, split-K, the 128-row kernel: ADVICE r3 - their padding
// rows held act(stale x W + b), harmless today, garbage for any future reader of those rows).
__global__ __launch_bounds__(256) void k_zero_rows_from(float* __restrict__ C, int ldc, int M, int N, const int32_t* __restrict__ rows_dev) {
    const int first = (__builtin_amdgcn_readfirstlane(rows_dev[0]) + 63) & ~63;
    const int row0 = blockIdx.x * 64;
    if (row0 < first) return;
    const int n4 = N >> 2;
    for (int idx = threadIdx.x; idx < 64 * n4; idx += 256) {
        const int row = row0 + idx / n4, j = 4 * (idx % n4);
        if (row < M) *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + j) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (N & 3)
        for (int idx = threadIdx.x; idx < 64 * (N & 3); idx += 256) {
            const int row = row0 + idx / (N & 3), j = (N & ~3) + idx % (N & 3);
            if (row < M) C[(size_t)row * ldc + j] = 0.f;
        }
}

// rows_dev (nullable, a_mode 0): device int32 with the TRUE number of rows of A / C (<= M; M then is the capacity the launch
// is sized for - a bucket of fusion_step.RaggedFusionStepper).  Rows from 64 * ceil(rows_dev / 64) on are ZERO on return
// (accumulate == 0) whichever kernel the shape dispatches to; rows between rows_dev and that boundary are unspecified.
extern "C" int mil_gemm_rows(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
                             int M, int N, int K, const float* bias, int act, const float* residual, int ldr, int accumulate,
                             float* workspace, size_t workspace_floats, const int32_t* rows_dev, void* stream) {
    if (rows_dev != nullptr && a_mode != 0) return MIL_EINVAL;
    int honoured = 0;
    const int rc = gemm_impl(A, lda, a_mode, B, ldb, b_mode, C, ldc, M, N, K, bias, act, residual, ldr, accumulate, workspace,
                             workspace_floats, nullptr, 0, AUX_NONE, stream, rows_dev, &honoured);
    if (rc != MIL_OK || rows_dev == nullptr || honoured || accumulate || M <= 0 || N <= 0) return rc;
    if ((ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 15)) return MIL_EINVAL;
    hipLaunchKernelGGL(k_zero_rows_from, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, C, ldc, M, N, rows_dev);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gemm_aux(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
                            int M, int N, int K, const float* bias, int act, const float* residual, int ldr,
                            int accumulate, float* workspace, size_t workspace_floats, float* aux, int ldaux, int aux_mode,
                            void* stream) {
    if (aux_mode < 0 || aux_mode > 2 || (aux_mode != AUX_NONE && (!aux || ldaux < N)) || (aux_mode != AUX_NONE && a_mode != 0))
        return MIL_EINVAL;
    return gemm_impl(A, lda, a_mode, B, ldb, b_mode, C, ldc, M, N, K, bias, act, residual, ldr, accumulate, workspace,
                     workspace_floats, aux, ldaux, aux_mode, stream);
}

// dW[N_out, K_in] (+)= (dY (.) act'(Y))^T . X,   db[N_out] (+)= column sums of dY (.) act'(Y):  the parameter half of a
// Linear layer's backward in one product launch + its split-K fold.
extern "C" size_t mil_linear_bwd_params_workspace_floats(int rows, int n_out, int k_in) {
    if (rows <= 0 || n_out <= 0 || k_in <= 0) return 0;
    int kc;
    int S = splitk_plan(n_out, k_in, rows, 1, &kc);
    if (S < 1) S = 1;
    const int S2 = mil_gemm_tn2_splits(rows, n_out, k_in);         // the low-VALU kernel's own row split (linear_nt2.hip)
    if (S2 > S) S = S2;
    return (size_t)S * n_out * k_in + (size_t)S * n_out;
}

int mil_gemm_tn2_rows(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx, int rows, int N, int K,
                      float* partial, float* cs_partial, const int32_t* rows_dev, void* stream);      // linear_nt2.hip

// rows_dev (nullable): device int32 with the true number of rows (<= rows: the capacity of a bucket) - the rows behind it
// carry zero gradients, and the tall-activation kernel then spreads only the true rows over its workgroups.
extern "C" int mil_linear_bwd_params_rows(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx,
                                          int rows, int n_out, int k_in, float* dW, int lddw, float* db, int accumulate,
                                          float* workspace, size_t workspace_floats, const int32_t* rows_dev, void* stream) {
    if (!dY || !X || !dW || !workspace || rows <= 0 || n_out < 4 || (n_out & 3) || k_in < 4 || (k_in & 3)) return MIL_EINVAL;
    if ((lddy & 3) || (ldx & 3) || (Y != nullptr && (ldy & 3))) return MIL_EINVAL;
    if (act != ACT_NONE && act != ACT_TANH && act != ACT_RELU) return MIL_EINVAL;
    if (act != ACT_NONE && Y == nullptr) return MIL_EINVAL;
    if (workspace_floats < mil_linear_bwd_params_workspace_floats(rows, n_out, k_in)) return MIL_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
#if !defined(LG_NO_TN2)
    if (mil_gemm_tn2_ok(lddy, Y ? ldy : lddy, ldx, rows, n_out, k_in) &&
        ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0) {
        // tall activation, whole 128 x 128 output tiles: the low-VALU split-rows kernel (linear_nt2.hip), same partial layout
        const int S2 = mil_gemm_tn2_splits(rows, n_out, k_in);
        float* cs2 = db != nullptr ? workspace + (size_t)S2 * n_out * k_in : nullptr;
        int rc = mil_gemm_tn2_rows(dY, lddy, act != ACT_NONE ? Y : nullptr, ldy, act, X, ldx, rows, n_out, k_in, workspace, cs2,
                                   rows_dev, stream);
        if (rc != MIL_OK) return rc;
        const size_t n2 = (size_t)n_out * k_in;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, workspace, S2, dW, lddw, n_out,
                           k_in, (const float*)nullptr, 0, (const float*)nullptr, 0, accumulate, (float*)nullptr, 0, 0,
                           (const float*)cs2, db, accumulate);
        MIL_CHECK_LAUNCH();
        return MIL_OK;
    }
#endif
    int kchunk;
    int S = splitk_plan(n_out, k_in, rows, 1, &kchunk);
    if (S < 1) { S = 1; kchunk = rows; }
    float* partial = workspace;
    float* cs_part = db != nullptr ? workspace + (size_t)S * n_out * k_in : nullptr;
    const GemmExtra ex{act != ACT_NONE ? Y : nullptr, cs_part, ldy, act};
    const dim3 grid((k_in + 127) / 128, (n_out + 127) / 128, S);
    hipLaunchKernelGGL((k_gemm<1, 1, true>), grid, dim3(256), 0, st, dY, lddy, X, ldx, dW, lddw, n_out, k_in, rows, kchunk,
                       (const float*)nullptr, 0, (const float*)nullptr, 0, accumulate, partial, (float*)nullptr, 0, 0,
                       GemmGroups{nullptr, 0, 0, 0, GRP_NONE, 1}, ex);
    MIL_CHECK_LAUNCH();
    const size_t n = (size_t)n_out * k_in;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, S, dW, lddw, n_out, k_in,
                       (const float*)nullptr, 0, (const float*)nullptr, 0, accumulate, (float*)nullptr, 0, 0,
                       (const float*)cs_part, db, accumulate);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_linear_bwd_params(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx,
                                     int rows, int n_out, int k_in, float* dW, int lddw, float* db, int accumulate,
                                     float* workspace, size_t workspace_floats, void* stream) {
    return mil_linear_bwd_params_rows(dY, lddy, Y, ldy, act, X, ldx, rows, n_out, k_in, dW, lddw, db, accumulate, workspace,
                                      workspace_floats, nullptr, stream);
}

#include "skinny_gemm.h"

// C_g = sum over the row chunks of a split grouped contraction (fixed order)
__global__ __launch_bounds__(256) void k_grouped_fold(const float* __restrict__ part, int splits, size_t per, float* __restrict__ C,
                                                      size_t total) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= total) return;
    const size_t g = i4 / per, o = i4 % per;
    const float* src = part + g * splits * per + o;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int sidx = 0;
    for (; sidx + 8 <= splits; sidx += 8) {              // eight partial tiles in flight per thread (one at a time: 0.7 TB/s)
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(sidx + u) * per);
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; sidx < splits; ++sidx) v += *reinterpret_cast<const f32x4*>(src + (size_t)sidx * per);
    *reinterpret_cast<f32x4*>(C + i4) = v;
}

extern "C" size_t mil_gemm_grouped_workspace_floats(int a_mode, int G, int max_group_rows, int M, int N) {
    if (a_mode != 1 || M > 128 || G <= 0) return 0;
    {
        int S64;
        if (tn64_plan(G, max_group_rows, M, N, &S64)) return S64 > 1 ? (size_t)G * S64 * M * N : 0;
    }
    const int tiles = G * ((N + 127) / 128);
    int S = (2 * MIL_NUM_CU) / (tiles > 0 ? tiles : 1);
    if (S > max_group_rows / 128) S = max_group_rows / 128;         // at least four 32-row slices per chunk
#ifndef LG_GRP_SMAX
#define LG_GRP_SMAX 32      // one bag of 4096 patches: 4 column tiles x 32 row chunks = 128 workgroups (8 chunks: 55 -> 19 us)
#endif
    if (S > LG_GRP_SMAX) S = LG_GRP_SMAX;
    return S >= 2 ? (size_t)G * S * M * N : 0;
}

// pad_rows > 0 (a_mode 0 only): C has pad_rows rows and those outside every group - the padding of a capacity bucket, whose
// bag lengths live on the device - must read ZERO.  With one group the few-columns kernels write those zeros themselves;
// any other case clears C first.
extern "C" int mil_gemm_grouped_pad(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
                                    const int32_t* grp_off, int G, int max_group_rows, int M, int N, int K, long strideB,
                                    long strideC, const float* bias, long strideBias, const float* residual, int ldr,
                                    float* workspace, size_t workspace_floats, int pad_rows, void* stream) {
    if (!A || !B || !C || !grp_off || G < 0 || max_group_rows < 0 || N <= 0 || pad_rows < 0) return MIL_EINVAL;
    if (pad_rows > 0 && a_mode != 0) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool skinny_nt = a_mode == 0 && b_mode == 0 && residual == nullptr && N <= 96 && (N % 32) == 0 && K > 0 && (K % LG_BK) == 0;
    const bool skinny_nn = a_mode == 0 && b_mode == 1 && K > 0 && K <= 96 && (K % LG_BK) == 0 && (N % 128) == 0 && (bias == nullptr || strideBias == 0);
    int pad_end = 0;
    if (pad_rows > 0) {
        if (G == 1 && max_group_rows >= pad_rows && (skinny_nt || skinny_nn)) {
            pad_end = pad_rows;
        } else {
            hipError_t e = hipMemsetAsync(C, 0, (size_t)pad_rows * ldc * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (G == 0 || max_group_rows == 0) return MIL_OK;
    if ((lda & 3) || (ldb & 3)) return MIL_EINVAL;
    GemmGroups gg{grp_off, strideB, strideC, strideBias, a_mode == 0 ? GRP_ROWS : GRP_CONTRACT, 1};
    if (a_mode == 0) {
        // C[rows_g, :N] = A[rows_g, :K] . B_g   (b_mode 0: B_g [N, K];  b_mode 1: B_g [K, N]);  K % 32 == 0
        if (K <= 0 || (K % LG_BK) != 0 || (b_mode == 1 && (N & 3))) return MIL_EINVAL;
        const dim3 grid((N + 127) / 128, (max_group_rows + 127) / 128, G);
        if (b_mode == 0 && residual == nullptr && N <= 96 && (N % 32) == 0) {
            // N = T x H absorbed vectors: 64-row workgroups, one MFMA tile per wave (skinny_gemm.h)
            const bool narrow = (long)G * ((max_group_rows + 63) / 64) < MIL_NUM_CU;      // few bags: 32-row workgroups
            const dim3 gs(narrow ? (max_group_rows + 31) / 32 : (max_group_rows + 63) / 64, G);
#define SKINNY_NT(NCTV)                                                                                                  \
    if (narrow) hipLaunchKernelGGL((k_skinny_nt<NCTV, 1>), gs, dim3(64 * NCTV), 0, st, A, lda, B, ldb, strideB, C, ldc, grp_off, K, bias, strideBias, pad_end); \
    else hipLaunchKernelGGL((k_skinny_nt<NCTV, 2>), gs, dim3(128 * NCTV), 0, st, A, lda, B, ldb, strideB, C, ldc, grp_off, K, bias, strideBias, pad_end);
            if (N == 32) { SKINNY_NT(1) } else if (N == 64) { SKINNY_NT(2) } else { SKINNY_NT(3) }
#undef SKINNY_NT
        } else if (b_mode == 1 && K <= 96 && (N % 128) == 0 && (bias == nullptr || strideBias == 0)) {
            // K = T x H: the whole contraction staged at once
            const dim3 gs(N / 128, (max_group_rows + 63) / 64, G);
            if (K == 32) hipLaunchKernelGGL(k_skinny_nn<32>, gs, dim3(256), 0, st, A, lda, B, ldb, strideB, C, ldc, grp_off, N, bias, residual, ldr, pad_end);
            else if (K == 64) hipLaunchKernelGGL(k_skinny_nn<64>, gs, dim3(256), 0, st, A, lda, B, ldb, strideB, C, ldc, grp_off, N, bias, residual, ldr, pad_end);
            else hipLaunchKernelGGL(k_skinny_nn<96>, gs, dim3(256), 0, st, A, lda, B, ldb, strideB, C, ldc, grp_off, N, bias, residual, ldr, pad_end);
        } else if (b_mode == 0)
            hipLaunchKernelGGL((k_gemm<0, 0>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, 0, N, K, K, bias, 0, residual, ldr, 0, (float*)nullptr, (float*)nullptr, 0, 0, gg);
        else
            hipLaunchKernelGGL((k_gemm<0, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, 0, N, K, K, bias, 0, residual, ldr, 0, (float*)nullptr, (float*)nullptr, 0, 0, gg);
    } else {
        // C_g[M, N] = A[rows_g, :M]^T . B[rows_g, :N]   (contraction over the group's rows)
        if (b_mode != 1 || M < 4 || (M & 3) || (N & 3) || bias || residual) return MIL_EINVAL;
        const size_t want = mil_gemm_grouped_workspace_floats(a_mode, G, max_group_rows, M, N);
        int S64;
        if (tn64_plan(G, max_group_rows, M, N, &S64) && strideC == (long)M * N && ldc == N && lda >= M &&
            (S64 == 1 || (workspace != nullptr && workspace_floats >= (size_t)G * S64 * M * N))) {
            // 64 x 64 tiles, ~3 workgroups per CU (gemm64.h: k_gemm64tn), then the fold of the splits
            hipLaunchKernelGGL(k_gemm64tn, dim3((N + 63) / 64, (M + 63) / 64, G * S64), dim3(256), 0, st, A, lda, B, ldb,
                               S64 > 1 ? workspace : C, M, N, grp_off, S64);
            if (S64 > 1) {
                MIL_CHECK_LAUNCH();
                const size_t total = (size_t)G * M * N;
                hipLaunchKernelGGL(k_grouped_fold, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, workspace, S64,
                                   (size_t)M * N, C, total);
            }
        } else if (workspace != nullptr && want > 0 && workspace_floats >= want && strideC == (long)M * N && ldc == N) {
            // few output tiles per group: split each group's rows over S workgroups, fold the partial tiles afterwards
            gg.splits = (int)(want / ((size_t)G * M * N));
            const dim3 grid((N + 127) / 128, gg.splits, G);
            hipLaunchKernelGGL((k_gemm<1, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, workspace, ldc, M, N, 0, 0, (const float*)nullptr, 0, (const float*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr, 0, 0, gg);
            MIL_CHECK_LAUNCH();
            const size_t total = (size_t)G * M * N;
            hipLaunchKernelGGL(k_grouped_fold, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, workspace, gg.splits,
                               (size_t)M * N, C, total);
        } else {
            const dim3 grid((N + 127) / 128, (M + 127) / 128, G);
            hipLaunchKernelGGL((k_gemm<1, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, 0, 0, (const float*)nullptr, 0, (const float*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr, 0, 0, gg);
        }
    }
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gemm_grouped(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
                                const int32_t* grp_off, int G, int max_group_rows, int M, int N, int K, long strideB,
                                long strideC, const float* bias, long strideBias, const float* residual, int ldr,
                                float* workspace, size_t workspace_floats, void* stream) {
    return mil_gemm_grouped_pad(A, lda, a_mode, B, ldb, b_mode, C, ldc, grp_off, G, max_group_rows, M, N, K, strideB, strideC, bias,
                                strideBias, residual, ldr, workspace, workspace_floats, 0, stream);
}

extern "C" size_t mil_colsum_workspace_floats(int M, int N) {
    const int nch = (M + 255) / 256;
    return nch > 1 ? (size_t)nch * N : 0;
}

extern "C" int mil_colsum(const float* Y, int ldy, int M, int N, float* out, int accumulate, float* workspace,
                          void* stream) {
    if (!Y || !out || M < 0 || N <= 0) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nch = (M + 255) / 256;
    if (nch > 1 && workspace != nullptr) {
        hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64, nch), dim3(256), 0, st, Y, ldy, M, N, 256, workspace, N, 0);
        MIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64, 1), dim3(256), 0, st, workspace, N, nch, N, nch, out, 0, accumulate);
    } else {
        hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64, 1), dim3(256), 0, st, Y, ldy, M, N, M > 0 ? M : 1, out, 0, accumulate);
    }
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act, void* stream) {
    if (!dy || !y || !dpre || act < 0 || act > 2) return MIL_EINVAL;
    if (n == 0) return MIL_OK;
    hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dpre, n, act);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
