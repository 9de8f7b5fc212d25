// Skinny grouped products of the multi-token absorbed attention (ops.multi_token_pool_attention / _rows_attention;
// model/sam/transformer.py:291-295,303-307 with T text tokens per bag).  One side of each product is the T x H <= 96
// absorbed vectors of a bag, so the 128 x 128 x 32 tile of k_gemm is the wrong shape: with N = 96 a quarter of its MFMAs
// multiply padding and one bag of 4096 patches is 32 workgroups; with K = 96 its pipeline is three slices long and the
// prologue / epilogue dominate (measured at 32 bags x 1024 patches: 61 and 75 us for 3.2 GFLOP = 20 us of MFMA time).
//
//   k_skinny_nt<NCT, RWT>:  C[rows_g, :32 NCT] = A[rows_g, :K] . B_g[32 NCT, K]^T + bias_g        (scores: K = 512)
//       workgroup = 64 rows x 32 NCT columns, one 32 x 32 MFMA tile per wave (2 x NCT waves), K walked in 32-deep
//       slices through a double-buffered LDS image, registers carry the next slice.
//   k_skinny_nn<KP>:   C[rows_g, :N] = A[rows_g, :KP] . B_g[KP, N] + bias + residual          (values: KP = 32 / 64 / 96)
//       workgroup = 64 rows x 128 columns, the whole contraction staged at once (no slice loop), 2 x 2 waves with
//       a 32 x 64 tile each.
// Same fragment conventions as k_gemm (32x32x2 fp32 MFMA, lane (r, h) takes k = 8t + 4h + jj).  Included by linear.hip.
#pragma once

// RWT = row blocks of 32 per workgroup: 2 (64 rows, 2 x NCT waves) in general, 1 (32 rows, NCT waves) when the launch
// would otherwise be under ~one workgroup per CU (a single bag of 4096 patches: 64 -> 128 workgroups)
template <int NCT, int RWT>
__global__ __launch_bounds__(64 * RWT * NCT) void k_skinny_nt(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                               int ldb, long strideB, float* __restrict__ C, int ldc,
                                                               const int32_t* __restrict__ grp_off, int K,
                                                               const float* __restrict__ bias, long strideBias, int pad_end) {
    constexpr int T = 64 * RWT * NCT, P = 32 * NCT, RW = 32 * RWT;
    constexpr int NA = (RW * 8 + T - 1) / T;       // float4 pieces of the RW x 32 A slice per thread
    constexpr int NB = (P * 8) / T;                // P x 8 pieces of the B slice: 4 / RWT per thread exactly
    constexpr int ASZ = RW * LG_KS, BSZ = P * LG_KS;
    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
    float* as = smem;
    float* bs = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / NCT, wc = wave % NCT;
    const int r = lane & 31, h = lane >> 5;
    const int g = blockIdx.y, goff = grp_off[g], M = grp_off[g + 1] - goff;
    const int i0 = blockIdx.x * RW;
    // pad_end > 0 (ONE group in a capacity bucket, segments.FusionBucket): the rows of C behind the group, up to pad_end, are
    // written as zeros here - they used to be a torch.zeros fill of the whole output in front of every product
    const int zend = pad_end > 0 && g == (int)gridDim.y - 1 ? pad_end - goff : 0;      // group-relative end of the zero rows
    if (i0 >= M) {
        for (int idx = threadIdx.x; idx < RW * P; idx += T) {
            const int row = i0 + idx / P;
            if (row < zend) C[(size_t)(goff + row) * ldc + idx % P] = 0.f;
        }
        return;
    }
    A += (size_t)goff * lda;
    C += (size_t)goff * ldc;
    B += (size_t)g * strideB;

    const float* asrc[NA];
    int aoff[NA];
    bool alive[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + T * i;
        alive[i] = idx < RW * 8;
        const int row = (idx >> 3) % RW, ch = idx & 7;
        asrc[i] = A + (size_t)min(i0 + row, M - 1) * lda + 4 * ch;
        aoff[i] = row * LG_KS + 4 * ch;
    }
    const float* bsrc[NB];
    int boff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + T * i, col = idx >> 3, ch = idx & 7;
        bsrc[i] = B + (size_t)col * ldb + 4 * ch;
        boff[i] = col * LG_KS + 4 * ch;
    }
    // two register sets in rotation: the loads of slice s + 3 are issued while slice s is multiplied and are stored two
    // slices later (one set - a lead of one slice = 16 MFMAs - left every slice waiting for its HBM round trip: 2.6 us per
    // slice at two workgroups per CU, 42 us for 64 MB)
    f32x4 ra[2][NA], rb[2][NB];
    auto load = [&](int set, int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) if (alive[i]) ra[set][i] = *reinterpret_cast<const f32x4*>(asrc[i] + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[set][i] = *reinterpret_cast<const f32x4*>(bsrc[i] + k0);
    };
    auto store = [&](int set, int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) if (alive[i]) *reinterpret_cast<f32x4*>(as + buf * ASZ + aoff[i]) = ra[set][i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(bs + buf * BSZ + boff[i]) = rb[set][i];
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nslice = K / LG_BK;
    load(0, 0);
    store(0, 0);
    load(1, min(1, nslice - 1) * LG_BK);
    load(0, min(2, nslice - 1) * LG_BK);
    __syncthreads();
    auto slice = [&](int s, auto set_c) {
        constexpr int set = decltype(set_c)::value;             // = (s + 1) & 1: holds slice s + 1, then takes slice s + 3
        const int buf = s & 1;
        const float* ap = as + buf * ASZ + (32 * wr + r) * LG_KS + 4 * h;
        const float* bp = bs + buf * BSZ + (32 * wc + r) * LG_KS + 4 * h;
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[t] = *reinterpret_cast<const f32x4*>(ap + 8 * t);
            fb[t] = *reinterpret_cast<const f32x4*>(bp + 8 * t);
        }
        if (s + 1 < nslice) store(set, buf ^ 1);
        load(set, min(s + 3, nslice - 1) * LG_BK);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t][jj], fb[t][jj], acc, 0, 0, 0);
        __syncthreads();
    };
    for (int s = 0; s < nslice; s += 2) {
        slice(s, std::integral_constant<int, 1>{});
        if (s + 1 < nslice) slice(s + 1, std::integral_constant<int, 0>{});
    }
    const int col = 32 * wc + r;
    const float bj = bias != nullptr ? bias[(size_t)g * strideBias + col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = i0 + 32 * wr + mfma32_row(i, h);
        if (row < M) C[(size_t)row * ldc + col] = acc[i] + bj;
        else if (row < zend) C[(size_t)row * ldc + col] = 0.f;
    }
}

template <int KP>
__global__ __launch_bounds__(256) void k_skinny_nn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                   long strideB, float* __restrict__ C, int ldc,
                                                   const int32_t* __restrict__ grp_off, int N,
                                                   const float* __restrict__ bias, const float* __restrict__ residual,
                                                   int ldr, int pad_end) {
    constexpr int AS = KP + 4;                      // A image row stride (words): 100 / 68 / 36
    __shared__ __attribute__((aligned(16))) float smem[64 * AS + KP * 128];
    float* as = smem;                               // [64][AS]   k-contiguous
    float* bs = smem + 64 * AS;                     // [KP][128]  k-major
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int g = blockIdx.z, goff = grp_off[g], M = grp_off[g + 1] - goff;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 128;
    const int zend = pad_end > 0 && g == (int)gridDim.z - 1 ? pad_end - goff : 0;      // as k_skinny_nt
    if (i0 >= M) {
        for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {                       // 64 rows x 128 columns in float4
            const int row = i0 + (idx >> 5), j = j0 + 4 * (idx & 31);
            if (row < zend && j < N) *reinterpret_cast<f32x4*>(C + (size_t)(goff + row) * ldc + j) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    A += (size_t)goff * lda;
    C += (size_t)goff * ldc;
    if (residual != nullptr) residual += (size_t)goff * ldr;
    B += (size_t)g * strideB;

    // stage everything: A 64 x KP (KP / 4 float4 per row), B KP x 128 (32 float4 per k row); all loads first
    constexpr int NAP = 64 * (KP / 4) / 256, NBP = KP * 32 / 256;
    f32x4 ra[NAP], rb[NBP];
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
        const int idx = tid + 256 * i, row = idx / (KP / 4), ch = idx % (KP / 4);
        ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)min(i0 + row, M - 1) * lda + 4 * ch);
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int idx = tid + 256 * i, kr = idx >> 5, c4 = idx & 31;
        rb[i] = *reinterpret_cast<const f32x4*>(B + (size_t)kr * ldb + j0 + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
        const int idx = tid + 256 * i, row = idx / (KP / 4), ch = idx % (KP / 4);
        *reinterpret_cast<f32x4*>(as + row * AS + 4 * ch) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int idx = tid + 256 * i, kr = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<f32x4*>(bs + kr * 128 + 4 * c4) = rb[i];
    }
    // the residual values of this lane's 2 x 16 outputs are requested before the products (they depend on nothing the
    // kernel computes): their trip runs under the MFMA phase instead of behind it (PMC: waves 47 % in memory waits)
    float rv[2][16];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = min(j0 + 64 * wc + 32 * b + r, N - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rc = min(i0 + 32 * wr + mfma32_row(i, h), M - 1);
            rv[b][i] = residual != nullptr ? residual[(size_t)rc * ldr + j] : 0.f;
        }
    }
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    const float* ap = as + (32 * wr + r) * AS + 4 * h;
    const float* bp = bs + 4 * h * 128 + 64 * wc + r;
#pragma unroll
    for (int t = 0; t < KP / 8; ++t) {
        const f32x4 fa = *reinterpret_cast<const f32x4*>(ap + 8 * t);
        float fb[2][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            fb[0][jj] = bp[(8 * t + jj) * 128];
            fb[1][jj] = bp[(8 * t + jj) * 128 + 32];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[jj], fb[0][jj], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[jj], fb[1][jj], acc[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + 64 * wc + 32 * b + r;
        if (j >= N) continue;
        const float bj = bias != nullptr ? bias[j] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i0 + 32 * wr + mfma32_row(i, h);
            if (row < M) C[(size_t)row * ldc + j] = acc[b][i] + bj + rv[b][i];
            else if (row < zend) C[(size_t)row * ldc + j] = 0.f;
        }
    }
}
