// 64-row tile of the fp32 GEMM - the default for tall products (a_mode 0) from half a round of the chip upwards:
//   C[M,N] (+)= epilogue( A[M,K] . B_op[K,N] )      b_mode 0: B [N, K] (y = x W^T),  b_mode 1: B [K, N] (dx = dy W)
// 64 x 128 tiles with an UNPADDED LDS image (48 KB for both stages instead of the 72 KB of k_gemm) fit THREE workgroups
// per CU: 768 slots of half-size tiles waste less of a launch's last round than 512 slots of 128 x 128 tiles (the text
// tower under learnable prompts multiplies ~10 k rows: 324 tiles for N = 512, 1296 for N = 2048), the third resident
// workgroup hides more of each other's prologue / epilogue, and nothing needs split-K with its fold launch.
// The k-contiguous images are [rows][32] with the 16-byte
// chunk index XOR-swizzled by (row >> 1) & 7 - the 16 lanes served together by a ds_read_b128 then cover all 64 banks -
// instead of k_gemm's 36-word row stride.  Workgroup 256 threads = 2 x 2 waves, wave tile 32 x 64 (two 32 x 32 MFMA
// tiles); pipeline, fragment convention (lane (r, h) takes k = 8t + 4h + jj) and epilogue as in k_gemm.  K % 32 == 0.
// Included by linear.hip.
#pragma once

__device__ __forceinline__ int g64_swz(int row, int chunk) { return row * 32 + 4 * (chunk ^ ((row >> 1) & 7)); }

template <int BMODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_gemm64(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
              int M, int N, int K, const float* __restrict__ bias, int act, const float* __restrict__ residual, int ldr,
              int accumulate, float* __restrict__ aux, int ldaux, int aux_mode, const int32_t* __restrict__ rows_dev) {
    constexpr int ASZ = 64 * 32;
    constexpr int BSZ = BMODE == 0 ? 128 * 32 : LG_BK * 128;
    // rows_dev (a capacity bucket whose true row count lives on the device; rows from it on are padding): tiles that lie
    // wholly in the padding write zeros and leave - the launch is sized for the capacity, its time follows the true count
    if (rows_dev != nullptr && (int)(blockIdx.y * 64) >= __builtin_amdgcn_readfirstlane(rows_dev[0])) {
        if (!accumulate)
            for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
                const int row = blockIdx.y * 64 + (idx >> 5), j = blockIdx.x * 128 + 4 * (idx & 31);
                if (row < M && j < N) *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + j) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        return;
    }
    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];       // 48 KB
    float* as = smem;
    float* bs = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 128;
    const int nslice = K / LG_BK;

    // staging maps.  A: rows (tid >> 3) + 32 i (i < 2), chunk tid & 7.  B (NT): rows (tid >> 3) + 32 i (i < 4), chunk tid & 7.
    // B (NN): k row (tid >> 5) + 8 i (i < 4), 16-byte column chunk tid & 31.
    const int srow = tid >> 3, sch = tid & 7;
    const float* asrc[2];
    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = srow + 32 * i;
        asrc[i] = A + (size_t)min(i0 + row, M - 1) * lda + 4 * sch;
        aoff[i] = g64_swz(row, sch);
    }
    const float* bsrc[4];
    int boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (BMODE == 0) {
            const int row = srow + 32 * i;
            bsrc[i] = B + (size_t)min(j0 + row, N - 1) * ldb + 4 * sch;
            boff[i] = g64_swz(row, sch);
        } else {
            const int kr = (tid >> 5) + 8 * i, c4 = tid & 31;
            bsrc[i] = B + (size_t)kr * ldb + min(j0 + 4 * c4, max(N - 4, 0));
            boff[i] = kr * 128 + 4 * c4;
        }
    }
    f32x4 ra[2], rb[4];
    auto a_load = [&](int i, int k0) { ra[i] = *reinterpret_cast<const f32x4*>(asrc[i] + k0); };
    auto b_load = [&](int i, int k0) {
        rb[i] = *reinterpret_cast<const f32x4*>(BMODE == 0 ? bsrc[i] + k0 : bsrc[i] + (size_t)k0 * ldb);
    };
    auto a_store = [&](int i, float* dst) { *reinterpret_cast<f32x4*>(dst + aoff[i]) = ra[i]; };
    auto b_store = [&](int i, float* dst) { *reinterpret_cast<f32x4*>(dst + boff[i]) = rb[i]; };

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

#pragma unroll
    for (int i = 0; i < 2; ++i) a_load(i, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) b_load(i, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) a_store(i, as);
#pragma unroll
    for (int i = 0; i < 4; ++i) b_store(i, bs);
    {
        const int k1 = min(1, nslice - 1) * LG_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) a_load(i, k1);
#pragma unroll
        for (int i = 0; i < 4; ++i) b_load(i, k1);
    }
    __syncthreads();
    const int arow = 32 * wi + r;
    for (int s = 0; s < nslice; ++s) {
        const int buf = s & 1;
        const int k2 = min(s + 2, nslice - 1) * LG_BK;
        const float* ab = as + buf * ASZ;
        const float* bb = bs + buf * BSZ;
        float* an = as + (buf ^ 1) * ASZ;
        float* bn = bs + (buf ^ 1) * BSZ;
        f32x4 fa[2], fb[2][2];                      // [register set] / [register set][tile]
        auto frag_a = [&](int t, int q) { fa[q] = *reinterpret_cast<const f32x4*>(ab + g64_swz(arow, 2 * t + h)); };
        auto frag_b = [&](int t, int q, int b) {
            if (BMODE == 0) {
                fb[q][b] = *reinterpret_cast<const f32x4*>(bb + g64_swz(64 * wj + 32 * b + r, 2 * t + h));
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) fb[q][b][jj] = bb[(8 * t + 4 * h + jj) * 128 + 64 * wj + 32 * b + r];
            }
        };
        frag_a(0, 0); frag_b(0, 0, 0); frag_b(0, 0, 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t & 1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int g = 4 * t + jj;
                if (g >= 2 && g < 8) {              // six staging pieces: next slice into LDS, reload with the slice after
                    const int pc = g - 2;
                    if (pc < 2) { a_store(pc, an); a_load(pc, k2); }
                    else { b_store(pc - 2, bn); b_load(pc - 2, k2); }
                }
                if (t < 3) {
                    if (jj == 0) frag_a(t + 1, q ^ 1);
                    if (jj == 1) frag_b(t + 1, q ^ 1, 0);
                    if (jj == 2) frag_b(t + 1, q ^ 1, 1);
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][jj], fb[q][0][jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][jj], fb[q][1][jj], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    const int rbase = i0 + 32 * wi;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + 64 * wj + 32 * b + r;
        if (j >= N) continue;
        const float bj = bias != nullptr ? bias[j] : 0.f;
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
            float v = acc[b][i] + bj;
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


// 64 x 64 tiles for products of a FEW HUNDRED rows (the frozen text tower at one bag x 10 prompts x 77 tokens = 770 rows:
// the learnable-prompt step of the one-bag-per-GPU regime; 96 such products per step).  64 x 128 tiles make 156 workgroups of
// 770 x 1536 and the 128 x 128 path 84 (+ split-K): 35 - 42 us for 1.2 - 1.6 GFLOP = 0.2 - 0.27 of peak.  Half-width tiles double
// the workgroups again (312 for 770 x 1536, 416 for 770 x 2048; 32 KB of LDS: five resident per CU), and blockIdx.z splits
// K when even that leaves CUs idle (770 x 512 with K = 2048: 104 tiles x 4 splits), the raw partial tiles going to the
// workspace for k_splitk_reduce and its epilogue.  Same pipeline, swizzle and fragment convention as k_gemm64; wave tile
// 32 x 32 (2 x 2 waves).  Measured on this kernel and left out: four independent accumulators per wave, 1 - 4 rotating
// register sets of prefetch (within 7 % of each other), 64-deep LDS stages (half the barriers: 6.72 - 6.84 against 6.75 - 6.80 ms
// for the learnable-prompt step) - what these launches wait for is neither the MFMA chain nor the loads nor the barrier alone.
#ifndef G64N_SETS
#define G64N_SETS 2      /* 1 / 2 / 3 / 4 register sets measured within 7 % of each other; 2 was the best */
#endif
template <int BMODE>
__global__ __launch_bounds__(256)
void k_gemm64n(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
               int M, int N, int K, int kchunk, float* __restrict__ partial, const float* __restrict__ bias, int act,
               const float* __restrict__ residual, int ldr, int accumulate, float* __restrict__ aux, int ldaux, int aux_mode) {
    constexpr int ASZ = 64 * 32;
    constexpr int BSZ = BMODE == 0 ? 64 * 32 : LG_BK * 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];       // 32 KB
    float* as = smem;
    float* bs = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    const int nslice = (kend - kbeg) / LG_BK;

    const int srow = tid >> 3, sch = tid & 7;
    const float* asrc[2];
    int aoff[2];
    const float* bsrc[2];
    int boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = srow + 32 * i;
        asrc[i] = A + (size_t)min(i0 + row, M - 1) * lda + 4 * sch + kbeg;
        aoff[i] = g64_swz(row, sch);
        if (BMODE == 0) {
            bsrc[i] = B + (size_t)min(j0 + row, N - 1) * ldb + 4 * sch + kbeg;
            boff[i] = g64_swz(row, sch);
        } else {
            const int kr = (tid >> 4) + 16 * i, c4 = tid & 15;
            bsrc[i] = B + (size_t)(kbeg + kr) * ldb + min(j0 + 4 * c4, max(N - 4, 0));
            boff[i] = kr * 64 + 4 * c4;
        }
    }
    // G64N_SETS register sets in rotation: the loads of slice q are issued G64N_SETS slices before the slice whose MFMAs
    // they are stored under (with mostly ONE workgroup of four waves per CU nothing else hides an L2 / Infinity Cache
    // round trip: one set - a lead of one 16-MFMA slice, 0.43 us - left the kernel waiting 1.5 us per slice)
    constexpr int NS = G64N_SETS;
    f32x4 ra[NS][2], rb[NS][2];
    auto a_load = [&](int set, int i, int k0) { ra[set][i] = *reinterpret_cast<const f32x4*>(asrc[i] + k0); };
    auto b_load = [&](int set, int i, int k0) {
        rb[set][i] = *reinterpret_cast<const f32x4*>(BMODE == 0 ? bsrc[i] + k0 : bsrc[i] + (size_t)k0 * ldb);
    };
    auto a_store = [&](int set, int i, float* dst) { *reinterpret_cast<f32x4*>(dst + aoff[i]) = ra[set][i]; };
    auto b_store = [&](int set, int i, float* dst) { *reinterpret_cast<f32x4*>(dst + boff[i]) = rb[set][i]; };

    // (one accumulator per wave: four independent ones - MFMA jj of a k-group into accumulator jj - measured the same, so
    // the dependent chain is not what these short launches wait for)
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) { a_load(0, i, 0); b_load(0, i, 0); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { a_store(0, i, as); b_store(0, i, bs); }
#pragma unroll
    for (int q = 1; q <= NS; ++q) {
        const int kq = min(q, nslice - 1) * LG_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) { a_load(q % NS, i, kq); b_load(q % NS, i, kq); }
    }
    __syncthreads();
    const int arow = 32 * wi + r;
    auto slice = [&](int s, auto set_c) {
        constexpr int set = decltype(set_c)::value;             // = (s + 1) % NS: holds slice s + 1, then takes slice s + 1 + NS
        const int buf = s & 1;
        const int k2 = min(s + 1 + NS, nslice - 1) * LG_BK;
        const float* ab = as + buf * ASZ;
        const float* bb = bs + buf * BSZ;
        float* an = as + (buf ^ 1) * ASZ;
        float* bn = bs + (buf ^ 1) * BSZ;
        f32x4 fa[2], fb[2];
        auto frag_a = [&](int t, int q) { fa[q] = *reinterpret_cast<const f32x4*>(ab + g64_swz(arow, 2 * t + h)); };
        auto frag_b = [&](int t, int q) {
            if (BMODE == 0) {
                fb[q] = *reinterpret_cast<const f32x4*>(bb + g64_swz(32 * wj + r, 2 * t + h));
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) fb[q][jj] = bb[(8 * t + 4 * h + jj) * 64 + 32 * wj + r];
            }
        };
        frag_a(0, 0); frag_b(0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = t & 1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int g = 4 * t + jj;
                if (g >= 2 && g < 6) {              // four staging pieces: slice s + 1 into LDS, its registers reloaded
                    const int pc = g - 2;
                    if (pc < 2) { a_store(set, pc, an); a_load(set, pc, k2); }
                    else { b_store(set, pc - 2, bn); b_load(set, pc - 2, k2); }
                }
                if (t < 3) {
                    if (jj == 0) frag_a(t + 1, q ^ 1);
                    if (jj == 1) frag_b(t + 1, q ^ 1);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][jj], fb[q][jj], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    for (int s = 0; s < nslice; s += NS) {
        slice(s, std::integral_constant<int, 1 % NS>{});
        if (NS > 1 && s + 1 < nslice) slice(s + 1, std::integral_constant<int, 2 % NS>{});
        if (NS > 2 && s + 2 < nslice) slice(s + 2, std::integral_constant<int, 3 % NS>{});
        if (NS > 3 && s + 3 < nslice) slice(s + 3, std::integral_constant<int, 4 % NS>{});
    }

    const int rbase = i0 + 32 * wi;
    const int j = j0 + 32 * wj + r;
    if (j >= N) return;
    if (partial != nullptr) {                       // split-K: the raw tile, epilogue in k_splitk_reduce
        float* o = partial + (size_t)blockIdx.z * M * N;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = rbase + mfma32_row(i, h);
            if (row < M) o[(size_t)row * N + j] = acc[i];
        }
        return;
    }
    const float bj = bias != nullptr ? bias[j] : 0.f;
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
        float v = acc[i] + bj;
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

// plan for k_gemm64n: products of at most 2048 rows whose 64 x 128 tiles would not fill 1.5 rounds; S = K splits
static bool small_tile_plan(int M, int N, int K, int a_mode, int* S_out, int* kchunk_out) {
    if (a_mode != 0 || M > 2048 || K < 256 || (K % LG_BK) != 0 || (N & 3)) return false;
    const long t128 = (long)((M + 63) / 64) * ((N + 127) / 128);
    if (t128 >= 3 * MIL_NUM_CU / 2) return false;                  // k_gemm64 takes it
    const long t = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (t < 16) return false;                                      // a handful of tiles: the split-K plan of the big tile
    // target number of workgroups: one per CU (end to end, the learnable-prompt step of one ragged bag: 7.33 / 6.83 / 6.80 /
    // 7.07 / 7.11 ms at 128 / 192 / 256 / 320 / 512 - a split costs its partial tiles and the fold launch)
    static const int slots = getenv("MIL_G64N_SLOTS") ? atoi(getenv("MIL_G64N_SLOTS")) : MIL_NUM_CU;
    int S = (int)((slots + t / 2) / t);
    if (S > K / 256) S = K / 256;                                  // at least eight 32-deep slices per split
    if (S < 1) S = 1;
    int kchunk = ((K + S - 1) / S + LG_BK - 1) / LG_BK * LG_BK;
    S = (K + kchunk - 1) / kchunk;
    *S_out = S;
    *kchunk_out = S > 1 ? kchunk : K;
    return true;
}


// The grouped contraction over a FEW LONG groups (one ragged bag per step: pooled vectors and the absorbed vectors'
// gradients of the multi-token attention, C_g[M <= 128, N] = A[rows_g, :M]^T . B[rows_g, :N]).  The 128 x 128 form gives a
// 10 000-row bag 4 column tiles x 32 row chunks = 128 workgroups of 12 slices, one per CU with nothing to overlap their
// latencies: 43 us + 10 us of fold for 1.2 GFLOP.  Here: 64 x 64 tiles of the output, K (the group's rows) split so that
// about three workgroups share a CU, each a handful of slices long (many short groups - 32 bags x 1024 rows - the same
// tiles with one or two splits: no rows of M padded to 128); both operands are k-major, so their [32][64] LDS
// images are read by consecutive lanes (conflict-free ds_read_b32).  blockIdx.z = group * S + split; the split's partial
// tile goes to ws[(group * S + split)][M][N] for k_grouped_fold.
__global__ __launch_bounds__(256)
void k_gemm64tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ ws, int M, int N,
                const int32_t* __restrict__ grp_off, int S) {
    constexpr int TSZ = LG_BK * 64;
    __shared__ __attribute__((aligned(16))) float smem[4 * TSZ];               // 32 KB: [2] A stages, [2] B stages
    float* as = smem;
    float* bs = smem + 2 * TSZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int g = blockIdx.z / S, sp = blockIdx.z % S;
    const int goff = grp_off[g], gn = grp_off[g + 1] - goff;
    const int chunk = ((gn + S - 1) / S + LG_BK - 1) / LG_BK * LG_BK;
    const int kbeg = min(gn, sp * chunk), kend = min(gn, kbeg + chunk);
    const int nslice = (kend - kbeg + LG_BK - 1) / LG_BK;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    A += (size_t)goff * lda;
    B += (size_t)goff * ldb;

    const int kr = tid >> 4, c4 = tid & 15;                    // staging: k rows kr, kr + 16; 16-byte column chunk c4
    const int acol = min(i0 + 4 * c4, max(M - 4, 0)), bcol = min(j0 + 4 * c4, max(N - 4, 0));
    f32x4 ra[2][2], rb[2][2];
    auto load = [&](int set, int sl) {                         // slice sl of this split; rows at / beyond kend contribute zero
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = kbeg + sl * LG_BK + kr + 16 * p;
            const bool ok = row < kend;
            const int rc = ok ? row : kend - 1;
            const f32x4 a = *reinterpret_cast<const f32x4*>(A + (size_t)rc * lda + acol);
            ra[set][p] = ok ? a : f32x4{0.f, 0.f, 0.f, 0.f};
            rb[set][p] = *reinterpret_cast<const f32x4*>(B + (size_t)rc * ldb + bcol);
        }
    };
    auto store = [&](int set, float* an, float* bn) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4*>(an + (kr + 16 * p) * 64 + 4 * c4) = ra[set][p];
            *reinterpret_cast<f32x4*>(bn + (kr + 16 * p) * 64 + 4 * c4) = rb[set][p];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (nslice > 0) {
        load(0, 0);
        store(0, as, bs);
        load(1, min(1, nslice - 1));
        load(0, min(2, nslice - 1));
        __syncthreads();
        auto slice = [&](int s, auto set_c) {
            constexpr int set = decltype(set_c)::value;         // = (s + 1) & 1: holds slice s + 1, then takes slice s + 3
            const int buf = s & 1;
            const float* ab = as + buf * TSZ;
            const float* bb = bs + buf * TSZ;
            float fa[2][4], fb[2][4];
            auto frag = [&](int t, int q) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    fa[q][jj] = ab[(8 * t + 4 * h + jj) * 64 + 32 * wi + r];
                    fb[q][jj] = bb[(8 * t + 4 * h + jj) * 64 + 32 * wj + r];
                }
            };
            frag(0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int q = t & 1;
                if (t == 1) { store(set, as + (buf ^ 1) * TSZ, bs + (buf ^ 1) * TSZ); load(set, min(s + 3, nslice - 1)); }
                if (t < 3) frag(t + 1, q ^ 1);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][jj], fb[q][jj], acc, 0, 0, 0);
            }
            __syncthreads();
        };
        for (int s = 0; s < nslice; s += 2) {
            slice(s, std::integral_constant<int, 1>{});
            if (s + 1 < nslice) slice(s + 1, std::integral_constant<int, 0>{});
        }
    }
    float* o = ws + (size_t)blockIdx.z * M * N;
    const int n = j0 + 32 * wj + r;
    if (n >= N) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = i0 + 32 * wi + mfma32_row(i, h);
        if (m < M) o[(size_t)m * N + n] = acc[i];
    }
}

// plan of k_gemm64tn: M <= 128 and groups of at least 512 rows; S = splits of a group's rows (1: straight into C, no fold)
static bool tn64_plan(int G, int max_group_rows, int M, int N, int* S_out) {
    static const int off = getenv("MIL_TN64_OFF") ? atoi(getenv("MIL_TN64_OFF")) : 0;      // A/B switch
    if (off || G <= 0 || max_group_rows < 512 || M > 128 || M < 4 || (M & 3) || (N & 3)) return false;
    const long tiles = (long)G * ((M + 63) / 64) * ((N + 63) / 64);
    int S = (int)((3 * MIL_NUM_CU + tiles / 2) / tiles);
    if (S > max_group_rows / 128) S = max_group_rows / 128;       // at least four slices per split
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    *S_out = S;
    return true;
}
