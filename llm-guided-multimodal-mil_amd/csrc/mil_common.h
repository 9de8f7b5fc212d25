// Shared device helpers for the gfx950 kernels.  CDNA4 only: 64-lane waves, MFMA, LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/mil_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MIL_WAVE 64
// Compute units of the current device, read once per process (256 on MI355X).  Host code only: grid sizing,
// split-K cost models, tile-quantisation thresholds.
static inline int mil_num_cu() {
    static int cached = 0;
    if (cached <= 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cached = n;
        else
            return 256;                     // no device visible (host-only sizing queries): the MI355X value
    }
    return cached;
}
#define MIL_NUM_CU (mil_num_cu())

// An operand larger than this is streamed from HBM whatever the access order (the Infinity Cache holds 256 MB, and a
// linear sweep over more than that leaves nothing useful behind): its loads carry the nontemporal hint.
#define MIL_STREAM_BYTES ((size_t)192 << 20)

#define MIL_CHECK_LAUNCH()                               \
    do {                                                 \
        hipError_t e_ = hipGetLastError();               \
        if (e_ != hipSuccess) return (int)e_;            \
    } while (0)

// Cross-lane primitives of the wave reductions below, none of which touches the LDS crossbar (ds_bpermute, what
// __shfl_xor compiles to, costs an LDS issue slot per step; k_apool_partial spent a third of its time in them):
//   swap32_add(x, y): lanes 0-31 get x[l] + x[l + 32], lanes 32-63 get y[l - 32] + y[l]      (v_permlane32_swap, gfx950)
//   swap16_add(x, y): even 16-lane rows get x[l] + x[l + 16], odd rows y[l - 16] + y[l]     (v_permlane16_swap, gfx950)
//   dpp_mov<ctrl>: row_ror:8 (0x128) = xor 8, row_half_mirror (0x141), quad_perm [1,0,3,2] (0xB1) / [2,3,0,1] (0x4E)
// The swaps go through inline asm: with the builtins (__builtin_amdgcn_permlane32_swap) this compiler adds the first
// result to itself (v_add v, v6, v6 after v_permlane32_swap v6, v2; checked on the box with tools/variants probes).
__device__ __forceinline__ float swap32_add(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float swap16_add(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float swap32_max(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return fmaxf(x, y);
}
__device__ __forceinline__ float swap16_max(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return fmaxf(x, y);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// value of lane l ^ 4: row_shl:4 into the lanes with bit 2 clear (banks 0, 2), row_shr:4 into the others
__device__ __forceinline__ float dpp_xor4(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x114, 0xf, 0xa, false);
    return __builtin_bit_cast(float, t);
}

// All-reduce over the 64 lanes of a wave, every lane ends with the result: quad permutations (xor 1, xor 2), the mirror
// inside 8 lanes, the rotation by 8 inside a 16-lane row, then the 16- and 32-lane swaps - six VALU instructions, no LDS
// crossbar (the butterfly of __shfl_xor steps this replaces was six ds_bpermute round trips; the row-per-wave kernels -
// LayerNorm, pool tiles, tails - issue two to four of them per row).
__device__ __forceinline__ float wave_allsum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x128>(v);
    v = swap16_add(v, v);
    return swap32_add(v, v);
}
__device__ __forceinline__ float wave_allmax(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    v = swap16_max(v, v);
    return swap32_max(v, v);
}
// All-reduce inside each 32-lane half (lanes l and l^32 stay separate).
__device__ __forceinline__ float half_allsum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Sum over each 32-lane half with DPP adds only (no LDS round trips: __shfl_xor lowers to ds_bpermute_b32, one LDS
// operation plus its latency per step - the gate epilogues ran 160 of them back to back).  The total of lanes 0-31 is
// valid in lane 31, that of lanes 32-63 in lane 63; other lanes hold partial sums.
__device__ __forceinline__ float half_sum_lane31(float v) {
    // row_shr:n = 0x110 + n (shift right inside a row of 16, zeros shifted in), row_bcast:15 = 0x142 (lane 15 of each
    // row to the next row; row_mask 0xa = rows 1 and 3 take it)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));
    return v;
}

// Sum 16 per-lane values over the 64 lanes of a wave: each step halves the number of values a lane carries while it
// doubles the lanes summed.  Afterwards lane l holds the wave total of value
// index k(l) = 8 b5 + 4 b4 + 2 b3 + b2 (b_i = bit i of l); wave_reduce16_owner(k) is one lane holding index k.
__device__ __forceinline__ float wave_reduce16(const float (&v)[16], int lane) {
    float w8[8], w4[4], w2[2];
    const bool s3 = lane & 8, s2 = lane & 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) w8[j] = swap32_add(v[j], v[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = swap16_add(w8[j], w8[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j) w2[j] = (s3 ? w4[j + 2] : w4[j]) + dpp_mov<0x128>(s3 ? w4[j] : w4[j + 2]);
    float r = (s2 ? w2[1] : w2[0]) + dpp_xor4(s2 ? w2[0] : w2[1]);
    r += dpp_mov<0x4E>(r);
    r += dpp_mov<0xB1>(r);
    return r;
}
// Eight values; lane l ends with the total of index 4 b5 + 2 b4 + b3.
__device__ __forceinline__ float wave_reduce8(const float (&v)[8], int lane) {
    float w4[4], w2[2];
    const bool s3 = lane & 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = swap32_add(v[j], v[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j) w2[j] = swap16_add(w4[j], w4[j + 2]);
    float r = (s3 ? w2[1] : w2[0]) + dpp_mov<0x128>(s3 ? w2[0] : w2[1]);
    r += dpp_mov<0x141>(r);          // 8-lane group: i <-> 7 - i, then the two quad permutations complete the sum
    r += dpp_mov<0xB1>(r);
    r += dpp_mov<0x4E>(r);
    return r;
}
// the total of index k (0..7) of wave_reduce8, made wave-uniform (k is a compile-time constant after unrolling)
__device__ __forceinline__ float wave_reduce8_get(float r, int k) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), ((k >> 2) & 1) * 32 + ((k >> 1) & 1) * 16 + (k & 1) * 8));
}
__device__ __forceinline__ int wave_reduce16_index(int lane) {
    return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// exp2-based forms: v_exp_f32 + v_rcp_f32 (1 ulp), abs error ~1e-7 on outputs in [-1, 1].  The reciprocal is the bare
// v_rcp_f32: __frcp_rn expands to the full IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup, ~11 VALU
// instructions per call) - in the gate epilogues (96-192 activations per lane, no MFMA left to hide under) that was
// two thirds of the instructions: the bf16 gate kernel spent as many cycles in its epilogue as in its main loop.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// torch.optim.Adam on one element (L2 weight decay folded into the gradient, bias-corrected, eps outside the sqrt);
// train_ddp.py:115-118.  lr_bc1 = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t).
__device__ __forceinline__ float adam_one(float& pi, float gi, float& mi, float& vi, float lr_bc1, float b1, float b2,
                                          float eps, float wd, float gscale, float bc2_sqrt) {
    // every product / sum pinned (no compiler-chosen FMA contraction): the stand-alone Adam kernel and the reduce kernel
    // that applies Adam on the spot (AdamFuse) must produce the same bits
    const float g = __fmaf_rn(wd, pi, __fmul_rn(gi, gscale));
    mi = __fmaf_rn(b1, mi, __fmul_rn(1.0f - b1, g));
    vi = __fmaf_rn(b2, vi, __fmul_rn(__fmul_rn(1.0f - b2, g), g));
    // torch: denom = sqrt(v)/sqrt(bc2) + eps; param -= (lr / bc1) * m / denom
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
    pi = __fsub_rn(pi, __fmul_rn(lr_bc1, __fdiv_rn(mi, denom)));
    return pi;
}

// Adam applied by the kernel that PRODUCES the final gradient (the split-K reduce of the image-only step at world size 1: no
// all-reduce between gradient and update): the thread that stores gradient element i of the flat buffer updates parameter i
// and its moments on the spot - one launch less per step.  param == NULL: off.
struct AdamFuse {
    float* param;             // flat parameters
    const float* grad_base;   // flat gradients: index = (address the gradient is stored at) - grad_base
    float* m;
    float* v;
    float lr_bc1, b1, b2, eps, wd, gscale, bc2_sqrt;   // lr_bc1: the host passes bc1 = 1 - beta1^t here; resolve turns it into lr / bc1
    const int* step_dev;      // or NULL.  Device counter of the steps ALREADY taken (a step replayed from a hipGraph): the
    float lr;                 // kernel forms the bias corrections from it instead of taking them from the host
    const float* lr_dev;      // or NULL.  Learning rate in device memory (a schedule changes it between replays of one graph)
    int* done;                // or NULL.  With step_dev: sign-off word (zero between launches) - the LAST workgroup of the grid to
                              // resolve advances *step_dev itself, so the step needs no one-thread increment launch behind it
};
// lr / bc1 and sqrt(bc2) of this launch (k_adam's expressions).  With a device step counter every workgroup derives the two
// bias corrections itself; with lr_dev the learning rate is read from device memory, so a captured graph follows the schedule.
__device__ __forceinline__ AdamFuse adam_fuse_resolve(const AdamFuse& ad, float* bcs /* __shared__ [2] */) {
    AdamFuse r = ad;
    if (ad.param == nullptr) return r;
    float bc1 = ad.lr_bc1;
    if (ad.step_dev != nullptr) {
        if (threadIdx.x == 0) {
            const double st = (double)(*ad.step_dev + 1);
            bcs[0] = (float)(1.0 - pow((double)ad.b1, st));
            bcs[1] = (float)sqrt(1.0 - pow((double)ad.b2, st));
        }
        __syncthreads();
        bc1 = bcs[0];
        r.bc2_sqrt = bcs[1];
        if (ad.done != nullptr && threadIdx.x == 0) {
            // every workgroup reads the step number before it signs off; the counter moves once all of them have
            const int prev = __hip_atomic_fetch_add(ad.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == (int)(gridDim.x * gridDim.y * gridDim.z) - 1) {
                __hip_atomic_store(ad.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(const_cast<int*>(ad.step_dev), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    const float lr = ad.lr_dev != nullptr ? *ad.lr_dev : ad.lr;
    r.lr_bc1 = lr / bc1;
    return r;
}
__device__ __forceinline__ void adam_fused(const AdamFuse& ad, const float* gptr, float g) {
    if (ad.param == nullptr) return;
    const size_t i = (size_t)(gptr - ad.grad_base);
    float p = ad.param[i], m = ad.m[i], v = ad.v[i];
    adam_one(p, g, m, v, ad.lr_bc1, ad.b1, ad.b2, ad.eps, ad.wd, ad.gscale, ad.bc2_sqrt);
    ad.param[i] = p;
    ad.m[i] = m;
    ad.v[i] = v;
}
// returns the updated parameters (the caller may have a bf16 shadow to refresh); g itself when there is no update
__device__ __forceinline__ f32x4 adam_fused4(const AdamFuse& ad, const float* gptr, f32x4 g) {
    if (ad.param == nullptr) return g;
    const size_t i = (size_t)(gptr - ad.grad_base);
    f32x4 p = *reinterpret_cast<const f32x4*>(ad.param + i), m = *reinterpret_cast<const f32x4*>(ad.m + i),
          v = *reinterpret_cast<const f32x4*>(ad.v + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_one(pe, g[e], me, ve, ad.lr_bc1, ad.b1, ad.b2, ad.eps, ad.wd, ad.gscale, ad.bc2_sqrt);
        p[e] = pe; m[e] = me; v[e] = ve;
    }
    *reinterpret_cast<f32x4*>(ad.param + i) = p;
    *reinterpret_cast<f32x4*>(ad.m + i) = m;
    *reinterpret_cast<f32x4*>(ad.v + i) = v;
    return p;
}

// Row of a 32x32 MFMA accumulator register: C/D layout col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)   (dtype independent on gfx950).
__device__ __forceinline__ int mfma32_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct TileMapJob {            // mil_build_tile_map's arguments as an optional rider of another launch (bag_len == NULL: none)
    const int32_t* bag_len;
    int B;
    int32_t* tile_map;
    int32_t* bag_tile_off;
    int32_t* rows_out;
    int T_cap;
};

// Tile map of a batch whose bag lengths live on the device (one workgroup of 256 threads; see k_build_tile_map in
// gated_pool.hip): shared by that kernel and by the generator launch that carries it as an extra workgroup (dropout.hip).
__device__ __forceinline__ void build_tile_map_block(const int32_t* __restrict__ bag_len, int B, int32_t* __restrict__ tile_map,
                                                     int32_t* __restrict__ bag_tile_off, int32_t* __restrict__ rows_out,
                                                     int T_cap) {
    __shared__ int s_row[1025], s_tile[1025];
    const int tid = threadIdx.x;
    if (tid == 0) {
        int r = 0, t = 0;
        for (int b = 0; b < B; ++b) {
            s_row[b] = r;
            s_tile[b] = t;
            const int n = max(bag_len[b], 0);
            r += n;
            t += (n + MIL_POOL_TILE - 1) / MIL_POOL_TILE;
        }
        s_row[B] = r;
        s_tile[B] = min(t, T_cap);
        rows_out[0] = r;
    }
    __syncthreads();
    for (int b = tid; b <= B; b += 256) bag_tile_off[b] = min(s_tile[b], T_cap);
    const int T = s_tile[B];
    for (int b = 0; b < B; ++b) {
        const int t0 = s_tile[b], t1 = min(s_tile[b + 1], T_cap), r0 = s_row[b], r1 = s_row[b + 1];
        for (int t = t0 + tid; t < t1; t += 256) {
            const int row0 = r0 + (t - t0) * MIL_POOL_TILE;
            reinterpret_cast<int4*>(tile_map)[t] = make_int4(b, row0, min(MIL_POOL_TILE, r1 - row0), 0);
        }
    }
    for (int t = T + tid; t < T_cap; t += 256) reinterpret_cast<int4*>(tile_map)[t] = make_int4(0, 0, 0, 0);
}
