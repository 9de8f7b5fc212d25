// K2, one-text-token form of the token->image attention (model/sam/transformer.py:291-295,113-118 with T = 1).
//
// With a single query per bag the N-sized K and V projections are not needed: per head h
//   score_h[n] = q_h . (Wk_h kin_n + bk_h) / sqrt(c)  =  (Wk_h^T q_h) . kin_n / sqrt(c)  + const     (kin = keys + pe)
//   out_h      = sum_n a_h[n] (Wv_h keys_n + bv_h)     =  Wv_h (sum_n a_h[n] keys_n) + bv_h          (sum_n a_h[n] = 1)
// so the projections are ABSORBED into H query vectors Qp[h] = Wk_h^T q_h (E long) and H pooled key vectors
// pooled[h] = sum_n a_h[n] keys_n; the constant q_h . bk_h drops out of the softmax (k_proj.bias gets exactly zero
// gradient, as it mathematically must).  The image side becomes an H-head attention POOL streamed over the keys -
// HBM-bound like the MIL pool - instead of two [N, 512] x [512, 256] GEMMs (2 x 8.6 GFLOP per 32 bags x 1024 patches)
// forward and four more backward.  The reference computes the same numbers through q/k/v Linear + softmax + matmul.
//
// Kernels (E = 512 = 64 lanes x 8, H <= 8, head dim c, rows tiled 64 per workgroup, split-N partials + merge):
//   k_absorb_query / _bwd     Qp[b][h][:] = sum_c qp[b][hc + c'] Wk[hc + c'][:]
//   k_apool_partial / _merge  online-softmax pool of the keys under H absorbed queries
//   k_apool_dots / _bwd_apply per row: recompute a_h[n] (MFMA dots); dkeys, and per-tile partial of dQp
//   k_value_proj / _bwd       o[b][hc + c'] = Wv[hc + c'] . pooled[b][h] + bv
#include "mil_common.h"
#include <cstdlib>
#include <cstring>

#define AP_H 8
#define AP_TILE 64

// ---------------------------------------------------------------------------------------------- absorbed query
// grid (B + pad blocks, H), block E/4 threads (float4 per thread).  Output layout: row b of qp (b = g T + t: token t of group
// g) goes to Qp[(g THp + t H + h)] - with T = 1, THp = H the plain [B, H, E]; with T text tokens per bag the [bags, THp, E]
// operand of the grouped products, whose rows T H .. THp - 1 of every group (padding up to a multiple of 32) are written as
// zeros by the blocks blockIdx.x >= B (they used to be a torch pad: a fill and a copy per call).  scale multiplies the
// result (the 1 / sqrt(c) of the scores).  bias / cb (both or neither): cb[g][t H + h] = scale * bias_h . qp[b][h], the
// column constant the other projection's bias adds to the scores (zeros in the padding), in the same launch.
__global__ void k_absorb_query(const float* __restrict__ qp, const float* __restrict__ Wk, int H, int C, int E,
                               float* __restrict__ Qp, int B, int T, int THp, float scale,
                               const float* __restrict__ bias, float* __restrict__ cb) {
    const int b = blockIdx.x, h = blockIdx.y, j4 = threadIdx.x;
    if (b >= B) {
        const int g = b - B;
        for (int r = T * H + h; r < THp; r += H) {
            *reinterpret_cast<f32x4*>(Qp + ((size_t)g * THp + r) * E + 4 * j4) = f32x4{0, 0, 0, 0};
            if (cb != nullptr && j4 == 0) cb[(size_t)g * THp + r] = 0.f;
        }
        return;
    }
    const int I = H * C;
    f32x4 acc = {0, 0, 0, 0};
    float cbv = 0.f;
    for (int c0 = 0; c0 < C; c0 += 16) {            // C is 32 or 64: 16 rows of Wk in flight per trip
        f32x4 wr[16];
        float q[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            wr[u] = *reinterpret_cast<const f32x4*>(Wk + (size_t)(h * C + c0 + u) * E + 4 * j4);
            q[u] = qp[(size_t)b * I + h * C + c0 + u];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += q[u] * wr[u];
        if (cb != nullptr && j4 == 0)
#pragma unroll
            for (int u = 0; u < 16; ++u) cbv += q[u] * bias[h * C + c0 + u];
    }
    const size_t orow = (size_t)(b / T) * THp + (b % T) * H + h;
    *reinterpret_cast<f32x4*>(Qp + orow * E + 4 * j4) = acc * scale;
    if (cb != nullptr && j4 == 0) cb[orow] = cbv * scale;        // the scores' column constant scale * bias_h . qp[b][h]
}

// dqp[b][hc + c'] = dQp[b][h] . Wk[hc + c'];   grid (B, H), 256 threads: 8 threads per c' (c <= 32) then shuffle
__global__ __launch_bounds__(256) void k_absorb_query_bwd_q(const float* __restrict__ dQp, const float* __restrict__ Wk,
                                                            int H, int C, int E, float* __restrict__ dqp, int T, int THp,
                                                            float scale, const float* __restrict__ bias,
                                                            const float* __restrict__ dcb) {
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int I = H * C;
    const int per = 256 / C;                       // threads per output (8 for C = 32, 4 for C = 64)
    const int c = tid / per, part = tid % per;
    const float* g = dQp + ((size_t)(b / T) * THp + (b % T) * H + h) * E;
    const float* w = Wk + (size_t)(h * C + c) * E;
    float v = 0.f;
    for (int j0 = 4 * part; j0 < E; j0 += 16 * per) {      // 4 x 16-byte loads of each operand in flight
        f32x4 gv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + 4 * per * u, E - 4);
            gv[u] = *reinterpret_cast<const f32x4*>(g + j);
            wv[u] = *reinterpret_cast<const f32x4*>(w + j);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + 4 * per * u < E) v += gv[u][0] * wv[u][0] + gv[u][1] * wv[u][1] + gv[u][2] * wv[u][2] + gv[u][3] * wv[u][3];
    }
    for (int m = per >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (part == 0) {
        if (dcb != nullptr) v += dcb[(size_t)(b / T) * THp + (b % T) * H + h] * bias[h * C + c];
        dqp[(size_t)b * I + h * C + c] = v * scale;
    }
}

// dWk[hc + c'][j] = sum_b qp[b][hc + c'] dQp[b][h][j];   grid H*C, block E/4
// grid H * C rows, block 4 x (E / 4) threads: the four thread groups take every fourth batch of 8 b's (with T text
// tokens per bag B is bags x T = a few hundred: one group walking all of them was 32 us of dependent latency) and
// fold through LDS in a fixed order.
#define AQ_GROUPS 4
__global__ void k_absorb_query_bwd_w(const float* __restrict__ qp, const float* __restrict__ dQp, int B, int H, int C,
                                     int E, float* __restrict__ dWk, int T, int THp, float scale,
                                     const float* __restrict__ dcb, float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) float red[(AQ_GROUPS - 1) * 1024];        // E <= 1024
    __shared__ float redb[AQ_GROUPS];
    const int E4 = E / 4;
    const int row = blockIdx.x, h = row / C, j4 = threadIdx.x % E4, grp = threadIdx.x / E4, I = H * C;
    f32x4 acc = {0, 0, 0, 0};
    float bacc = 0.f;
    for (int b0 = 8 * grp; b0 < B; b0 += 8 * AQ_GROUPS) {
        f32x4 gv[8];
        float q[8], dc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = min(b0 + u, B - 1);
            const size_t orow = (size_t)(b / T) * THp + (b % T) * H + h;
            gv[u] = *reinterpret_cast<const f32x4*>(dQp + orow * E + 4 * j4);
            q[u] = b0 + u < B ? qp[(size_t)b * I + row] : 0.f;
            dc[u] = dcb != nullptr && j4 == 0 ? dcb[orow] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc += q[u] * gv[u]; bacc += q[u] * dc[u]; }
    }
    if (grp > 0) *reinterpret_cast<f32x4*>(red + (grp - 1) * 1024 + 4 * j4) = acc;
    if (j4 == 0) redb[grp] = bacc;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int g = 0; g < AQ_GROUPS - 1; ++g) acc += *reinterpret_cast<const f32x4*>(red + g * 1024 + 4 * j4);
        *reinterpret_cast<f32x4*>(dWk + (size_t)row * E + 4 * j4) = acc * scale;
        if (dbias != nullptr && j4 == 0) dbias[row] = ((redb[0] + redb[1]) + (redb[2] + redb[3])) * scale;
    }
}

// ---------------------------------------------------------------------------------------------- absorbed pool, forward
// One workgroup (256 threads) per 64-key tile: tile_map[g] = {bag, key0, nkeys}.  Wave w owns rows w, w+4, ... (16 rows);
// lane l holds columns 4l + 256q (q < 2) of a row.  Partial per tile: acc [H][E] = sum_n exp(s_h[n] - m_h) keys_n, then
// (m_h, l_h) [H][2] with m_h the TILE maximum.
// Two phases (round 2; the one-pass online softmax it replaces re-scaled 8 x 512 accumulators per key row and evaluated
// the eight exponentials of a row on all 64 lanes: ~260 VALU instructions per row, 33 us per site at 32 x 1024 keys
// against 7 us of memory time):
//   1. every load of the wave's 16 rows is issued up front (32 + 32 16-byte loads in flight per lane) and the rows STAY
//      in registers; the eight head scores of a row (8 FMAs per lane and head, one 10-step transposing wave reduction)
//      go to LDS;
//   2. thread (row, head) pairs form the tile maxima, the 64 x 8 probabilities (one exponential each) and their sums
//      through LDS; then each wave accumulates its 16 register-resident rows with plain FMAs - no re-scaling - and the
//      four waves' accumulators (same reference maximum) are summed through LDS.
//
// LN = true (round 4): the keys are not read but MADE here - keys_n = LayerNorm(x_n + o[bag]) of the image->token attention
// of the block in front (model/sam/transformer.py:303-309 with one text token per bag: every patch receives the same row
// o[bag], see ops.layer_norm_bag_row) - written to `y` (+ mean / rstd to `stats`) on their way into the registers the pool
// reads them from: the LayerNorm launch and the pool's own read of its 64 MB output are gone.  Same arithmetic, in the same
// order, as k_layernorm_fwd512.  Padding tiles of a capacity bucket (nkeys < 0) write zero rows (finite: later passes read
// whole buckets).
struct LnbrFwd {
    const float* x;
    const float* o;
    const float* gamma;
    const float* beta;
    float eps;
    float* y;
    float* stats;
};
template <bool LN>
__global__ __launch_bounds__(256, 2) void k_apool_partial(const float* __restrict__ keys, const float* __restrict__ pe,
                                                       const float* __restrict__ Qp, const int32_t* __restrict__ k_off,
                                                       const int32_t* __restrict__ tile_map, float scale,
                                                       float* __restrict__ pacc, float* __restrict__ pml, LnbrFwd ln) {
    constexpr int E = 512, NQ = 2, RW = AP_TILE / 4;
    __shared__ __attribute__((aligned(16))) float red[3 * AP_H * E];
    __shared__ __attribute__((aligned(16))) float s_lds[AP_TILE][AP_H];      // scores, then probabilities
    __shared__ float mh_lds[4][AP_H], lh_lds[4][AP_H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int b = tile_map[3 * g], key0 = tile_map[3 * g + 1], nkeys = tile_map[3 * g + 2];
    const int pos0 = key0 - k_off[b];
    f32x4 kv[RW][NQ];
    if (LN) {
        if (nkeys <= 0) {
            for (int rr = wave; rr < -nkeys; rr += 4) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    *reinterpret_cast<f32x4*>(ln.y + (size_t)(key0 + rr) * E + 256 * q + 4 * lane) = f32x4{0, 0, 0, 0};
                if (lane == 0) { ln.stats[2 * (size_t)(key0 + rr)] = 0.f; ln.stats[2 * (size_t)(key0 + rr) + 1] = 0.f; }
            }
            return;                                                       // no bag reads this tile's partial
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int rr = min(wave + 4 * i, nkeys - 1);
#pragma unroll
            for (int q = 0; q < NQ; ++q) kv[i][q] = *reinterpret_cast<const f32x4*>(ln.x + (size_t)(key0 + rr) * E + 256 * q + 4 * lane);
        }
        f32x4 ov[NQ], gm[NQ], bt[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            ov[q] = *reinterpret_cast<const f32x4*>(ln.o + (size_t)b * E + 256 * q + 4 * lane);
            gm[q] = *reinterpret_cast<const f32x4*>(ln.gamma + 256 * q + 4 * lane);
            bt[q] = *reinterpret_cast<const f32x4*>(ln.beta + 256 * q + 4 * lane);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const f32x4 v0 = kv[i][0] + ov[0], v1 = kv[i][1] + ov[1];
            const f32x4 t = v0 + v1;
            const float mean = wave_allsum((t[0] + t[1]) + (t[2] + t[3])) / E;
            const f32x4 d0 = v0 - mean, d1 = v1 - mean;
            const f32x4 sq = d0 * d0 + d1 * d1;
            const float rstd = 1.0f / sqrtf(wave_allsum((sq[0] + sq[1]) + (sq[2] + sq[3])) / E + ln.eps);
            kv[i][0] = d0 * rstd * gm[0] + bt[0];
            kv[i][1] = d1 * rstd * gm[1] + bt[1];
            if (wave + 4 * i < nkeys) {                                    // wave-uniform
                const size_t row = (size_t)(key0 + wave + 4 * i);
#pragma unroll
                for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(ln.y + row * E + 256 * q + 4 * lane) = kv[i][q];
                if (lane == 0) { ln.stats[2 * row] = mean; ln.stats[2 * row + 1] = rstd; }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int rr = max(min(wave + 4 * i, nkeys - 1), 0);          // rows past the tile end: clamped, weight 0 below
#pragma unroll
            for (int q = 0; q < NQ; ++q) kv[i][q] = *reinterpret_cast<const f32x4*>(keys + (size_t)(key0 + rr) * E + 256 * q + 4 * lane);
        }
    }
    {
        f32x4 qv[AP_H][NQ];
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                qv[h][q] = scale * *reinterpret_cast<const f32x4*>(Qp + ((size_t)b * AP_H + h) * E + 256 * q + 4 * lane);
        // the positional rows (a 2 MB table, L2-resident) in batches of four, so that the keys (128 registers), the
        // queries (64) and one batch (32) leave room for two workgroups per CU
#pragma unroll
        for (int i0 = 0; i0 < RW; i0 += 4) {
            f32x4 pv[4][NQ];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = max(min(wave + 4 * (i0 + u), nkeys - 1), 0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) pv[u][q] = *reinterpret_cast<const f32x4*>(pe + (size_t)(pos0 + rr) * E + 256 * q + 4 * lane);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                float d8[AP_H];
#pragma unroll
                for (int h = 0; h < AP_H; ++h) {
                    f32x4 t = (kv[i][0] + pv[u][0]) * qv[h][0];
                    t += (kv[i][1] + pv[u][1]) * qv[h][1];
                    d8[h] = (t[0] + t[1]) + (t[2] + t[3]);
                }
                const float tot = wave_reduce8(d8, lane);          // lane 8k holds the score of head k
                if ((lane & 7) == 0) s_lds[wave + 4 * i][lane >> 3] = wave + 4 * i < nkeys ? tot : -INFINITY;
            }
        }
    }
    __syncthreads();
    // thread (row r0 = tid >> 3 and r0 + 32, head h = tid & 7): tile maximum, probabilities, their sum
    const int h_ = tid & 7, r0 = tid >> 3;
    const float s0 = s_lds[r0][h_], s1 = s_lds[32 + r0][h_];
    float m = fmaxf(s0, s1);
    m = fmaxf(m, dpp_mov<0x128>(m));          // lanes with the same head: xor 8, 16, 32 without the LDS crossbar
    m = swap16_max(m, m);
    m = swap32_max(m, m);
    if (lane < AP_H) mh_lds[wave][lane] = m;
    __syncthreads();
    const float mt = fmaxf(fmaxf(mh_lds[0][h_], mh_lds[1][h_]), fmaxf(mh_lds[2][h_], mh_lds[3][h_]));
    const float p0 = s0 == -INFINITY ? 0.f : __expf(s0 - mt), p1 = s1 == -INFINITY ? 0.f : __expf(s1 - mt);
    s_lds[r0][h_] = p0;
    s_lds[32 + r0][h_] = p1;
    float l = p0 + p1;
    l += dpp_mov<0x128>(l);
    l = swap16_add(l, l);
    l = swap32_add(l, l);
    if (lane < AP_H) lh_lds[wave][lane] = l;
    __syncthreads();
    f32x4 acc[AP_H][NQ];
#pragma unroll
    for (int h = 0; h < AP_H; ++h)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[h][q] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(&s_lds[wave + 4 * i][0]);      // wave-uniform: LDS broadcast
        const f32x4 pb = *reinterpret_cast<const f32x4*>(&s_lds[wave + 4 * i][4]);
#pragma unroll
        for (int h = 0; h < AP_H; ++h) {
            const float p = h < 4 ? pa[h & 3] : pb[h & 3];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[h][q] += p * kv[i][q];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                *reinterpret_cast<f32x4*>(red + ((wave - 1) * AP_H + h) * E + 256 * q + 4 * lane) = acc[h][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 v = acc[h][q];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4*>(red + (w * AP_H + h) * E + 256 * q + 4 * lane);
                *reinterpret_cast<f32x4*>(pacc + ((size_t)g * AP_H + h) * E + 256 * q + 4 * lane) = v;
            }
        if (lane < AP_H) {                                        // lane = head (h_ == lane in wave 0)
            pml[((size_t)g * AP_H + lane) * 2] = mt;
            pml[((size_t)g * AP_H + lane) * 2 + 1] = (lh_lds[0][lane] + lh_lds[1][lane]) + (lh_lds[2][lane] + lh_lds[3][lane]);
        }
    }
}

// merge over a bag's tiles: pooled[b][h][:], lse[b][h].   grid (B, H), block E/4 threads
__global__ void k_apool_merge(const float* __restrict__ pacc, const float* __restrict__ pml,
                              const int32_t* __restrict__ bag_tile_off, int E, float* __restrict__ pooled,
                              float* __restrict__ lse) {
    const int b = blockIdx.x, h = blockIdx.y, j4 = threadIdx.x;
    const int g0 = bag_tile_off[b], g1 = bag_tile_off[b + 1];
    float m = -INFINITY;
    {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {                 // eight loads in flight per pass: the merge is pure latency
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pml[((size_t)(g + u) * AP_H + h) * 2];
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, t[u]);
        }
        for (; g < g1; ++g) m = fmaxf(m, pml[((size_t)g * AP_H + h) * 2]);
    }
    float l = 0.f;
    f32x4 acc = {0, 0, 0, 0};
    {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float2 ml[8];
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ml[u] = *reinterpret_cast<const float2*>(pml + ((size_t)(g + u) * AP_H + h) * 2);
                t[u] = *reinterpret_cast<const f32x4*>(pacc + ((size_t)(g + u) * AP_H + h) * E + 4 * j4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float sc = __expf(ml[u].x - m);
                l += sc * ml[u].y;
                acc += sc * t[u];
            }
        }
        for (; g < g1; ++g) {
            const float sc = __expf(pml[((size_t)g * AP_H + h) * 2] - m);
            l += sc * pml[((size_t)g * AP_H + h) * 2 + 1];
            acc += sc * *reinterpret_cast<const f32x4*>(pacc + ((size_t)g * AP_H + h) * E + 4 * j4);
        }
    }
    const float inv = g1 > g0 ? 1.0f / l : 0.f;
    *reinterpret_cast<f32x4*>(pooled + ((size_t)b * AP_H + h) * E + 4 * j4) = acc * inv;
    if (j4 == 0) lse[b * AP_H + h] = g1 > g0 ? m + logf(l) : -INFINITY;
}

// ---------------------------------------------------------------------------------------------- absorbed pool, backward
// Per row n of a tile: a_h = exp(scale Qp_h . kin_n - lse_h);  da_h = dpooled_h . keys_n;  ds_h = a_h (da_h - cdot_h);
//   dkeys_n = sum_h (a_h dpooled_h + scale ds_h Qp_h);   dQp_h += scale ds_h kin_n  (per-tile partial, merged per bag)
// Two kernels: the per-row dot products (16 per row; as wave-wide reductions they cost 16 shuffle trees per row) are a skinny
// MFMA product: for a 16-row group  acc[16 x 16] = keys[16 x 512] . [Qp | dpooled]^T + pe[16 x 512] . [Qp | 0]^T
// (v_mfma_f32_16x16x4_f32, operands straight from global memory: lane (r, kq) loads 16 bytes of row r at
// k = 16t + 4kq and feeds four MFMAs), so columns 0-7 hold Qp_h . kin_n and columns 8-15 dpooled_h . keys_n.
//   ad[n][h] = a_h[n],   ad[n][8 + h] = scale * a_h[n] (da_h[n] - cdot_h)
// Round 2: the operands no longer come "straight from global memory".  In that form lane (r, kq) loaded 16 bytes of ITS
// row, so the 16 lanes of a quarter-wave touched 16 different rows = 16 cache lines per quarter, 64 tag look-ups per load
// instruction where a contiguous 1 KB needs 8 - the kernel ran at the L1's tag rate (25 us per site for 64 MB that take 7;
// splitting the contraction over twice the waves changed nothing).  Now the workgroup streams K in chunks of 64 floats:
// every wave instruction loads 4 rows x 256 contiguous bytes (keys and pe rows of the 64-key tile, the bag's 16-row
// [Qp | dpooled] operand once for all four waves), registers -> LDS (row stride 68 floats: the 16 lanes of a fragment read
// hit 64 different banks), double-buffered with the next chunk's loads in flight under this chunk's 32 MFMAs per wave.
#define AD_KC 64
#define AD_LS 68
__global__ __launch_bounds__(256) void k_apool_dots(const float* __restrict__ keys, const float* __restrict__ pe,
                                                    const float* __restrict__ Qp, const float* __restrict__ lse,
                                                    const float* __restrict__ dpooled, const float* __restrict__ pooled,
                                                    const int32_t* __restrict__ k_off, const int32_t* __restrict__ tile_map,
                                                    float scale, float* __restrict__ ad) {
    constexpr int E = 512, NCH = E / AD_KC;
    __shared__ __attribute__((aligned(16))) float lds[2][(2 * AP_TILE + 16) * AD_LS];      // [stage][keys 64 | pe 64 | B 16][68]
    __shared__ float cd_lds[AP_H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int b = tile_map[3 * g], key0 = tile_map[3 * g + 1], nkeys = tile_map[3 * g + 2];
    const int pos0 = key0 - k_off[b];
    {
        // cdot_h = dpooled_h . pooled_h of this bag (the softmax backward's row constant): 32 lanes per head; 2 x 16 KB from
        // L2, under the first chunk's loads - it used to be a launch of its own (k_rowdot) in front of this kernel
        const int hh = tid >> 5, part = tid & 31;
        const float* dpv = dpooled + ((size_t)b * AP_H + hh) * E + 16 * part;
        const float* pv = pooled + ((size_t)b * AP_H + hh) * E + 16 * part;
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(dpv + 4 * u), b4 = *reinterpret_cast<const f32x4*>(pv + 4 * u);
            v += (a4[0] * b4[0] + a4[1] * b4[1]) + (a4[2] * b4[2] + a4[3] * b4[3]);
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (part == 0) cd_lds[hh] = v;
    }
    const bool active = 16 * wave < nkeys;                       // wave-uniform
    // staging map: thread -> (row srow + 16 i, 16-byte chunk sc) of the tile's [64][64] chunk; B: row srow (0..15), chunk sc
    const int srow = tid >> 4, sc = tid & 15;
    const float* kp[4];
    const float* pp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rowc = max(min(srow + 16 * i, nkeys - 1), 0);
        kp[i] = keys + (size_t)(key0 + rowc) * E + 4 * sc;
        pp[i] = pe + (size_t)(pos0 + rowc) * E + 4 * sc;
    }
    const float* bsrc = (srow < 8 ? Qp + ((size_t)b * AP_H + srow) * E : dpooled + ((size_t)b * AP_H + (srow - 8)) * E) + 4 * sc;
    f32x4 rk[4], rp[4], rb;
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rk[i] = *reinterpret_cast<const f32x4*>(kp[i] + AD_KC * c);
            rp[i] = *reinterpret_cast<const f32x4*>(pp[i] + AD_KC * c);
        }
        rb = *reinterpret_cast<const f32x4*>(bsrc + AD_KC * c);
    };
    auto swrite = [&](int st) {
        float* base = lds[st];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(base + (srow + 16 * i) * AD_LS + 4 * sc) = rk[i];
            *reinterpret_cast<f32x4*>(base + (AP_TILE + srow + 16 * i) * AD_LS + 4 * sc) = rp[i];
        }
        *reinterpret_cast<f32x4*>(base + (2 * AP_TILE + srow) * AD_LS + 4 * sc) = rb;
    };
    const int r = lane & 15, kq = lane >> 4;
    const float qmask = r < 8 ? 1.f : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    gload(0);
    swrite(0);
    __syncthreads();
    for (int c = 0; c < NCH; ++c) {
        const int st = c & 1;
        if (c + 1 < NCH) gload(c + 1);
        if (active) {
            const float* ak = lds[st] + (16 * wave + r) * AD_LS + 4 * kq;
            const float* ap = ak + AP_TILE * AD_LS;
            const float* bq = lds[st] + (2 * AP_TILE + r) * AD_LS + 4 * kq;
#pragma unroll
            for (int tt = 0; tt < AD_KC / 16; ++tt) {
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(ak + 16 * tt);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(ap + 16 * tt);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bq + 16 * tt);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[jj], bb[jj], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[jj], bb[jj] * qmask, acc, 0, 0, 0);
                }
            }
        }
        if (c + 1 < NCH) swrite(st ^ 1);
        __syncthreads();
    }
    if (!active) return;
    // lane (c, g4) holds column c of rows 4 g4 + i: columns c < 8 pair with c + 8
    const int h = r & 7;
    const float ls = lse[b * AP_H + h], cd = cd_lds[h];         // written before the loop's first barrier
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float other = __shfl_xor(acc[i], 8);
        const int row = 16 * wave + 4 * kq + i;
        if (r < 8 && row < nkeys) {
            const float a = __expf(acc[i] * scale - ls);
            float* o = ad + (size_t)(key0 + row) * 16;
            o[h] = a;
            o[8 + h] = a * (other - cd) * scale;
        }
    }
}

// dkeys_n = sum_h (a_h dpooled_h + ds_h Qp_h);  per-tile partial of dQp_h = sum_n ds_h kin_n.  Column-parallel: lane l
// owns columns 4l..4l+3 and 256+4l..+3, the four waves split the tile's rows; a_h / ds_h come from k_apool_dots.
// DK = false (round 4): only the per-tile partial of dQp - the rank-16 update of dkeys is left to the reader of dkeys
// (k_lnbr_bwd_r16 adds it to the gradient it loads), so neither dkeys_acc is read nor dkeys written here.
template <bool DK>
__global__ __launch_bounds__(256) void k_apool_bwd_apply(const float* __restrict__ keys, const float* __restrict__ pe,
                                                         const float* __restrict__ Qp, const float* __restrict__ dpooled,
                                                         const float* __restrict__ ad, const int32_t* __restrict__ k_off,
                                                         const int32_t* __restrict__ tile_map,
                                                         const float* __restrict__ dkeys_acc, float* __restrict__ dkeys,
                                                         float* __restrict__ pdq) {
    constexpr int E = 512, NQ = 2;
    __shared__ __attribute__((aligned(16))) float red[3 * AP_H * E];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int b = tile_map[3 * g], key0 = tile_map[3 * g + 1], nkeys = tile_map[3 * g + 2];
    const int pos0 = key0 - k_off[b];
    f32x4 qv[AP_H][NQ], dp[AP_H][NQ], dq[AP_H][NQ];
#pragma unroll
    for (int h = 0; h < AP_H; ++h)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (DK) {
                qv[h][q] = *reinterpret_cast<const f32x4*>(Qp + ((size_t)b * AP_H + h) * E + 256 * q + 4 * lane);
                dp[h][q] = *reinterpret_cast<const f32x4*>(dpooled + ((size_t)b * AP_H + h) * E + 256 * q + 4 * lane);
            }
            dq[h][q] = f32x4{0, 0, 0, 0};
        }
    if (!DK) {
        // rows of the tile four at a time per wave: 8 + 8 16-byte loads in flight per lane
        for (int r0 = wave; r0 < nkeys; r0 += 16) {
            f32x4 kin[4][NQ];
            float ds[4][AP_H];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = min(r0 + 4 * u, nkeys - 1);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    kin[u][q] = *reinterpret_cast<const f32x4*>(keys + (size_t)(key0 + rr) * E + 256 * q + 4 * lane) +
                                *reinterpret_cast<const f32x4*>(pe + (size_t)(pos0 + rr) * E + 256 * q + 4 * lane);
                const float* adr = ad + (size_t)(key0 + rr) * 16 + 8;      // wave-uniform: scalar loads
#pragma unroll
                for (int h = 0; h < AP_H; ++h) ds[u][h] = r0 + 4 * u < nkeys ? adr[h] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < AP_H; ++h)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) dq[h][q] += ds[u][h] * kin[u][q];
        }
    }
    if (DK && nkeys < 0) {
        // padding tile of a capacity bucket (mil_build_fusion_segs): rows key0 .. key0 - nkeys - 1 lie beyond every bag; their
        // gradient is exactly zero and is written here, so the caller need not clear dkeys
        for (int rr = wave; rr < -nkeys; rr += 4)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                *reinterpret_cast<f32x4*>(dkeys + (size_t)(key0 + rr) * E + 256 * q + 4 * lane) = f32x4{0, 0, 0, 0};
    }
    for (int rr = wave; DK && rr < nkeys; rr += 4) {
        const float* adr = ad + (size_t)(key0 + rr) * 16;          // wave-uniform: scalar loads
        float a[AP_H], ds[AP_H];
#pragma unroll
        for (int h = 0; h < AP_H; ++h) { a[h] = adr[h]; ds[h] = adr[8 + h]; }
        f32x4 kin[NQ], out[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            kin[q] = *reinterpret_cast<const f32x4*>(keys + (size_t)(key0 + rr) * E + 256 * q + 4 * lane) +
                     *reinterpret_cast<const f32x4*>(pe + (size_t)(pos0 + rr) * E + 256 * q + 4 * lane);
            // dkeys_acc: the gradient the keys receive from their other consumer, folded in here instead of in a
            // separate [N, 512] add
            out[q] = dkeys_acc != nullptr
                         ? *reinterpret_cast<const f32x4*>(dkeys_acc + (size_t)(key0 + rr) * E + 256 * q + 4 * lane)
                         : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                out[q] += a[h] * dp[h][q] + ds[h] * qv[h][q];
                dq[h][q] += ds[h] * kin[q];
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(dkeys + (size_t)(key0 + rr) * E + 256 * q + 4 * lane) = out[q];
    }
    if (wave > 0) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                *reinterpret_cast<f32x4*>(red + ((wave - 1) * AP_H + h) * E + 256 * q + 4 * lane) = dq[h][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 v = dq[h][q];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4*>(red + (w * AP_H + h) * E + 256 * q + 4 * lane);
                *reinterpret_cast<f32x4*>(pdq + ((size_t)g * AP_H + h) * E + 256 * q + 4 * lane) = v;
            }
    }
}

// dQp[b][h][:] = sum over the bag's tiles.  grid (B, H), 1024 threads = 8 groups x E / 4 column threads: group q sums the
// tiles g0 + q, g0 + q + 8, ... (four loads in flight), the eight partial sums are folded through LDS in group order.
__global__ __launch_bounds__(1024) void k_apool_bwd_merge(const float* __restrict__ pdq, const int32_t* __restrict__ bag_tile_off,
                                                          int E, float* __restrict__ dQp) {
    __shared__ __attribute__((aligned(16))) float red[8][512];
    const int b = blockIdx.x, h = blockIdx.y, j4 = threadIdx.x & 127, q = threadIdx.x >> 7;
    f32x4 acc = {0, 0, 0, 0};
    const int g0 = bag_tile_off[b], g1 = bag_tile_off[b + 1];
    int g = g0 + q;
    for (; g + 24 < g1; g += 32) {
        f32x4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(pdq + ((size_t)(g + 8 * u) * AP_H + h) * E + 4 * j4);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += t[u];
    }
    for (; g < g1; g += 8) acc += *reinterpret_cast<const f32x4*>(pdq + ((size_t)g * AP_H + h) * E + 4 * j4);
    *reinterpret_cast<f32x4*>(&red[q][4 * j4]) = acc;
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int u = 1; u < 8; ++u) acc += *reinterpret_cast<const f32x4*>(&red[u][4 * j4]);
        *reinterpret_cast<f32x4*>(dQp + ((size_t)b * AP_H + h) * E + 4 * j4) = acc;
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm(x + o[bag]) backward
// with the pool's rank-16 update folded into the gradient it loads (round 4).  The keys the pool read were y = LN(x + o[bag])
// (k_apool_partial<true>); their gradient is
//     dy_n = dy_acc_n (the other consumer of y)  +  sum_h (a_h[n] dpooled_h + ds_h[n] Qp_h)       (k_apool_bwd_apply<true>'s sum)
// and this kernel is its only reader - so the sum is formed HERE, per row, in the registers that feed the LayerNorm backward,
// instead of being written to and read back from a [N, 512] tensor (one read of dy_acc + one write of dkeys + one read by
// the LayerNorm backward: 3 x 64 MB per site at 32 x 1024 keys).  One workgroup per 64-key tile of the pool's tile map (a
// tile lies inside one bag: no second-bag slot as in k_layernorm_bagrow_bwd), wave w rows w, w + 4, ..; lane l owns columns
// 4l..4l+3 and 256+4l..; the bag's 16 vectors [dpooled | Qp] stay in 128 registers.  Per-tile partials
// part[g][3][E] = {dgamma, dbeta, do}; padding tiles (nkeys < 0) write the zero gradient of their rows.
__global__ __launch_bounds__(256, 2) void k_lnbr_bwd_r16(const float* __restrict__ x, const float* __restrict__ o,
                                                         const float* __restrict__ gamma, const float* __restrict__ stats,
                                                         const float* __restrict__ dy_acc, const float* __restrict__ ad,
                                                         const float* __restrict__ Qp, const float* __restrict__ dpooled,
                                                         const int32_t* __restrict__ tile_map, float* __restrict__ dx,
                                                         float* __restrict__ part) {
    constexpr int E = 512, NQ = 2;
    __shared__ __attribute__((aligned(16))) float red[3][3][E];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int b = tile_map[3 * g], key0 = tile_map[3 * g + 1], nkeys = tile_map[3 * g + 2];
    if (nkeys <= 0) {
        for (int rr = wave; rr < -nkeys; rr += 4)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                *reinterpret_cast<f32x4*>(dx + (size_t)(key0 + rr) * E + 256 * q + 4 * lane) = f32x4{0, 0, 0, 0};
        return;
    }
    f32x4 qv[AP_H][NQ], dp[AP_H][NQ], gm[NQ], ov[NQ], dg[NQ], db[NQ], d0[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h) {
            qv[h][q] = *reinterpret_cast<const f32x4*>(Qp + ((size_t)b * AP_H + h) * E + 256 * q + 4 * lane);
            dp[h][q] = *reinterpret_cast<const f32x4*>(dpooled + ((size_t)b * AP_H + h) * E + 256 * q + 4 * lane);
        }
        gm[q] = *reinterpret_cast<const f32x4*>(gamma + 256 * q + 4 * lane);
        ov[q] = *reinterpret_cast<const f32x4*>(o + (size_t)b * E + 256 * q + 4 * lane);
        dg[q] = db[q] = d0[q] = f32x4{0, 0, 0, 0};
    }
    f32x4 xn[NQ], dn[NQ];                                             // next row's loads, in flight under this row's arithmetic
    auto issue = [&](int rr) {
        const int rc = min(rr, nkeys - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            xn[q] = *reinterpret_cast<const f32x4*>(x + (size_t)(key0 + rc) * E + 256 * q + 4 * lane);
            dn[q] = dy_acc != nullptr ? *reinterpret_cast<const f32x4*>(dy_acc + (size_t)(key0 + rc) * E + 256 * q + 4 * lane)
                                      : f32x4{0, 0, 0, 0};
        }
    };
    issue(wave);
    for (int rr = wave; rr < nkeys; rr += 4) {
        const size_t row = (size_t)(key0 + rr);
        const float* adr = ad + row * 16;                              // wave-uniform: scalar loads
        float a[AP_H], ds[AP_H];
#pragma unroll
        for (int h = 0; h < AP_H; ++h) { a[h] = adr[h]; ds[h] = adr[8 + h]; }
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        f32x4 xc[NQ], d[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { xc[q] = xn[q]; d[q] = dn[q]; }
        issue(rr + 4);
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) d[q] += a[h] * dp[h][q] + ds[h] * qv[h][q];
        f32x4 xh[NQ], gg[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            xh[q] = (xc[q] + ov[q] - mean) * rstd;
            gg[q] = d[q] * gm[q];
            dg[q] += d[q] * xh[q];
            db[q] += d[q];
        }
        const f32x4 t1 = gg[0] + gg[1], t2 = gg[0] * xh[0] + gg[1] * xh[1];
        const float s1 = wave_allsum((t1[0] + t1[1]) + (t1[2] + t1[3])) / E;
        const float s2 = wave_allsum((t2[0] + t2[1]) + (t2[2] + t2[3])) / E;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 v = rstd * (gg[q] - s1 - xh[q] * s2);
            *reinterpret_cast<f32x4*>(dx + row * E + 256 * q + 4 * lane) = v;
            d0[q] += v;
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            *reinterpret_cast<f32x4*>(&red[wave - 1][0][256 * q + 4 * lane]) = dg[q];
            *reinterpret_cast<f32x4*>(&red[wave - 1][1][256 * q + 4 * lane]) = db[q];
            *reinterpret_cast<f32x4*>(&red[wave - 1][2][256 * q + 4 * lane]) = d0[q];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                dg[q] += *reinterpret_cast<const f32x4*>(&red[w][0][256 * q + 4 * lane]);
                db[q] += *reinterpret_cast<const f32x4*>(&red[w][1][256 * q + 4 * lane]);
                d0[q] += *reinterpret_cast<const f32x4*>(&red[w][2][256 * q + 4 * lane]);
            }
            *reinterpret_cast<f32x4*>(part + ((size_t)g * 3 + 0) * E + 256 * q + 4 * lane) = dg[q];
            *reinterpret_cast<f32x4*>(part + ((size_t)g * 3 + 1) * E + 256 * q + 4 * lane) = db[q];
            *reinterpret_cast<f32x4*>(part + ((size_t)g * 3 + 2) * E + 256 * q + 4 * lane) = d0[q];
        }
    }
}

// The whole backward of the pair in ONE pass over the rows (round 4): everything k_apool_dots, k_apool_bwd_apply<false> and
// k_lnbr_bwd_r16 do for a key row is row-local once the row sits in registers - the keys y_n = xhat_n gamma + beta are
// RECOMPUTED from x (the LayerNorm backward needs xhat_n anyway), the 16 dot products of the row (scores against the eight
// absorbed queries, dpooled_h . y_n) are per-lane partial sums + one 16-value transposing wave reduction, the softmax
// weights a_h and t_h = a_h (da_h - cdot_h) come back to every lane through 16 v_readlane, then the rank-16 update, the dQp
// partial and the LayerNorm backward run on the same registers.  Reads x + dy_acc (+ pe from L2), writes dx: 3 x 64 MB per
// site where the three kernels moved 5 (and the round-3 form 7).  The bag's 16 vectors [scale Qp | dpooled] live in LDS
// (32 KB, 16-byte reads, conflict-free; 64 KB per row and wave = 14 us of LDS time per site next to 37 us of HBM time); the
// same 32 KB fold the four waves' dQp partials afterwards.
// LN = false: the plain absorbed pool (the first block: its keys are the image projection, no norm in front): x holds the
// keys themselves, dx = dy_acc + the rank-16 update, no dgamma / dbeta / do partials - the one-pass form of k_apool_dots +
// k_apool_bwd_apply<true>.
template <bool LN>
__global__ __launch_bounds__(256, 2) void k_lnbr_apool_bwd_one(const float* __restrict__ x, const float* __restrict__ o,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ stats, const float* __restrict__ dy_acc,
                                                               const float* __restrict__ pe, const float* __restrict__ Qp,
                                                               const float* __restrict__ dpooled, const float* __restrict__ pooled,
                                                               const float* __restrict__ lse, const int32_t* __restrict__ k_off,
                                                               const int32_t* __restrict__ tile_map, float scale,
                                                               float* __restrict__ dx, float* __restrict__ part,
                                                               float* __restrict__ pdq) {
    constexpr int E = 512, NQ = 2;
    __shared__ __attribute__((aligned(16))) float V[16 * E];          // [scale Qp_h | dpooled_h]; afterwards the dQp fold
    __shared__ __attribute__((aligned(16))) float red3[3][3][E];
    __shared__ float cd_lds[AP_H];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int b = tile_map[3 * g], key0 = tile_map[3 * g + 1], nkeys = tile_map[3 * g + 2];
    if (nkeys <= 0) {
        for (int rr = wave; rr < -nkeys; rr += 4)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                *reinterpret_cast<f32x4*>(dx + (size_t)(key0 + rr) * E + 256 * q + 4 * lane) = f32x4{0, 0, 0, 0};
        return;
    }
    const int pos0 = key0 - k_off[b];
    // two rows of loads in flight per wave (buffers A, B: rows rr and rr + 4), each re-issued for the row 8 further on as soon
    // as its values are consumed: one row of arithmetic (~1500 cycles per wave, two waves per SIMD) does not cover HBM latency
    f32x4 xa[NQ], da_[NQ], pa[NQ], xb[NQ], db_[NQ], pb[NQ];
    auto issue = [&](int rr, f32x4 (&xn)[NQ], f32x4 (&dn)[NQ], f32x4 (&pn)[NQ]) {
        const int rc = min(rr, nkeys - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            xn[q] = *reinterpret_cast<const f32x4*>(x + (size_t)(key0 + rc) * E + 256 * q + 4 * lane);
            dn[q] = dy_acc != nullptr ? *reinterpret_cast<const f32x4*>(dy_acc + (size_t)(key0 + rc) * E + 256 * q + 4 * lane)
                                      : f32x4{0, 0, 0, 0};
            pn[q] = *reinterpret_cast<const f32x4*>(pe + (size_t)(pos0 + rc) * E + 256 * q + 4 * lane);
        }
    };
    issue(wave, xa, da_, pa);
    issue(wave + 4, xb, db_, pb);
    {
        // the bag's 16 vectors -> LDS, cdot_h = dpooled_h . pooled_h (32 lanes per head, as k_apool_dots)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 256 + tid, row = idx >> 7, c4 = idx & 127;
            f32x4 v;
            if (row < AP_H) v = scale * *reinterpret_cast<const f32x4*>(Qp + ((size_t)b * AP_H + row) * E + 4 * c4);
            else v = *reinterpret_cast<const f32x4*>(dpooled + ((size_t)b * AP_H + row - AP_H) * E + 4 * c4);
            *reinterpret_cast<f32x4*>(V + row * E + 4 * c4) = v;
        }
        const int hh = tid >> 5, pt = tid & 31;
        const float* dpv = dpooled + ((size_t)b * AP_H + hh) * E + 16 * pt;
        const float* pv = pooled + ((size_t)b * AP_H + hh) * E + 16 * pt;
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(dpv + 4 * u), b4 = *reinterpret_cast<const f32x4*>(pv + 4 * u);
            v += (a4[0] * b4[0] + a4[1] * b4[1]) + (a4[2] * b4[2] + a4[3] * b4[3]);
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (pt == 0) cd_lds[hh] = v;
    }
    f32x4 gm[NQ], bt[NQ], ov[NQ], dg[NQ], db[NQ], d0[NQ], dq[AP_H][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (LN) {
            gm[q] = *reinterpret_cast<const f32x4*>(gamma + 256 * q + 4 * lane);
            bt[q] = *reinterpret_cast<const f32x4*>(beta + 256 * q + 4 * lane);
            ov[q] = *reinterpret_cast<const f32x4*>(o + (size_t)b * E + 256 * q + 4 * lane);
        }
        dg[q] = db[q] = d0[q] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < AP_H; ++h) dq[h][q] = f32x4{0, 0, 0, 0};
    }
    __syncthreads();
    // wave_reduce16 leaves value index k(l) = 8 b5 + 4 b4 + 2 b3 + b2 on lane l: lanes < 32 the score of head hl, lanes >= 32
    // dpooled_hl . y of the same head
    const int hl = ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    const float lse_l = lse[b * AP_H + hl], cd_l = cd_lds[hl];
    const float* Vl = V + 4 * lane;
    auto one_row = [&](int rr, f32x4 (&xn)[NQ], f32x4 (&dn)[NQ], f32x4 (&pn)[NQ]) {
        const size_t row = (size_t)(key0 + rr);
        float mean = 0.f, rstd = 1.f;
        if (LN) { mean = stats[2 * row]; rstd = stats[2 * row + 1]; }
        f32x4 xh[NQ], y[NQ], kin[NQ], d[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (LN) {
                xh[q] = (xn[q] + ov[q] - mean) * rstd;
                y[q] = xh[q] * gm[q] + bt[q];
            } else {
                y[q] = xn[q];
            }
            kin[q] = y[q] + pn[q];
            d[q] = dn[q];
        }
        issue(rr + 8, xn, dn, pn);
        float p[16];
#pragma unroll
        for (int h = 0; h < AP_H; ++h) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(Vl + h * E), q1 = *reinterpret_cast<const f32x4*>(Vl + h * E + 256);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(Vl + (AP_H + h) * E), e1 = *reinterpret_cast<const f32x4*>(Vl + (AP_H + h) * E + 256);
            const f32x4 t = kin[0] * q0 + kin[1] * q1, u = y[0] * e0 + y[1] * e1;
            p[h] = (t[0] + t[1]) + (t[2] + t[3]);
            p[AP_H + h] = (u[0] + u[1]) + (u[2] + u[3]);
        }
        const float r = wave_reduce16(p, lane);
        const float av = __expf(r - lse_l);                                    // lanes < 32: a_hl
        const float hi = lane < 32 ? 0.f : r;
        const float da = swap32_add(hi, hi);                                    // lanes < 32: da_hl (from lane + 32)
        const float tv = av * (da - cd_l);
        float ah[AP_H], th[AP_H];
#pragma unroll
        for (int h = 0; h < AP_H; ++h) {
            const int src = ((h >> 2) & 1) * 16 + ((h >> 1) & 1) * 8 + (h & 1) * 4;
            ah[h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, av), src));
            th[h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tv), src));
        }
#pragma unroll
        for (int h = 0; h < AP_H; ++h) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(Vl + h * E), q1 = *reinterpret_cast<const f32x4*>(Vl + h * E + 256);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(Vl + (AP_H + h) * E), e1 = *reinterpret_cast<const f32x4*>(Vl + (AP_H + h) * E + 256);
            d[0] += ah[h] * e0 + th[h] * q0;
            d[1] += ah[h] * e1 + th[h] * q1;
            dq[h][0] += th[h] * kin[0];
            dq[h][1] += th[h] * kin[1];
        }
        if (!LN) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(dx + row * E + 256 * q + 4 * lane) = d[q];
            return;
        }
        f32x4 gg[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            gg[q] = d[q] * gm[q];
            dg[q] += d[q] * xh[q];
            db[q] += d[q];
        }
        const f32x4 t1 = gg[0] + gg[1], t2 = gg[0] * xh[0] + gg[1] * xh[1];
        const float s1 = wave_allsum((t1[0] + t1[1]) + (t1[2] + t1[3])) / E;
        const float s2 = wave_allsum((t2[0] + t2[1]) + (t2[2] + t2[3])) / E;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 v = rstd * (gg[q] - s1 - xh[q] * s2);
            *reinterpret_cast<f32x4*>(dx + row * E + 256 * q + 4 * lane) = v;
            d0[q] += v;
        }
    };
    for (int rr = wave; rr < nkeys; rr += 8) {
        one_row(rr, xa, da_, pa);
        if (rr + 4 < nkeys) one_row(rr + 4, xb, db_, pb);
    }
    // fold of the four waves: dgamma / dbeta / do through red3; dQp through the 32 KB of V in two rounds (3, 2 -> 1, 0; 1 -> 0)
    if (LN && wave > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            *reinterpret_cast<f32x4*>(&red3[wave - 1][0][256 * q + 4 * lane]) = dg[q];
            *reinterpret_cast<f32x4*>(&red3[wave - 1][1][256 * q + 4 * lane]) = db[q];
            *reinterpret_cast<f32x4*>(&red3[wave - 1][2][256 * q + 4 * lane]) = d0[q];
        }
    }
    __syncthreads();                                                       // every wave is done with V
    if (wave >= 2) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(V + ((wave - 2) * AP_H + h) * E + 256 * q + 4 * lane) = dq[h][q];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) dq[h][q] += *reinterpret_cast<const f32x4*>(V + (wave * AP_H + h) * E + 256 * q + 4 * lane);
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(V + h * E + 256 * q + 4 * lane) = dq[h][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int h = 0; h < AP_H; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4 v = dq[h][q] + *reinterpret_cast<const f32x4*>(V + h * E + 256 * q + 4 * lane);
                *reinterpret_cast<f32x4*>(pdq + ((size_t)g * AP_H + h) * E + 256 * q + 4 * lane) = v * scale;
            }
        if (LN) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    dg[q] += *reinterpret_cast<const f32x4*>(&red3[w][0][256 * q + 4 * lane]);
                    db[q] += *reinterpret_cast<const f32x4*>(&red3[w][1][256 * q + 4 * lane]);
                    d0[q] += *reinterpret_cast<const f32x4*>(&red3[w][2][256 * q + 4 * lane]);
                }
                *reinterpret_cast<f32x4*>(part + ((size_t)g * 3 + 0) * E + 256 * q + 4 * lane) = dg[q];
                *reinterpret_cast<f32x4*>(part + ((size_t)g * 3 + 1) * E + 256 * q + 4 * lane) = db[q];
                *reinterpret_cast<f32x4*>(part + ((size_t)g * 3 + 2) * E + 256 * q + 4 * lane) = d0[q];
            }
        }
    }
}

// Fold of the per-tile partials of k_lnbr_bwd_r16 and of k_apool_bwd_apply in ONE launch; every sum in a fixed order.  A job =
// one output vector of E floats; it is split over 4 workgroups (128 columns each) of 1024 threads = 32 tile groups x 32 column
// threads (16-byte columns), group q sums the tiles g0 + q, g0 + q + 32, ... four loads in flight - the walk over the tiles
// is pure latency: one workgroup per vector and 8 groups took 11 us for the 512 tiles of dgamma, this form 16 trips of 4.
//   jobs [0, 2):              dgamma / dbeta = sum over the real tiles [0, bag_tile_off[B])
//   jobs [2, 2 + B):          do[bag]        = sum over the bag's tiles
//   jobs [2 + B, 2 + B + BH): dQp[bag][h]    = sum over the bag's tiles            (k_apool_bwd_merge's sum)
__global__ __launch_bounds__(1024) void k_lnbr_apool_fold(const float* __restrict__ part, const float* __restrict__ pdq,
                                                          const int32_t* __restrict__ bag_tile_off, int B,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ d_o, float* __restrict__ dQp) {
    constexpr int E = 512;
    __shared__ __attribute__((aligned(16))) float red[32][128];
    const int j4 = threadIdx.x & 31, q = threadIdx.x >> 5, job = blockIdx.x >> 2, c0 = (blockIdx.x & 3) * 128 + 4 * j4;
    const float* src;
    size_t stride;
    int g0, g1;
    float* dst;
    if (job < 2) {
        src = part + (size_t)job * E; stride = (size_t)3 * E; g0 = 0; g1 = bag_tile_off[B]; dst = job == 0 ? dgamma : dbeta;
    } else if (job < 2 + B) {
        const int bag = job - 2;
        src = part + (size_t)2 * E; stride = (size_t)3 * E; g0 = bag_tile_off[bag]; g1 = bag_tile_off[bag + 1]; dst = d_o + (size_t)bag * E;
    } else {
        const int bh = job - 2 - B, bag = bh / AP_H, h = bh % AP_H;
        src = pdq + (size_t)h * E; stride = (size_t)AP_H * E; g0 = bag_tile_off[bag]; g1 = bag_tile_off[bag + 1];
        dst = dQp + (size_t)bh * E;
    }
    f32x4 acc = {0, 0, 0, 0};
    int g = g0 + q;
    for (; g + 96 < g1; g += 128) {
        f32x4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(g + 32 * u) * stride + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += t[u];
    }
    for (; g < g1; g += 32) acc += *reinterpret_cast<const f32x4*>(src + (size_t)g * stride + c0);
    *reinterpret_cast<f32x4*>(&red[q][4 * j4]) = acc;
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int u = 1; u < 32; ++u) acc += *reinterpret_cast<const f32x4*>(&red[u][4 * j4]);
        *reinterpret_cast<f32x4*>(dst + c0) = acc;
    }
}

// ---------------------------------------------------------------------------------------------- value projection
// o[b][hc + c'] = Wv[hc + c'] . pooled[b][h] + bv[hc + c'];   grid (B, H), 256 threads
__global__ __launch_bounds__(256) void k_value_proj(const float* __restrict__ pooled, const float* __restrict__ Wv,
                                                    const float* __restrict__ bv, int H, int C, int E,
                                                    float* __restrict__ o, int T, int THp) {
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, I = H * C;
    const int per = 256 / C, c = tid / per, part = tid % per;
    const float* pv = pooled + ((size_t)(b / T) * THp + (b % T) * H + h) * E;
    const float* w = Wv + (size_t)(h * C + c) * E;
    float v = 0.f;
    for (int j0 = 4 * part; j0 < E; j0 += 16 * per) {
        f32x4 gv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + 4 * per * u, E - 4);
            gv[u] = *reinterpret_cast<const f32x4*>(pv + j);
            wv[u] = *reinterpret_cast<const f32x4*>(w + j);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + 4 * per * u < E) v += gv[u][0] * wv[u][0] + gv[u][1] * wv[u][1] + gv[u][2] * wv[u][2] + gv[u][3] * wv[u][3];
    }
    for (int m = per >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (part == 0) o[(size_t)b * I + h * C + c] = v + bv[h * C + c];
}

// dpooled[b][h][:] = sum_c' do[b][hc + c'] Wv[hc + c'][:]     (same form as k_absorb_query)
// dWv[hc + c'][:]   = sum_b do[b][hc + c'] pooled[b][h][:]    (same form as k_absorb_query_bwd_w)

// One launch for both halves of the absorbed query's backward (they only share their inputs): workgroups [0, B H) form
// dqp (k_absorb_query_bwd_q's arithmetic on the first 256 threads), workgroups behind them one row of dWk each
// (k_absorb_query_bwd_w).  The token-side chain is a string of ~4 us launches whose cost is the launch itself.
__global__ __launch_bounds__(512) void k_absorb_query_bwd_both(const float* __restrict__ qp, const float* __restrict__ Wk,
                                                               const float* __restrict__ dQp, int B, int H, int C, int E,
                                                               float* __restrict__ dqp, float* __restrict__ dWk, int T,
                                                               int THp, float scale, const float* __restrict__ bias,
                                                               const float* __restrict__ dcb, float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) float red[(AQ_GROUPS - 1) * 1024];
    __shared__ float redb[AQ_GROUPS];
    const int nq = B * H, tid = threadIdx.x, I = H * C;
    if ((int)blockIdx.x < nq) {
        if (tid >= 256) return;
        const int b = blockIdx.x / H, h = blockIdx.x % H;
        const int per = 256 / C, c = tid / per, part = tid % per;
        const float* g = dQp + ((size_t)(b / T) * THp + (b % T) * H + h) * E;
        const float* w = Wk + (size_t)(h * C + c) * E;
        float v = 0.f;
        for (int j0 = 4 * part; j0 < E; j0 += 16 * per) {
            f32x4 gv[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + 4 * per * u, E - 4);
                gv[u] = *reinterpret_cast<const f32x4*>(g + j);
                wv[u] = *reinterpret_cast<const f32x4*>(w + j);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + 4 * per * u < E) v += gv[u][0] * wv[u][0] + gv[u][1] * wv[u][1] + gv[u][2] * wv[u][2] + gv[u][3] * wv[u][3];
        }
        for (int m = per >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (part == 0) {
            if (dcb != nullptr) v += dcb[(size_t)(b / T) * THp + (b % T) * H + h] * bias[h * C + c];
            dqp[(size_t)b * I + h * C + c] = v * scale;
        }
        return;
    }
    const int E4 = E / 4;
    const int row = blockIdx.x - nq, h = row / C, j4 = tid % E4, grp = tid / E4;
    f32x4 acc = {0, 0, 0, 0};
    float bacc = 0.f;
    if (grp < AQ_GROUPS)
        for (int b0 = 8 * grp; b0 < B; b0 += 8 * AQ_GROUPS) {
            f32x4 gv[8];
            float q[8], dc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = min(b0 + u, B - 1);
                const size_t orow = (size_t)(b / T) * THp + (b % T) * H + h;
                gv[u] = *reinterpret_cast<const f32x4*>(dQp + orow * E + 4 * j4);
                q[u] = b0 + u < B ? qp[(size_t)b * I + row] : 0.f;
                dc[u] = dcb != nullptr && j4 == 0 ? dcb[orow] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += q[u] * gv[u]; bacc += q[u] * dc[u]; }
        }
    if (grp > 0 && grp < AQ_GROUPS) *reinterpret_cast<f32x4*>(red + (grp - 1) * 1024 + 4 * j4) = acc;
    if (j4 == 0 && grp < AQ_GROUPS) redb[grp] = bacc;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int g = 0; g < AQ_GROUPS - 1; ++g) acc += *reinterpret_cast<const f32x4*>(red + g * 1024 + 4 * j4);
        *reinterpret_cast<f32x4*>(dWk + (size_t)row * E + 4 * j4) = acc * scale;
        if (dbias != nullptr && j4 == 0) dbias[row] = ((redb[0] + redb[1]) + (redb[2] + redb[3])) * scale;
    }
}

// The value projection's whole backward in one launch (three results that only share `do`):
//   workgroups [0, B H):           dpooled[b][h][:] = sum_c' do[b][hc + c'] Wv[hc + c'][:]      (k_absorb_query's form)
//   workgroups [B H, B H + H C):   dWv[row][:]      = sum_b do[b][row] pooled[b][h][:]         (k_absorb_query_bwd_w's form)
//   last workgroup:                dbv[i]           = sum_b do[b][i]
// 512 threads, E = 512.
__global__ __launch_bounds__(512) void k_value_proj_bwd(const float* __restrict__ dO, const float* __restrict__ Wv,
                                                        const float* __restrict__ pooled, int B, int H, int C, int E,
                                                        float* __restrict__ dpooled, float* __restrict__ dWv,
                                                        float* __restrict__ dbv) {
    __shared__ __attribute__((aligned(16))) float red[(AQ_GROUPS - 1) * 1024];
    const int nq = B * H, nw = H * C, tid = threadIdx.x, I = H * C, E4 = E / 4;
    if ((int)blockIdx.x < nq) {
        if (tid >= E4) return;
        const int b = blockIdx.x / H, h = blockIdx.x % H, j4 = tid;
        f32x4 acc = {0, 0, 0, 0};
        for (int c0 = 0; c0 < C; c0 += 16) {
            f32x4 wr[16];
            float q[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                wr[u] = *reinterpret_cast<const f32x4*>(Wv + (size_t)(h * C + c0 + u) * E + 4 * j4);
                q[u] = dO[(size_t)b * I + h * C + c0 + u];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += q[u] * wr[u];
        }
        *reinterpret_cast<f32x4*>(dpooled + ((size_t)b * H + h) * E + 4 * j4) = acc;
        return;
    }
    if ((int)blockIdx.x < nq + nw) {
        const int row = blockIdx.x - nq, h = row / C, j4 = tid % E4, grp = tid / E4;
        f32x4 acc = {0, 0, 0, 0};
        if (grp < AQ_GROUPS)
            for (int b0 = 8 * grp; b0 < B; b0 += 8 * AQ_GROUPS) {
                f32x4 gv[8];
                float q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = min(b0 + u, B - 1);
                    gv[u] = *reinterpret_cast<const f32x4*>(pooled + ((size_t)b * H + h) * E + 4 * j4);
                    q[u] = b0 + u < B ? dO[(size_t)b * I + row] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += q[u] * gv[u];
            }
        if (grp > 0 && grp < AQ_GROUPS) *reinterpret_cast<f32x4*>(red + (grp - 1) * 1024 + 4 * j4) = acc;
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 0; g < AQ_GROUPS - 1; ++g) acc += *reinterpret_cast<const f32x4*>(red + g * 1024 + 4 * j4);
            *reinterpret_cast<f32x4*>(dWv + (size_t)row * E + 4 * j4) = acc;
        }
        return;
    }
    if (dbv != nullptr)
        for (int i = tid; i < I; i += 512) {
            float v = 0.f;
            int b = 0;
            for (; b + 8 <= B; b += 8) {                    // eight rows in flight (the caller keeps B <= 64 on this path)
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = dO[(size_t)(b + e) * I + i];
#pragma unroll
                for (int e = 0; e < 8; ++e) v += t[e];
            }
            for (; b < B; ++b) v += dO[(size_t)b * I + i];
            dbv[i] = v;
        }
}

// k_apool_merge followed by k_value_proj in one launch: grid (B, H), 1024 threads = 8 groups x E / 4 column threads.
// Group q merges the bag's tiles g0 + q, g0 + q + 8, ... against the bag-wide maximum (formed first by all threads), the
// eight partial sums are folded through LDS in group order; pooled[b][h] goes to global memory (the backward needs it) and
// stays in LDS for the head's C outputs of the value projection (first 256 threads).  With ONE long bag per batch (the
// authors' regime: ~200 tiles) the single-group form walked every tile in turn: 31 us per attention site.
#define AMV_G 8
__global__ __launch_bounds__(1024) void k_apool_merge_value(const float* __restrict__ pacc, const float* __restrict__ pml,
                                                            const int32_t* __restrict__ bag_tile_off, int E,
                                                            float* __restrict__ pooled, float* __restrict__ lse,
                                                            const float* __restrict__ Wv, const float* __restrict__ bv, int C,
                                                            float* __restrict__ o) {
    __shared__ __attribute__((aligned(16))) float pl[AMV_G][512];
    __shared__ float lred[AMV_G], mred[16];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, H = AP_H, I = H * C;
    const int j4 = tid & 127, q = tid >> 7;                   // E == 512: 128 column threads per group
    const int g0 = bag_tile_off[b], g1 = bag_tile_off[b + 1];
    // the value projection's weights (first 256 threads: `per` threads per output, 512 / per floats each) depend on nothing:
    // requested now, they arrive under the merge - behind the last barrier they were four more round trips
    const int per = 256 / C, vc = (tid & 255) / per, vpart = (tid & 255) % per;
    f32x4 wvr[16];                                             // C = 32: 16 pieces of 16 bytes per thread, C = 64: 8
    if (tid < 256) {
        const float* w = Wv + (size_t)(h * C + vc) * E;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = 4 * vpart + 4 * per * u;
            wvr[u] = j < E ? *reinterpret_cast<const f32x4*>(w + j) : f32x4{0, 0, 0, 0};
        }
    }
    float m = -INFINITY;
    for (int g = g0 + tid; g < g1; g += 1024) m = fmaxf(m, pml[((size_t)g * AP_H + h) * 2]);
    m = wave_allmax(m);
    if ((tid & 63) == 0) mred[tid >> 6] = m;
    __syncthreads();
    m = mred[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, mred[w]);
    float l = 0.f;
    f32x4 acc = {0, 0, 0, 0};
    {
        int g = g0 + q;
        for (; g + 3 * AMV_G < g1; g += 4 * AMV_G) {          // four tiles of this group in flight
            float2 ml[4];
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ml[u] = *reinterpret_cast<const float2*>(pml + ((size_t)(g + AMV_G * u) * AP_H + h) * 2);
                t[u] = *reinterpret_cast<const f32x4*>(pacc + ((size_t)(g + AMV_G * u) * AP_H + h) * E + 4 * j4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sc = __expf(ml[u].x - m);
                l += sc * ml[u].y;
                acc += sc * t[u];
            }
        }
        for (; g < g1; g += AMV_G) {
            const float sc = __expf(pml[((size_t)g * AP_H + h) * 2] - m);
            l += sc * pml[((size_t)g * AP_H + h) * 2 + 1];
            acc += sc * *reinterpret_cast<const f32x4*>(pacc + ((size_t)g * AP_H + h) * E + 4 * j4);
        }
    }
    *reinterpret_cast<f32x4*>(&pl[q][4 * j4]) = acc;
    if (j4 == 0) lred[q] = l;
    __syncthreads();
    if (q == 0) {                                              // pl[0] is this group's own slot: nobody else reads it
        f32x4 v = acc;
        float lt = lred[0];
#pragma unroll
        for (int u = 1; u < AMV_G; ++u) { v += *reinterpret_cast<const f32x4*>(&pl[u][4 * j4]); lt += lred[u]; }
        const float inv = g1 > g0 ? 1.0f / lt : 0.f;
        v = v * inv;
        *reinterpret_cast<f32x4*>(pooled + ((size_t)b * AP_H + h) * E + 4 * j4) = v;
        if (j4 == 0) lse[b * AP_H + h] = g1 > g0 ? m + logf(lt) : -INFINITY;
        *reinterpret_cast<f32x4*>(&pl[0][4 * j4]) = v;
    }
    __syncthreads();
    if (tid >= 256) return;
    const int c = vc, part = vpart;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {                              // same pieces in the same order as the four-at-a-time loop
        const int j = 4 * part + 4 * per * u;
        if (j < E) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(&pl[0][j]);
            v += gv[0] * wvr[u][0] + gv[1] * wvr[u][1] + gv[2] * wvr[u][2] + gv[3] * wvr[u][3];
        }
    }
    if (4 * per * 16 < E) {                                     // C = 64: the second half of the row, one batch
        const float* w = Wv + (size_t)(h * C + c) * E;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = 4 * part + 4 * per * (16 + u);
            wvr[u] = j < E ? *reinterpret_cast<const f32x4*>(w + j) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = 4 * part + 4 * per * (16 + u);
            if (j < E) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(&pl[0][j]);
                v += gv[0] * wvr[u][0] + gv[1] * wvr[u][1] + gv[2] * wvr[u][2] + gv[3] * wvr[u][3];
            }
        }
    }
    for (int m2 = per >> 1; m2 >= 1; m2 >>= 1) v += __shfl_xor(v, m2);
    if (part == 0) o[(size_t)b * I + h * C + c] = v + bv[h * C + c];
}

// ---------------------------------------------------------------------------------------------- multi-token absorbed attention
// With T text tokens per bag the same absorption yields T x H query (or key / value) vectors per bag and the image side
// becomes skinny grouped products (mil_gemm_grouped) around a softmax on a [rows, T*H] score matrix (column c = t H + h,
// padded to a multiple of 32 with zeros):
//   token -> image: softmax over the ROWS (patches) of a bag, per column         k_grp_col_softmax / _bwd
//   image -> token: softmax over the T tokens of a row, per head                k_row_softmax_t / _bwd
// Column softmax: grid (G, ceil(ld / 32)), 1024 threads = 32 row lanes x 32 columns; a thread keeps its <= 64 values of
// the column in registers (groups of up to 2048 rows: one read of the slab), longer groups fall back to re-reading.
// Workgroup = 1024 threads = CW columns x (1024 / CW) row lanes; a thread keeps its rows of the column in registers
// (CS_KEEP of them), so a group of up to (1024 / CW) * CS_KEEP rows is read once.  CW = 32 covers 2048 rows (bench bags),
// CW = 8 8192 and CW = 2 32768 (the authors' bags reach ~15 000 patches); longer groups take the re-reading loop.
#define CS_KEEP 64

// reduce over the row lanes of a column: lanes of a wave that share (lane % CW), then the 16 waves through LDS
template <int CW, bool MAX>
__device__ __forceinline__ float cs_reduce(float v, float (*red)[33], int cl) {
    if (CW < 64) {
#pragma unroll
        for (int o = CW; o < 64; o <<= 1) {
            const float t = __shfl_xor(v, o);
            v = MAX ? fmaxf(v, t) : v + t;
        }
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                    // the previous use of red is over
    if ((threadIdx.x & 63) < CW) red[wave][cl] = v;
    __syncthreads();
    float r = red[0][cl];
#pragma unroll
    for (int i = 1; i < 16; ++i) r = MAX ? fmaxf(r, red[i][cl]) : r + red[i][cl];
    return r;
}

template <int CW>
__global__ __launch_bounds__(1024) void k_grp_col_softmax(float* __restrict__ S, int ld, const int32_t* __restrict__ grp_off,
                                                          int TH) {
    constexpr int RL = 1024 / CW;
    __shared__ float red[16][33];
    const int g = blockIdx.x, cl = threadIdx.x % CW, c = blockIdx.y * CW + cl, rl = threadIdx.x / CW;
    const int r0 = grp_off[g], r1 = grp_off[g + 1], n = r1 - r0;
    const bool live = c < TH && c < ld;
    const bool fits = n <= RL * CS_KEEP;
    float v[CS_KEEP];
    float m = -INFINITY;
    if (live) {
        if (fits) {
#pragma unroll
            for (int i = 0; i < CS_KEEP; ++i) {
                const int row = r0 + rl + RL * i;
                v[i] = row < r1 ? S[(size_t)row * ld + c] : -INFINITY;
                m = fmaxf(m, v[i]);
            }
        } else {
            for (int row = r0 + rl; row < r1; row += RL) m = fmaxf(m, S[(size_t)row * ld + c]);
        }
    }
    m = cs_reduce<CW, true>(m, red, cl);
    float l = 0.f;
    if (live) {
        if (fits) {
#pragma unroll
            for (int i = 0; i < CS_KEEP; ++i) { v[i] = expf(v[i] - m); l += v[i]; }      // exp(-inf) = 0 for the padding slots
        } else {
            for (int row = r0 + rl; row < r1; row += RL) l += expf(S[(size_t)row * ld + c] - m);
        }
    }
    l = cs_reduce<CW, false>(l, red, cl);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (c >= ld) return;
    if (live && fits) {
#pragma unroll
        for (int i = 0; i < CS_KEEP; ++i) {
            const int row = r0 + rl + RL * i;
            if (row < r1) S[(size_t)row * ld + c] = v[i] * inv;
        }
    } else {
        for (int row = r0 + rl; row < r1; row += RL) {
            float* p = S + (size_t)row * ld + c;
            *p = live ? expf(*p - m) * inv : 0.f;
        }
    }
}

// dS = A (dA - sum_rows A dA)  per (group, column); columns >= TH get 0
template <int CW>
__global__ __launch_bounds__(1024) void k_grp_col_softmax_bwd(const float* __restrict__ A, const float* __restrict__ dA, int ld,
                                                              const int32_t* __restrict__ grp_off, int TH,
                                                              float* __restrict__ dS) {
    constexpr int RL = 1024 / CW;
    __shared__ float red[16][33];
    const int g = blockIdx.x, cl = threadIdx.x % CW, c = blockIdx.y * CW + cl, rl = threadIdx.x / CW;
    const int r0 = grp_off[g], r1 = grp_off[g + 1], n = r1 - r0;
    const bool live = c < TH && c < ld;
    const bool fits = n <= RL * (CS_KEEP / 2);
    float a[CS_KEEP / 2], d[CS_KEEP / 2];
    float cd = 0.f;
    if (live) {
        if (fits) {
#pragma unroll
            for (int i = 0; i < CS_KEEP / 2; ++i) {
                const int row = r0 + rl + RL * i;
                a[i] = row < r1 ? A[(size_t)row * ld + c] : 0.f;
                d[i] = row < r1 ? dA[(size_t)row * ld + c] : 0.f;
                cd += a[i] * d[i];
            }
        } else {
            for (int row = r0 + rl; row < r1; row += RL) cd += A[(size_t)row * ld + c] * dA[(size_t)row * ld + c];
        }
    }
    cd = cs_reduce<CW, false>(cd, red, cl);
    if (c >= ld) return;
    if (live && fits) {
#pragma unroll
        for (int i = 0; i < CS_KEEP / 2; ++i) {
            const int row = r0 + rl + RL * i;
            if (row < r1) dS[(size_t)row * ld + c] = a[i] * (d[i] - cd);
        }
    } else {
        for (int row = r0 + rl; row < r1; row += RL) {
            const size_t o = (size_t)row * ld + c;
            dS[o] = live ? A[o] * (dA[o] - cd) : 0.f;
        }
    }
}

// A FEW LONG groups (one ragged bag of ~10 000 patches per GPU and step): the forms above give a group's column to ONE
// workgroup, two columns wide for such a group - 48 workgroups reading 8 bytes of every 384-byte row (36 us forward, 51 us
// backward for 3.5 MB).  Row-parallel form in two launches, every row read as a whole by consecutive lanes:
//   k_gcs_stats  grid (G, NCH): workgroup (g, ch) = rows [ch RCH, (ch + 1) RCH) of group g, thread (row lane, column):
//                online column maximum / sum (forward) or sum of A dA (backward) -> ws[g][ch][.][ld]
//   k_gcs_apply  grid (G, NCH): folds the NCH partials of its group (same order everywhere), then rewrites its rows.
// ld <= 128.
#define GCS_NCH_MAX 64
template <bool BWD>
__global__ __launch_bounds__(1024) void k_gcs_stats(const float* __restrict__ S, const float* __restrict__ dA, int ld,
                                                    const int32_t* __restrict__ grp_off, int TH, int RCH,
                                                    float* __restrict__ ws) {
    __shared__ float red[2][32][129];
    const int g = blockIdx.x, ch = blockIdx.y, RL = 1024 / ld;
    const int c = threadIdx.x % ld, rl = threadIdx.x / ld;
    const int r0 = grp_off[g], r1 = grp_off[g + 1];
    const int a0 = r0 + ch * RCH, a1 = min(r1, a0 + RCH);
    float m = -INFINITY, l = 0.f;
    if (rl < RL && c < TH) {
        if (BWD) {
            for (int row = a0 + rl; row < a1; row += RL) l += S[(size_t)row * ld + c] * dA[(size_t)row * ld + c];
        } else {
            for (int row = a0 + rl; row < a1; row += RL) {
                const float v = S[(size_t)row * ld + c];
                if (v > m) { l = l * expf(m - v) + 1.f; m = v; }          // exp(-inf - v) = 0 on the first row
                else l += expf(v - m);
            }
        }
    }
    if (rl < RL) { red[0][rl][c] = m; red[1][rl][c] = l; }
    __syncthreads();
    if (rl == 0 && c < ld) {
        float M = -INFINITY, L = 0.f;
        if (BWD) {
            for (int i = 0; i < RL; ++i) L += red[1][i][c];
        } else {
            for (int i = 0; i < RL; ++i) M = fmaxf(M, red[0][i][c]);
            for (int i = 0; i < RL; ++i) L += red[0][i][c] > -INFINITY ? red[1][i][c] * expf(red[0][i][c] - M) : 0.f;
        }
        float* o = ws + ((size_t)g * gridDim.y + ch) * 2 * ld;
        o[c] = M;
        o[ld + c] = L;
    }
}

template <bool BWD>
__global__ __launch_bounds__(1024) void k_gcs_apply(float* __restrict__ S, const float* __restrict__ A, const float* __restrict__ dA,
                                                    int ld, const int32_t* __restrict__ grp_off, int TH, int RCH,
                                                    const float* __restrict__ ws) {
    __shared__ float cm[128], cl_[128];
    __shared__ float red[2][32][129];
    const int g = blockIdx.x, ch = blockIdx.y, RL = 1024 / ld, NCH = gridDim.y;
    const int c = threadIdx.x % ld, rl = threadIdx.x / ld;
    const int r0 = grp_off[g], r1 = grp_off[g + 1];
    const int a0 = r0 + ch * RCH, a1 = min(r1, a0 + RCH);
    if (a0 >= a1) return;                                   // workgroup-uniform: no rows here
    // fold the group's NCH partial statistics: row lane rl takes partials rl, rl + RL, ... (a single lane walking all of them
    // was a chain of ~100 dependent round trips: 31 us per launch), then the lanes are folded in a fixed order
    if (rl < RL) {
        const float* w = ws + (size_t)g * NCH * 2 * ld;
        float M = -INFINITY, L = 0.f;
        for (int i = rl; i < NCH; i += RL) {
            const float mi = w[(size_t)i * 2 * ld + c], li = w[(size_t)i * 2 * ld + ld + c];
            if (BWD) {
                L += li;
            } else if (mi > -INFINITY) {
                if (mi > M) { L = L * expf(M - mi) + li; M = mi; }
                else L += li * expf(mi - M);
            }
        }
        red[0][rl][c] = M;
        red[1][rl][c] = L;
    }
    __syncthreads();
    if (rl == 0) {
        float M = -INFINITY, L = 0.f;
        if (BWD) {
            for (int i = 0; i < RL; ++i) L += red[1][i][c];
        } else {
            for (int i = 0; i < RL; ++i) M = fmaxf(M, red[0][i][c]);
            for (int i = 0; i < RL; ++i) L += red[0][i][c] > -INFINITY ? red[1][i][c] * expf(red[0][i][c] - M) : 0.f;
        }
        cm[c] = M;
        cl_[c] = BWD ? L : (L > 0.f ? 1.0f / L : 0.f);
    }
    __syncthreads();
    if (rl >= RL) return;
    const bool live = c < TH;
    const float M = cm[c], L = cl_[c];
    for (int row = a0 + rl; row < a1; row += RL) {
        const size_t o = (size_t)row * ld + c;
        if (BWD) S[o] = live ? A[o] * (dA[o] - L) : 0.f;
        else S[o] = live ? expf(S[o] - M) * L : 0.f;
    }
}

static bool gcs_plan(int G, int max_group_rows, int ld, int* nch, int* rch) {
    if (G > 8 || max_group_rows <= 2048 || ld > 128 || ld < 32) return false;
    int n = (max_group_rows + 255) / 256;
    if (n > GCS_NCH_MAX) n = GCS_NCH_MAX;
    *nch = n;
    *rch = (max_group_rows + n - 1) / n;
    return true;
}
extern "C" size_t mil_grp_col_softmax_workspace_floats(int G, int max_group_rows, int ld) {
    int nch, rch;
    return gcs_plan(G, max_group_rows, ld, &nch, &rch) ? (size_t)G * nch * 2 * ld : 0;
}

// Row softmax over the T tokens of each (row, head): column t H + h.  Thread = (row, h); T <= 16.  In place.
__global__ __launch_bounds__(256) void k_row_softmax_t(float* __restrict__ S, int ld, int R, int T, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= R * H) return;
    const int row = idx / H, h = idx % H;
    float* p = S + (size_t)row * ld + h;
    float v[16], m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 16; ++t) { v[t] = t < T ? p[t * H] : -INFINITY; m = fmaxf(m, v[t]); }
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { v[t] = t < T ? expf(v[t] - m) : 0.f; l += v[t]; }
    const float inv = 1.0f / l;
#pragma unroll
    for (int t = 0; t < 16; ++t) if (t < T) p[t * H] = v[t] * inv;
    for (int c = T * H + h; c < ld; c += H) S[(size_t)row * ld + c] = 0.f;        // padding columns
}

__global__ __launch_bounds__(256) void k_row_softmax_t_bwd(const float* __restrict__ A, const float* __restrict__ dA, int ld,
                                                           int R, int T, int H, float* __restrict__ dS) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= R * H) return;
    const int row = idx / H, h = idx % H;
    const size_t base = (size_t)row * ld + h;
    float a[16], d[16], cd = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        a[t] = t < T ? A[base + t * H] : 0.f;
        d[t] = t < T ? dA[base + t * H] : 0.f;
        cd += a[t] * d[t];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) if (t < T) dS[base + t * H] = a[t] * (d[t] - cd);
    for (int c = T * H + h; c < ld; c += H) dS[(size_t)row * ld + c] = 0.f;
}

#define AP_CHECK(cond) do { if (!(cond)) return MIL_EINVAL; } while (0)

// T text tokens per group, output rows in the grouped products' padded layout [B / T, THp, E] (THp >= T H: rows T H .. THp - 1
// of every group written as zeros), result scaled by `scale`.  T = 1, THp = H, scale = 1: the plain [B, H, E].
// bias [H C] / cb [B / T, THp] (both or neither): cb = scale * bias_h . qp[b][h] in the same layout.
extern "C" int mil_absorb_query_pad(const float* qp, const float* Wk, int B, int H, int C, int E, int T, int THp, float scale,
                                    const float* bias, float* Qp, float* cb, void* stream) {
    AP_CHECK(qp && Wk && Qp && B >= 0 && H > 0 && C > 0 && (C % 16) == 0 && E > 0 && (E & 3) == 0 && E <= 4096);
    AP_CHECK(T >= 1 && THp >= T * H && (B % T) == 0 && (bias != nullptr) == (cb != nullptr));
    if (B == 0) return MIL_OK;
    const int pad_blocks = THp > T * H ? B / T : 0;
    hipLaunchKernelGGL(k_absorb_query, dim3(B + pad_blocks, H), dim3(E / 4), 0, (hipStream_t)stream, qp, Wk, H, C, E, Qp, B, T, THp,
                       scale, bias, cb);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
extern "C" int mil_absorb_query(const float* qp, const float* Wk, int B, int H, int C, int E, float* Qp, void* stream) {
    return mil_absorb_query_pad(qp, Wk, B, H, C, E, 1, H, 1.0f, nullptr, Qp, nullptr, stream);
}

// dcb [B / T, THp] (the gradient of cb; with bias) adds scale * dcb bias to dqp and yields dbias [H C] (with dWk).
extern "C" int mil_absorb_query_bwd_pad(const float* qp, const float* Wk, const float* dQp, int B, int H, int C, int E, int T,
                                        int THp, float scale, const float* bias, const float* dcb, float* dqp, float* dWk,
                                        float* dbias, void* stream) {
    AP_CHECK(qp && Wk && dQp && B > 0 && H > 0 && (C == 32 || C == 64) && E > 0 && (E & 3) == 0 && E <= 1024);
    AP_CHECK(T >= 1 && THp >= T * H && (B % T) == 0);
    AP_CHECK((dcb == nullptr || bias != nullptr) && (dbias == nullptr || (dcb != nullptr && dWk != nullptr)));
    hipStream_t st = (hipStream_t)stream;
    if (dqp != nullptr && dWk != nullptr && E == 512) {          // both halves in one launch
        hipLaunchKernelGGL(k_absorb_query_bwd_both, dim3(B * H + H * C), dim3(512), 0, st, qp, Wk, dQp, B, H, C, E, dqp, dWk, T, THp,
                           scale, bias, dcb, dbias);
        MIL_CHECK_LAUNCH();
        return MIL_OK;
    }
    if (dqp != nullptr) {
        hipLaunchKernelGGL(k_absorb_query_bwd_q, dim3(B, H), dim3(256), 0, st, dQp, Wk, H, C, E, dqp, T, THp, scale, bias, dcb);
        MIL_CHECK_LAUNCH();
    }
    if (dWk != nullptr) {
        hipLaunchKernelGGL(k_absorb_query_bwd_w, dim3(H * C), dim3(AQ_GROUPS * (E / 4)), 0, st, qp, dQp, B, H, C, E, dWk, T, THp, scale,
                           dcb, dbias);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}
extern "C" int mil_absorb_query_bwd(const float* qp, const float* Wk, const float* dQp, int B, int H, int C, int E,
                                    float* dqp, float* dWk, void* stream) {
    return mil_absorb_query_bwd_pad(qp, Wk, dQp, B, H, C, E, 1, H, 1.0f, nullptr, nullptr, dqp, dWk, nullptr, stream);
}

extern "C" int mil_absorbed_pool_fwd(const float* keys, const float* pe, const float* Qp, const int32_t* k_off,
                                     const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles, int B, int H,
                                     int C, int E, float* pooled, float* lse, float* workspace, void* stream) {
    AP_CHECK(keys && pe && Qp && k_off && tile_map && bag_tile_off && pooled && lse && workspace);
    AP_CHECK(H == AP_H && E == 512 && C > 0 && B >= 0 && ntiles >= 0);
    if (B == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    float* pacc = workspace;
    float* pml = workspace + (size_t)ntiles * AP_H * E;
    const float scale = 1.0f / sqrtf((float)C);
    if (ntiles > 0) {
        hipLaunchKernelGGL(k_apool_partial<false>, dim3(ntiles), dim3(256), 0, st, keys, pe, Qp, k_off, tile_map, scale, pacc, pml,
                           LnbrFwd{});
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_apool_merge, dim3(B, AP_H), dim3(E / 4), 0, st, pacc, pml, bag_tile_off, E, pooled, lse);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// mil_absorbed_pool_fwd with the value projection (o = Wv pooled + bv, model/sam/transformer.py:441-448 with the
// projections absorbed) formed by the merge launch itself.
extern "C" int mil_absorbed_pool_value_fwd(const float* keys, const float* pe, const float* Qp, const int32_t* k_off,
                                           const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles, int B, int H,
                                           int C, int E, const float* Wv, const float* bv, float* pooled, float* lse, float* o,
                                           float* workspace, void* stream) {
    AP_CHECK(keys && pe && Qp && k_off && tile_map && bag_tile_off && pooled && lse && workspace && Wv && bv && o);
    AP_CHECK(H == AP_H && E == 512 && (C == 32 || C == 64) && B >= 0 && ntiles >= 0);
    if (B == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    float* pacc = workspace;
    float* pml = workspace + (size_t)ntiles * AP_H * E;
    const float scale = 1.0f / sqrtf((float)C);
    if (ntiles > 0) {
        hipLaunchKernelGGL(k_apool_partial<false>, dim3(ntiles), dim3(256), 0, st, keys, pe, Qp, k_off, tile_map, scale, pacc, pml,
                           LnbrFwd{});
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_apool_merge_value, dim3(B, AP_H), dim3(1024), 0, st, pacc, pml, bag_tile_off, E, pooled, lse, Wv, bv, C, o);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// The value projection's backward in one launch: dpooled [B, H, E], dWv [H C, E], dbv [H C] (nullable) from do [B, H C].
extern "C" int mil_value_proj_bwd(const float* dO, const float* Wv, const float* pooled, int B, int H, int C, int E,
                                  float* dpooled, float* dWv, float* dbv, void* stream) {
    AP_CHECK(dO && Wv && pooled && dpooled && dWv && B > 0 && H > 0 && (C == 32 || C == 64) && E == 512);
    hipLaunchKernelGGL(k_value_proj_bwd, dim3(B * H + H * C + 1), dim3(512), 0, (hipStream_t)stream, dO, Wv, pooled, B, H, C, E,
                       dpooled, dWv, dbv);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

static bool lnbr_bwd_one_pass() {
    const char* e = getenv("MIL_LNBR_BWD");            // read per call (a call is a capture-time event under a hipGraph)
    return e == nullptr || strcmp(e, "r16") != 0;
}

extern "C" int mil_absorbed_pool_bwd(const float* keys, const float* pe, const float* Qp, const float* lse,
                                     const float* dpooled, const float* pooled, const int32_t* k_off,
                                     const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles, int n_keys, int B,
                                     int H, int C, int E, const float* dkeys_acc, float* dkeys, float* dQp,
                                     float* workspace, void* stream) {
    AP_CHECK(keys && pe && Qp && lse && dpooled && pooled && k_off && tile_map && bag_tile_off && dkeys && dQp && workspace);
    AP_CHECK(H == AP_H && E == 512 && C > 0 && B >= 0 && ntiles >= 0 && n_keys >= 0);
    if (B == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)C);
    float* pdq = workspace;                                    // [ntiles][H][E]
    float* ad = workspace + (size_t)ntiles * AP_H * E;         // [n_keys][16]
    if (ntiles > 0 && lnbr_bwd_one_pass()) {
        // one pass over the rows (round 4; MIL_LNBR_BWD=r16 keeps the dots + apply pair): keys, dkeys_acc in, dkeys out
        hipLaunchKernelGGL(k_lnbr_apool_bwd_one<false>, dim3(ntiles), dim3(256), 0, st, keys, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, dkeys_acc, pe, Qp, dpooled, pooled, lse, k_off, tile_map, scale,
                           dkeys, (float*)nullptr, pdq);
        MIL_CHECK_LAUNCH();
    } else if (ntiles > 0) {
        hipLaunchKernelGGL(k_apool_dots, dim3(ntiles), dim3(256), 0, st, keys, pe, Qp, lse, dpooled, pooled, k_off, tile_map,
                           scale, ad);
        MIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_apool_bwd_apply<true>, dim3(ntiles), dim3(256), 0, st, keys, pe, Qp, dpooled, ad, k_off, tile_map,
                           dkeys_acc, dkeys, pdq);
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_apool_bwd_merge, dim3(B, AP_H), dim3(1024), 0, st, pdq, bag_tile_off, E, dQp);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// LayerNorm(x + o[bag]) + the next attention site's absorbed pool + value projection, forward: y [rows, E] (the keys, every
// row the tile map covers - padding rows zero), stats [rows, 2], pooled, lse, o_attn.  workspace: ntiles H (E + 2) floats.
extern "C" int mil_lnbr_absorbed_pool_value_fwd(const float* x, const float* o, const float* gamma, const float* beta, float eps,
                                                const float* pe, const float* Qp, const int32_t* k_off, const int32_t* tile_map,
                                                const int32_t* bag_tile_off, int ntiles, int B, int H, int C, int E,
                                                const float* Wv, const float* bv, float* y, float* stats, float* pooled,
                                                float* lse, float* o_attn, float* workspace, void* stream) {
    AP_CHECK(x && o && gamma && beta && pe && Qp && k_off && tile_map && bag_tile_off && Wv && bv && y && stats && pooled && lse);
    AP_CHECK(o_attn && workspace && H == AP_H && E == 512 && (C == 32 || C == 64) && B >= 0 && ntiles >= 0 && eps > 0.f);
    if (B == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    float* pacc = workspace;
    float* pml = workspace + (size_t)ntiles * AP_H * E;
    const float scale = 1.0f / sqrtf((float)C);
    if (ntiles > 0) {
        hipLaunchKernelGGL(k_apool_partial<true>, dim3(ntiles), dim3(256), 0, st, (const float*)nullptr, pe, Qp, k_off, tile_map, scale,
                           pacc, pml, LnbrFwd{x, o, gamma, beta, eps, y, stats});
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_apool_merge_value, dim3(B, AP_H), dim3(1024), 0, st, pacc, pml, bag_tile_off, E, pooled, lse, Wv, bv, C,
                       o_attn);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// Its backward (dpooled from mil_value_proj_bwd): dx [rows, E], do [B, E], dgamma, dbeta [E], dQp [B, H, E].  y = the keys the
// forward wrote; dy_acc (nullable) = the gradient y receives from its other consumer.  Two launches: k_lnbr_apool_bwd_one +
// one fold of all partials (MIL_LNBR_BWD=r16: the three-kernel form it replaces - per-row dots, per-tile dQp partials, the
// LayerNorm backward with the pool's rank-16 update folded into its load - kept as the cross-check of the tests).
// workspace: ntiles H E + 16 n_keys + 3 ntiles E floats.
extern "C" int mil_lnbr_absorbed_pool_bwd(const float* x, const float* o, const float* gamma, const float* beta, const float* stats, const float* y,
                                          const float* pe, const float* Qp, const float* lse, const float* dpooled,
                                          const float* pooled, const int32_t* k_off, const int32_t* tile_map,
                                          const int32_t* bag_tile_off, int ntiles, int n_keys, int B, int H, int C, int E,
                                          const float* dy_acc, float* dx, float* d_o, float* dgamma, float* dbeta, float* dQp,
                                          float* workspace, void* stream) {
    AP_CHECK(x && o && gamma && beta && stats && y && pe && Qp && lse && dpooled && pooled && k_off && tile_map && bag_tile_off);
    AP_CHECK(dx && d_o && dgamma && dbeta && dQp && workspace && H == AP_H && E == 512 && C > 0 && B >= 0 && ntiles >= 0 && n_keys >= 0);
    if (B == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)C);
    float* pdq = workspace;                                    // [ntiles][H][E]
    float* ad = pdq + (size_t)ntiles * AP_H * E;               // [n_keys][16]
    float* part = ad + (size_t)16 * n_keys;                    // [ntiles][3][E]
    if (ntiles > 0 && lnbr_bwd_one_pass()) {
        hipLaunchKernelGGL(k_lnbr_apool_bwd_one<true>, dim3(ntiles), dim3(256), 0, st, x, o, gamma, beta, stats, dy_acc, pe, Qp, dpooled, pooled,
                           lse, k_off, tile_map, scale, dx, part, pdq);
        MIL_CHECK_LAUNCH();
    } else if (ntiles > 0) {
        hipLaunchKernelGGL(k_apool_dots, dim3(ntiles), dim3(256), 0, st, y, pe, Qp, lse, dpooled, pooled, k_off, tile_map, scale, ad);
        MIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_apool_bwd_apply<false>, dim3(ntiles), dim3(256), 0, st, y, pe, Qp, dpooled, (const float*)ad, k_off,
                           tile_map, (const float*)nullptr, (float*)nullptr, pdq);
        MIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_lnbr_bwd_r16, dim3(ntiles), dim3(256), 0, st, x, o, gamma, stats, dy_acc, (const float*)ad, Qp, dpooled,
                           tile_map, dx, part);
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_lnbr_apool_fold, dim3(4 * (2 + B + B * AP_H)), dim3(1024), 0, st, (const float*)part, (const float*)pdq, bag_tile_off,
                       B, dgamma, dbeta, d_o, dQp);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// pooled in the grouped layout [B / T, THp, E] (row t H + h of group b / T; mil_absorb_query_pad): T = 1, THp = H is [B, H, E]
extern "C" int mil_value_proj_pad(const float* pooled, const float* Wv, const float* bv, int B, int H, int C, int E, int T,
                                  int THp, float* o, void* stream) {
    AP_CHECK(pooled && Wv && bv && o && B >= 0 && H > 0 && (C == 32 || C == 64) && E > 0);
    AP_CHECK(T >= 1 && THp >= T * H && (B % T) == 0);
    if (B == 0) return MIL_OK;
    hipLaunchKernelGGL(k_value_proj, dim3(B, H), dim3(256), 0, (hipStream_t)stream, pooled, Wv, bv, H, C, E, o, T, THp);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
extern "C" int mil_value_proj(const float* pooled, const float* Wv, const float* bv, int B, int H, int C, int E, float* o,
                              void* stream) {
    return mil_value_proj_pad(pooled, Wv, bv, B, H, C, E, 1, H, o, stream);
}

extern "C" int mil_grp_col_softmax_ws(float* S, int ld, const int32_t* grp_off, int G, int max_group_rows, int TH,
                                      float* ws, void* stream) {
    AP_CHECK(S && grp_off && G >= 0 && ld > 0 && TH > 0 && TH <= ld && max_group_rows >= 0);
    if (G == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    int nch, rch;
    if (ws != nullptr && gcs_plan(G, max_group_rows, ld, &nch, &rch)) {
        hipLaunchKernelGGL(k_gcs_stats<false>, dim3(G, nch), dim3(1024), 0, st, (const float*)S, (const float*)nullptr, ld, grp_off, TH, rch, ws);
        MIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_gcs_apply<false>, dim3(G, nch), dim3(1024), 0, st, S, (const float*)nullptr, (const float*)nullptr, ld, grp_off,
                           TH, rch, (const float*)ws);
        MIL_CHECK_LAUNCH();
        return MIL_OK;
    }
    // narrowest row-lane count whose registers hold the longest group (wider column blocks coalesce better)
    if (max_group_rows > 8192 && max_group_rows <= 32768)
        hipLaunchKernelGGL(k_grp_col_softmax<2>, dim3(G, (ld + 1) / 2), dim3(1024), 0, st, S, ld, grp_off, TH);
    else if (max_group_rows > 2048 && max_group_rows <= 8192)
        hipLaunchKernelGGL(k_grp_col_softmax<8>, dim3(G, (ld + 7) / 8), dim3(1024), 0, st, S, ld, grp_off, TH);
    else
        hipLaunchKernelGGL(k_grp_col_softmax<32>, dim3(G, (ld + 31) / 32), dim3(1024), 0, st, S, ld, grp_off, TH);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_grp_col_softmax(float* S, int ld, const int32_t* grp_off, int G, int max_group_rows, int TH,
                                   void* stream) {
    return mil_grp_col_softmax_ws(S, ld, grp_off, G, max_group_rows, TH, nullptr, stream);
}

extern "C" int mil_grp_col_softmax_bwd_ws(const float* A, const float* dA, int ld, const int32_t* grp_off, int G,
                                          int max_group_rows, int TH, float* dS, float* ws, void* stream) {
    AP_CHECK(A && dA && dS && grp_off && G >= 0 && ld > 0 && TH > 0 && TH <= ld && max_group_rows >= 0);
    if (G == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    int nch, rch;
    if (ws != nullptr && gcs_plan(G, max_group_rows, ld, &nch, &rch)) {
        hipLaunchKernelGGL(k_gcs_stats<true>, dim3(G, nch), dim3(1024), 0, st, A, dA, ld, grp_off, TH, rch, ws);
        MIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_gcs_apply<true>, dim3(G, nch), dim3(1024), 0, st, dS, A, dA, ld, grp_off, TH, rch, (const float*)ws);
        MIL_CHECK_LAUNCH();
        return MIL_OK;
    }
    if (max_group_rows > 4096 && max_group_rows <= 16384)
        hipLaunchKernelGGL(k_grp_col_softmax_bwd<2>, dim3(G, (ld + 1) / 2), dim3(1024), 0, st, A, dA, ld, grp_off, TH, dS);
    else if (max_group_rows > 1024 && max_group_rows <= 4096)
        hipLaunchKernelGGL(k_grp_col_softmax_bwd<8>, dim3(G, (ld + 7) / 8), dim3(1024), 0, st, A, dA, ld, grp_off, TH, dS);
    else
        hipLaunchKernelGGL(k_grp_col_softmax_bwd<32>, dim3(G, (ld + 31) / 32), dim3(1024), 0, st, A, dA, ld, grp_off, TH, dS);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_grp_col_softmax_bwd(const float* A, const float* dA, int ld, const int32_t* grp_off, int G,
                                       int max_group_rows, int TH, float* dS, void* stream) {
    return mil_grp_col_softmax_bwd_ws(A, dA, ld, grp_off, G, max_group_rows, TH, dS, nullptr, stream);
}

extern "C" int mil_row_softmax_t(float* S, int ld, int R, int T, int H, void* stream) {
    AP_CHECK(S && R >= 0 && T > 0 && T <= 16 && H > 0 && T * H <= ld);
    if (R == 0) return MIL_OK;
    hipLaunchKernelGGL(k_row_softmax_t, dim3((unsigned)(((size_t)R * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, ld,
                       R, T, H);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_row_softmax_t_bwd(const float* A, const float* dA, int ld, int R, int T, int H, float* dS, void* stream) {
    AP_CHECK(A && dA && dS && R >= 0 && T > 0 && T <= 16 && H > 0 && T * H <= ld);
    if (R == 0) return MIL_OK;
    hipLaunchKernelGGL(k_row_softmax_t_bwd, dim3((unsigned)(((size_t)R * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       A, dA, ld, R, T, H, dS);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
