// K2: multi-head attention cores of the SAM two-way transformer (model/sam/transformer.py:428-450,
// after the q/k/v projections and before out_proj) and of the CLIP text blocks (clip/model.py:171-184),
// plus LayerNorm and the positional-encoding add.  Softmax(q k^T / sqrt(c)) v per head, per bag.
//
// Segments: query rows [q_off[b], q_off[b+1]) attend to key rows [k_off[b], k_off[b+1]) of the same bag b.
// Two regimes, both HBM/latency-bound (head dim c = 32 or 64, so no MFMA: the contraction is over c):
//   "rows":  one thread per (query row, head), keys looped with an online softmax.  Used when a bag has
//            few keys per query: image->token attention (N queries x T<=16 text tokens), token self
//            attention, CLIP's causal 77 x 77.
//   "pool":  few queries (T <= 16) over many keys (token->image attention: an 8-head attention POOL over
//            the patches).  Split over 64-key tiles like the MIL pool: thread = channel j of the internal
//            width I (head = j / c), scores by a c-lane shuffle reduction, online-softmax partials, merge.
#include "mil_common.h"

#define AT_MAXT 16
#define AT_KT 64

template <int C>
__device__ __forceinline__ float head_allsum(float v) {
    if (C == 64) return wave_allsum(v);
    return half_allsum(v);          // C == 32: the 32-lane half that holds this head
}

// ================================================================================ rows: forward
// grid = ceil(total query rows * H / 256).  q, k, v: [rows, I] row-major; o: [Tq, I]; lse: [Tq, H].
template <int C>
__global__ __launch_bounds__(256) void k_attn_rows_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const int32_t* __restrict__ q_off,
                                                       const int32_t* __restrict__ k_off, const int32_t* __restrict__ q_bag,
                                                       int Tq, int H, int causal, float scale, float* __restrict__ o,
                                                       float* __restrict__ lse) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tq * H) return;
    const int row = idx / H, h = idx % H, I = H * C;
    const int b = q_bag[row];
    const int kb = k_off[b];
    int nk = k_off[b + 1] - kb;
    if (causal) nk = min(nk, row - q_off[b] + 1);
    float qr[C];
#pragma unroll
    for (int e = 0; e < C; e += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(q + (size_t)row * I + h * C + e);
        qr[e] = t[0] * scale; qr[e + 1] = t[1] * scale; qr[e + 2] = t[2] * scale; qr[e + 3] = t[3] * scale;
    }
    float m = -INFINITY, l = 0.f, acc[C];
#pragma unroll
    for (int e = 0; e < C; ++e) acc[e] = 0.f;
    for (int j = 0; j < nk; ++j) {
        const float* kr = k + (size_t)(kb + j) * I + h * C;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < C; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(kr + e);
            s += qr[e] * t[0] + qr[e + 1] * t[1] + qr[e + 2] * t[2] + qr[e + 3] * t[3];
        }
        const float mn = fmaxf(m, s);
        const float alpha = expf(m - mn), p = expf(s - mn);
        l = l * alpha + p;
        const float* vr = v + (size_t)(kb + j) * I + h * C;
#pragma unroll
        for (int e = 0; e < C; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(vr + e);
            acc[e] = acc[e] * alpha + p * t[0];
            acc[e + 1] = acc[e + 1] * alpha + p * t[1];
            acc[e + 2] = acc[e + 2] * alpha + p * t[2];
            acc[e + 3] = acc[e + 3] * alpha + p * t[3];
        }
        m = mn;
    }
    const float inv = nk > 0 ? 1.0f / l : 0.f;
#pragma unroll
    for (int e = 0; e < C; e += 4) {
        f32x4 t = {acc[e] * inv, acc[e + 1] * inv, acc[e + 2] * inv, acc[e + 3] * inv};
        *reinterpret_cast<f32x4*>(o + (size_t)row * I + h * C + e) = t;
    }
    if (lse != nullptr) lse[(size_t)row * H + h] = nk > 0 ? m + logf(l) : -INFINITY;
}

// ================================================================================ rows: backward (Tk <= 16 per bag)
// Phase 1 (thread = (row, head) of a 32-row block): recompute p_t, ds_t; dq row.  Phase 2 (thread = channel):
// per-workgroup partial dk[t][j] = sum_rows ds[row,h,t] q[row][j] scale, dv[t][j] = sum_rows p[row,h,t] do[row][j].
// Workgroups never straddle bags (host pads the block map): blk_map[g] = {bag, row0, nrows}.
// Partials: [nblk][2][AT_MAXT][I] summed per bag by k_attn_rows_bwd_reduce.
template <int C>
__global__ __launch_bounds__(256) void k_attn_rows_bwd(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                const float* __restrict__ o, const float* __restrict__ dout,
                                const float* __restrict__ lse, const int32_t* __restrict__ k_off,
                                const int32_t* __restrict__ blk_map, int H, float scale, float* __restrict__ dq,
                                float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // p [32][H][16], ds [32][H][16]
    const int tid = threadIdx.x, I = H * C;
    const int b = blk_map[3 * blockIdx.x], row0 = blk_map[3 * blockIdx.x + 1], nrows = blk_map[3 * blockIdx.x + 2];
    const int kb = k_off[b], nk = k_off[b + 1] - kb;
    float* p_l = lds;
    float* ds_l = lds + 32 * H * AT_MAXT;
    for (int idx = tid; idx < 32 * H; idx += blockDim.x) {
        const int rr = idx / H, h = idx % H;
        float pt[AT_MAXT], dst[AT_MAXT];
#pragma unroll
        for (int t = 0; t < AT_MAXT; ++t) { pt[t] = 0.f; dst[t] = 0.f; }
        if (rr < nrows) {
            const size_t row = (size_t)(row0 + rr);
            float qr[C], dor[C];
            float delta = 0.f;
#pragma unroll
            for (int e = 0; e < C; ++e) {
                qr[e] = q[row * I + h * C + e];
                dor[e] = dout[row * I + h * C + e];
                delta += dor[e] * o[row * I + h * C + e];
            }
            const float ls = lse[row * H + h];
            float dqr[C];
#pragma unroll
            for (int e = 0; e < C; ++e) dqr[e] = 0.f;
#pragma unroll
            for (int t = 0; t < AT_MAXT; ++t) {
                if (t < nk) {
                    const float* kr = k + (size_t)(kb + t) * I + h * C;
                    const float* vr = v + (size_t)(kb + t) * I + h * C;
                    float s = 0.f, dp = 0.f;
#pragma unroll
                    for (int e = 0; e < C; ++e) { s += qr[e] * kr[e]; dp += dor[e] * vr[e]; }
                    const float p = expf(s * scale - ls);
                    // one key: softmax == 1 identically, its score gradient is exactly 0 (the T=1 degeneracy)
                    const float d = nk == 1 ? 0.f : p * (dp - delta);
                    pt[t] = p;
                    dst[t] = d * scale;
#pragma unroll
                    for (int e = 0; e < C; ++e) dqr[e] += d * scale * kr[e];
                }
            }
#pragma unroll
            for (int e = 0; e < C; ++e) dq[row * I + h * C + e] = dqr[e];
        }
#pragma unroll
        for (int t = 0; t < AT_MAXT; ++t) {
            p_l[(rr * H + h) * AT_MAXT + t] = pt[t];
            ds_l[(rr * H + h) * AT_MAXT + t] = dst[t];
        }
    }
    __syncthreads();
    for (int j = tid; j < I; j += blockDim.x) {
        const int h = j / C;
        float dk[AT_MAXT], dv[AT_MAXT];
#pragma unroll
        for (int t = 0; t < AT_MAXT; ++t) { dk[t] = 0.f; dv[t] = 0.f; }
        for (int rr = 0; rr < nrows; ++rr) {
            const float qv = q[(size_t)(row0 + rr) * I + j], dov = dout[(size_t)(row0 + rr) * I + j];
#pragma unroll
            for (int t = 0; t < AT_MAXT; ++t) {
                dk[t] += ds_l[(rr * H + h) * AT_MAXT + t] * qv;
                dv[t] += p_l[(rr * H + h) * AT_MAXT + t] * dov;
            }
        }
        float* pp = part + (size_t)blockIdx.x * 2 * AT_MAXT * I;
#pragma unroll
        for (int t = 0; t < AT_MAXT; ++t) {
            pp[t * I + j] = dk[t];
            pp[(AT_MAXT + t) * I + j] = dv[t];
        }
    }
}

// dk[kb + t][j] = sum over the bag's blocks of part[g][0][t][j]; same for dv.  grid = (B, 2 * AT_MAXT), block = I threads
__global__ void k_attn_rows_bwd_reduce(const float* __restrict__ part, const int32_t* __restrict__ bag_blk_off,
                                       const int32_t* __restrict__ k_off, int I, float* __restrict__ dk,
                                       float* __restrict__ dv) {
    const int b = blockIdx.x, which = blockIdx.y / AT_MAXT, t = blockIdx.y % AT_MAXT;
    const int kb = k_off[b], nk = k_off[b + 1] - kb;
    if (t >= nk) return;
    for (int j = threadIdx.x; j < I; j += blockDim.x) {
        float vsum = 0.f;
        for (int g = bag_blk_off[b]; g < bag_blk_off[b + 1]; ++g)
            vsum += part[((size_t)g * 2 + which) * AT_MAXT * I + (size_t)t * I + j];
        (which == 0 ? dk : dv)[(size_t)(kb + t) * I + j] = vsum;
    }
}

// ================================================================================ pool: forward (T <= 16 queries, many keys)
// One workgroup (I threads) per 64-key tile of one bag: tile_map[g] = {bag, key0, nkeys}.  Thread j = channel.
// Partials per tile: acc [T][I], (m, l) [T][H][2].  TM = compile-time bound on the queries per bag; slots
// t >= T run on zero queries (harmless, never stored) so the key loop has no data-dependent branches.
template <int C, int TM>
__global__ __launch_bounds__(512) void k_attn_pool_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const int32_t* __restrict__ q_off,
                                                       const int32_t* __restrict__ tile_map, int H, float scale,
                                                       float* __restrict__ pacc, float* __restrict__ pml) {
    const int j = threadIdx.x, I = H * C, h = j / C;
    const int b = tile_map[3 * blockIdx.x], key0 = tile_map[3 * blockIdx.x + 1], nkeys = tile_map[3 * blockIdx.x + 2];
    const int qb = q_off[b], T = q_off[b + 1] - qb;
    float qv[TM], m[TM], l[TM], acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        qv[t] = t < T ? q[(size_t)(qb + t) * I + j] * scale : 0.f;
        m[t] = -INFINITY; l[t] = 0.f; acc[t] = 0.f;
    }
#pragma unroll 1
    for (int kk = 0; kk < nkeys; ++kk) {
        const float kv = k[(size_t)(key0 + kk) * I + j], vv = v[(size_t)(key0 + kk) * I + j];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const float s = head_allsum<C>(qv[t] * kv);
            const float mn = fmaxf(m[t], s);
            const float alpha = expf(m[t] - mn), p = expf(s - mn);
            l[t] = l[t] * alpha + p;
            acc[t] = acc[t] * alpha + p * vv;
            m[t] = mn;
        }
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        if (t < T) {
            pacc[((size_t)blockIdx.x * AT_MAXT + t) * I + j] = acc[t];
            if ((j % C) == 0) {
                pml[(((size_t)blockIdx.x * AT_MAXT + t) * H + h) * 2] = m[t];
                pml[(((size_t)blockIdx.x * AT_MAXT + t) * H + h) * 2 + 1] = l[t];
            }
        }
    }
}

// merge: grid = (B, T_max), block = I threads.  o[qb + t][j], lse[qb + t][h]
template <int C>
__global__ __launch_bounds__(512) void k_attn_pool_merge(const float* __restrict__ pacc, const float* __restrict__ pml,
                                  const int32_t* __restrict__ q_off, const int32_t* __restrict__ bag_tile_off, int H,
                                  float* __restrict__ o, float* __restrict__ lse) {
    const int b = blockIdx.x, t = blockIdx.y, j = threadIdx.x, I = H * C, h = j / C;
    const int qb = q_off[b], T = q_off[b + 1] - qb;
    if (t >= T) return;
    const int g0 = bag_tile_off[b], g1 = bag_tile_off[b + 1];
    float m = -INFINITY;
    for (int g = g0; g < g1; ++g) m = fmaxf(m, pml[(((size_t)g * AT_MAXT + t) * H + h) * 2]);
    float l = 0.f, acc = 0.f;
    for (int g = g0; g < g1; ++g) {
        const float sc = expf(pml[(((size_t)g * AT_MAXT + t) * H + h) * 2] - m);
        l += sc * pml[(((size_t)g * AT_MAXT + t) * H + h) * 2 + 1];
        acc += sc * pacc[((size_t)g * AT_MAXT + t) * I + j];
    }
    o[(size_t)(qb + t) * I + j] = g1 > g0 ? acc / l : 0.f;
    if ((j % C) == 0) lse[(size_t)(qb + t) * H + h] = g1 > g0 ? m + logf(l) : -INFINITY;
}

// ================================================================================ pool: backward
// Same tiling.  Per key: dv = sum_t p do_t, dk = sum_t ds q_t scale; per tile partial dq[t][j] = sum_keys ds k scale.
// Slots t >= T carry q = do = 0, lse = delta = 0: p = 1, dp = 0, ds = 0, so they add nothing.
template <int C, int TM>
__global__ __launch_bounds__(512) void k_attn_pool_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ o,
                                                       const float* __restrict__ dout, const float* __restrict__ lse,
                                                       const int32_t* __restrict__ q_off,
                                                       const int32_t* __restrict__ tile_map, int H, float scale,
                                                       float* __restrict__ dk, float* __restrict__ dv,
                                                       float* __restrict__ pdq) {
    const int j = threadIdx.x, I = H * C, h = j / C;
    const int b = tile_map[3 * blockIdx.x], key0 = tile_map[3 * blockIdx.x + 1], nkeys = tile_map[3 * blockIdx.x + 2];
    const int qb = q_off[b], T = q_off[b + 1] - qb;
    float qv[TM], dov[TM], ls[TM], delta[TM], dq[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const bool live = t < T;
        const size_t qr = (size_t)(qb + (live ? t : 0)) * I + j;
        qv[t] = live ? q[qr] : 0.f;
        dov[t] = live ? dout[qr] : 0.f;
        ls[t] = live ? lse[(size_t)(qb + t) * H + h] : 0.f;
        delta[t] = head_allsum<C>(dov[t] * (live ? o[qr] : 0.f));
        dq[t] = 0.f;
    }
#pragma unroll 1
    for (int kk = 0; kk < nkeys; ++kk) {
        const size_t kr = (size_t)(key0 + kk) * I + j;
        const float kv = k[kr], vv = v[kr];
        float dkv = 0.f, dvv = 0.f;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const float s = head_allsum<C>(qv[t] * kv) * scale;
            const float dp = head_allsum<C>(dov[t] * vv);
            const float p = expf(s - ls[t]);
            const float d = p * (dp - delta[t]) * scale;
            dvv += p * dov[t];
            dkv += d * qv[t];
            dq[t] += d * kv;
        }
        dk[kr] = dkv;
        dv[kr] = dvv;
    }
#pragma unroll
    for (int t = 0; t < TM; ++t)
        if (t < T) pdq[((size_t)blockIdx.x * AT_MAXT + t) * I + j] = dq[t];
}

// ---- pool form without wave reductions in the key loop (used for all T <= 16; the kernels above remain as the
// reference implementation of the tiling).  The per-key, per-query dot products over the head's c channels are
// the expensive part of the shuffle version (80-160 cross-lane steps per key); here they are plain per-thread
// loops: phase A thread = (key n of the tile, head h) holds that key's c channels in registers and reads the
// queries from LDS (a wave has one h: broadcast reads), phase B does the softmax statistics of the tile with one
// wave per (t, h), phase C thread = channel j accumulates over the keys with the probabilities read from LDS as
// b128 over four keys.  Partials keep the layout of k_attn_pool_fwd / _bwd, so merge and reduce are shared.
template <int C>
__global__ __launch_bounds__(512) void k_attn_pool_fwd_lds(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const int32_t* __restrict__ q_off,
                                                           const int32_t* __restrict__ tile_map, int H, float scale,
                                                           float* __restrict__ pacc, float* __restrict__ pml) {
    __shared__ __attribute__((aligned(16))) float lds[AT_MAXT * 512 + 8 * AT_MAXT * 64];     // qs [16][I], sbuf [H][16][64]; I <= 512, H <= 8
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, I = H * C, nw = I >> 6;
    const int b = tile_map[3 * blockIdx.x], key0 = tile_map[3 * blockIdx.x + 1], nkeys = tile_map[3 * blockIdx.x + 2];
    const int qb = q_off[b], T = q_off[b + 1] - qb;
    float* qs = lds;
    float* sbuf = lds + AT_MAXT * I;
    for (int idx = tid; idx < T * I; idx += I) qs[idx] = q[(size_t)qb * I + idx] * scale;
    __syncthreads();
    // phase A: scores
    for (int pr = tid; pr < 64 * H; pr += I) {
        const int n = pr & 63, h = pr >> 6;
        f32x4 kr[C / 4];
        const float* krow = k + (size_t)(key0 + min(n, nkeys - 1)) * I + h * C;
#pragma unroll
        for (int c = 0; c < C / 4; ++c) kr[c] = *reinterpret_cast<const f32x4*>(krow + 4 * c);
        for (int t = 0; t < T; ++t) {
            const float* qt = qs + t * I + h * C;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < C / 4; ++c) {
                const f32x4 qq = *reinterpret_cast<const f32x4*>(qt + 4 * c);
                s += kr[c][0] * qq[0] + kr[c][1] * qq[1] + kr[c][2] * qq[2] + kr[c][3] * qq[3];
            }
            sbuf[(h * AT_MAXT + t) * 64 + n] = n < nkeys ? s : -INFINITY;
        }
    }
    __syncthreads();
    // phase B: per (t, h) max / sum over the tile's keys, scores -> probabilities
    for (int it = wave; it < T * H; it += nw) {
        const int t = it / H, h = it % H;
        float* sp = sbuf + (h * AT_MAXT + t) * 64;
        const float sv = sp[lane];
        const float m = wave_allmax(sv);
        const float pv = lane < nkeys ? expf(sv - m) : 0.f;
        const float l = wave_allsum(pv);
        sp[lane] = pv;
        if (lane == 0) {
            pml[(((size_t)blockIdx.x * AT_MAXT + t) * H + h) * 2] = m;
            pml[(((size_t)blockIdx.x * AT_MAXT + t) * H + h) * 2 + 1] = l;
        }
    }
    __syncthreads();
    // phase C: acc[t][j] = sum_n p[t][h][n] v[n][j]
    const int j = tid, h = j / C;
    float acc[AT_MAXT];
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) acc[t] = 0.f;
    for (int n0 = 0; n0 < nkeys; n0 += 4) {
        float vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] = v[(size_t)(key0 + min(n0 + u, nkeys - 1)) * I + j];     // p = 0 beyond nkeys
#pragma unroll
        for (int t = 0; t < AT_MAXT; ++t) {
            if (t < T) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(sbuf + (h * AT_MAXT + t) * 64 + n0);
                acc[t] += p4[0] * vv[0] + p4[1] * vv[1] + p4[2] * vv[2] + p4[3] * vv[3];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t)
        if (t < T) pacc[((size_t)blockIdx.x * AT_MAXT + t) * I + j] = acc[t];
}

template <int C>
__global__ __launch_bounds__(512) void k_attn_pool_bwd_lds(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ o,
                                                           const float* __restrict__ dout, const float* __restrict__ lse,
                                                           const int32_t* __restrict__ q_off,
                                                           const int32_t* __restrict__ tile_map, int H, float scale,
                                                           float* __restrict__ dk, float* __restrict__ dv,
                                                           float* __restrict__ pdq) {
    // qs, dos [16][I]; pbuf, dsbuf [H][16][64]; lsd [16][H][2]   (I <= 512, H <= 8: 129 KB)
    __shared__ __attribute__((aligned(16))) float lds[2 * AT_MAXT * 512 + 2 * 8 * AT_MAXT * 64 + AT_MAXT * 8 * 2];
    const int tid = threadIdx.x, I = H * C;
    const int b = tile_map[3 * blockIdx.x], key0 = tile_map[3 * blockIdx.x + 1], nkeys = tile_map[3 * blockIdx.x + 2];
    const int qb = q_off[b], T = q_off[b + 1] - qb;
    float* qs = lds;
    float* dos = qs + AT_MAXT * I;
    float* pbuf = dos + AT_MAXT * I;
    float* dsbuf = pbuf + H * AT_MAXT * 64;
    float* lsd = dsbuf + H * AT_MAXT * 64;
    for (int idx = tid; idx < T * I; idx += I) {
        qs[idx] = q[(size_t)qb * I + idx];
        dos[idx] = dout[(size_t)qb * I + idx];
    }
    {   // delta[t][h] = do_t^h . o_t^h ; thread j contributes one channel, head_allsum folds the head's c lanes
        const int j = tid, h = j / C;
        for (int t = 0; t < T; ++t) {
            const float d = head_allsum<C>(dout[(size_t)(qb + t) * I + j] * o[(size_t)(qb + t) * I + j]);
            if ((j % C) == 0) {
                lsd[(t * H + h) * 2] = lse[(size_t)(qb + t) * H + h];
                lsd[(t * H + h) * 2 + 1] = d;
            }
        }
    }
    __syncthreads();
    for (int pr = tid; pr < 64 * H; pr += I) {
        const int n = pr & 63, h = pr >> 6;
        f32x4 kr[C / 4], vr[C / 4];
        const size_t rowoff = (size_t)(key0 + min(n, nkeys - 1)) * I + h * C;
#pragma unroll
        for (int c = 0; c < C / 4; ++c) {
            kr[c] = *reinterpret_cast<const f32x4*>(k + rowoff + 4 * c);
            vr[c] = *reinterpret_cast<const f32x4*>(v + rowoff + 4 * c);
        }
        for (int t = 0; t < T; ++t) {
            const float* qt = qs + t * I + h * C;
            const float* dt = dos + t * I + h * C;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < C / 4; ++c) {
                const f32x4 qq = *reinterpret_cast<const f32x4*>(qt + 4 * c);
                const f32x4 dd = *reinterpret_cast<const f32x4*>(dt + 4 * c);
                s += kr[c][0] * qq[0] + kr[c][1] * qq[1] + kr[c][2] * qq[2] + kr[c][3] * qq[3];
                dp += vr[c][0] * dd[0] + vr[c][1] * dd[1] + vr[c][2] * dd[2] + vr[c][3] * dd[3];
            }
            const float pv = n < nkeys ? expf(s * scale - lsd[(t * H + h) * 2]) : 0.f;
            pbuf[(h * AT_MAXT + t) * 64 + n] = pv;
            dsbuf[(h * AT_MAXT + t) * 64 + n] = pv * (dp - lsd[(t * H + h) * 2 + 1]) * scale;
        }
    }
    __syncthreads();
    const int j = tid, h = j / C;
    float qv[AT_MAXT], dov[AT_MAXT], dq[AT_MAXT];
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) {
        qv[t] = t < T ? qs[t * I + j] : 0.f;
        dov[t] = t < T ? dos[t * I + j] : 0.f;
        dq[t] = 0.f;
    }
    for (int n0 = 0; n0 < nkeys; n0 += 4) {
        float kq[4], dkk[4], dvv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kq[u] = k[(size_t)(key0 + min(n0 + u, nkeys - 1)) * I + j];
            dkk[u] = 0.f;
            dvv[u] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < AT_MAXT; ++t) {
            if (t < T) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(pbuf + (h * AT_MAXT + t) * 64 + n0);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dsbuf + (h * AT_MAXT + t) * 64 + n0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    dkk[u] += d4[u] * qv[t];
                    dvv[u] += p4[u] * dov[t];
                }
                dq[t] += d4[0] * kq[0] + d4[1] * kq[1] + d4[2] * kq[2] + d4[3] * kq[3];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (n0 + u < nkeys) {
                dk[(size_t)(key0 + n0 + u) * I + j] = dkk[u];
                dv[(size_t)(key0 + n0 + u) * I + j] = dvv[u];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t)
        if (t < T) pdq[((size_t)blockIdx.x * AT_MAXT + t) * I + j] = dq[t];
}

// dq[qb + t][j] = sum over the bag's tiles.  grid = (B, T_max), block = I
__global__ void k_attn_pool_bwd_reduce(const float* __restrict__ pdq, const int32_t* __restrict__ q_off,
                                       const int32_t* __restrict__ bag_tile_off, int I, float* __restrict__ dq) {
    const int b = blockIdx.x, t = blockIdx.y, j = threadIdx.x;
    const int qb = q_off[b], T = q_off[b + 1] - qb;
    if (t >= T) return;
    float acc = 0.f;
    for (int g = bag_tile_off[b]; g < bag_tile_off[b + 1]; ++g) acc += pdq[((size_t)g * AT_MAXT + t) * I + j];
    dq[(size_t)(qb + t) * I + j] = acc;
}

// ================================================================================ LayerNorm (rows x E), one wave per row
// y = (x - mean) * rstd * gamma + beta, biased variance, eps inside the sqrt (nn.LayerNorm, eps 1e-5).
template <int NE>     // E = 64 * NE
__global__ __launch_bounds__(256) void k_layernorm_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int rows, float eps,
                                                       float* __restrict__ y, float* __restrict__ stats,
                                                       const float* __restrict__ o, const int32_t* __restrict__ row_bag) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), E = 64 * NE;
    if (row >= rows) return;
    float v[NE];
    float s = 0.f;
    if (o != nullptr) {
        // LayerNorm(x + o[bag of row]): the per-bag row of the one-token image->token attention (k_add_bag_row) added on
        // the way in - the sum is never written to memory
        const float* orow = o + (size_t)row_bag[row] * E;
#pragma unroll
        for (int e = 0; e < NE; ++e) { v[e] = x[(size_t)row * E + lane + 64 * e] + orow[lane + 64 * e]; s += v[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) { v[e] = x[(size_t)row * E + lane + 64 * e]; s += v[e]; }
    }
    const float mean = wave_allsum(s) / E;
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) { const float d = v[e] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_allsum(ss) / E + eps);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int c = lane + 64 * e;
        y[(size_t)row * E + c] = (v[e] - mean) * rstd * gamma[c] + beta[c];
    }
    if (stats != nullptr && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// E = 512 with 16-byte accesses: lane l owns columns 4 l .. 4 l + 3 and 256 + 4 l .. (two loads and two stores per row where
// the generic kernel issues eight 4-byte ones of each); same arithmetic, another summation order inside the row.
__global__ __launch_bounds__(256) void k_layernorm_fwd512(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int rows, float eps,
                                                          float* __restrict__ y, float* __restrict__ stats,
                                                          const float* __restrict__ o, const int32_t* __restrict__ row_bag) {
    constexpr int E = 512;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) v[q] = *reinterpret_cast<const f32x4*>(x + (size_t)row * E + 256 * q + 4 * lane);
    if (o != nullptr) {
        const float* orow = o + (size_t)row_bag[row] * E;
#pragma unroll
        for (int q = 0; q < 2; ++q) v[q] += *reinterpret_cast<const f32x4*>(orow + 256 * q + 4 * lane);
    }
    const f32x4 t = v[0] + v[1];
    const float mean = wave_allsum((t[0] + t[1]) + (t[2] + t[3])) / E;
    f32x4 d[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) d[q] = v[q] - mean;
    const f32x4 sq = d[0] * d[0] + d[1] * d[1];
    const float rstd = 1.0f / sqrtf(wave_allsum((sq[0] + sq[1]) + (sq[2] + sq[3])) / E + eps);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 256 * q + 4 * lane);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 256 * q + 4 * lane);
        *reinterpret_cast<f32x4*>(y + (size_t)row * E + 256 * q + 4 * lane) = d[q] * rstd * g + b;
    }
    if (stats != nullptr && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)) (+ dres), g = dy gamma;  per-workgroup partial dgamma/dbeta [nblk][2][E].
// PARAMS = false: frozen gamma / beta (the CLIP tower): no parameter sums, no partials.  dres (optional): gradient that
// reached x along the residual branch around the norm - added here instead of by a separate elementwise launch.
// (Two rows per pass and wave were measured: no gain at 770 rows, 20 % slower at 32 768.)
template <int NE, bool PARAMS>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ dy, const float* __restrict__ stats,
                                                       const float* __restrict__ dres, int rows, int rows_per_blk,
                                                       float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[PARAMS ? 4 : 1][2][64 * NE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, E = 64 * NE;
    float dg[NE], db[NE], gm[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { dg[e] = 0.f; db[e] = 0.f; gm[e] = gamma[lane + 64 * e]; }
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    for (int row = r0 + w; row < r1; row += 4) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        float xh[NE], g[NE], rs[NE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            const float d = dy[(size_t)row * E + c];
            xh[e] = (x[(size_t)row * E + c] - mean) * rstd;
            rs[e] = dres != nullptr ? dres[(size_t)row * E + c] : 0.f;
            g[e] = d * gm[e];
            s1 += g[e];
            s2 += g[e] * xh[e];
            if (PARAMS) { dg[e] += d * xh[e]; db[e] += d; }
        }
        s1 = wave_allsum(s1) / E;
        s2 = wave_allsum(s2) / E;
#pragma unroll
        for (int e = 0; e < NE; ++e) dx[(size_t)row * E + lane + 64 * e] = rstd * (g[e] - s1 - xh[e] * s2) + rs[e];
    }
    if (PARAMS) {
#pragma unroll
        for (int e = 0; e < NE; ++e) { red[w][0][lane + 64 * e] = dg[e]; red[w][1][lane + 64 * e] = db[e]; }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * E; c += 256) {
            const int which = c / E, cc = c % E;
            part[((size_t)blockIdx.x * 2 + which) * E + cc] = red[0][which][cc] + red[1][which][cc] + red[2][which][cc] + red[3][which][cc];
        }
    }
}

// dgamma[c] = sum_b part[b][0][c], dbeta[c] = sum_b part[b][1][c]: workgroup = 64 columns of the 2 E, its 16 waves take
// every 16th partial (four loads in flight each) and fold through LDS - one launch instead of two column sums.
__global__ __launch_bounds__(1024) void k_layernorm_param_fold(const float* __restrict__ part, int nb, int E,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;              // column of the [2 E] row: < E -> dgamma, else dbeta
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (c < 2 * E) {
        const size_t st = (size_t)2 * E;
        int b = g;
        for (; b + 48 < nb; b += 64) {
            v0 += part[(size_t)b * st + c];
            v1 += part[(size_t)(b + 16) * st + c];
            v2 += part[(size_t)(b + 32) * st + c];
            v3 += part[(size_t)(b + 48) * st + c];
        }
        for (; b < nb; b += 16) v0 += part[(size_t)b * st + c];
    }
    red[g][lane] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (g == 0 && c < 2 * E) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += red[i][lane];
        if (c < E) dgamma[c] = v; else dbeta[c - E] = v;
    }
}

// Backward of LayerNorm(x + o[bag of row]) (k_layernorm_fwd with o): dx as k_layernorm_bwd with xhat recomputed from
// x + o, and - since d(x + o) reaches o as the sum over the bag's rows - the per-bag column sums of dx in the same pass
// (replaces k_segment_colsum's second trip over [rows, E]).  A workgroup's row range may cross ONE bag boundary (the host
// guarantees every bag has at least rows_per_blk rows): sums go to slot 2 for the bag of the block's first row, slot 3
// for the next bag.  part [nblk][4][E] = {dgamma, dbeta, do(first bag), do(second bag)}; folded in fixed order by
// k_layernorm_bagrow_fold (no atomics).
template <int NE>
__global__ __launch_bounds__(256) void k_layernorm_bagrow_bwd(const float* __restrict__ x, const float* __restrict__ o,
                                                              const int32_t* __restrict__ row_bag,
                                                              const float* __restrict__ gamma, const float* __restrict__ dy,
                                                              const float* __restrict__ stats, int rows, int rows_per_blk,
                                                              float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4][4][64 * NE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, E = 64 * NE;
    float dg[NE], db[NE], d0[NE], d1[NE], gm[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { dg[e] = 0.f; db[e] = 0.f; d0[e] = 0.f; d1[e] = 0.f; gm[e] = gamma[lane + 64 * e]; }
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    const int fb = r0 < rows ? row_bag[r0] : 0;
    for (int row = r0 + w; row < r1; row += 4) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        const int bag = row_bag[row];
        const float* orow = o + (size_t)bag * E;
        float xh[NE], g[NE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            const float d = dy[(size_t)row * E + c];
            xh[e] = (x[(size_t)row * E + c] + orow[c] - mean) * rstd;
            g[e] = d * gm[e];
            s1 += g[e];
            s2 += g[e] * xh[e];
            dg[e] += d * xh[e];
            db[e] += d;
        }
        s1 = wave_allsum(s1) / E;
        s2 = wave_allsum(s2) / E;
        if (bag == fb) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float v = rstd * (g[e] - s1 - xh[e] * s2);
                dx[(size_t)row * E + lane + 64 * e] = v;
                d0[e] += v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float v = rstd * (g[e] - s1 - xh[e] * s2);
                dx[(size_t)row * E + lane + 64 * e] = v;
                d1[e] += v;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        red[w][0][lane + 64 * e] = dg[e];
        red[w][1][lane + 64 * e] = db[e];
        red[w][2][lane + 64 * e] = d0[e];
        red[w][3][lane + 64 * e] = d1[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 4 * E; c += 256) {
        const int which = c / E, cc = c % E;
        part[((size_t)blockIdx.x * 4 + which) * E + cc] = red[0][which][cc] + red[1][which][cc] + red[2][which][cc] + red[3][which][cc];
    }
}

// Fold of k_layernorm_bagrow_bwd's partials, one launch: workgroups [0, 2E/64) sum dgamma / dbeta over all blocks (as
// k_layernorm_param_fold), workgroups behind them take (bag, 64 columns): the blocks whose row range meets the bag's rows,
// slot 2 where the bag is the block's first bag, slot 3 where it is the second.
__global__ __launch_bounds__(1024) void k_layernorm_bagrow_fold(const float* __restrict__ part, int nb, int E, int rows_per_blk,
                                                               const int32_t* __restrict__ row_off,
                                                               const int32_t* __restrict__ row_bag, int B,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ d_o) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int nA = (2 * E + 63) / 64;
    const size_t st = (size_t)4 * E;
    float v = 0.f;
    if ((int)blockIdx.x < nA) {
        const int c = blockIdx.x * 64 + lane;              // column of the [2 E] row: < E -> dgamma, else dbeta
        if (c < 2 * E) {
            float v1 = 0.f, v2 = 0.f, v3 = 0.f;       // four loads in flight per wave (the sums are latency-bound)
            int b = g;
            for (; b + 48 < nb; b += 64) {
                v += part[(size_t)b * st + c];
                v1 += part[(size_t)(b + 16) * st + c];
                v2 += part[(size_t)(b + 32) * st + c];
                v3 += part[(size_t)(b + 48) * st + c];
            }
            for (; b < nb; b += 16) v += part[(size_t)b * st + c];
            v = (v + v1) + (v2 + v3);
        }
        red[g][lane] = v;
        __syncthreads();
        if (g == 0 && c < 2 * E) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += red[i][lane];
            if (c < E) dgamma[c] = t; else dbeta[c - E] = t;
        }
        return;
    }
    const int q = blockIdx.x - nA, ncb = E / 64;
    const int bag = q / ncb, c = (q % ncb) * 64 + lane;
    const int lo = row_off[bag], hi = row_off[bag + 1];
    if (hi > lo) {
        const int b0 = lo / rows_per_blk, b1 = (hi - 1) / rows_per_blk;
        int b = b0 + g;
        // four (bag lookup, partial) pairs in flight per wave: with ONE long bag per batch (the authors' regime) a single
        // workgroup column walks all 512 blocks, two dependent loads each - 18 us when taken one at a time
        for (; b + 48 <= b1; b += 64) {
            int rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rb[u] = row_bag[(b + 16 * u) * rows_per_blk];
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = part[(size_t)(b + 16 * u) * st + (size_t)(rb[u] == bag ? 2 : 3) * E + c];
            v += (t[0] + t[1]) + (t[2] + t[3]);
        }
        for (; b <= b1; b += 16) {
            const int slot = row_bag[b * rows_per_blk] == bag ? 2 : 3;
            v += part[(size_t)b * st + (size_t)slot * E + c];
        }
    }
    red[g][lane] = v;
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][lane];
        d_o[(size_t)bag * E + c] = t;
    }
}

// The token stream has a few dozen rows: one workgroup of 16 waves, wave w takes rows w, w + 16, w + 32, w + 48 with
// all loads issued up front, and the parameter gradients are folded in LDS and written directly (no partials, no
// column-sum launches).
template <int NE, bool RES>
__global__ __launch_bounds__(1024) void k_layernorm_bwd_small(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ dy, const float* __restrict__ stats,
                                                              const float* __restrict__ dres, int rows,
                                                              float* __restrict__ dx, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    __shared__ float red[16][2][64 * NE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, E = 64 * NE;
    float dg[NE], db[NE], gm[NE], xv[4][NE], dv[4][NE], rv[RES ? 4 : 1][RES ? NE : 1], mean[4], rstd[4];
#pragma unroll
    for (int e = 0; e < NE; ++e) { dg[e] = 0.f; db[e] = 0.f; gm[e] = gamma[lane + 64 * e]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = min(w + 16 * i, rows - 1);
        mean[i] = stats[2 * row];
        rstd[i] = stats[2 * row + 1];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            xv[i][e] = x[(size_t)row * E + lane + 64 * e];
            dv[i][e] = dy[(size_t)row * E + lane + 64 * e];
            if (RES) rv[i][e] = dres[(size_t)row * E + lane + 64 * e];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = w + 16 * i;
        if (row >= rows) break;
        float xh[NE], g[NE], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            xh[e] = (xv[i][e] - mean[i]) * rstd[i];
            g[e] = dv[i][e] * gm[e];
            s1 += g[e];
            s2 += g[e] * xh[e];
            dg[e] += dv[i][e] * xh[e];
            db[e] += dv[i][e];
        }
        s1 = wave_allsum(s1) / E;
        s2 = wave_allsum(s2) / E;
#pragma unroll
        for (int e = 0; e < NE; ++e) dx[(size_t)row * E + lane + 64 * e] = rstd[i] * (g[e] - s1 - xh[e] * s2) + (RES ? rv[i][e] : 0.f);
    }
    if (dgamma == nullptr) return;          // frozen parameters (uniform over the workgroup)
#pragma unroll
    for (int e = 0; e < NE; ++e) { red[w][0][lane + 64 * e] = dg[e]; red[w][1][lane + 64 * e] = db[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * E; c += 1024) {
        const int which = c / E, cc = c % E;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) v += red[ww][which][cc];
        (which == 0 ? dgamma : dbeta)[cc] = v;
    }
}

// ================================================================================ keys + pe (positional table row = index within the bag)
// out[row] = x[row] + pe[row - row_off[bag]]   (model/sam/transformer.py:292,304: k = keys + key_pe with
// key_pe = self.pe[:, :N] of model/aggregator.py:99-106,190).  One float4 per thread.
__global__ __launch_bounds__(256) void k_add_pe(const float* __restrict__ x, const float* __restrict__ pe,
                                                const int32_t* __restrict__ row_bag, const int32_t* __restrict__ row_off,
                                                size_t n4, int E4, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int row = (int)(i / E4), c4 = (int)(i % E4);
    const int pos = row - row_off[row_bag[row]];
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * i);
    const f32x4 p = *reinterpret_cast<const f32x4*>(pe + ((size_t)pos * E4 + c4) * 4);
    *reinterpret_cast<f32x4*>(out + 4 * i) = a + p;
}

// out[row] = x[row] + o[row_bag[row]]: a per-bag row broadcast over the bag's rows.  With ONE text token per bag
// the image->token attention (sam/transformer.py:303-307) has softmax == 1, so its output is the same projected
// token for every patch of the bag: this add replaces the q projection, the attention core and the per-patch
// out projection.  One float4 per thread.
__global__ __launch_bounds__(256) void k_add_bag_row(const float* __restrict__ x, const float* __restrict__ o,
                                                     const int32_t* __restrict__ row_bag, size_t n4, int E4,
                                                     float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int row = (int)(i / E4), c4 = (int)(i % E4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * i);
    const f32x4 b = *reinterpret_cast<const f32x4*>(o + ((size_t)row_bag[row] * E4 + c4) * 4);
    *reinterpret_cast<f32x4*>(out + 4 * i) = a + b;
}

// out[b][j] = sum over the rows of bag b of Y[row][j]   (backward of k_add_bag_row w.r.t. o).
// grid = (ceil(E / 64), B, nchunk): chunk ch sums rows ch, ch + nchunk, ... of 4-row lanes into part[ch][b][E];
// a second launch with nchunk = 1 over the partials folds them (fixed order, no atomics).
__global__ __launch_bounds__(256) void k_segment_colsum(const float* __restrict__ Y, const int32_t* __restrict__ row_off,
                                                        int E, int nchunk, int rows_per_chunk, float* __restrict__ out,
                                                        int B) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int b = blockIdx.y, ch = blockIdx.z;
    const int r0 = row_off[b] + ch * rows_per_chunk, r1 = min(row_off[b + 1], r0 + rows_per_chunk);
    float v0 = 0.f, v1 = 0.f;
    if (c < E) {
        int i = r0 + g;
        for (; i + 4 < r1; i += 8) { v0 += Y[(size_t)i * E + c]; v1 += Y[(size_t)(i + 4) * E + c]; }
        for (; i < r1; i += 4) v0 += Y[(size_t)i * E + c];
    }
    red[g][threadIdx.x & 63] = v0 + v1;
    __syncthreads();
    if (g == 0 && c < E)
        out[((size_t)ch * B + b) * E + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Sinusoidal table of model/aggregator.py:99-106, built on the device once: pe[p][2i] = sin(p * w_i),
// pe[p][2i+1] = cos(p * w_i), w_i = exp(2i * -(ln 1e4 / E)).  fp32 argument as in the reference.
__global__ __launch_bounds__(256) void k_sinusoid_pe(float* __restrict__ pe, int n, int E) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * (E / 2)) return;
    const int p = (int)(idx / (E / 2)), i = (int)(idx % (E / 2));
    const float w = expf((float)(2 * i) * -(logf(10000.0f) / (float)E));
    const float a = (float)p * w;
    pe[(size_t)p * E + 2 * i] = sinf(a);
    pe[(size_t)p * E + 2 * i + 1] = cosf(a);
}

// ================================================================================ host entry points
#define DISPATCH_C(C_, ...)                      \
    do {                                         \
        if ((C_) == 32) { constexpr int CC = 32; __VA_ARGS__; } \
        else { constexpr int CC = 64; __VA_ARGS__; }            \
    } while (0)
#define DISPATCH_CT(C_, T_, ...)                                                     \
    do {                                                                             \
        if ((T_) <= 1) { constexpr int TT = 1; DISPATCH_C(C_, __VA_ARGS__); }         \
        else if ((T_) <= 4) { constexpr int TT = 4; DISPATCH_C(C_, __VA_ARGS__); }    \
        else if ((T_) <= 8) { constexpr int TT = 8; DISPATCH_C(C_, __VA_ARGS__); }    \
        else { constexpr int TT = 16; DISPATCH_C(C_, __VA_ARGS__); }                  \
    } while (0)

extern "C" int mil_attn_rows_fwd(const float* q, const float* k, const float* v, const int32_t* q_off,
                                 const int32_t* k_off, const int32_t* q_bag, int Tq, int H, int C, int causal,
                                 float* o, float* lse, void* stream) {
    if (!q || !k || !v || !q_off || !k_off || !q_bag || !o) return MIL_EINVAL;
    if ((C != 32 && C != 64) || H <= 0 || Tq < 0) return MIL_EINVAL;
    if (Tq == 0) return MIL_OK;
    const float scale = 1.0f / sqrtf((float)C);
    const int grid = (Tq * H + 255) / 256;
    DISPATCH_C(C, hipLaunchKernelGGL((k_attn_rows_fwd<CC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, q, k, v, q_off,
                                     k_off, q_bag, Tq, H, causal, scale, o, lse));
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_attn_rows_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                 const float* lse, const int32_t* k_off, const int32_t* blk_map,
                                 const int32_t* bag_blk_off, int nblk, int B, int H, int C, float* dq, float* dk,
                                 float* dv, float* workspace, void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !k_off || !blk_map || !bag_blk_off || !dq || !dk || !dv || !workspace)
        return MIL_EINVAL;
    if ((C != 32 && C != 64) || H <= 0 || nblk < 0 || B < 0) return MIL_EINVAL;
    if (nblk == 0) return MIL_OK;
    const float scale = 1.0f / sqrtf((float)C);
    const int I = H * C;
    const size_t shm = (size_t)2 * 32 * H * AT_MAXT * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_C(C, hipLaunchKernelGGL((k_attn_rows_bwd<CC>), dim3(nblk), dim3(256), shm, st, q, k, v, o, dout, lse, k_off,
                                     blk_map, H, scale, dq, workspace));
    MIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_attn_rows_bwd_reduce, dim3(B, 2 * AT_MAXT), dim3(min(I, 256)), 0, st, workspace, bag_blk_off,
                       k_off, I, dk, dv);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ================================================================================ rows: GENERAL backward (any Tq, Tk per bag)
// The shapes no shipped run needs a fast kernel for (`--alignment_base CT`, sam/transformer.py:78-86: 160 CT tokens as
// queries over N patches, N patches over 160 tokens, self-attention over 160 tokens): plain thread-per-(row, head) loops.
//   k_attn_gen_bwd_q   thread (query row, head): D = dO . O, then over the bag's keys  p = exp(scale q . k - lse),
//                      ds = p (dO . v - D),  dq += scale ds k;  D goes to dws [Tq, H]
//   k_attn_gen_bwd_kv  thread (key row, head): over the bag's queries  dv += p dO,  dk += scale ds q
template <int C>
__global__ __launch_bounds__(256) void k_attn_gen_bwd_q(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ o,
                                                        const float* __restrict__ dout, const float* __restrict__ lse,
                                                        const int32_t* __restrict__ k_off, const int32_t* __restrict__ q_bag,
                                                        int Tq, int H, float scale, float* __restrict__ dq,
                                                        float* __restrict__ dws) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tq * H) return;
    const int row = idx / H, h = idx % H, I = H * C;
    const int b = q_bag[row], kb = k_off[b], nk = k_off[b + 1] - kb;
    float qr[C], gr[C], acc[C];
    float D = 0.f;
#pragma unroll
    for (int e = 0; e < C; ++e) {
        qr[e] = q[(size_t)row * I + h * C + e] * scale;
        gr[e] = dout[(size_t)row * I + h * C + e];
        D += gr[e] * o[(size_t)row * I + h * C + e];
        acc[e] = 0.f;
    }
    dws[(size_t)row * H + h] = D;
    const float ls = lse[(size_t)row * H + h];
    for (int j = 0; j < nk; ++j) {
        const float* kr = k + (size_t)(kb + j) * I + h * C;
        const float* vr = v + (size_t)(kb + j) * I + h * C;
        float sdot = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < C; ++e) { sdot += qr[e] * kr[e]; dp += gr[e] * vr[e]; }
        const float ds = expf(sdot - ls) * (dp - D);
#pragma unroll
        for (int e = 0; e < C; ++e) acc[e] += ds * kr[e];
    }
#pragma unroll
    for (int e = 0; e < C; ++e) dq[(size_t)row * I + h * C + e] = acc[e] * scale;
}

template <int C>
__global__ __launch_bounds__(256) void k_attn_gen_bwd_kv(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ dout,
                                                         const float* __restrict__ lse, const float* __restrict__ dws,
                                                         const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_bag,
                                                         int Tk, int H, float scale, float* __restrict__ dk,
                                                         float* __restrict__ dv) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tk * H) return;
    const int row = idx / H, h = idx % H, I = H * C;
    const int b = k_bag[row], qb = q_off[b], nq = q_off[b + 1] - qb;
    float kr[C], vr[C], ak[C], av[C];
#pragma unroll
    for (int e = 0; e < C; ++e) {
        kr[e] = k[(size_t)row * I + h * C + e] * scale;
        vr[e] = v[(size_t)row * I + h * C + e];
        ak[e] = 0.f;
        av[e] = 0.f;
    }
    for (int i = 0; i < nq; ++i) {
        const float* qr = q + (size_t)(qb + i) * I + h * C;
        const float* gr = dout + (size_t)(qb + i) * I + h * C;
        float sdot = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < C; ++e) { sdot += qr[e] * kr[e]; dp += gr[e] * vr[e]; }
        const float p = expf(sdot - lse[(size_t)(qb + i) * H + h]);
        const float ds = p * (dp - dws[(size_t)(qb + i) * H + h]);
#pragma unroll
        for (int e = 0; e < C; ++e) { av[e] += p * gr[e]; ak[e] += ds * qr[e]; }
    }
#pragma unroll
    for (int e = 0; e < C; ++e) {
        dk[(size_t)row * I + h * C + e] = ak[e] * scale;
        dv[(size_t)row * I + h * C + e] = av[e];
    }
}

extern "C" int mil_attn_rows_bwd_general(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                         const float* lse, const int32_t* q_off, const int32_t* k_off,
                                         const int32_t* q_bag, const int32_t* k_bag, int Tq, int Tk, int H, int C, float* dq,
                                         float* dk, float* dv, float* workspace, void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !q_off || !k_off || !q_bag || !k_bag || !dq || !dk || !dv || !workspace)
        return MIL_EINVAL;
    if ((C != 32 && C != 64) || H <= 0 || Tq < 0 || Tk < 0) return MIL_EINVAL;
    const float scale = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (Tq > 0) {
        DISPATCH_C(C, hipLaunchKernelGGL((k_attn_gen_bwd_q<CC>), dim3((Tq * H + 255) / 256), dim3(256), 0, st, q, k, v, o, dout, lse,
                                         k_off, q_bag, Tq, H, scale, dq, workspace));
        MIL_CHECK_LAUNCH();
    }
    if (Tk > 0) {
        DISPATCH_C(C, hipLaunchKernelGGL((k_attn_gen_bwd_kv<CC>), dim3((Tk * H + 255) / 256), dim3(256), 0, st, q, k, v, dout, lse,
                                         (const float*)workspace, q_off, k_bag, Tk, H, scale, dk, dv));
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

extern "C" int mil_attn_pool_fwd_mh(const float* q, const float* k, const float* v, const int32_t* q_off,
                                    const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles, int B, int Tmax,
                                    int H, int C, float* o, float* lse, float* workspace, void* stream) {
    if (!q || !k || !v || !q_off || !tile_map || !bag_tile_off || !o || !lse || !workspace) return MIL_EINVAL;
    if ((C != 32 && C != 64) || H <= 0 || H > 8 || H * C > 512 || Tmax <= 0 || Tmax > AT_MAXT) return MIL_EINVAL;
    if (B == 0) return MIL_OK;
    const float scale = 1.0f / sqrtf((float)C);
    const int I = H * C;
    float* pacc = workspace;
    float* pml = workspace + (size_t)ntiles * AT_MAXT * I;
    hipStream_t st = (hipStream_t)stream;
    if (ntiles > 0) {
        DISPATCH_C(C, hipLaunchKernelGGL((k_attn_pool_fwd_lds<CC>), dim3(ntiles), dim3(I), 0, st, q, k, v, q_off, tile_map,
                                         H, scale, pacc, pml));
        MIL_CHECK_LAUNCH();
    }
    DISPATCH_C(C, hipLaunchKernelGGL((k_attn_pool_merge<CC>), dim3(B, Tmax), dim3(I), 0, st, pacc, pml, q_off, bag_tile_off,
                                     H, o, lse));
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_attn_pool_bwd_mh(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                    const float* lse, const int32_t* q_off, const int32_t* tile_map,
                                    const int32_t* bag_tile_off, int ntiles, int B, int Tmax, int H, int C, float* dq,
                                    float* dk, float* dv, float* workspace, void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !q_off || !tile_map || !bag_tile_off || !dq || !dk || !dv || !workspace)
        return MIL_EINVAL;
    if ((C != 32 && C != 64) || H <= 0 || H > 8 || H * C > 512 || Tmax <= 0 || Tmax > AT_MAXT) return MIL_EINVAL;
    if (B == 0) return MIL_OK;
    const float scale = 1.0f / sqrtf((float)C);
    const int I = H * C;
    hipStream_t st = (hipStream_t)stream;
    if (ntiles > 0) {
        DISPATCH_C(C, hipLaunchKernelGGL((k_attn_pool_bwd_lds<CC>), dim3(ntiles), dim3(I), 0, st, q, k, v, o, dout, lse,
                                         q_off, tile_map, H, scale, dk, dv, workspace));
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_attn_pool_bwd_reduce, dim3(B, Tmax), dim3(I), 0, st, workspace, q_off, bag_tile_off, I, dq);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

static int layernorm_fwd_impl(const float* x, const float* gamma, const float* beta, int rows, int E, float eps, float* y,
                              float* stats, const float* o, const int32_t* row_bag, void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || E <= 0 || (E % 64) != 0 || E > 512) return MIL_EINVAL;
    if (rows == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((rows + 3) / 4), blk(256);
    switch (E / 64) {
        case 1: hipLaunchKernelGGL(k_layernorm_fwd<1>, grid, blk, 0, st, x, gamma, beta, rows, eps, y, stats, o, row_bag); break;
        case 2: hipLaunchKernelGGL(k_layernorm_fwd<2>, grid, blk, 0, st, x, gamma, beta, rows, eps, y, stats, o, row_bag); break;
        case 4: hipLaunchKernelGGL(k_layernorm_fwd<4>, grid, blk, 0, st, x, gamma, beta, rows, eps, y, stats, o, row_bag); break;
        case 8: hipLaunchKernelGGL(k_layernorm_fwd512, grid, blk, 0, st, x, gamma, beta, rows, eps, y, stats, o, row_bag); break;
        default: return MIL_EINVAL;
    }
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int E, float eps,
                                 float* y, float* stats, void* stream) {
    return layernorm_fwd_impl(x, gamma, beta, rows, E, eps, y, stats, nullptr, nullptr, stream);
}

extern "C" int mil_layernorm_bagrow_fwd(const float* x, const float* o, const int32_t* row_bag, const float* gamma,
                                        const float* beta, int rows, int E, float eps, float* y, float* stats,
                                        void* stream) {
    if (!o || !row_bag) return MIL_EINVAL;
    return layernorm_fwd_impl(x, gamma, beta, rows, E, eps, y, stats, o, row_bag, stream);
}

extern "C" int mil_layernorm_bagrow_bwd(const float* x, const float* o, const int32_t* row_bag, const int32_t* row_off,
                                        int B, const float* gamma, const float* dy, const float* stats, int rows, int E,
                                        float* dx, float* d_o, float* dgamma, float* dbeta, float* workspace,
                                        void* stream) {
    if (!x || !o || !row_bag || !row_off || !gamma || !dy || !stats || !dx || !d_o || !dgamma || !dbeta || !workspace)
        return MIL_EINVAL;
    if (rows <= 0 || B <= 0 || (E % 64) != 0 || E > 512) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nb0 = mil_layernorm_bwd_blocks(rows);
    const int rpb = (rows + nb0 - 1) / nb0;
    const int nb = (rows + rpb - 1) / rpb;                 // every block owns at least one row
    const dim3 grid(nb), blk(256);
    switch (E / 64) {
        case 1: hipLaunchKernelGGL(k_layernorm_bagrow_bwd<1>, grid, blk, 0, st, x, o, row_bag, gamma, dy, stats, rows, rpb, dx, workspace); break;
        case 2: hipLaunchKernelGGL(k_layernorm_bagrow_bwd<2>, grid, blk, 0, st, x, o, row_bag, gamma, dy, stats, rows, rpb, dx, workspace); break;
        case 4: hipLaunchKernelGGL(k_layernorm_bagrow_bwd<4>, grid, blk, 0, st, x, o, row_bag, gamma, dy, stats, rows, rpb, dx, workspace); break;
        case 8: hipLaunchKernelGGL(k_layernorm_bagrow_bwd<8>, grid, blk, 0, st, x, o, row_bag, gamma, dy, stats, rows, rpb, dx, workspace); break;
        default: return MIL_EINVAL;
    }
    MIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_layernorm_bagrow_fold, dim3((2 * E + 63) / 64 + B * (E / 64)), dim3(1024), 0, st, workspace, nb, E, rpb,
                       row_off, row_bag, B, dgamma, dbeta, d_o);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_layernorm_bagrow_rows_per_block(int rows) {
    if (rows <= 0) return 1;
    const int nb0 = mil_layernorm_bwd_blocks(rows);
    return (rows + nb0 - 1) / nb0;
}

extern "C" int mil_layernorm_bwd_blocks(int rows) {
    int nb = (rows + 15) / 16;          // 16 rows per workgroup (four per wave): enough workgroups for a few hundred rows
    if (nb > 512) nb = 512;             // tall inputs: two workgroups per CU; more only adds partial-sum traffic
    return nb < 1 ? 1 : nb;
}

#define LNB_LAUNCH(NEV)                                                                                                  \
    if (params) hipLaunchKernelGGL((k_layernorm_bwd<NEV, true>), grid, blk, 0, st, x, gamma, dy, stats, dres, rows, rpb, dx, workspace); \
    else hipLaunchKernelGGL((k_layernorm_bwd<NEV, false>), grid, blk, 0, st, x, gamma, dy, stats, dres, rows, rpb, dx, workspace);

extern "C" int mil_layernorm_bwd_res(const float* x, const float* gamma, const float* dy, const float* stats,
                                     const float* dres, int rows, int E, float* dx, float* dgamma, float* dbeta,
                                     float* workspace, void* stream) {
    if (!x || !gamma || !dy || !stats || !dx) return MIL_EINVAL;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return MIL_EINVAL;
    const bool params = dgamma != nullptr;
    if (params && !workspace) return MIL_EINVAL;
    if (rows <= 0 || (E % 64) != 0 || E > 512) return MIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (rows <= 64) {
        const dim3 g1(1), b1(1024);
#define LNS_LAUNCH(NEV)                                                                                                   \
    if (dres != nullptr) hipLaunchKernelGGL((k_layernorm_bwd_small<NEV, true>), g1, b1, 0, st, x, gamma, dy, stats, dres, rows, dx, dgamma, dbeta); \
    else hipLaunchKernelGGL((k_layernorm_bwd_small<NEV, false>), g1, b1, 0, st, x, gamma, dy, stats, dres, rows, dx, dgamma, dbeta);
        switch (E / 64) {
            case 1: LNS_LAUNCH(1); break;
            case 2: LNS_LAUNCH(2); break;
            case 4: LNS_LAUNCH(4); break;
            case 8: LNS_LAUNCH(8); break;
            default: return MIL_EINVAL;
        }
        MIL_CHECK_LAUNCH();
        return MIL_OK;
    }
    const int nb = mil_layernorm_bwd_blocks(rows);
    const int rpb = (rows + nb - 1) / nb;
    const dim3 grid(nb), blk(256);
    switch (E / 64) {
        case 1: LNB_LAUNCH(1); break;
        case 2: LNB_LAUNCH(2); break;
        case 4: LNB_LAUNCH(4); break;
        case 8: LNB_LAUNCH(8); break;
        default: return MIL_EINVAL;
    }
    MIL_CHECK_LAUNCH();
    if (!params) return MIL_OK;
    // partials are [nb][2][E]: both parameter gradients folded by one launch (fixed order)
    hipLaunchKernelGGL(k_layernorm_param_fold, dim3((2 * E + 63) / 64), dim3(1024), 0, st, workspace, nb, E, dgamma, dbeta);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* stats, int rows, int E,
                                 float* dx, float* dgamma, float* dbeta, float* workspace, void* stream) {
    if (!dgamma || !dbeta || !workspace) return MIL_EINVAL;
    return mil_layernorm_bwd_res(x, gamma, dy, stats, nullptr, rows, E, dx, dgamma, dbeta, workspace, stream);
}

// ---------------------------------------------------------------------------------------------- device-side segments
// Every map the one-note fusion step reads, rebuilt ON THE DEVICE from the bags' true lengths, for a CAPACITY bucket of
// `cap` patch rows (the authors' regime: one ragged bag per GPU whose length changes every step, dataset.py:366-393; a
// captured hipGraph of the step is keyed by the bucket, not by the lengths).  Row layout of the bucket:
//   patch rows   [0, cap):  bag b at [k_off[b], k_off[b + 1]), rows >= k_off[B] are padding (zero weight, zero gradient)
//   token rows   [cap, cap + B P):  bag b at cap + b P .. + P   (the multi-modal bag of aggregator.py:192, no concat copy)
// Outputs:
//   k_off [B + 1], k_bag [cap] (padding rows: B - 1, a valid index for row-wise kernels whose gradient there is zero)
//   tile64 [T64][3] = {bag, row0, nkeys}: 64-key tiles of the absorbed attention pool, bag_tile64_off [B + 1]; tiles past
//     the last real one are padding tiles {0, row0, -count} over the rows [N, cap) (the kernels emit neutral partials for
//     nkeys <= 0, no bag's merge reads them, the apply pass zeroes those rows' gradient) or {0, 0, 0}
//   ds_zero [cap + B P]: the ABMIL score-gradient buffer of the bucket; its padding rows are zeroed here
//   tile32 [T32][4] = {bag, row0, nrows, 0}: the ABMIL pool's tiles, per bag its patch tiles then its token tiles
//     (bags.BagLayout.two_segment), bag_tile32_off [B + 1], padding tiles {0, 0, 0, 0}
//   row_bag [cap + B P]: bag of every row of the multi-modal bag, -1 for padding rows (mil_gate_bwd_input_pool)
//   rows_out [1] = sum of the lengths
struct FusionTail {              // the static row segments behind the `cap` patch rows: seg s holds rows[s] rows per bag
    int nseg;
    int rows[4];
};
__global__ __launch_bounds__(1024) void k_build_fusion_segs(const int32_t* __restrict__ len_dev, int B, FusionTail tail, int cap,
                                                            int32_t* __restrict__ k_off, int32_t* __restrict__ k_bag,
                                                            int32_t* __restrict__ tile64, int32_t* __restrict__ bag_tile64_off,
                                                            int T64, int32_t* __restrict__ tile32,
                                                            int32_t* __restrict__ bag_tile32_off, int T32,
                                                            int32_t* __restrict__ row_bag, int32_t* __restrict__ rows_out,
                                                            float* __restrict__ ds_zero) {
    __shared__ int s_row[1025], s_t64[1025], s_t32[1025];
    const int tid = threadIdx.x;
    int tokt = 0, tail_rows = 0;                    // tail tiles / tail rows per bag
    int seg_base[4], seg_t0[4];
    for (int sgm = 0; sgm < tail.nseg; ++sgm) {
        seg_base[sgm] = cap + B * tail_rows;        // segment s of bag b starts at seg_base[s] + b rows[s]
        seg_t0[sgm] = tokt;
        tokt += (tail.rows[sgm] + MIL_POOL_TILE - 1) / MIL_POOL_TILE;
        tail_rows += tail.rows[sgm];
    }
    if (tid == 0) {
        int r = 0, a = 0, c = 0;
        for (int b = 0; b < B; ++b) {
            s_row[b] = r;
            s_t64[b] = a;
            s_t32[b] = c;
            const int n = min(max(len_dev[b], 0), cap - r);              // never beyond the bucket
            r += n;
            a += (n + 63) / 64;
            c += (n + MIL_POOL_TILE - 1) / MIL_POOL_TILE + tokt;
        }
        s_row[B] = r;
        s_t64[B] = min(a, T64);
        s_t32[B] = min(c, T32);
        rows_out[0] = r;
    }
    __syncthreads();
    for (int b = tid; b <= B; b += 1024) {
        k_off[b] = s_row[b];
        bag_tile64_off[b] = min(s_t64[b], T64);
        bag_tile32_off[b] = min(s_t32[b], T32);
    }
    const int N = s_row[B];
    for (int b = 0; b < B; ++b) {
        const int r0 = s_row[b], r1 = s_row[b + 1];
        for (int r = r0 + tid; r < r1; r += 1024) { k_bag[r] = b; row_bag[r] = b; }
        const int a0 = s_t64[b], a1 = min(s_t64[b + 1], T64);
        for (int t = a0 + tid; t < a1; t += 1024) {
            const int row0 = r0 + (t - a0) * 64;
            tile64[3 * t] = b; tile64[3 * t + 1] = row0; tile64[3 * t + 2] = min(64, r1 - row0);
        }
        const int c0 = s_t32[b], c1 = min(s_t32[b + 1], T32), np = (r1 - r0 + MIL_POOL_TILE - 1) / MIL_POOL_TILE;
        for (int t = c0 + tid; t < c1; t += 1024) {
            const int j = t - c0;
            if (j < np) {
                const int row0 = r0 + j * MIL_POOL_TILE;
                reinterpret_cast<int4*>(tile32)[t] = make_int4(b, row0, min(MIL_POOL_TILE, r1 - row0), 0);
            } else {
                int sgm = tail.nseg - 1;
                while (sgm > 0 && j - np < seg_t0[sgm]) --sgm;           // the tail segment this tile belongs to
                const int first = seg_base[sgm] + b * tail.rows[sgm];
                const int row0 = first + (j - np - seg_t0[sgm]) * MIL_POOL_TILE;
                reinterpret_cast<int4*>(tile32)[t] = make_int4(b, row0, min(MIL_POOL_TILE, first + tail.rows[sgm] - row0), 0);
            }
        }
        for (int sgm = 0; sgm < tail.nseg; ++sgm)
            for (int r = tid; r < tail.rows[sgm]; r += 1024) row_bag[seg_base[sgm] + b * tail.rows[sgm] + r] = b;
    }
    for (int r = N + tid; r < cap; r += 1024) { k_bag[r] = B - 1; row_bag[r] = -1; }
    // behind the real 64-key tiles: PADDING tiles {0, row0, -count} that cover the rows [N, cap) - empty for every pool
    // kernel (nkeys <= 0), and the backward's apply pass writes the zero gradient of those rows from them (no fill launch)
    for (int t = s_t64[B] + tid; t < T64; t += 1024) {
        const int i = t - s_t64[B];
        const int r0 = i == 0 ? N : ((N + 63) / 64 + (i - 1)) * 64;       // first: N .. next multiple of 64, then whole blocks
        const int r1 = i == 0 ? min(cap, (N + 63) / 64 * 64) : min(cap, r0 + 64);
        const bool live = r0 < cap && r1 > r0;
        tile64[3 * t] = 0; tile64[3 * t + 1] = live ? r0 : 0; tile64[3 * t + 2] = live ? -(r1 - r0) : 0;
    }
    for (int r = N + tid; r < cap; r += 1024) ds_zero[r] = 0.f;           // score gradient of padding rows
    for (int t = s_t32[B] + tid; t < T32; t += 1024) reinterpret_cast<int4*>(tile32)[t] = make_int4(0, 0, 0, 0);
}

static int build_fusion_segs_impl(const int32_t* len_dev, int B, FusionTail tail, int cap, int32_t* k_off, int32_t* k_bag,
                                  int32_t* tile64, int32_t* bag_tile64_off, int T64, int32_t* tile32,
                                  int32_t* bag_tile32_off, int T32, int32_t* row_bag, int32_t* rows_out, float* ds_zero,
                                  void* stream) {
    if (!len_dev || !k_off || !k_bag || !tile64 || !bag_tile64_off || !tile32 || !bag_tile32_off || !row_bag || !rows_out ||
        !ds_zero)
        return MIL_EINVAL;
    int tokt = 0;
    if (tail.nseg < 1 || tail.nseg > 4) return MIL_EINVAL;
    for (int sgm = 0; sgm < tail.nseg; ++sgm) {
        if (tail.rows[sgm] <= 0) return MIL_EINVAL;
        tokt += (tail.rows[sgm] + MIL_POOL_TILE - 1) / MIL_POOL_TILE;
    }
    if (B <= 0 || B > 1024 || cap <= 0 || T64 < cap / 64 + B + 2 || T32 < cap / MIL_POOL_TILE + B * (1 + tokt)) return MIL_EINVAL;
    if (reinterpret_cast<uintptr_t>(tile32) & 15) return MIL_EINVAL;
    hipLaunchKernelGGL(k_build_fusion_segs, dim3(1), dim3(1024), 0, (hipStream_t)stream, len_dev, B, tail, cap, k_off, k_bag, tile64,
                       bag_tile64_off, T64, tile32, bag_tile32_off, T32, row_bag, rows_out, ds_zero);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_build_fusion_segs(const int32_t* len_dev, int B, int P, int cap, int32_t* k_off, int32_t* k_bag,
                                     int32_t* tile64, int32_t* bag_tile64_off, int T64, int32_t* tile32,
                                     int32_t* bag_tile32_off, int T32, int32_t* row_bag, int32_t* rows_out, float* ds_zero,
                                     void* stream) {
    FusionTail tail{1, {P, 0, 0, 0}};
    return build_fusion_segs_impl(len_dev, B, tail, cap, k_off, k_bag, tile64, bag_tile64_off, T64, tile32, bag_tile32_off, T32,
                                  row_bag, rows_out, ds_zero, stream);
}

// The same for a multi-modal bag with SEVERAL static segments behind the patch rows (model/aggregator.py:173: the CT +
// pathology bag [x_CT2CI | x_CI2CT | x_Pth2CI | x_CI2Pth] = patch rows, then P text-from-CT tokens, D CT tokens, P
// text-from-pathology tokens per bag): seg_rows[s] rows per bag in segment s, nseg <= 4; rows of segment s of bag b at
// cap + B (seg_rows[0] + .. + seg_rows[s-1]) + b seg_rows[s].  T32 >= cap / 32 + B (1 + sum ceil(seg_rows / 32)).
extern "C" int mil_build_fusion_segs_tail(const int32_t* len_dev, int B, int nseg, const int32_t* seg_rows, int cap,
                                          int32_t* k_off, int32_t* k_bag, int32_t* tile64, int32_t* bag_tile64_off, int T64,
                                          int32_t* tile32, int32_t* bag_tile32_off, int T32, int32_t* row_bag,
                                          int32_t* rows_out, float* ds_zero, void* stream) {
    if (!seg_rows || nseg < 1 || nseg > 4) return MIL_EINVAL;
    FusionTail tail{nseg, {0, 0, 0, 0}};
    for (int sgm = 0; sgm < nseg; ++sgm) tail.rows[sgm] = seg_rows[sgm];           // a HOST array (launch arguments)
    return build_fusion_segs_impl(len_dev, B, tail, cap, k_off, k_bag, tile64, bag_tile64_off, T64, tile32, bag_tile32_off, T32,
                                  row_bag, rows_out, ds_zero, stream);
}

extern "C" int mil_add_pe(const float* x, const float* pe, const int32_t* row_bag, const int32_t* row_off, int rows, int E,
                          float* out, void* stream) {
    if (!x || !pe || !row_bag || !row_off || !out || rows < 0 || E <= 0 || (E & 3)) return MIL_EINVAL;
    if (rows == 0) return MIL_OK;
    const size_t n4 = (size_t)rows * (E / 4);
    hipLaunchKernelGGL(k_add_pe, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, pe, row_bag,
                       row_off, n4, E / 4, out);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// CT feature map -> tokens (sam/transformer.py:86-98).  ct [B][C][D][HW] fp32 contiguous (the CT encoder's output
// [B, 512, 160, h, w], model/aggregator.py:139-140, HW = h * w):
//   reduce != 0 (resnetMC3_18): tokens[b][d][c] = mean over HW                      -> out [B * D, C]
//   reduce == 0 (medicalNet):   tokens[b][d * HW + s][c] = ct[b][c][d][s]           -> out [B * D * HW, C]
// One workgroup per (b, 32-channel tile): the 32 x T sums go through LDS so that the token rows leave as 128-byte runs.
__global__ __launch_bounds__(256) void k_ct_map_tokens(const float* __restrict__ ct, float* __restrict__ out, int C, int D, int HW,
                                                       int reduce) {
    extern __shared__ float tile[];                 // [32][TT + 1]
    const int T = reduce ? D : D * HW;              // tokens per bag
    const int inner = reduce ? HW : 1;
    const int b = blockIdx.y, c0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TT = 256;                         // tokens per pass
    const float inv = 1.0f / (float)inner;
    for (int t0 = 0; t0 < T; t0 += TT) {
        // wave w sums channels c0 + w, w + 4, ...; lane = token
        for (int cc = wave; cc < 32 && c0 + cc < C; cc += 4) {
            const float* src = ct + ((size_t)(b * C + c0 + cc) * T) * inner;
            for (int t = t0 + lane; t < min(T, t0 + TT); t += 64) {
                float v = 0.f;
                const float* q = src + (size_t)t * inner;
                for (int i = 0; i < inner; ++i) v += q[i];
                tile[cc * (TT + 1) + (t - t0)] = v * inv;
            }
        }
        __syncthreads();
        for (int k = tid; k < 32 * TT; k += 256) {
            const int t = t0 + k / 32, cc = k % 32;
            if (t < T && c0 + cc < C) out[((size_t)b * T + t) * C + c0 + cc] = tile[cc * (TT + 1) + (t - t0)];
        }
        __syncthreads();
    }
}

extern "C" int mil_ct_map_tokens(const float* ct, int B, int C, int D, int HW, int reduce, float* out, void* stream) {
    if (!ct || !out || B < 0 || C <= 0 || D <= 0 || HW <= 0) return MIL_EINVAL;
    if (B == 0) return MIL_OK;
    hipLaunchKernelGGL(k_ct_map_tokens, dim3((C + 31) / 32, B), dim3(256), 32 * 257 * sizeof(float), (hipStream_t)stream, ct, out, C, D,
                       HW, reduce);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_sinusoid_pe(float* pe, int n, int E, void* stream) {
    if (!pe || n < 0 || E <= 0 || (E & 1)) return MIL_EINVAL;
    if (n == 0) return MIL_OK;
    const size_t cnt = (size_t)n * (E / 2);
    hipLaunchKernelGGL(k_sinusoid_pe, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pe, n, E);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ================================================================================ CLIP text front/back ends
// x[s][p] = token_embedding[ids[s][p]] + positional_embedding[p]      (clip/model.py:340-342)
__global__ __launch_bounds__(256) void k_embed_tokens(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                      const float* __restrict__ pos, size_t n4, int ctx, int W4,
                                                      float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t tok = i / W4;
    const int c4 = (int)(i % W4), p = (int)(tok % ctx);
    const f32x4 e = *reinterpret_cast<const f32x4*>(table + ((size_t)ids[tok] * W4 + c4) * 4);
    const f32x4 q = *reinterpret_cast<const f32x4*>(pos + ((size_t)p * W4 + c4) * 4);
    *reinterpret_cast<f32x4*>(out + 4 * i) = e + q;
}

// out[s] = x[s][argmax_p ids[s][p]]  (the EOT token has the largest id, clip/model.py:348-350; first maximum
// on ties, as torch.argmax).  One workgroup (64 threads) per sequence.
__global__ __launch_bounds__(64) void k_gather_eot(const int64_t* __restrict__ ids, const float* __restrict__ x, int ctx, int W,
                                                   float* __restrict__ out) {
    const int s = blockIdx.x, lane = threadIdx.x;
    long long best = -1;
    int bi = 0;
    for (int p = lane; p < ctx; p += 64) {
        const long long v = ids[(size_t)s * ctx + p];
        if (v > best) { best = v; bi = p; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const long long ob = __shfl_xor(best, m);
        const int oi = __shfl_xor(bi, m);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    for (int c = lane; c < W; c += 64) out[(size_t)s * W + c] = x[((size_t)s * ctx + bi) * W + c];
}

extern "C" int mil_embed_tokens(const int64_t* ids, const float* table, const float* pos, int nseq, int ctx, int W,
                                float* out, void* stream) {
    if (!ids || !table || !pos || !out || nseq < 0 || ctx <= 0 || W <= 0 || (W & 3)) return MIL_EINVAL;
    if (nseq == 0) return MIL_OK;
    const size_t n4 = (size_t)nseq * ctx * (W / 4);
    hipLaunchKernelGGL(k_embed_tokens, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, table,
                       pos, n4, ctx, W / 4, out);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_gather_eot(const int64_t* ids, const float* x, int nseq, int ctx, int W, float* out, void* stream) {
    if (!ids || !x || !out || nseq < 0 || ctx <= 0 || W <= 0) return MIL_EINVAL;
    if (nseq == 0) return MIL_OK;
    hipLaunchKernelGGL(k_gather_eot, dim3(nseq), dim3(64), 0, (hipStream_t)stream, ids, x, ctx, W, out);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_add_bag_row(const float* x, const float* o, const int32_t* row_bag, int rows, int E, float* out,
                               void* stream) {
    if (!x || !o || !row_bag || !out || rows < 0 || E <= 0 || (E & 3)) return MIL_EINVAL;
    if (rows == 0) return MIL_OK;
    const size_t n4 = (size_t)rows * (E / 4);
    hipLaunchKernelGGL(k_add_bag_row, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, o, row_bag,
                       n4, E / 4, out);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_segment_colsum(const float* Y, const int32_t* row_off, int B, int max_rows, int E, float* out,
                                  float* workspace, void* stream) {
    if (!Y || !row_off || !out || B < 0 || E <= 0 || max_rows < 0) return MIL_EINVAL;
    if (B == 0) return MIL_OK;
    hipStream_t st = (hipStream_t)stream;
    int nchunk = (max_rows + 255) / 256;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 1 && workspace == nullptr) nchunk = 1;
    const int rpc = nchunk > 1 ? 256 : (max_rows > 0 ? max_rows : 1);
    if (nchunk == 1) {
        hipLaunchKernelGGL(k_segment_colsum, dim3((E + 63) / 64, B, 1), dim3(256), 0, st, Y, row_off, E, 1, rpc, out, B);
        MIL_CHECK_LAUNCH();
        return MIL_OK;
    }
    hipLaunchKernelGGL(k_segment_colsum, dim3((E + 63) / 64, B, nchunk), dim3(256), 0, st, Y, row_off, E, nchunk, rpc,
                       workspace, B);
    MIL_CHECK_LAUNCH();
    // fold: view the partials [nchunk][B*E] as a matrix with nchunk rows
    return mil_colsum(workspace, B * E, nchunk, B * E, out, 0, nullptr, stream);
}

// ================================================================================ full-sequence attention backward
// Backward of the rows-form attention for sequences of up to 80 tokens per (bag, head) - the CLIP text blocks when the
// prompt context is learnable (model/dim1/CLIP.py:29-62: CoOp trains ctx through the frozen tower).  One workgroup
// per (bag, head): scores, probabilities and dS live in LDS ([T][T] floats), q/k/v/do rows are re-read from L2.
//   p = softmax(q k^T scale (+ causal mask));  dv_j = sum_i p_ij do_i;  dS_ij = p_ij (do_i . v_j - delta_i);
//   dq_i = scale sum_j dS_ij k_j;  dk_j = scale sum_i dS_ij q_i.
#define AS_MAXT 96          // padded sequence length: three 32-row MFMA tiles
#define AS_THREADS 512      // 8 waves per (sequence, head) workgroup
#define AS_NW (AS_THREADS / 64)
#define AS_PS 100           // row stride of the score / probability image (100 mod 32 = 4: conflict-free b128 rows)

// One workgroup (8 waves) per (sequence, head); q / k / v (and do) rows of the head are staged once in LDS as
// [96][C + 4] images (rows >= T zero), every product runs on v_mfma_f32_32x32x2_f32:
//   NT  S = Q K^T, dP = dO V^T     both operands read with ds_read_b128 (lane (r, h): 16 bytes at k = 8t + 4h)
//   NN  O = P V,  dQ = dS K        A with b128 from the probability image, B k-major with ds_read_b32
//   TN  dV = P^T dO, dK = dS^T Q   both k-major
// Tile (it, jt) of S is skipped when the causal mask empties it; k-ranges are trimmed the same way.
template <int C>
struct SeqTiles {
    static constexpr int RS = C + 4;
    // acc (32 x 32 tile) += A[32 it.., k] B[32 jt.., k]^T over k in [0, C)     (both images row-major, k contiguous)
    static __device__ __forceinline__ void nt(const float* A, int it, const float* B, int jt, int r, int h, f32x16& acc) {
#pragma unroll
        for (int t = 0; t < C / 8; ++t) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(A + (32 * it + r) * RS + 8 * t + 4 * h);
            const f32x4 fb = *reinterpret_cast<const f32x4*>(B + (32 * jt + r) * RS + 8 * t + 4 * h);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[jj], fb[jj], acc, 0, 0, 0);
        }
    }
    // acc += P[32 it.., k] V[k, 32 ct..] over k in [8 t0, 8 t1)      (P image stride AS_PS, V image stride RS)
    static __device__ __forceinline__ void nn(const float* P, int it, const float* V, int ct, int t0, int t1, int r, int h,
                                              f32x16& acc) {
        for (int t = t0; t < t1; ++t) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(P + (32 * it + r) * AS_PS + 8 * t + 4 * h);
            float fb[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) fb[jj] = V[(8 * t + 4 * h + jj) * RS + 32 * ct + r];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[jj], fb[jj], acc, 0, 0, 0);
        }
    }
    // acc += P[k, 32 jt..]^T X[k, 32 ct..] over k in [8 t0, 8 t1)
    static __device__ __forceinline__ void tn(const float* P, int jt, const float* X, int ct, int t0, int t1, int r, int h,
                                              f32x16& acc) {
        for (int t = t0; t < t1; ++t) {
            float fa[4], fb[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                fa[jj] = P[(8 * t + 4 * h + jj) * AS_PS + 32 * jt + r];
                fb[jj] = X[(8 * t + 4 * h + jj) * RS + 32 * ct + r];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[jj], fb[jj], acc, 0, 0, 0);
        }
    }
    // stage rows [r0, r0 + T) of head h (C floats each, row stride ld) as [96][RS], zero beyond T, times mul
    static __device__ __forceinline__ void stage(const float* __restrict__ src, int r0, int T, int ld, int hoff, float mul,
                                                 float* dst, int tid) {
#pragma unroll
        for (int idx = tid; idx < AS_MAXT * (C / 4); idx += AS_THREADS) {
            const int row = idx / (C / 4), c4 = idx % (C / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < T) v = *reinterpret_cast<const f32x4*>(src + (size_t)(r0 + row) * ld + hoff + 4 * c4) * mul;
            *reinterpret_cast<f32x4*>(dst + row * RS + 4 * c4) = v;
        }
    }
};

template <int C>
__global__ __launch_bounds__(AS_THREADS) void k_attn_seq_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int ld,
                                                      const int32_t* __restrict__ q_off, int H, int causal, float scale,
                                                      float* __restrict__ o, float* __restrict__ lse) {
    using TL = SeqTiles<C>;
    constexpr int RS = TL::RS;
    // the score image overlays q and k once every wave holds its S tiles in registers (2 x 26 KB >= 38 KB): 78 KB per
    // workgroup for C = 64, i.e. two workgroups per CU
    __shared__ __attribute__((aligned(16))) float smem[3 * AS_MAXT * RS];
    static_assert(2 * AS_MAXT * RS >= AS_MAXT * AS_PS || C == 32, "score image must fit over q and k");
    __shared__ __attribute__((aligned(16))) float ps32[C == 32 ? AS_MAXT * AS_PS : 4];
    float* qs = smem;
    float* ks = smem + AS_MAXT * RS;
    float* vs = smem + 2 * AS_MAXT * RS;
    float* ps = C == 32 ? ps32 : smem;
    __shared__ float linv[AS_MAXT];
    const int b = blockIdx.x, hh = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, I = H * C;
    const int r = lane & 31, h = lane >> 5;
    const int r0 = q_off[b], T = q_off[b + 1] - r0;
    TL::stage(q, r0, T, ld, hh * C, scale, qs, tid);
    TL::stage(k, r0, T, ld, hh * C, 1.f, ks, tid);
    TL::stage(v, r0, T, ld, hh * C, 1.f, vs, tid);
    __syncthreads();
    const int ntile = (T + 31) / 32;
    f32x16 sacc[2];                          // tiles wave and wave + 8 of the 3 x 3 score grid
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int tile = wave + AS_NW * u, it = tile / 3, jt = tile % 3;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[u][i] = 0.f;
        if (tile < 9 && it < ntile && jt < ntile && !(causal && jt > it)) TL::nt(qs, it, ks, jt, r, h, sacc[u]);
    }
    __syncthreads();                         // q, k are dead: their space becomes the score image
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int tile = wave + AS_NW * u, it = tile / 3, jt = tile % 3;
        if (tile < 9 && it < ntile && jt < ntile && !(causal && jt > it)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) ps[(32 * it + mfma32_row(i, h)) * AS_PS + 32 * jt + r] = sacc[u][i];
        }
    }
    __syncthreads();
    // softmax of row i over j <= lim: one wave per row, lanes over j (two chunks of 64 cover 96)
    for (int i = wave; i < 32 * ntile; i += AS_NW) {
        const int lim = i < T ? (causal ? i : T - 1) : -1;
        float* pr = ps + i * AS_PS;
        const float s0 = lane <= lim ? pr[lane] : -INFINITY, s1 = (lane < 32 && 64 + lane <= lim) ? pr[64 + lane] : -INFINITY;
        const float m = wave_allmax(fmaxf(s0, s1));
        const float p0 = lane <= lim ? __expf(s0 - m) : 0.f, p1 = (lane < 32 && 64 + lane <= lim) ? __expf(s1 - m) : 0.f;
        const float l = wave_allsum(p0 + p1);
        pr[lane] = p0;
        if (lane < 32) pr[64 + lane] = p1;
        if (lane == 0) {
            linv[i] = lim >= 0 ? 1.0f / l : 0.f;
            if (i < T && lse != nullptr) lse[(size_t)(r0 + i) * H + hh] = m + logf(l);
        }
    }
    __syncthreads();
    for (int tile = wave; tile < 3 * (C / 32); tile += AS_NW) {
        const int it = tile / (C / 32), ct = tile % (C / 32);
        if (it >= ntile) continue;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        TL::nn(ps, it, vs, ct, 0, causal ? 4 * (it + 1) : 4 * ntile, r, h, acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * it + mfma32_row(i, h);
            if (row < T) o[(size_t)(r0 + row) * I + hh * C + 32 * ct + r] = acc[i] * linv[row];
        }
    }
}

// Backward of the above (recomputes P from the saved lse).
//   dV = P^T dO;  dP = dO V^T;  dS = P (dP - delta) scale;  dQ = dS K;  dK = dS^T Q,   delta_i = dO_i . O_i
template <int C>
__global__ __launch_bounds__(AS_THREADS) void k_attn_seq_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int ld, const float* __restrict__ o,
                                                      const float* __restrict__ dout, const float* __restrict__ lse,
                                                      const int32_t* __restrict__ q_off, int H, int causal, float scale,
                                                      float* __restrict__ dq, float* __restrict__ dk,
                                                      float* __restrict__ dv, int ldd) {
    using TL = SeqTiles<C>;
    constexpr int RS = TL::RS;
    __shared__ __attribute__((aligned(16))) float qs[AS_MAXT * RS], ks[AS_MAXT * RS], vs[AS_MAXT * RS], dos[AS_MAXT * RS];
    __shared__ __attribute__((aligned(16))) float ps[AS_MAXT * AS_PS];
    __shared__ float delta[AS_MAXT], lrow[AS_MAXT];
    const int b = blockIdx.x, hh = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, I = H * C;
    const int r = lane & 31, h = lane >> 5;
    const int r0 = q_off[b], T = q_off[b + 1] - r0;
    TL::stage(q, r0, T, ld, hh * C, 1.f, qs, tid);
    TL::stage(k, r0, T, ld, hh * C, 1.f, ks, tid);
    TL::stage(v, r0, T, ld, hh * C, 1.f, vs, tid);
    TL::stage(dout, r0, T, I, hh * C, 1.f, dos, tid);
    for (int i = wave; i < AS_MAXT; i += AS_NW) {
        float d = 0.f;
        if (i < T && lane < C) d = dout[(size_t)(r0 + i) * I + hh * C + lane] * o[(size_t)(r0 + i) * I + hh * C + lane];
        d = wave_allsum(d);
        if (lane == 0) { delta[i] = d; lrow[i] = i < T ? lse[(size_t)(r0 + i) * H + hh] : 0.f; }
    }
    __syncthreads();
    const int ntile = (T + 31) / 32;
    // P = exp(scale S - lse), masked; empty tiles are written as zeros (dV / dK read whole k-ranges)
    for (int tile = wave; tile < 9; tile += AS_NW) {
        const int it = tile / 3, jt = tile % 3;
        if (it >= ntile || jt >= ntile) continue;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const bool live = !(causal && jt > it);
        if (live) TL::nt(qs, it, ks, jt, r, h, acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * it + mfma32_row(i, h), col = 32 * jt + r;
            const bool keep = live && row < T && col < T && (!causal || col <= row);
            ps[row * AS_PS + col] = keep ? __expf(acc[i] * scale - lrow[row]) : 0.f;
        }
    }
    __syncthreads();
    for (int tile = wave; tile < 3 * (C / 32); tile += AS_NW) {           // dV[j][c] = sum_i P[i][j] dO[i][c]
        const int jt = tile / (C / 32), ct = tile % (C / 32);
        if (jt >= ntile) continue;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        TL::tn(ps, jt, dos, ct, causal ? 4 * jt : 0, 4 * ntile, r, h, acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * jt + mfma32_row(i, h);
            if (row < T) dv[(size_t)(r0 + row) * ldd + hh * C + 32 * ct + r] = acc[i];
        }
    }
    __syncthreads();
    for (int tile = wave; tile < 9; tile += AS_NW) {                      // dS = P (dO V^T - delta) scale, in place
        const int it = tile / 3, jt = tile % 3;
        if (it >= ntile || jt >= ntile || (causal && jt > it)) continue;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        TL::nt(dos, it, vs, jt, r, h, acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * it + mfma32_row(i, h);
            float* pp = ps + row * AS_PS + 32 * jt + r;
            *pp = *pp * (acc[i] - delta[row]) * scale;
        }
    }
    __syncthreads();
    for (int tile = wave; tile < 6 * (C / 32); tile += AS_NW) {           // dQ = dS K  and  dK = dS^T Q
        const bool is_q = tile < 3 * (C / 32);
        const int tt = is_q ? tile : tile - 3 * (C / 32);
        const int rt = tt / (C / 32), ct = tt % (C / 32);
        if (rt >= ntile) continue;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (is_q) TL::nn(ps, rt, ks, ct, 0, causal ? 4 * (rt + 1) : 4 * ntile, r, h, acc);
        else TL::tn(ps, rt, qs, ct, causal ? 4 * rt : 0, 4 * ntile, r, h, acc);
        float* dst = is_q ? dq : dk;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * rt + mfma32_row(i, h);
            if (row < T) dst[(size_t)(r0 + row) * ldd + hh * C + 32 * ct + r] = acc[i];
        }
    }
}

// QuickGELU y = x sigmoid(1.702 x) and its backward from the pre-activation (clip/model.py:162-164)
__global__ __launch_bounds__(256) void k_quickgelu(const float* __restrict__ x, const float* __restrict__ dy,
                                                   float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i], s = 1.0f / (1.0f + expf(-1.702f * xv));
    out[i] = dy == nullptr ? xv * s : dy[i] * s * (1.0f + 1.702f * xv * (1.0f - s));
}

extern "C" int mil_attn_seq_fwd(const float* q, const float* k, const float* v, int ld, const int32_t* q_off, int B, int Tmax,
                                int H, int C, int causal, float* o, float* lse, void* stream) {
    if (!q || !k || !v || !q_off || !o) return MIL_EINVAL;
    if ((C != 32 && C != 64) || H <= 0 || B < 0 || Tmax <= 0 || Tmax > AS_MAXT || ld < H * C || (ld & 3)) return MIL_EINVAL;
    if (B == 0) return MIL_OK;
    const float scale = 1.0f / sqrtf((float)C);
    DISPATCH_C(C, hipLaunchKernelGGL((k_attn_seq_fwd<CC>), dim3(B, H), dim3(AS_THREADS), 0, (hipStream_t)stream, q, k, v, ld, q_off,
                                     H, causal, scale, o, lse));
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_attn_seq_bwd(const float* q, const float* k, const float* v, int ld, const float* o, const float* dout,
                                const float* lse, const int32_t* q_off, int B, int Tmax, int H, int C, int causal,
                                float* dq, float* dk, float* dv, int ldd, void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !q_off || !dq || !dk || !dv) return MIL_EINVAL;
    if ((C != 32 && C != 64) || H <= 0 || B < 0 || Tmax <= 0 || Tmax > AS_MAXT) return MIL_EINVAL;
    if (ld < H * C || (ld & 3) || ldd < H * C) return MIL_EINVAL;
    if (B == 0) return MIL_OK;
    const float scale = 1.0f / sqrtf((float)C);
    DISPATCH_C(C, hipLaunchKernelGGL((k_attn_seq_bwd<CC>), dim3(B, H), dim3(AS_THREADS), 0, (hipStream_t)stream, q, k, v, ld, o,
                                     dout, lse, q_off, H, causal, scale, dq, dk, dv, ldd));
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_quickgelu(const float* x, const float* dy, float* out, size_t n, void* stream) {
    if (!x || !out) return MIL_EINVAL;
    if (n == 0) return MIL_OK;
    hipLaunchKernelGGL(k_quickgelu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, out, n);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
