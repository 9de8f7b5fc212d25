// Dropout keep-masks as packed bit tensors (train mode of the hot path).
//
// Reference: model/dim1/ABMIL.py:26,49 (`self.dropout1 = nn.Dropout(0.5)`, applied to the bag x before the gate AND before
// the pooled sum, :59) and model/aggregator.py:128-131 (`nn.Dropout(0.25)` in front of the head's Linear).  torch draws
// these masks from its Philox stream; which stream position an element gets is an implementation detail of ATen, so the
// contract kept here is the distribution (independent Bernoulli keeps, survivors scaled by 1/(1-p)) and Philox4x32-10 as
// the generator - the masks themselves are an explicit input/output (bit tensors), so the oracle can be run on exactly
// the mask the kernels used (tests/test_gpu_dropout.py) and a caller can supply its own.
//
// Layout: bits[row * (cols/32) + (col >> 5)], bit (col & 31) set = element kept.  One bit per element: the three kernels
// that consume the patches (k_gate_fwd, k_pool_partial, k_gate_bwd_dw) re-read 1/32 of the bytes of x (2 MiB for
// 32 x 1024 x 512) instead of a materialised dropped copy (64 MiB written and read), and the backward sees the forward's
// mask by construction.
//
// Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants), key = (seed_lo, seed_hi), counter =
// (block_lo, block_hi, offset_lo, offset_hi): `offset` is the stream position (the trainer uses the step number, host
// scalar or a device counter so a hipGraph replay draws fresh masks), `block` the index of the 128-bit output block:
//   p_drop = 0.5 : word w of the bit tensor = output word (w & 3) of block (w >> 2)            (1 random bit / element)
//   p_drop = 0.25: word w = ~(out[2 (w & 1)] & out[2 (w & 1) + 1]) of block (w >> 1)          (dropped iff 2 bits set)
//   otherwise    : element e kept iff out[e & 3] of block (e >> 2) >= p_drop * 2^32            (32 random bits / element)
#include "mil_common.h"
#include "philox.h"

// mode 0: p = 0.5, mode 1: p = 0.25, mode 2: generic threshold.  One thread per 128-bit Philox block.
__global__ __launch_bounds__(256) void k_dropout_keep_bits(uint32_t* __restrict__ bits, size_t nwords, int mode,
                                                           uint32_t thr, uint32_t seed_lo, uint32_t seed_hi,
                                                           uint64_t offset, const int32_t* __restrict__ offset_dev) {
    const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (offset_dev != nullptr) offset += (uint64_t)(uint32_t)offset_dev[0];
    const uint32_t o_lo = (uint32_t)offset, o_hi = (uint32_t)(offset >> 32);
    if (mode == 0) {
        const size_t w0 = blk * 4;
        if (w0 >= nwords) return;
        const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), o_lo, o_hi, seed_lo, seed_hi);
        if (w0 + 4 <= nwords) {
            *reinterpret_cast<uint4*>(bits + w0) = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
        } else {
            for (int i = 0; i < 4 && w0 + i < nwords; ++i) bits[w0 + i] = r.v[i];
        }
    } else if (mode == 1) {
        const size_t w0 = blk * 2;
        if (w0 >= nwords) return;
        const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), o_lo, o_hi, seed_lo, seed_hi);
        bits[w0] = ~(r.v[0] & r.v[1]);
        if (w0 + 1 < nwords) bits[w0 + 1] = ~(r.v[2] & r.v[3]);
    } else {
        // one output word = 32 elements = 8 Philox blocks; thread = word
        if (blk >= nwords) return;
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint64_t b = (uint64_t)blk * 8 + q;
            const philox4 r = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), o_lo, o_hi, seed_lo, seed_hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) word |= (r.v[e] >= thr ? 1u : 0u) << (4 * q + e);
        }
        bits[blk] = word;
    }
}

extern "C" int mil_dropout_keep_bits(uint32_t* bits, int rows, int cols, float p_drop, uint64_t seed, uint64_t offset,
                                     const int32_t* offset_dev, void* stream) {
    if (!bits || rows < 0 || cols <= 0 || (cols % 32) != 0 || !(p_drop >= 0.f) || !(p_drop < 1.f)) return MIL_EINVAL;
    const size_t nwords = (size_t)rows * (cols / 32);
    if (nwords == 0) return MIL_OK;
    int mode = 2;
    size_t nthreads = nwords;
    if (p_drop == 0.5f) { mode = 0; nthreads = (nwords + 3) / 4; }
    else if (p_drop == 0.25f) { mode = 1; nthreads = (nwords + 1) / 2; }
    const double t = (double)p_drop * 4294967296.0;
    const uint32_t thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    hipLaunchKernelGGL(k_dropout_keep_bits, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bits,
                       nwords, mode, thr, (uint32_t)seed, (uint32_t)(seed >> 32), offset, offset_dev);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// Both keep-bit tensors of a training step in ONE launch: the patch bits (p = 0.5, one Philox block = 4 words) by the first
// workgroups, the head's [B, L/32] words (p = 0.25, one block = 2 words) by the workgroups behind them - the same words the
// two stand-alone launches write (same block numbering per tensor, same keys), one launch of 4 us less per step where the
// forward kernel does not draw them itself (bucketed one-bag steps, bf16 x).
__global__ __launch_bounds__(256) void k_dropout_keep_bits_pair(uint32_t* __restrict__ xbits, size_t nx, uint32_t xs_lo,
                                                                uint32_t xs_hi, uint32_t* __restrict__ mbits, size_t nm,
                                                                uint32_t ms_lo, uint32_t ms_hi, uint64_t offset,
                                                                const int32_t* __restrict__ offset_dev, unsigned xblocks,
                                                                int32_t* __restrict__ advance, int32_t* __restrict__ done,
                                                                uint32_t mdelta, TileMapJob tm) {
    if (tm.bag_len != nullptr && blockIdx.x == gridDim.x - 1) {
        // the LAST workgroup of the launch is not a generator block: it builds the step's tile map (a 5 us launch of its own
        // in front of every one-ragged-bag step otherwise)
        build_tile_map_block(tm.bag_len, tm.B, tm.tile_map, tm.bag_tile_off, tm.rows_out, tm.T_cap);
        return;
    }
    if (advance != nullptr) {
        // the counter is read by ONE thread and handed to the others through LDS: that thread signs off only after its own
        // read has returned, so no wave of a workgroup can see the value the last workgroup to sign off writes
        __shared__ uint32_t off_lds;
        if (threadIdx.x == 0) off_lds = offset_dev != nullptr ? (uint32_t)offset_dev[0] : 0u;
        __syncthreads();
        offset += (uint64_t)off_lds;
    } else if (offset_dev != nullptr) {
        offset += (uint64_t)(uint32_t)offset_dev[0];
    }
    if (advance != nullptr && threadIdx.x == 0) {
        // every workgroup has read the counter before it signs off; the last one to do so advances it (the module route's
        // separate counter launch rides here: mil_dropout_keep_bits_pair)
        const int prev = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (int)gridDim.x - 1 - (tm.bag_len != nullptr ? 1 : 0)) {
            __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(advance, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const uint32_t o_lo = (uint32_t)offset, o_hi = (uint32_t)(offset >> 32);
    if (blockIdx.x < xblocks) {
        const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x, w0 = blk * 4;
        if (w0 >= nx) return;
        const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), o_lo, o_hi, xs_lo, xs_hi);
        if (w0 + 4 <= nx) {
            *reinterpret_cast<uint4*>(xbits + w0) = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
        } else {
            for (int i = 0; i < 4 && w0 + i < nx; ++i) xbits[w0 + i] = r.v[i];
        }
    } else {
        const size_t blk = (size_t)(blockIdx.x - xblocks) * 256 + threadIdx.x, w0 = blk * 2;
        if (w0 >= nm) return;
        const uint64_t mo = offset + mdelta;              // the head's words mdelta stream positions behind the patch bits
        const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)mo, (uint32_t)(mo >> 32), ms_lo, ms_hi);
        mbits[w0] = ~(r.v[0] & r.v[1]);
        if (w0 + 1 < nm) mbits[w0 + 1] = ~(r.v[2] & r.v[3]);
    }
}

// Internal (step.hip, gated_pool.hip): mil_dropout_keep_bits(xbits, R, L, 0.5, seed ..) + (mbits, B, L, 0.25, mseed ..)
static int keep_bits_pair_impl(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed,
                               uint64_t offset, const int32_t* offset_dev, int32_t* advance, int32_t* done, uint32_t mdelta,
                               void* stream, TileMapJob tm = TileMapJob{nullptr, 0, nullptr, nullptr, nullptr, 0}) {
    if ((R > 0 && !xbits) || (B > 0 && !mbits) || R < 0 || B < 0 || L <= 0 || (L % 32) != 0) return MIL_EINVAL;
    if ((advance != nullptr) != (done != nullptr)) return MIL_EINVAL;
    const size_t nx = (size_t)R * (L / 32), nm = (size_t)B * (L / 32);
    if (nx + nm == 0 && tm.bag_len == nullptr) return MIL_OK;
    const unsigned xblocks = (unsigned)(((nx + 3) / 4 + 255) / 256), mblocks = (unsigned)(((nm + 1) / 2 + 255) / 256);
    hipLaunchKernelGGL(k_dropout_keep_bits_pair, dim3(xblocks + mblocks + (tm.bag_len != nullptr ? 1 : 0)), dim3(256), 0,
                       (hipStream_t)stream, xbits, nx, (uint32_t)seed, (uint32_t)(seed >> 32), mbits, nm, (uint32_t)mseed,
                       (uint32_t)(mseed >> 32), offset, offset_dev, xblocks, advance, done, mdelta, tm);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
int dropout_keep_bits_pair(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed, uint64_t offset,
                           const int32_t* offset_dev, void* stream) {
    if (!xbits || !mbits) return MIL_EINVAL;
    return keep_bits_pair_impl(xbits, R, mbits, B, L, seed, mseed, offset, offset_dev, nullptr, nullptr, 0u, stream);
}
// Internal (step.hip): dropout_keep_bits_pair + mil_build_tile_map in ONE launch (the map is built by an extra workgroup).
int dropout_keep_bits_pair_tilemap(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed,
                                   uint64_t offset, const int32_t* offset_dev, const TileMapJob& tm, void* stream) {
    if (!xbits || !mbits || !tm.bag_len || !tm.tile_map || !tm.bag_tile_off || !tm.rows_out || tm.B <= 0 || tm.B > 1024 ||
        tm.T_cap < 0)
        return MIL_EINVAL;
    return keep_bits_pair_impl(xbits, R, mbits, B, L, seed, mseed, offset, offset_dev, nullptr, nullptr, 0u, stream, tm);
}
// The module route's three launches (patch keep bits, the pass counter's increment, the head's keep words) as one: both
// tensors drawn at stream position offset + offset_dev[0] (keys seed / mseed; the head's words mdelta positions further: 1 =
// where the module route's separate launch drew them, behind the counter's increment), then - by the last workgroup to sign
// off - advance[0] += 1 (advance, done: both or neither; done: a zero word between launches).  R == 0 or B == 0: that tensor only.
extern "C" int mil_dropout_keep_bits_pair(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed,
                                          uint64_t offset, const int32_t* offset_dev, int32_t* advance, int32_t* done,
                                          int mdelta, void* stream) {
    if (mdelta < 0) return MIL_EINVAL;
    return keep_bits_pair_impl(xbits, R, mbits, B, L, seed, mseed, offset, offset_dev, advance, done, (uint32_t)mdelta, stream);
}

// dx[row][col] = keep ? dx * scale : 0 in place (autograd route: backward through the patch dropout when the gradient
// of the bag rows is needed and no consumer kernel can fold the mask in).
__global__ __launch_bounds__(256) void k_dropout_apply_bits(float* __restrict__ t, const uint32_t* __restrict__ bits,
                                                            size_t n4, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // float4 index: 8 per bits word
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<f32x4*>(t)[i];
    const uint32_t m = bits[i >> 3] >> (4 * (i & 7));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ((m >> e) & 1u) ? v[e] * scale : 0.f;
    reinterpret_cast<f32x4*>(t)[i] = v;
}

extern "C" int mil_dropout_apply_bits(float* t, const uint32_t* bits, int rows, int cols, float scale, void* stream) {
    if (!t || !bits || rows < 0 || cols <= 0 || (cols % 32) != 0) return MIL_EINVAL;
    const size_t n4 = (size_t)rows * cols / 4;
    if (n4 == 0) return MIL_OK;
    hipLaunchKernelGGL(k_dropout_apply_bits, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, bits, n4,
                       scale);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
