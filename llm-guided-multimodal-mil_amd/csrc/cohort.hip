// HBM-resident cohort: the per-epoch patch drop and the per-step feed of a capacity bucket, both on the device.
//
// Reference: dataset.py:366-393 loads one `<patient>.npy` bag [n, 768] per item on the host, keeps a SORTED RANDOM SUBSET of
// int(n * 0.9) (biopsies) / int(n * 0.8) (resections) of its rows (`sorted(random.sample(range(n), k))`, :374-381) and hands
// it to the step through DataLoader workers + pinned memory + `.cuda(non_blocking=True)` (train_ddp.py:193,274-293): 28 MB
// per bag per step over PCIe.  An MI355X has 288 GB of HBM - the whole cohort (1 000 bags x <= 48 MB) fits - so here the
// bags are loaded ONCE into one flat device buffer [total_rows, F] with a row-offset table, and
//   * k_patch_drop_select draws, for EVERY bag of the cohort in one launch per epoch, the sorted keep-index list of that
//     epoch: a uniformly random k-subset = the rows with the k smallest Philox4x32-10 keys (ties by row number), found by a
//     4-pass radix select over keys that are re-generated instead of stored, then written in ascending row order by a
//     block-wide scan;
//   * k_cohort_feed gathers the kept rows of this step's bags straight into the bucket's static input buffer (slot.x), and
//     in the same launch writes the bag lengths, labels, token ids / cached text embeddings of those bags to the bucket -
//     one launch in front of the replayed step, no host copy, no zero-pad, no D2D hop.
// The distribution is the reference's (uniform over k-subsets, ascending order, k = int(n * keep) computed on the host with
// the reference's own expression); the stream is Philox, not Python's Mersenne Twister, so WHICH subset is drawn differs -
// parity of the step is stated on identical rows (tests/test_gpu_cohort.py), the selection itself is pinned bit for bit
// by the numpy restatement in oracle/cohort.py.
#include "mil_common.h"
#include "philox.h"

#define SEL_THREADS 1024
#define COHORT_KEY_XOR 0x70617463685F6472ull   // "patch_dr": another key than the dropout streams of the same seed

// keys of rows 4 q .. 4 q + 3 of bag `bag` in epoch `epoch`
__device__ __forceinline__ philox4 cohort_keys(uint32_t q, uint32_t bag, uint32_t ep_lo, uint32_t ep_hi, uint32_t k0, uint32_t k1) {
    return philox4x32_10(q, bag, ep_lo, ep_hi, k0, k1);
}

// inclusive scan of one int per thread over the 1024 threads of the workgroup; returns the inclusive value, *total = sum
__device__ __forceinline__ int block_scan_incl(int v, int* wsum /* __shared__ [16] */, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    __syncthreads();                 // wsum may still be read by the previous scan
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
        const int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return v + base;
}

// One workgroup per bag (table slot s = blockIdx.x, cohort bag number bag0 + s).  sel[out_off[s] + j] = row_off[s] + the
// j-th kept row of the bag, ascending.
__global__ __launch_bounds__(SEL_THREADS) void k_patch_drop_select(const int32_t* __restrict__ row_off,
                                                                   const int32_t* __restrict__ keep,
                                                                   const int32_t* __restrict__ out_off, uint32_t bag0, uint32_t k0,
                                                                   uint32_t k1, uint32_t ep_lo, uint32_t ep_hi,
                                                                   int32_t* __restrict__ sel) {
    __shared__ int hist[256];
    __shared__ int wsum[16];
    __shared__ int pick[2];          // {bin, rows in the bins below it}
    const int slot = blockIdx.x, tid = threadIdx.x;
    const uint32_t bag = bag0 + blockIdx.x;          // the bag's number in the cohort: part of its Philox counter
    const int r0 = row_off[slot], n = row_off[slot + 1] - r0;
    int k = keep[slot];
    k = k < 0 ? 0 : (k > n ? n : k);
    int32_t* out = sel + out_off[slot];
    if (k == 0 || n <= 0) return;
    const int per = (((n + SEL_THREADS - 1) / SEL_THREADS) + 3) & ~3;        // rows per thread, whole Philox blocks
    const int c0 = tid * per, c1 = min(n, c0 + per);
    if (k == n) {                    // nothing dropped (keep fraction 1: validation / --augmentation 0)
        for (int i = c0; i < c1; ++i) out[i] = r0 + i;
        return;
    }
    // T = the k-th smallest key: radix select, one byte per pass from the top; keys are re-drawn every pass
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = c0; i < c1; i += 4) {
            const philox4 r = cohort_keys((uint32_t)(i >> 2), bag, ep_lo, ep_hi, k0, k1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < c1 && (r.v[e] & mask) == prefix) atomicAdd(&hist[(r.v[e] >> shift) & 255u], 1);
        }
        __syncthreads();
        int h = 0, v = 0;            // waves 0-3: inclusive scan of the 256 bins
        if (tid < 256) {
            const int lane = tid & 63;
            h = v = hist[tid];
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const int t = __shfl_up(v, dd);
                if (lane >= dd) v += t;
            }
            if (lane == 63) wsum[tid >> 6] = v;
        }
        __syncthreads();
        if (tid < 256) {
            int base = 0;
            for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
            const int incl = v + base, excl = incl - h;
            if (excl < remaining && remaining <= incl) { pick[0] = tid; pick[1] = excl; }
        }
        __syncthreads();
        prefix |= (uint32_t)pick[0] << shift;
        mask |= 255u << shift;
        remaining -= pick[1];
        __syncthreads();
    }
    const uint32_t T = prefix;
    const int need = remaining;      // rows with key == T to take (the first `need` of them in row order), >= 1
    int less = 0, eq = 0;
    for (int i = c0; i < c1; i += 4) {
        const philox4 r = cohort_keys((uint32_t)(i >> 2), bag, ep_lo, ep_hi, k0, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < c1) { less += r.v[e] < T; eq += r.v[e] == T; }
    }
    int tot;
    const int less_ex = block_scan_incl(less, wsum, &tot) - less;
    const int eq_ex = block_scan_incl(eq, wsum, &tot) - eq;
    int pos = less_ex + min(eq_ex, need), eqr = eq_ex;
    for (int i = c0; i < c1; i += 4) {
        const philox4 r = cohort_keys((uint32_t)(i >> 2), bag, ep_lo, ep_hi, k0, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e >= c1) break;
            const uint32_t key = r.v[e];
            bool take = key < T;
            if (key == T) { take = eqr < need; ++eqr; }
            if (take) out[pos++] = r0 + i + e;
        }
    }
}

extern "C" int mil_patch_drop_select(const int32_t* row_off, const int32_t* keep, const int32_t* out_off, int nbags, int bag0,
                                     uint64_t seed, uint64_t epoch, int32_t* sel, void* stream) {
    if (!row_off || !keep || !out_off || !sel || nbags < 0 || bag0 < 0) return MIL_EINVAL;
    if (nbags == 0) return MIL_OK;
    const uint64_t key = seed ^ COHORT_KEY_XOR;
    hipLaunchKernelGGL(k_patch_drop_select, dim3((unsigned)nbags), dim3(SEL_THREADS), 0, (hipStream_t)stream, row_off, keep,
                       out_off, (uint32_t)bag0, (uint32_t)key, (uint32_t)(key >> 32), (uint32_t)epoch, (uint32_t)(epoch >> 32), sel);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ---- feed ----------------------------------------------------------------------------------------------------------
#define FEED_ROWS_PER_WAVE 4
#define FEED_ROWS_PER_BLOCK 16

// Output row r (0 .. total) of the step = row r - prefix[b] of bag b's kept rows; the bags sit back to back in dst.
// Every wave moves FEED_ROWS_PER_WAVE whole rows: the row numbers first, then all its 16-byte loads (nontemporal: a cohort
// row is read once per epoch and the cohort is far larger than the Infinity Cache), then the stores (plain: the step reads
// them next).  The last workgroup writes the lengths and the per-bag side tables instead.
__global__ __launch_bounds__(256) void k_cohort_feed(const float* __restrict__ cohort, const int32_t* __restrict__ sel,
                                                     const mil_cohort_feed_desc d, float* __restrict__ dst,
                                                     int32_t* __restrict__ len_dev) {
    const int nblk = (int)gridDim.x - 1;
    if ((int)blockIdx.x == nblk) {
        if (len_dev != nullptr && (int)threadIdx.x < d.nb) len_dev[d.dst_bag0 + threadIdx.x] = d.rows[threadIdx.x];
        for (int a = 0; a < d.naux; ++a) {
            const uint32_t* tab = (const uint32_t*)d.aux_table[a];
            uint32_t* out = (uint32_t*)d.aux_dst[a];
            const int w = d.aux_words[a];
            for (int b = 0; b < d.nb; ++b) {
                const uint32_t* src = tab + (size_t)d.bag_id[b] * w;
                uint32_t* o = out + (size_t)(d.dst_bag0 + b) * w;
                for (int i = threadIdx.x; i < w; i += 256) o[i] = src[i];
            }
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L4 = d.L >> 2;
    int total = 0;
    for (int b = 0; b < d.nb; ++b) total += d.rows[b];
    const int rbase = blockIdx.x * FEED_ROWS_PER_BLOCK + wave * FEED_ROWS_PER_WAVE;
    size_t src_row[FEED_ROWS_PER_WAVE];
#pragma unroll
    for (int j = 0; j < FEED_ROWS_PER_WAVE; ++j) {
        const int r = rbase + j;
        int b = 0, p = 0;
        while (b + 1 < d.nb && r >= p + d.rows[b]) { p += d.rows[b]; ++b; }
        const int local = r - p;
        src_row[j] = 0;
        if (r < total) src_row[j] = sel != nullptr ? (size_t)sel[d.sel_off[b] + local] : (size_t)d.src_row0[b] + local;
    }
    for (int c = lane; c < L4; c += 64) {
        f32x4 v[FEED_ROWS_PER_WAVE];
#pragma unroll
        for (int j = 0; j < FEED_ROWS_PER_WAVE; ++j)
            if (rbase + j < total)
                v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(cohort + src_row[j] * (size_t)d.L) + c);
#pragma unroll
        for (int j = 0; j < FEED_ROWS_PER_WAVE; ++j)
            if (rbase + j < total)
                *(reinterpret_cast<f32x4*>(dst + (size_t)(d.dst_row0 + rbase + j) * d.L) + c) = v[j];
    }
}

extern "C" int mil_cohort_feed(const float* cohort, const int32_t* sel, const mil_cohort_feed_desc* d, float* dst,
                               int32_t* len_dev, void* stream) {
    if (!cohort || !d || !dst || d->struct_bytes != sizeof(mil_cohort_feed_desc)) return MIL_EINVAL;
    if (d->nb < 1 || d->nb > MIL_FEED_MAX_BAGS || d->L <= 0 || (d->L & 3) || d->dst_row0 < 0 || d->dst_bag0 < 0) return MIL_EINVAL;
    if (d->naux < 0 || d->naux > MIL_FEED_MAX_AUX) return MIL_EINVAL;
    long total = 0;
    for (int b = 0; b < d->nb; ++b) {
        if (d->rows[b] < 0 || d->bag_id[b] < 0 || (sel ? d->sel_off[b] < 0 : d->src_row0[b] < 0)) return MIL_EINVAL;
        total += d->rows[b];
    }
    for (int a = 0; a < d->naux; ++a)
        if (!d->aux_table[a] || !d->aux_dst[a] || d->aux_words[a] <= 0) return MIL_EINVAL;
    if (total > 0x7fffffffL) return MIL_EINVAL;
    const unsigned nblk = (unsigned)((total + FEED_ROWS_PER_BLOCK - 1) / FEED_ROWS_PER_BLOCK);
    hipLaunchKernelGGL(k_cohort_feed, dim3(nblk + 1), dim3(256), 0, (hipStream_t)stream, cohort, sel, *d, dst, len_dev);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
