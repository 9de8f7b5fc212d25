// Scratch microbenchmark 3 (round 4): what the bf16 weight-gradient kernels' CONSUMER side costs by itself - transposing
// fragment reads (ds_read_b64_tr_b16, two per 32x32x16 operand fragment) feeding v_mfma_f32_32x32x16_bf16, no staging, no
// global memory.  Wave tile A x B fragments per k-step (2 x 4 = the product kernels; 4 x 4 = a 256-accumulator wave), NW waves
// per workgroup (4 = one per SIMD, 8 = two), fragments prefetched one k-step ahead.  Reports SIMD cycles per MFMA (ideal 32)
// and LDS fragment bytes per clock and CU.  PLAIN=1 replaces the transposing reads by ds_read_b128 of the same volume.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_tr_mix.hip -o tools/mfma_tr_mix && tools/mfma_tr_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#define S 160
template <int PLAIN>
__device__ __forceinline__ u16x8 frag(const u16* img, int row, int col, int lane) {
    if (PLAIN) return *reinterpret_cast<const u16x8*>(img + ((lane & 31) * S + 8 * (lane >> 5)) + col);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + row * S + col));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (row + 4) * S + col));
    u16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
template <int TA, int TB, int NW, int PLAIN>
__global__ __launch_bounds__(64 * NW) void k_loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) u16 lds[64 * S * 5];          // A image + 4 panels, 64 rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * S * 5; i += blockDim.x) lds[i] = (u16)(0x3c00 + (i % 61));
    __syncthreads();
    const int h = lane >> 5, tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int acol = 16 * tg + 4 * tp, bcol = 16 * tg + 4 * tp;
    const u16* ai = lds;
    const u16* bi = lds + 64 * S * (1 + (wave & 3));
    f32x16 acc[TA][TB];
    for (int a = 0; a < TA; ++a) for (int b = 0; b < TB; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u16x8 fa[2][TA], fb[2][TB];
        auto frags = [&](int ks, int q) {
            const int row = 16 * ks + 8 * h + tq;
#pragma unroll
            for (int a = 0; a < TA; ++a) fa[q][a] = frag<PLAIN>(ai, row, acol + 32 * (a & 3), lane);
#pragma unroll
            for (int b = 0; b < TB; ++b) fb[q][b] = frag<PLAIN>(bi, row, bcol + 32 * b, lane);
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks & 1;
            if (ks < 3) frags(ks + 1, q ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[q][a]),
                                                                        __builtin_bit_cast(bf16x8, fb[q][b]), acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < TA; ++a) for (int b = 0; b < TB; ++b) for (int i = 0; i < 16; ++i) s += acc[a][b][i];
    out[blockIdx.x * 64 * NW + tid] = s;
    if (lane == 0 && blockIdx.x == 0) out[(1 << 20) + wave] = (float)(t1 - t0);
}
template <int TA, int TB, int NW, int PLAIN>
void run(float* out, const char* name) {
    const int iters = 2000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_loop<TA, TB, NW, PLAIN>), dim3(256), dim3(64 * NW), 0, 0, out, iters);
        hipDeviceSynchronize();
        float hh[8]; hipMemcpy(hh, out + (1 << 20), 4 * NW, hipMemcpyDeviceToHost);
        float h = 0.f;
        for (int w = 0; w < NW; ++w) h = hh[w] > h ? hh[w] : h;
        best = h < best ? h : best;
    }
    const double mf = iters * 4.0 * TA * TB;                       // MFMAs per wave
    const double per = best / mf / (NW / 4.0);                     // SIMD cycles per MFMA (NW / 4 waves share a SIMD)
    const double bytes = iters * 4.0 * (TA + TB) * 1024.0 * NW;    // fragment bytes read by the workgroup
    printf("%-52s : %6.1f SIMD cycles per MFMA (ideal 32), %5.1f fragment B / memtime tick / CU\n", name, per, bytes / best);
}
int main() {
    float* out; hipMalloc(&out, ((1 << 20) + 16) * 4);
    run<2, 4, 4, 0>(out, "tr reads, 2 x 4 wave tile, 4 waves (kept kernel)");
    run<2, 4, 8, 0>(out, "tr reads, 2 x 4 wave tile, 8 waves (w8 kernel)");
    run<4, 4, 4, 0>(out, "tr reads, 4 x 4 wave tile, 4 waves (256 accumulators)");
    run<2, 2, 8, 0>(out, "tr reads, 2 x 2 wave tile, 8 waves");
    run<2, 4, 4, 1>(out, "b128 reads, 2 x 4 wave tile, 4 waves");
    run<2, 4, 8, 1>(out, "b128 reads, 2 x 4 wave tile, 8 waves");
    run<4, 4, 4, 1>(out, "b128 reads, 4 x 4 wave tile, 4 waves");
    return 0;
}
