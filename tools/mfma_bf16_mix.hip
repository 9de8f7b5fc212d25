// Scratch microbenchmark 3: v_mfma_f32_32x32x16_bf16 with 12 accumulator tiles per wave (the bf16 gate kernel's
// register shape), 2 waves per SIMD: bare, with the kernel's 16 ds_read_b128 per 24 MFMAs (interleaved or as blocks), and
// with a slice barrier.  Reports SIMD cycles per MFMA (ideal 32) and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int BAR>
__global__ __launch_bounds__(512) void k_loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[(256 + 384) * 32 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < (256 + 384) * 32 * 2; i += blockDim.x) lds[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 24));
    __syncthreads();
    f32x16 acc[2][6];
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 6; ++c) for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.f;
    const int fx = (r >> 2) & 3;
    const unsigned short* xa0 = lds + (64 * wr + r) * 32;
    const unsigned short* wb0 = lds + 2 * 256 * 32 + (96 * wc + r) * 32;
    u16x8 a[2], b[6];
    a[0] = *reinterpret_cast<const u16x8*>(xa0); a[1] = *reinterpret_cast<const u16x8*>(xa0 + 32 * 32);
    for (int j = 0; j < 6; ++j) b[j] = *reinterpret_cast<const u16x8*>(wb0 + ((j & 1) * 192 + 32 * (j >> 1)) * 32);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        const unsigned short* xa = xa0 + buf * 256 * 32;
        const unsigned short* wb = wb0 + buf * 384 * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = 8 * ((2 * ks + h) ^ fx);
            if (MODE == 1) {            // block read, then cluster
                a[0] = *reinterpret_cast<const u16x8*>(xa + ch); a[1] = *reinterpret_cast<const u16x8*>(xa + 32 * 32 + ch);
#pragma unroll
                for (int j = 0; j < 6; ++j) b[j] = *reinterpret_cast<const u16x8*>(wb + ((j & 1) * 192 + 32 * (j >> 1)) * 32 + ch);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[j]), acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[j]), acc[1][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a2 = 0; a2 < 2; ++a2) for (int c = 0; c < 6; ++c) for (int i = 0; i < 16; ++i) s += acc[a2][c][i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && blockIdx.x == 0) { out[(1 << 20) + wave] = (float)(t1 - t0); out[(1 << 20) + 8 + wave] = (float)(r1 - r0); }
}
template <int MODE, int BAR>
void run(float* out, const char* name) {
    const int iters = 20000;
    float best = 1e30f, clk = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_loop<MODE, BAR>), dim3(256), dim3(512), 0, 0, out, iters);
        hipDeviceSynchronize();
        float hh[16]; hipMemcpy(hh, out + (1 << 20), 64, hipMemcpyDeviceToHost);
        float h = 0.f; int w0 = 0;
        for (int w = 0; w < 8; ++w) if (hh[w] > h) { h = hh[w]; w0 = w; }
        if (h < best) { best = h; clk = hh[w0] / hh[8 + w0] * 100.0f; }
    }
    printf("%-44s : %.1f cycles per MFMA (SIMD time), clock %.0f MHz -> %.0f TFLOP/s chip\n", name, best / (iters * 24.0) / 2, clk,
           256 * 4 * 32768.0 / (best / (iters * 24.0) / 2) * clk * 1e6 / 1e12);
}
int main() {
    float* out; hipMalloc(&out, ((1 << 20) + 32) * 4);
    run<0, 0>(out, "bare (operands in registers)");
    run<0, 1>(out, "bare + barrier per 24 MFMAs");
    run<1, 0>(out, "8 ds_read_b128 block per 12 MFMAs");
    run<1, 1>(out, "8 ds_read_b128 block per 12 MFMAs + barrier");
    return 0;
}
