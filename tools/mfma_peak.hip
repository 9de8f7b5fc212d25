// Scratch: bare fp32 MFMA loop, to find the peak the chip actually sustains (clock under load).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k_peak(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < NACC; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[1 << 20] = (float)(t1 - t0); out[(1 << 20) + 1] = (float)(r1 - r0); }
}
int main() {
    float* out; hipMalloc(&out, ((1 << 20) + 16) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        const int iters = 2000;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_peak<6>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float h[2]; hipMemcpy(h, out + (1 << 20), 8, hipMemcpyDeviceToHost);
            double flops = 256.0 * (threads / 64) * iters * 24 * (2.0 * 32 * 32 * 2);
            printf("threads %d rep %d: %.3f ms  %.1f TFLOP/s  clock %.3f GHz\n", threads, rep, ms, flops / ms / 1e9, h[0] / h[1] * 0.1);
        }
    }
    return 0;
}
