// Scratch microbenchmark: does VALU / LDS / VMEM work hide in the shadow of v_mfma_f32_32x32x2_f32 on gfx950?
// Loop body: 4 independent MFMAs, NV dependent-free v_fma_f32 (or other fillers) placed after EACH MFMA.  One workgroup per
// CU, 256 or 512 threads (1 or 2 waves per SIMD).  Reports shader cycles per MFMA (ideal 64, or 32 per wave-pair slot).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int KIND>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = a * (i + 1);
    lds[threadIdx.x] = a;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) f[v & 7] = __builtin_fmaf(f[v & 7], 1.0001f, 0.5f);                       // v_fma_f32
                if (KIND == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[v & 7]) : "v"(b));          // integer VALU
                if (KIND == 2) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((threadIdx.x & 1023) * 4)); asm volatile("" ::"v"(t)); }
                if (KIND == 3) asm volatile("s_nop 0");
            }
            if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[(1 << 20) + (threadIdx.x >> 6)] = (float)(t1 - t0);
}
template <int NV, int KIND>
void run(float* out, const char* name) {
    for (int threads : {256, 512}) {
        const int iters = 4000;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL((k_mix<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
            hipDeviceSynchronize();
            float hh[8]; hipMemcpy(hh, out + (1 << 20), 32, hipMemcpyDeviceToHost);
            float h = 0.f;
            for (int w = 0; w < threads / 64; ++w) h = hh[w] > h ? hh[w] : h;       // the slowest wave: the SIMD is busy until then
            best = h < best ? h : best;
        }
        const double per_mfma = best / (iters * 4.0) / (threads / 256);     // SIMD-cycles per MFMA issued on that SIMD
        printf("%-10s fillers/MFMA %2d  waves/SIMD %d : %.1f cycles per MFMA (SIMD time)\n", name, NV, threads / 256, per_mfma);
    }
}
int main() {
    float* out; hipMalloc(&out, ((1 << 20) + 16) * 4);
    run<0, 0>(out, "none");
    run<4, 0>(out, "v_fma"); run<8, 0>(out, "v_fma"); run<12, 0>(out, "v_fma"); run<16, 0>(out, "v_fma");
    run<4, 1>(out, "v_and"); run<8, 1>(out, "v_and"); run<16, 1>(out, "v_and");
    run<2, 2>(out, "ds_read"); run<4, 2>(out, "ds_read");
    run<8, 3>(out, "s_nop"); run<16, 3>(out, "s_nop");
    return 0;
}
