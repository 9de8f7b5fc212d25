// Scratch microbenchmark: how many bytes per second can the CUs of an MI355X pull through their vector-memory path
// (TA / L1 / L2 fabric), from an L2-resident buffer, from Infinity Cache and from HBM, with ordinary 16-byte loads to
// registers and with LDS-DMA (global_load_lds_dwordx4)?  The bf16 gate kernels move 26-48 bytes per CU and MFMA clock at
// full matrix rate; this gives the ceiling they run into.      hipcc -O3 --offload-arch=gfx950 cu_load_bw.hip -o cu_load_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// every workgroup streams `span` bytes starting at (blockIdx * stride) % total, `iters` times; U loads in flight per thread
template <int U, bool NT = true>
__global__ __launch_bounds__(512) void k_reg(const u32x4* __restrict__ src, size_t stride16, size_t span16, size_t total16,
                                            int iters, unsigned* out) {
    const size_t base = ((size_t)blockIdx.x * stride16) % total16;
    u32x4 acc = {0, 0, 0, 0};
    const size_t per = (size_t)blockDim.x * U;
    for (int it = 0; it < iters; ++it) {
        for (size_t o = 0; o + per <= span16; o += per) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4* p = src + base + o + (size_t)u * blockDim.x + threadIdx.x;
                v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = acc.x;
}

__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_byte_addr_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte_addr_uniform) : "memory");
}

template <int U>
__global__ __launch_bounds__(512) void k_dma(const u32x4* __restrict__ src, size_t stride16, size_t span16, size_t total16,
                                            int iters, unsigned* out) {
    __shared__ __attribute__((aligned(16))) u32x4 ring[8 * 512];      // 64 KB
    const size_t base = ((size_t)blockIdx.x * stride16) % total16;
    const int wave = threadIdx.x >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ring;
    const size_t per = (size_t)blockDim.x * U;
    for (int it = 0; it < iters; ++it) {
        for (size_t o = 0; o + per <= span16; o += per) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                dma16(src + base + o + (size_t)u * blockDim.x + threadIdx.x,
                      __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((u & 7) * 512 + wave * 64) * 16)));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U / 2) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ring[threadIdx.x].x == 0x12345678u) out[threadIdx.x] = 1;
}

template <typename K>
static void run(const char* name, K kern, int wgs, int threads, const u32x4* src, size_t stride, size_t span, size_t total, unsigned* out) {
    const int iters = (int)((size_t)(1u << 31) / ((size_t)wgs * span) + 1);      // ~2 GB moved per launch at least
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, src, stride / 16, span / 16, total / 16, iters, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double bytes = (double)wgs * span * iters;
    printf("%-58s wgs %4d x %3d thr: %7.2f TB/s  (%.1f GB/s per CU)\n", name, wgs, threads, bytes / best * 1e-9, bytes / best * 1e-6 / 256);
    fflush(stdout);
}

// one pass per launch (iters = 1), the launch repeated: nothing is re-read inside a workgroup
template <typename K>
static void run1(const char* name, K kern, int wgs, int threads, const u32x4* src, size_t stride, size_t span, size_t total, unsigned* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, src, stride / 16, span / 16, total / 16, 1, out);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, src, stride / 16, span / 16, total / 16, 1, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * span * reps;
    printf("%-66s : %7.2f TB/s  (%.1f us per pass)\n", name, bytes / ms * 1e-9, ms / reps * 1e3);
    fflush(stdout);
}

int main() {
    const size_t total = (size_t)2 << 30;      // 2 GiB
    u32x4* src; unsigned* out;
    if (hipMalloc(&src, total) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, total);
    hipDeviceSynchronize();
    const size_t KB = 1024, MB = 1024 * 1024;
    // (a) all workgroups re-read the SAME 768 KB (the gate weights of config 5): L2-resident
    run("L2: shared 768 KB, 16-B loads x8 in flight", k_reg<8>, 256, 512, src, 0, 768 * KB, total, out);
    run("L2: shared 768 KB, 16-B loads x16 in flight", k_reg<16>, 256, 512, src, 0, 768 * KB, total, out);
    run("L2: shared 768 KB, 16-B loads x8, 2 workgroups/CU", k_reg<8>, 512, 512, src, 0, 768 * KB, total, out);
    run("L2: shared 768 KB, 16-B loads x8, 256 threads", k_reg<8>, 256, 256, src, 0, 768 * KB, total, out);
    run("L2: shared 768 KB, LDS-DMA x8 in flight", k_dma<8>, 256, 512, src, 0, 768 * KB, total, out);
    run("L2: shared 768 KB, LDS-DMA x16 in flight", k_dma<16>, 256, 512, src, 0, 768 * KB, total, out);
    // (b) per-workgroup private 256 KB, re-read: L2-resident per XCD (32 x 256 KB = 8 MB > 4 MB L2 -> mostly Infinity Cache)
    run("MALL: private 256 KB per workgroup, 16-B loads x8", k_reg<8>, 256, 512, src, 256 * KB, 256 * KB, total, out);
    run("L2: private 64 KB per workgroup, 16-B loads x8", k_reg<8>, 256, 512, src, 64 * KB, 64 * KB, total, out);
    run("L2: private 64 KB per workgroup, LDS-DMA x8", k_dma<8>, 256, 512, src, 64 * KB, 64 * KB, total, out);
    // (c) streaming 2 GiB from HBM, once
    run("HBM: 8 MB per workgroup streamed, 16-B loads x8", k_reg<8>, 256, 512, src, 8 * MB, 8 * MB, total, out);
    run("HBM: 4 MB per workgroup streamed, 2 wg/CU, 16-B loads x8", k_reg<8>, 512, 512, src, 4 * MB, 4 * MB, total, out);
    run("HBM: 8 MB per workgroup, 16-B loads x8, plain (no nt hint)", k_reg<8, false>, 256, 512, src, 8 * MB, 8 * MB, total, out);
    run("HBM: 2 MB per workgroup, 4 wg/CU x 256 thr, x16, plain", k_reg<16, false>, 1024, 256, src, 2 * MB, 2 * MB, total, out);
    // the pool pass's pattern: one short workgroup per 64 KB (32 rows x 2 KB), every byte read once per launch
    run1("HBM: 512 MB, 8192 short workgroups x 64 KB x 256 thr, x16, plain", k_reg<16, false>, 8192, 256, src, 64 * KB, 64 * KB, 512 * MB, out);
    run1("HBM: 512 MB, 8192 short workgroups x 64 KB x 256 thr, x16, nt", k_reg<16, true>, 8192, 256, src, 64 * KB, 64 * KB, 512 * MB, out);
    run1("HBM: 512 MB, 2048 workgroups x 256 KB x 256 thr, x16, nt", k_reg<16, true>, 2048, 256, src, 256 * KB, 256 * KB, 512 * MB, out);
    run1("HBM: 512 MB, 1024 workgroups x 512 KB x 512 thr, x8, nt", k_reg<8, true>, 1024, 512, src, 512 * KB, 512 * KB, 512 * MB, out);
    run1("HBM: 512 MB, 512 workgroups x 1 MB x 512 thr, x8, nt", k_reg<8, true>, 512, 512, src, 1 * MB, 1 * MB, 512 * MB, out);
    run1("HBM: 512 MB, 256 workgroups x 2 MB x 512 thr, x8, nt", k_reg<8, true>, 256, 512, src, 2 * MB, 2 * MB, 512 * MB, out);
    run("HBM: 512 MB region re-streamed (pool_roofline's size), 2 MB per wg, nt", k_reg<8, true>, 256, 512, src, 2 * MB, 2 * MB, 512 * MB, out);
    run("HBM: 512 MB region re-streamed, 2 MB per wg, plain", k_reg<8, false>, 256, 512, src, 2 * MB, 2 * MB, 512 * MB, out);
    run("HBM: 8 MB per workgroup streamed, LDS-DMA x8", k_dma<8>, 256, 512, src, 8 * MB, 8 * MB, total, out);
    return 0;
}
