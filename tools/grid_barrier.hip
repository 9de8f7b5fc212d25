// Cost of a grid-wide barrier between dependent phases inside ONE persistent kernel on MI355X (would a single token-side
// kernel beat a chain of ~100 launches of 5-6 us each?).  G workgroups of 256 threads, all resident; per phase every
// workgroup does a little memory work (reads 16 KB written by the others in the previous phase, writes 4 KB) and then crosses
// a sense-reversing barrier built on one atomic counter in device memory.
// build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o tools/grid_barrier ; run: ./tools/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_sync(unsigned* counter, unsigned nwg, unsigned phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                     // release this workgroup's writes
        atomicAdd(counter, 1u);
        const unsigned target = nwg * (phase + 1);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_phases(float* buf, unsigned* counter, int phases, int work) {
    const unsigned nwg = gridDim.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        if (work) {
            const float* src = buf + (size_t)((p & 1) * nwg + (blockIdx.x * 7 + p) % nwg) * 1024;
            for (int i = threadIdx.x; i < 1024; i += 256) acc += src[i] + src[(i + 512) & 1023];
            buf[(size_t)(((p + 1) & 1) * nwg + blockIdx.x) * 1024 + threadIdx.x] = acc;
        }
        grid_sync(counter, nwg, (unsigned)p);
    }
    if (acc == 123.456f) buf[0] = acc;
}

int main() {
    for (int G : {32, 64, 128, 256}) {
        for (int work : {0, 1}) {
            float* buf; unsigned* counter;
            hipMalloc(&buf, (size_t)2 * G * 1024 * 4);
            hipMemset(buf, 0, (size_t)2 * G * 1024 * 4);
            hipMalloc(&counter, 4);
            const int phases = 200;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(counter, 0, 4);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_phases, dim3(G), dim3(256), 0, 0, buf, counter, phases, work);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("G = %3d workgroups, work %d: %.2f us per phase\n", G, work, best * 1e3f / phases);
            hipFree(buf); hipFree(counter);
        }
    }
    return 0;
}
