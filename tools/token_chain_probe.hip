// What would ONE persistent kernel make of the token-side chain?  A chain of P dependent layers y = W_p x (32 rows, 512 -> 512,
// fp32; 16 distinct weight matrices = 16 MB cycling, so they come from L2 / Infinity Cache like the model's 38 MB of weights)
// run (a) as P kernel launches captured in a hipGraph and (b) as one kernel of G workgroups x 512 threads that crosses a
// grid barrier (one atomic counter, sense by phase number) between layers.  Prints us per layer.
// build: hipcc --offload-arch=gfx950 -O3 tools/token_chain_probe.hip -o tools/token_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ROWS 32
#define DIM 512

// columns [c0, c0 + nc) of y for all 32 rows: thread (row r = tid & 31, column group cg = tid >> 5): 16 groups
__device__ __forceinline__ void layer_cols(const float* __restrict__ x, const float* __restrict__ W, float* __restrict__ y,
                                           int c0, int nc, int tid) {
    const int r = tid & 31, cg = tid >> 5;
    for (int c = c0 + cg; c < c0 + nc; c += 16) {
        const float4* xr = reinterpret_cast<const float4*>(x + r * DIM);
        const float4* wr = reinterpret_cast<const float4*>(W + (size_t)c * DIM);
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < DIM / 4; ++k) {
            const float4 a = xr[k], b = wr[k];
            acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
        y[r * DIM + c] = acc * 0.04f;
    }
}

// work = 0: nothing but the boundary (launch or barrier); 1: a few columns per workgroup (2 per 16-thread-group pass)
__global__ __launch_bounds__(512) void k_layer(const float* x, const float* W, float* y, int cols_per_wg, int work) {
    if (work) layer_cols(x, W, y, blockIdx.x * cols_per_wg, work == 2 ? cols_per_wg : (cols_per_wg < 16 ? cols_per_wg : 16), threadIdx.x);
}

__device__ __forceinline__ void grid_sync(unsigned* counter, unsigned nwg, unsigned phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        const unsigned target = nwg * (phase + 1);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void k_chain(float* xa, float* xb, const float* Wall, int P, unsigned* counter, int work) {
    const unsigned G = gridDim.x;
    const int cols = DIM / G;
    for (int p = 0; p < P; ++p) {
        const float* x = (p & 1) ? xb : xa;
        float* y = (p & 1) ? xa : xb;
        // the activations were written by other workgroups in the previous phase: read them past the (non-coherent) L1
        if (work) layer_cols(x, Wall + (size_t)(p & 15) * DIM * DIM, y, blockIdx.x * cols, work == 2 ? cols : (cols < 16 ? cols : 16), threadIdx.x);
        grid_sync(counter, G, (unsigned)p);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every thread: drop stale L1 lines of the activations
    }
}

int main() {
    float *xa, *xb, *W; unsigned* counter;
    hipMalloc(&xa, ROWS * DIM * 4); hipMalloc(&xb, ROWS * DIM * 4); hipMalloc(&W, (size_t)16 * DIM * DIM * 4); hipMalloc(&counter, 4);
    std::vector<float> h(ROWS * DIM, 0.5f), hw((size_t)16 * DIM * DIM, 0.01f);
    hipMemcpy(xa, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int P = 64;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 1, 2})
    for (int G : {8, 16, 32, 64}) {
        // (a) graph of P launches
        hipGraph_t graph; hipGraphExec_t exec;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int p = 0; p < P; ++p)
            hipLaunchKernelGGL(k_layer, dim3(G), dim3(512), 0, st, (p & 1) ? xb : xa, W + (size_t)(p & 15) * DIM * DIM, (p & 1) ? xa : xb, DIM / G, work);
        hipStreamEndCapture(st, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        float best_a = 1e9f, best_b = 1e9f;
        for (int rep = 0; rep < 8; ++rep) {
            hipEventRecord(e0, st); hipGraphLaunch(exec, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best_a) best_a = ms;
        }
        for (int rep = 0; rep < 8; ++rep) {
            hipMemsetAsync(counter, 0, 4, st);
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(k_chain, dim3(G), dim3(512), 0, st, xa, xb, W, P, counter, work);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best_b) best_b = ms;
        }
        std::vector<float> out(8);
        hipMemcpy(out.data(), xa, 32, hipMemcpyDeviceToHost);
        printf("work %d  G = %2d: graph of %d launches %.2f us / layer; persistent kernel %.2f us / layer   (y[0] = %g)\n", work, G, P,
               best_a * 1e3 / P, best_b * 1e3 / P, out[0]);
        hipGraphExecDestroy(exec); hipGraphDestroy(graph);
    }
    return 0;
}
