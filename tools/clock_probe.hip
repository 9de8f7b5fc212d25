// Which shader clock do SHORT kernels run at?  One wave per workgroup times a fixed chain of dependent FMAs with clock64() (shader
// cycles) and wall_clock64() (constant 100 MHz): (a) launched into an idle chip, (b) as the last of a train of 200 tiny dependent
// launches, (c) right after 30 ms of full-chip FMA load.  build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>

__global__ void k_probe(float* out, unsigned long long* t, int n) {
    float v = out[threadIdx.x];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) v = fmaf(v, 1.000001f, 0.5f);
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
__global__ void k_tiny(float* out) { out[threadIdx.x] += 1.f; }
__global__ void k_load(float* out, int n) {
    float v = out[threadIdx.x & 63];
    for (int i = 0; i < n; ++i) v = fmaf(v, 1.000001f, 0.5f);
    if (v == 123.f) out[0] = v;
}

static void report(const char* what, unsigned long long* t) {
    unsigned long long h[2];
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-48s %8llu shader cycles in %7.2f us  ->  %.0f MHz\n", what, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
}

int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 4096); hipMemset(out, 0, 4096); hipMalloc(&t, 16);
    const int n = 20000;
    hipDeviceSynchronize(); usleep(200000);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize();
    report("idle chip, one wave", t);
    usleep(200000);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), 0, 0, out, t, n); hipDeviceSynchronize();
    report("idle chip, 256 workgroups x 256 threads", t);
    usleep(200000);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, 0, out);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize();
    report("after a train of 200 tiny launches, one wave", t);
    usleep(200000);
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k_load, dim3(2048), dim3(256), 0, 0, out, 200000);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize();
    report("after ~30 ms of full-chip FMA load, one wave", t);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), 0, 0, out, t, n); hipDeviceSynchronize();
    report("right behind it, 256 workgroups x 256 threads", t);
    return 0;
}
