// Shader clock seen by a small kernel launched into an otherwise idle GPU, the way the one-hop streaming step is:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/idle_clock.hip -o idle_clock && ./idle_clock
// s_memtime counts shader cycles, s_memrealtime a constant 100 MHz: their ratio over a loop of dependent adds is the
// clock the wave ran at.  Mode "spaced": one 1-wave kernel every ~100 us (the streaming pattern); "busy": the same kernel
// right after 50 ms of a chip-filling load.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void probe(unsigned long long* out, int iters)
{
    float x = threadIdx.x * 1e-3f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (x == 123.456f) out[2] = 1;
}
__global__ void load(float* buf, int iters)
{
    float x = buf[threadIdx.x];
    for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
    buf[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 64); float* b; hipMalloc(&b, 4 * 4096 * 256);
    unsigned long long h[3];
    auto run = [&](const char* name) {
        double mhz = 0; double us = 0;
        for (int i = 0; i < 200; i++) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 20000);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            mhz += 100.0 * (double) h[0] / (double) h[1]; us += (double) h[1] / 100.0;
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        printf("%s: shader clock %.0f MHz over a %.0f us probe\n", name, mhz / 200, us / 200);
    };
    run("idle, spaced launches");
    for (int k = 0; k < 20; k++) hipLaunchKernelGGL(load, dim3(4096), dim3(256), 0, 0, b, 200000);
    hipDeviceSynchronize();
    run("right after a load  ");
    // probe while a load is running on another stream
    hipStream_t s2; hipStreamCreate(&s2);
    for (int k = 0; k < 40; k++) hipLaunchKernelGGL(load, dim3(2048), dim3(256), 0, s2, b, 200000);
    run("beside a load       ");
    hipDeviceSynchronize();
    return 0;
}
