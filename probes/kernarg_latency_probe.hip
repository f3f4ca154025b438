// Probe: what does a block pay before its first kernel-argument-dependent instruction?  A kernel with a ConvK-sized by-value argument stamps the shader
// clock at entry (s_memtime needs no argument), after its first use of an argument (the scalar loads of the kernarg segment have returned), and after
// ~600 dependent VALU instructions (the length of conv_igemm_dma_kernel's prologue) -- launched the way the train step launches (back to back, and inside
// a hipGraph).   hipcc --offload-arch=gfx950 -O3 -std=c++17 kernarg_latency_probe.hip -o kernarg_latency_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Big { unsigned long long* dbg; int v[80]; };      // 328 bytes, like msc_conv::ConvK

__global__ __launch_bounds__(512) void probe(Big p) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    int x = p.v[3] + (int)threadIdx.x;                   // first use of the kernarg segment
    asm volatile("" : "+v"(x));
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 150; ++i) { x = x * 3 + p.v[i & 63]; x ^= x >> 3; x += i; x = x * 5 + 1; }      // ~600 dependent VALU
    asm volatile("" : "+v"(x));
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        p.dbg[blockIdx.x * 4 + 0] = t0; p.dbg[blockIdx.x * 4 + 1] = t1; p.dbg[blockIdx.x * 4 + 2] = t2; p.dbg[blockIdx.x * 4 + 3] = (unsigned long long)x;
    }
}

int main() {
    const int blocks = 256;
    Big b = {};
    hipMalloc(&b.dbg, blocks * 4 * 8 * 64);
    for (int i = 0; i < 80; ++i) b.v[i] = i * 7 + 1;
    std::vector<unsigned long long> h(blocks * 4);
    auto report = [&](const char* what) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), b.dbg, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> a(blocks), c(blocks);
        unsigned long long e0 = ~0ull, e1 = 0;
        for (int i = 0; i < blocks; ++i) { a[i] = h[i * 4 + 1] - h[i * 4]; c[i] = h[i * 4 + 2] - h[i * 4 + 1]; e0 = std::min(e0, h[i * 4]); e1 = std::max(e1, h[i * 4]); }
        std::sort(a.begin(), a.end()); std::sort(c.begin(), c.end());
        printf("%-34s entry -> first argument use: median %5llu clk (min %5llu, max %5llu) | 600 dependent VALU: median %5llu clk | block entries spread over %6llu clk\n", what, a[blocks / 2], a[0],
               a[blocks - 1], c[blocks / 2], e1 - e0);
    };
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, b); }
    report("eager, third of three launches");
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, st, b);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    report("hipGraph, last of 8 kernel nodes");
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    hipEventRecord(ea, st); for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, st); hipEventRecord(eb, st); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb);
    printf("graph of 8 probe kernels: %.2f us per kernel\n", ms * 1e3f / 160);
    return 0;
}
