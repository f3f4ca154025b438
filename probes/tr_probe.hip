// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (used to design the transposing operand reads of
// the weight-gradient kernel).  LDS holds u16 value = element index; lane t supplies byte address 8*t.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // row-major [rows][stride] image: lane t -> row (t>>2)&3 (+4*(t>>4) rows), 8-byte chunk (t&3)
    unsigned addr = (unsigned)(uintptr_t)lds;   // LDS base (32-bit)
    addr += stride_bytes ? ((lane >> 2) * stride_bytes + (lane & 3) * 8) : lane * 8;
    unsigned long long r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)(r >> (16 * j));
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {0, 32, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride_bytes=%d (0: lane t reads bytes 8t..8t+7)\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
