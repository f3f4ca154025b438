// Probe: HBM/L2 -> LDS fill rate of buffer_load_dwordx4 ... lds on gfx950 as a function of the contiguous
// segment each tile row contributes per stage (64 B = 16 rows per wave-instruction, 128 B = 8 rows, 256 B = 4 rows),
// with the row pitch of a real NHWC operand (512 B) and a working set that stays in L2 / MALL.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(u32x4_t srd, char* lds, unsigned voff, int soff) {
    const unsigned a = (unsigned)(size_t)(lds_ptr_t)lds;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(a), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int SEG, int LPW, int NST>
__global__ __launch_bounds__(256) void fill(const char* buf, unsigned bytes, int iters, int pitch, int rows_total, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[NST * LPW * 4 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long b = (unsigned long long)buf;
    u32x4_t srd; srd.x = (unsigned)b; srd.y = (unsigned)(b >> 32) & 0xffffu; srd.z = bytes; srd.w = 0x00020000u;
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per wave-instruction
    const int row0 = (blockIdx.x * 97) % (rows_total - LPW * 4 * RPI);
    unsigned voff[LPW];
    for (int i = 0; i < LPW; ++i) voff[i] = (unsigned)((row0 + (i * 4 + wid) * RPI + lane / LPR) * pitch + (lane % LPR) * 16);
    int segs = pitch / SEG;
    for (int it = 0; it < iters; ++it) {
        char* st = smem + (it % NST) * (LPW * 4 * 1024);
        const int soff = (it % segs) * SEG;
#pragma unroll
        for (int i = 0; i < LPW; ++i) dma16(srd, st + (i * 4 + wid) * 1024, voff[i], soff);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 1) * LPW) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && sink) sink[blockIdx.x] = *(float*)smem;
}
template <int SEG, int LPW, int NST>
void run(const char* name, char* d, unsigned bytes, int pitch, int rows, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((fill<SEG, LPW, NST>), dim3(blocks), dim3(256), 0, 0, d, bytes, iters, pitch, rows, (float*)nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gb = (double)blocks * iters * LPW * 4 * 1024 / 1e9;
    printf("%-34s blocks %5d  %8.1f GB/s  (%.1f B/clk/CU @2.4GHz)\n", name, blocks, gb / (ms * 1e-3), gb / (ms * 1e-3) / 256 / 2.4);
}
int main() {
    const int pitch = 512, rows = 65536;             // 32 MB operand: L2/MALL resident after the first pass
    const unsigned bytes = (unsigned)pitch * rows;
    char* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
    for (int blocks : {256, 512, 1024}) {
        run<64, 4, 4>("seg 64B  16KB/stage 4 stages", d, bytes, pitch, rows, blocks);
        run<128, 4, 4>("seg 128B 16KB/stage 4 stages", d, bytes, pitch, rows, blocks);
        run<256, 4, 4>("seg 256B 16KB/stage 4 stages", d, bytes, pitch, rows, blocks);
        run<128, 8, 2>("seg 128B 32KB/stage 2 stages", d, bytes, pitch, rows, blocks);
        run<128, 8, 4>("seg 128B 32KB/stage 4 stages", d, bytes, pitch, rows, blocks);
        run<64, 2, 8>("seg 64B   8KB/stage 8 stages", d, bytes, pitch, rows, blocks);
        run<64, 2, 4>("seg 64B   8KB/stage 4 stages", d, bytes, pitch, rows, blocks);
    }
    return 0;
}
