// Device helpers shared by the MFMA kernels of the conv family (igemm.hip, bottleneck.hip): MFMA wrappers per element type,
// buffer descriptors, the inline-asm LDS-DMA wave instruction, counted s_waitcnt and the raw barrier.
#pragma once
#include "common.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB_OFF = 0x80000000u;   // >= any buffer extent we accept: the load returns zeros

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ void raw_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from buffer offsets `voff` (+ scalar `soff`) to LDS bytes
// [lds_wave_base, +1024).  Issued through inline asm on purpose: a compiler-visible LDS-DMA makes hipcc put
// s_waitcnt vmcnt(0) in front of every ds_read of the loop (it cannot disambiguate the ring slots), which
// serialises the pipeline; hidden from it, the ring is ordered by our own counted vmcnt + s_barrier.
// M0 (the DMA's LDS base) is compiler-reserved: saved, set and restored inside the one statement.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t make_srd(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    u32x4_t r;
    r.x = (unsigned)b; r.y = (unsigned)(b >> 32) & 0xffffu; r.z = bytes; r.w = 0x00020000u;   // raw buffer, stride 0
    return r;
}
__device__ __forceinline__ void dma16(u32x4_t srd, char* lds_wave_base, unsigned voff, int soff) {
    const unsigned lds_addr = (unsigned)(size_t)(lds_ptr_t)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
                 : "memory");
}
// 16 bytes per lane to buffer offset `voff`; a lane with voff = OOB_OFF stores nothing (range-checked raw buffer) -- a per-lane condition
// without a divergent branch (behind one, hipcc no longer proves the operands of the LDS-DMA statements that follow wave-uniform).  The
// s_nop covers the hazard of overwriting the data registers of a store wider than 64 bits right after it (the compiler does not see it).
__device__ __forceinline__ void bstore16(u32x4_t srd, unsigned voff, uint4 v) {
    u32x4_t t, r;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    r.x = __builtin_amdgcn_readfirstlane(srd.x); r.y = __builtin_amdgcn_readfirstlane(srd.y);      // (uniform already; spelled out for the compiler)
    r.z = __builtin_amdgcn_readfirstlane(srd.z); r.w = __builtin_amdgcn_readfirstlane(srd.w);
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(t), "v"(voff), "s"(r) : "memory");
}

}  // namespace
