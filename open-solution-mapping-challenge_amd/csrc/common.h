// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the U-Net hot path.
// wave = 64 lanes; MFMA fragments as documented for gfx950:
//   v_mfma_f32_16x16x32_bf16 : A lane l -> row l&15, k = 8*(l>>4)+j (j<8); B lane l -> col l&15, same k
//   v_mfma_f32_16x16x4_f32   : A lane l -> row l&15, k = l>>4      ; B lane l -> col l&15, same k
//   C/D (both)               : col = l&15, row = 4*(l>>4)+r (r<4)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define MSC_OK 0
#define MSC_ERR_ARG (-1)
#define MSC_ERR_HIP (-2)
#define MSC_ERR_UNSUPPORTED (-3)

// dtype enum of the C ABI (include/msc.h)
#define MSC_F32 0
#define MSC_BF16 1
#define MSC_F16 2
static inline bool msc_dtype_ok(int dtype) { return dtype == MSC_F32 || dtype == MSC_BF16 || dtype == MSC_F16; }
static inline int msc_dtype_size(int dtype) { return dtype == MSC_F32 ? 4 : 2; }
static inline int msc_dtype_vec(int dtype) { return 16 / msc_dtype_size(dtype); }      // elements per 16-byte vector

// Per-channel BatchNorm sums are accumulated with fp32 atomics into one slot per XCD ([MSC_BN_SLOTS][C][2] floats, zeroed by the
// caller): every contribution to an address comes from ONE XCD, so the line stays in that XCD's L2 (a same-address atomic
// from different XCDs migrates the line through the fabric: 5-7x slower, probes/xcd_atomic_probe.hip) and the consumer's
// prologue reads 8 values per channel instead of launching a reduction.  The slot is the hardware XCC id: any workgroup ->
// XCD placement gives the right total.
#define MSC_BN_SLOTS 8
#ifdef __HIPCC__
__device__ __forceinline__ int msc_xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (MSC_BN_SLOTS - 1); }   // HW_REG_XCC_ID[3:0]
#endif

extern thread_local char msc_err_buf[512];
int msc_fail(int code, const char* fmt, ...);
int msc_check_launch(const char* what);

typedef uint16_t bf16_t;  // raw bfloat16 bits
struct f16_t { uint16_t bits; };   // raw IEEE binary16 bits (a distinct type, so the kernel templates can tell the two apart)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16, round to nearest even (NaN stays NaN): gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction); the integer sequence it replaces was 5-6 VALU operations per element in every epilogue
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_hw){lo, hi}, bf16x2_hw));
}

__device__ __forceinline__ float f16_to_f32(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.bits); }
__device__ __forceinline__ f16_t f32_to_f16(float f) {      // round to nearest even, overflow -> inf (v_cvt_f16_f32)
    f16_t r;
    r.bits = __builtin_bit_cast(uint16_t, (_Float16)f);
    return r;
}
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    return (uint32_t)f32_to_f16(lo).bits | ((uint32_t)f32_to_f16(hi).bits << 16);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<f16_t> {
    static __device__ __forceinline__ float load(const f16_t* p) { return f16_to_f32(*p); }
    static __device__ __forceinline__ void store(f16_t* p, float v) { *p = f32_to_f16(v); }
    static __device__ __forceinline__ f16_t from(float v) { return f32_to_f16(v); }
};
template <> struct ElemIO<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct ElemIO<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ bf16_t from(float v) { return f32_to_bf16(v); }
};

// One 16-byte global store per lane, the form every tensor-producing kernel writes its output with.  MSC_STORE_WT selects the cache
// policy (MI355X_MICROARCH.md, "stores of each flavour"): 0 = plain (default), 1 = sc1 (write-through, line dropped from the L2), 2 =
// sc0 sc1, 3 = nt.  Measured (round 4, profiles/r4_run8_store_policy_ab.txt): write-through avoids the write-back burst at the kernel
// boundary but LOSES -- train step 11.09 -> 11.40 ms, forward 2.29 -> 2.45 ms: the boundary's write-back leaves the lines clean in the
// producing XCD's L2, where the XCD-aware tile order lets the next kernel's blocks find them; sc1 drops them.
#ifndef MSC_STORE_WT
#define MSC_STORE_WT 0
#endif
typedef unsigned msc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* p, uint4 v) {
#if MSC_STORE_WT == 0
    *reinterpret_cast<uint4*>(p) = v;
#else
    msc_u32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
#if MSC_STORE_WT == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
#elif MSC_STORE_WT == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(t) : "memory");
#endif
#endif
}

// Training-mode BatchNorm2d coefficients of one channel from the per-XCD partial sums (double slots[MSC_BN_SLOTS][C][2] = sum, sum of squares):
// batch mean / biased variance -> scale = gamma * invstd, shift = beta - mean * scale.  `publish` (one block per launch): the coefficients,
// save_mean / save_invstd for the backward and the running statistics (unbiased variance) go to memory.  Shared by msc_bn_apply and by the
// convolution that applies a BatchNorm to its INPUT on load (msc_conv_desc.in_bn): the two paths produce the same coefficients.
struct BnFwdFin {
    const double* slots; double count; const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; float* scale; float* shift; float* save_mean; float* save_invstd;
};
__device__ __forceinline__ void bn_fwd_coeffs(const BnFwdFin& f, int C, int c, bool publish, float& sc, float& sh) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int x = 0; x < MSC_BN_SLOTS; ++x) {
        const double2 v = *reinterpret_cast<const double2*>(f.slots + ((long)x * C + c) * 2);
        s1 += v.x; s2 += v.y;
    }
    const double mean = s1 / f.count;
    double var = s2 / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    sc = g * invstd;
    sh = b - (float)mean * sc;
    if (publish) {
        f.scale[c] = sc;
        f.shift[c] = sh;
        if (f.save_mean) f.save_mean[c] = (float)mean;
        if (f.save_invstd) f.save_invstd[c] = invstd;
        if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
        if (f.running_var) {
            const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
        }
    }
}

// 16-byte vector of T as floats: 4 x f32 or 8 x bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* v) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void unpack(uint4 t, float* v) {      // a raw 16-byte load, converted later
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* v) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
    static __device__ __forceinline__ void store(float* p, const float* v) { store16(p, pack(v)); }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(uint4 t, float* v) {
        uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) { unpack(*reinterpret_cast<const uint4*>(p), v); }
    static __device__ __forceinline__ uint4 pack(const float* v) {
        uint4 t;
        t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]);
        t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
        return t;
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) { store16(p, pack(v)); }
};

template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(uint4 t, float* v) {
        uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x2 h = __builtin_bit_cast(f16x2, w[i]);
            v[2 * i] = (float)h[0];
            v[2 * i + 1] = (float)h[1];
        }
    }
    static __device__ __forceinline__ void load(const f16_t* p, float* v) { unpack(*reinterpret_cast<const uint4*>(p), v); }
    static __device__ __forceinline__ uint4 pack(const float* v) {
        uint4 t;
        t.x = pack_f16x2(v[0], v[1]); t.y = pack_f16x2(v[2], v[3]);
        t.z = pack_f16x2(v[4], v[5]); t.w = pack_f16x2(v[6], v[7]);
        return t;
    }
    static __device__ __forceinline__ void store(f16_t* p, const float* v) { store16(p, pack(v)); }
};

// Sum over the 16 lanes of a DPP row (the lanes that share lane >> 4: the 16 pixels of an MFMA fragment column), result in every lane: four VALU adds with
// DPP operands (two quad permutes, two row rotations).  __shfl_xor compiles to ds_bpermute_b32 -- a round trip through the LDS crossbar per value and step;
// the statistics epilogue of the convolutions did 128 of them per wave (round 6: +2-3 us on a 12 us layer3 launch, tools/conv_cfg_table.py --stats).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));       // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));       // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));      // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));      // row_ror:8
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
