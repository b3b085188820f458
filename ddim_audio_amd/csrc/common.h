// Shared device/host helpers for libddimx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddimx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kGroups = 8;  // GroupNorm groups everywhere on this path (reference models/diffusion.py:19)

// dtype codes of the C ABI (include/ddimx.h)
enum { DT_F32 = 0, DT_BF16 = 1 };

__host__ __device__ constexpr int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

__device__ __forceinline__ float silu_f(float v) {
    // x * sigmoid(x); v_exp_f32 / v_rcp_f32 are ~1 ulp, far inside the fp32 parity tolerance
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

__device__ __forceinline__ float gelu_new_f(float v) {
    // 0.5 v (1 + tanh(sqrt(2/pi) (v + 0.044715 v^3)))  (transformers activations.py:59-66)
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return 0.5f * v * (1.0f + tanhf(u));
}

// ---- 16-byte piece <-> floats ------------------------------------------------------------------
template <typename T> struct Piece;  // a 16-byte run of elements
template <> struct Piece<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Piece<__bf16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint32_t pk(float lo, float hi) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
        bf2 r = {(__bf16)lo, (__bf16)hi};  // round-to-nearest-even (v_cvt_pk_bf16_f32)
        return __builtin_bit_cast(uint32_t, r);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
    }
};

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace ddimx
