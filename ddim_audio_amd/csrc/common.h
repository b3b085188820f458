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
// 16-byte non-temporal load / store (`global_load/store_dwordx4 ... nt`): for tensors that are streamed once and are larger than the
// 256 MiB Infinity Cache, so that they do not push lines that WILL be re-read out of it (the element-wise passes of a training step
// at 32 samples: -5 % per pass, profiles/r04/nt/)
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load16(const void* p) {
    const nt_u32x4 v = __builtin_nontemporal_load((const nt_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store16(void* p, const uint4 v) {
    nt_u32x4 w;
    w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, (nt_u32x4*)p);
}
// host side: does one tensor of an element-wise pass exceed the Infinity Cache?  (DDIMX_NT=0 turns the non-temporal paths off: A/B)
static inline int nt_streaming(size_t tensor_bytes) {
    static const int on = getenv("DDIMX_NT") ? atoi(getenv("DDIMX_NT")) : 1;
    return on && tensor_bytes > ((size_t)256 << 20) ? 1 : 0;
}
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


// ---- packed-f32 pairs: arithmetic on float2 vectors lowers to v_pk_mul/add/fma_f32 (one VALU issue per two
// elements).  The element-wise prologue/epilogue of the conv kernels is VALU-issue bound, so this matters.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ f32x2_t silu2(f32x2_t v) {
    const f32x2_t t = v * (-1.4426950408889634f);
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2_t d = e + 1.0f;
    f32x2_t r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return v * r;
}
// d/du SiLU(u) = s (1 + u (1 - s)), s = sigmoid(u)   (same arithmetic as train_kernels.hip's dsilu_f, two lanes)
__device__ __forceinline__ f32x2_t dsilu2(f32x2_t u) {
    const f32x2_t t = u * (-1.4426950408889634f);
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2_t d = e + 1.0f;
    f32x2_t sg;
    sg.x = __builtin_amdgcn_rcpf(d.x);
    sg.y = __builtin_amdgcn_rcpf(d.y);
    f32x2_t r;
    r.x = sg.x * fmaf(u.x, 1.0f - sg.x, 1.0f);
    r.y = sg.y * fmaf(u.y, 1.0f - sg.y, 1.0f);
    return r;
}
__device__ __forceinline__ f32x2_t fma2(f32x2_t a, f32x2_t b, f32x2_t c) {
    return __builtin_elementwise_fma(a, b, c);
}

template <typename T> struct Pairs;  // 16-byte piece <-> float2 pairs
template <> struct Pairs<float> {
    static constexpr int N = 2;
    static __device__ __forceinline__ void unpack(const uint4& v, f32x2_t* f) {
        f[0].x = __uint_as_float(v.x); f[0].y = __uint_as_float(v.y);
        f[1].x = __uint_as_float(v.z); f[1].y = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const f32x2_t* f) {
        return make_uint4(__float_as_uint(f[0].x), __float_as_uint(f[0].y), __float_as_uint(f[1].x), __float_as_uint(f[1].y));
    }
};
template <> struct Pairs<__bf16> {
    static constexpr int N = 4;
    static __device__ __forceinline__ f32x2_t up(uint32_t w) {
        f32x2_t r;
        r.x = __uint_as_float(w << 16);
        r.y = __uint_as_float(w & 0xffff0000u);
        return r;
    }
    static __device__ __forceinline__ void unpack(const uint4& v, f32x2_t* f) {
        f[0] = up(v.x); f[1] = up(v.y); f[2] = up(v.z); f[3] = up(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const f32x2_t* f) {
        return make_uint4(Piece<__bf16>::pk(f[0].x, f[0].y), Piece<__bf16>::pk(f[1].x, f[1].y),
                          Piece<__bf16>::pk(f[2].x, f[2].y), Piece<__bf16>::pk(f[3].x, f[3].y));
    }
};


// ---- buffer resources: hardware bounds check (offset >= num_records loads zeros / drops the store), so
// padding and ragged tile edges need no branches around memory instructions ------------------------------
typedef __attribute__((vector_size(16))) unsigned int u32x4_t;
constexpr unsigned kOOB = 0xFFFFFFFFu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned off, const uint4& v) {
    const u32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, 0);
}


// ---- LDS-DMA (global -> LDS, 16 B per lane, no VGPR destination).  Written as inline asm on purpose: with the
// builtin, hipcc orders every later ds_read behind the pending DMA with s_waitcnt vmcnt(0) (it cannot prove the
// addresses disjoint), which serialises the ring.  As asm the instruction is invisible to the waitcnt pass; the
// caller waits with a counted s_waitcnt vmcnt(N) and a barrier before any wave reads the bytes
// (cdna_hip_programming.md section 5.7).  lds_addr: wave-uniform LDS byte address (lane i lands at +16*i).
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

// Sum over aligned groups of 8 or 16 adjacent lanes by DPP (quad swaps, then the half-row / row mirrors): four VALU
// instructions instead of ds_bpermute round trips; every lane of a group ends with the same bits (each step adds two values
// that are uniform over the sub-groups it joins).  G = 2: the lane pair (l, l + 32).
template <int CTRL> __device__ __forceinline__ float dpp_mov_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int G> __device__ __forceinline__ float group_sum(float v) {
    static_assert(G == 2 || G == 8 || G == 16, "lane group");
    if constexpr (G == 2) {
        return v + __shfl_xor(v, 32, 64);
    } else {
        v += dpp_mov_f<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_mov_f<0x4E>(v);   // quad_perm [2,3,0,1]
        v += dpp_mov_f<0x141>(v);  // row_half_mirror
        if constexpr (G == 16) v += dpp_mov_f<0x140>(v);  // row_mirror
        return v;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace ddimx
