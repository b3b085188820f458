// Non-MFMA kernels of libddimx: HBM-bound edge convolutions, GroupNorm finalisation, the residual
// pass, small dense layers, LayerNorm, sampler / loss / EMA elementwise kernels and weight packing.
// gfx950 only.  Every reduction is a fixed-order tree (no float atomics): results are reproducible.
#include <stdlib.h>
#include "kernels.h"
#include "conv_wreg.h"
#include "gn_fused.h"

namespace ddimx {

// =====================================================================================================
// in-conv: Conv2d(cin -> C0, k3, p1) reading NCHW fp32, writing NHWC T (+ per-channel stats partials)
// =====================================================================================================
constexpr int kInPixPerBlock = 1024;
int conv_in_nparts(int H, int W) { return (H * W + kInPixPerBlock - 1) / kInPixPerBlock; }

template <typename T>
__global__ void __launch_bounds__(256) conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ out,
                                                      float* __restrict__ stats, int cin, int C0, int H, int W,
                                                      int groups) {
    constexpr int EPB = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                       // [cin*9][C0]
    float* red = lds + cin * 9 * C0;       // [4 waves][C0][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OPP = C0 / EPB, PPP = 256 / OPP;
    const int b = blockIdx.y, part = blockIdx.x;
    for (int i = tid; i < cin * 9 * C0; i += 256) {
        const int co = i % C0, r = i / C0;  // r = ci*9 + tap
        wl[i] = w[(size_t)co * cin * 9 + r];
    }
    __syncthreads();
    const int c = tid % OPP, pslot = tid / OPP;
    float s[EPB], q[EPB], bv[EPB];
#pragma unroll
    for (int j = 0; j < EPB; ++j) { s[j] = q[j] = 0.f; bv[j] = bias[c * EPB + j]; }
    const int HW = H * W;
    const float* xb = x + (size_t)b * cin * HW;
    for (int it = 0; it < kInPixPerBlock / PPP; ++it) {
        const int pix = part * kInPixPerBlock + it * PPP + pslot;
        if (pix >= HW) break;
        const int py = pix / W, px = pix % W;
        float acc[EPB];
#pragma unroll
        for (int j = 0; j < EPB; ++j) acc[j] = bv[j];
        for (int ci = 0; ci < cin; ++ci) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int gy = py + k / 3 - 1, gx = px + k % 3 - 1;
                float v = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = xb[(size_t)ci * HW + (size_t)gy * W + gx];
                const float* wr = wl + (ci * 9 + k) * C0 + c * EPB;
#pragma unroll
                for (int j = 0; j < EPB; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
            }
        }
        uint4 pv = Piece<T>::pack(acc);
        Piece<T>::unpack(pv, acc);  // statistics of the stored values
        *(uint4*)(out + ((size_t)b * HW + pix) * C0 + c * EPB) = pv;
#pragma unroll
        for (int j = 0; j < EPB; ++j) { s[j] += acc[j]; q[j] = fmaf(acc[j], acc[j], q[j]); }
    }
    if (stats) {
        for (int o = OPP; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < EPB; ++j) { s[j] += __shfl_xor(s[j], o, 64); q[j] += __shfl_xor(q[j], o, 64); }
        }
        if (lane < OPP) {
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                red[(wave * C0 + c * EPB + j) * 2 + 0] = s[j];
                red[(wave * C0 + c * EPB + j) * 2 + 1] = q[j];
            }
        }
        __syncthreads();
        if (groups) {  // group-format partials (gn_fused.h)
            if (wave == 0) gn_bins_store<4>(red, C0 * 2, C0, 0, C0, stats + ((size_t)b * gridDim.x + part) * kGnSlab, lane);
        } else {
            for (int i = tid; i < C0 * 2; i += 256) {
                const float t = red[i] + red[C0 * 2 + i] + red[2 * C0 * 2 + i] + red[3 * C0 * 2 + i];
                stats[(((size_t)b * gridDim.x + part) * C0) * 2 + i] = t;
            }
        }
    }
}


// ---- fast path (C0 = 32, cin = 2): one lane = one pixel x all 32 output channels.  Input reads are
// coalesced along W, the 576 weights are wave-uniform (scalar loads), each lane stores 64 contiguous bytes.
template <typename T, int C0, int CIN>
__global__ void __launch_bounds__(256) conv_in_fast_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ out,
                                                           float* __restrict__ stats, int H, int W, int groups) {
    constexpr int EPB = Piece<T>::N;
    constexpr int ROWB = C0 * (int)sizeof(T), PCS = ROWB / 16;  // bytes / 16-byte pieces per pixel
    __shared__ float red[4][C0 * 2];  // per-wave (sum, sumsq) per channel
    // per-wave staging tile: a lane computes one pixel (ROWB contiguous bytes), but a store instruction should write
    // contiguous memory across the lanes -> pieces go through LDS (row stride ROWB + 16 keeps the b128 accesses conflict-free)
    __shared__ __attribute__((aligned(16))) char otile[4][64 * (ROWB + 16)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, part = blockIdx.x;
    const int HW = H * W;
    const float* xb = x + (size_t)b * CIN * HW;
    float s[C0], q[C0];
#pragma unroll
    for (int c = 0; c < C0; ++c) s[c] = q[c] = 0.f;
    for (int it = 0; it < kInPixPerBlock / 256; ++it) {
        const int pix0 = part * kInPixPerBlock + it * 256 + wave * 64;  // first pixel of this wave (uniform)
        if (pix0 >= HW) break;
        const bool valid = pix0 + lane < HW;  // ragged last wave: idle lanes still help with the stores below
        const int pix = valid ? pix0 + lane : HW - 1;
        const int py = pix / W, px = pix % W;
        float v[CIN * 9];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int gy = py + k / 3 - 1, gx = px + k % 3 - 1;
                v[ci * 9 + k] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[(size_t)ci * HW + (size_t)gy * W + gx] : 0.f;
            }
        char* my = otile[wave] + lane * (ROWB + 16);
#pragma unroll
        for (int c0 = 0; c0 < C0; c0 += EPB) {
            float acc[EPB];
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                float a = bias[c0 + j];
#pragma unroll
                for (int r = 0; r < CIN * 9; ++r) a = fmaf(v[r], w[(c0 + j) * CIN * 9 + r], a);
                acc[j] = a;
            }
            const uint4 pv = Piece<T>::pack(acc);
            Piece<T>::unpack(pv, acc);
            *(uint4*)(my + (c0 / EPB) * 16) = pv;
            if (valid) {
#pragma unroll
                for (int j = 0; j < EPB; ++j) { s[c0 + j] += acc[j]; q[c0 + j] = fmaf(acc[j], acc[j], q[c0 + j]); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the wave's 64 pixels are consecutive in memory: store them as PCS instructions of 1 KiB contiguous each
        char* obase = (char*)(out + ((size_t)b * HW + pix0) * C0);
        const int npix = HW - pix0 < 64 ? HW - pix0 : 64;
#pragma unroll
        for (int k = 0; k < PCS; ++k) {
            const int idx = k * 64 + lane;           // piece index inside the wave's tile
            const int p = idx / PCS, pc = idx % PCS;
            const uint4 vv = *(const uint4*)(otile[wave] + p * (ROWB + 16) + pc * 16);
            if (p < npix) *(uint4*)(obase + (size_t)idx * 16) = vv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (stats) {
#pragma unroll
        for (int c = 0; c < C0; ++c) {
            const float ts = wave_sum(s[c]), tq = wave_sum(q[c]);
            if (lane == 0) { red[wave][c * 2] = ts; red[wave][c * 2 + 1] = tq; }
        }
        __syncthreads();
        if (groups) {  // group-format partials (gn_fused.h)
            if (wave == 0) gn_bins_store<4>(&red[0][0], C0 * 2, C0, 0, C0, stats + ((size_t)b * gridDim.x + part) * kGnSlab, lane);
        } else if (tid < C0 * 2) {
            stats[(((size_t)b * gridDim.x + part) * C0) * 2 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        }
    }
}

// ---- MFMA path (C0 = 32, cin = 2): the 18-term dot products run on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: A = weights,
// rows = cout; B = im2col of the input, columns = 32 consecutive pixels).  MFMA step i multiplies tap i, its two k are the two
// input channels: lane half h handles channel h (any assignment of the 18 products to (step, half) is valid as long as A and B
// agree), so the tap geometry is a compile-time constant and the channel one per-lane offset; the bias is a tenth step against
// a column of ones (exact).  A wave walks 32-pixel blocks; the nine input loads of the block two ahead are issued before the current
// block is multiplied and stored: the lane-per-pixel kernel above is a chain of {18 loads, wait -- which also waits for the
// previous stores, vmcnt is in-order -- 576 FMAs, LDS, 4 stores} per 64 pixels and reaches 1.7 TB/s of the 8.  Loads are
// unconditional (padding taps read the centre pixel and are zeroed by select): a load under a branch is waited for at once.
template <typename T>
__global__ void __launch_bounds__(256, 4) conv_in_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, T* __restrict__ out,
                                                              float* __restrict__ stats, int H, int W, int groups) {
    constexpr int C0 = 32, CIN = 2, KT = CIN * 9;
    constexpr int ROWB = C0 * (int)sizeof(T), PCS = ROWB / 16;  // bytes / 16-byte pieces per pixel
    constexpr int NBLK = kInPixPerBlock / (4 * 32);              // 32-pixel blocks per wave
    __shared__ float red[4][C0 * 2];
    __shared__ __attribute__((aligned(16))) char otile[4][32 * (ROWB + 16)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, part = blockIdx.x;
    const int HW = H * W;
    const float* xb = x + (size_t)b * CIN * HW;
    float wa[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wa[i] = w[l31 * KT + h * 9 + i];
    const float wbias = h == 0 ? bias[l31] : 0.f;
    const int hoff = h * HW;  // this lane half's input channel (32-bit element offsets against the uniform base xb)
    const int wshift = (W & (W - 1)) == 0 ? __builtin_ctz(W) : -1;  // uniform
    // accumulator register r of a lane holds cout 8*(r/4) + 4*h + r%4 of pixel l31 (32x32 MFMA output layout)
    float s[16], q[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = q[r] = 0.f;
    float nxt[2][9];  // the blocks one and two ahead
    auto load_block = [&](int blk, float (&dst)[9]) __attribute__((always_inline)) {
        const int p = part * kInPixPerBlock + (blk * 4 + wave) * 32 + l31;
        const int pc = p < HW ? p : HW - 1;
        const int py = wshift >= 0 ? pc >> wshift : pc / W, px = pc - py * W;
        const int ctr = hoff + pc;  // the pixel itself: always a valid element
        const bool vy[3] = {py > 0, true, py < H - 1}, vx[3] = {px > 0, true, px < W - 1};
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int dy = i / 3 - 1, dx = i % 3 - 1;
            const bool inb = vy[dy + 1] && vx[dx + 1];
            const float v = xb[inb ? ctr + dy * W + dx : ctr];
            dst[i] = inb ? v : 0.f;
        }
    };
    load_block(0, nxt[0]);
    load_block(1, nxt[1]);
    char* const tile = otile[wave];
#pragma unroll 1
    for (int blk = 0; blk < NBLK; ++blk) {
        const int pix0 = part * kInPixPerBlock + (blk * 4 + wave) * 32;  // first pixel of this block (uniform per wave)
        if (pix0 >= HW) break;
        const bool valid = pix0 + l31 < HW;
        float cur[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { cur[i] = nxt[0][i]; nxt[0][i] = nxt[1][i]; }
        load_block(blk + 2, nxt[1]);  // (past the workgroup's range / the image: clamped addresses, results unused)
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wbias, 1.0f, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 9; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i], cur[i], acc, 0, 0, 0);
        char* my = tile + l31 * (ROWB + 16);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float f[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
            if constexpr (sizeof(T) == 2) {
                const uint32_t lo = Piece<__bf16>::pk(f[0], f[1]), hi = Piece<__bf16>::pk(f[2], f[3]);
                *(uint2*)(my + (8 * qd + 4 * h) * 2) = make_uint2(lo, hi);
                f[0] = __uint_as_float(lo << 16); f[1] = __uint_as_float(lo & 0xffff0000u);  // the values as stored
                f[2] = __uint_as_float(hi << 16); f[3] = __uint_as_float(hi & 0xffff0000u);
            } else {
                *(float4*)(my + (8 * qd + 4 * h) * 4) = make_float4(f[0], f[1], f[2], f[3]);
            }
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[4 * qd + j] += f[j]; q[4 * qd + j] = fmaf(f[j], f[j], q[4 * qd + j]); }
            }
        }
        // LDS operations of one wave execute in order: the writes above are visible to the reads below without a fence (a
        // wavefront-scope fence also waits for the global stores of the previous block)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // the block's 32 pixels are consecutive in memory: PCS / 2 store instructions of 1 KiB contiguous each
        char* obase = (char*)(out + ((size_t)b * HW + pix0) * C0);
        const int npix = HW - pix0 < 32 ? HW - pix0 : 32;
#pragma unroll
        for (int k = 0; k < PCS / 2; ++k) {
            const int idx = k * 64 + lane;           // piece index inside the block
            const int pp = idx / PCS, pc = idx % PCS;
            const uint4 vv = *(const uint4*)(tile + pp * (ROWB + 16) + pc * 16);
            if (pp < npix) *(uint4*)(obase + (size_t)idx * 16) = vv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile has been read before the next block overwrites it
        __builtin_amdgcn_wave_barrier();
    }
    if (stats) {
        // lanes of one half hold the same 16 couts for 32 different pixels: butterflies over the five pixel bits
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { s[r] += __shfl_xor(s[r], o, 64); q[r] += __shfl_xor(q[r], o, 64); }
            if (l31 == 0) {
                const int c = 8 * (r / 4) + 4 * h + (r % 4);
                red[wave][c * 2] = s[r]; red[wave][c * 2 + 1] = q[r];
            }
        }
        __syncthreads();
        if (groups) {  // group-format partials (gn_fused.h)
            if (wave == 0) gn_bins_store<4>(&red[0][0], C0 * 2, C0, 0, C0, stats + ((size_t)b * gridDim.x + part) * kGnSlab, lane);
        } else if (tid < C0 * 2) {
            stats[(((size_t)b * gridDim.x + part) * C0) * 2 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        }
    }
}

hipError_t conv_in_launch(int dtype, const float* x, const float* w, const float* bias, void* out, float* stats, int B,
                          int cin, int C0, int H, int W, hipStream_t s, int groups) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    const int opp = C0 / epb;
    if (C0 % epb || opp > 64 || (opp & (opp - 1)) || (groups && C0 % kGroups)) return hipErrorInvalidValue;
    dim3 grid(conv_in_nparts(H, W), B);
    if (C0 == 32 && cin == 2) {
        static const bool lane_per_pixel = getenv("DDIMX_CONV_IN_VALU") != nullptr;  // A/B hook: the round-1 kernel
        if (lane_per_pixel) {
            if (dtype == DT_BF16)
                hipLaunchKernelGGL((conv_in_fast_kernel<__bf16, 32, 2>), grid, dim3(256), 0, s, x, w, bias, (__bf16*)out, stats, H, W, groups);
            else
                hipLaunchKernelGGL((conv_in_fast_kernel<float, 32, 2>), grid, dim3(256), 0, s, x, w, bias, (float*)out, stats, H, W, groups);
            return hipGetLastError();
        }
        if (dtype == DT_BF16)
            hipLaunchKernelGGL(conv_in_mfma_kernel<__bf16>, grid, dim3(256), 0, s, x, w, bias, (__bf16*)out, stats, H, W, groups);
        else
            hipLaunchKernelGGL(conv_in_mfma_kernel<float>, grid, dim3(256), 0, s, x, w, bias, (float*)out, stats, H, W, groups);
        return hipGetLastError();
    }
    const size_t lds = (size_t)(cin * 9 * C0 + 4 * C0 * 2) * 4;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(conv_in_kernel<__bf16>, grid, dim3(256), lds, s, x, w, bias, (__bf16*)out, stats, cin, C0, H, W, groups);
    else
        hipLaunchKernelGGL(conv_in_kernel<float>, grid, dim3(256), lds, s, x, w, bias, (float*)out, stats, cin, C0, H, W, groups);
    return hipGetLastError();
}

// =====================================================================================================
// out-conv: Conv2d(C0 -> cout, k3, p1) on (a + b) NHWC T, writing NCHW fp32
// =====================================================================================================
constexpr int kOutTH = 8, kOutTW = 32;

template <typename T>
__global__ void __launch_bounds__(256) conv_out_kernel(const T* __restrict__ a, const T* __restrict__ b2,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ out, int C0, int cout, int H, int W,
                                                       int tiles_x, int tiles_y) {
    constexpr int EPB = Piece<T>::N;
    constexpr int IH = kOutTH + 2, IW = kOutTW + 2;
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [IH*IW][C0+1]
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
    const int y0 = ty * kOutTH, x0 = tx * kOutTW;
    const int CPP = C0 / EPB, LS = C0 + 1;
    for (int i = tid; i < IH * IW * CPP; i += 256) {
        const int c = i % CPP, pix = i / CPP;
        const int gy = y0 - 1 + pix / IW, gx = x0 - 1 + pix % IW;
        float f[EPB];
#pragma unroll
        for (int j = 0; j < EPB; ++j) f[j] = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t g = (((size_t)b * H + gy) * W + gx) * C0 + c * EPB;
            float k[EPB];
            Piece<T>::unpack(*(const uint4*)(a + g), f);
            Piece<T>::unpack(*(const uint4*)(b2 + g), k);
#pragma unroll
            for (int j = 0; j < EPB; ++j) f[j] += k[j];  // x + hidden[0], kept in fp32
        }
#pragma unroll
        for (int j = 0; j < EPB; ++j) tile[pix * LS + c * EPB + j] = f[j];
    }
    __syncthreads();
    const int py = tid / kOutTW, px = tid % kOutTW;
    float acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = (o < cout) ? bias[o] : 0.f;
    for (int ci = 0; ci < C0; ++ci) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float v = tile[((py + k / 3) * IW + px + k % 3) * LS + ci];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < cout) acc[o] = fmaf(v, w[((size_t)k * cout + o) * C0 + ci], acc[o]);
        }
    }
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < cout) out[(((size_t)b * cout + o) * H + gy) * W + gx] = acc[o];
    }
}


// ---- fast path (C0 = 32, cout = 2): (a + b) staged once per tile into LDS in the activation dtype with a
// pixel stride of C0*es+16 bytes (conflict-free b128 reads), one lane = one output pixel, weights wave-uniform.
template <typename T, int C0, int COUT>
__global__ void __launch_bounds__(256) conv_out_fast_kernel(const T* __restrict__ a, const T* __restrict__ b2,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int H, int W, int tiles_x,
                                                            int tiles_y) {
    constexpr int EPB = Piece<T>::N, ES = sizeof(T);
    constexpr int IH = kOutTH + 2, IW = kOutTW + 2;
    constexpr int CPP = C0 / EPB, PS = C0 * ES + 16;
    __shared__ __attribute__((aligned(16))) char tile[IH * IW * PS];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
    const int y0 = ty * kOutTH, x0 = tx * kOutTW;
    // halo staging: all loads of the tile are issued before the first use, unconditionally (out-of-image pieces read a clamped
    // address and are zeroed by select) -- under `if (inside) load` every piece was a round trip of its own, six in a row per thread
    constexpr int NPIECE = IH * IW * CPP, NIT = (NPIECE + 255) / 256;
    uint4 va[NIT], vb[NIT];
    bool inb[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i0 = tid + u * 256, i = i0 < NPIECE ? i0 : NPIECE - 1;
        const int c = i % CPP, pix = i / CPP;
        const int gy = y0 - 1 + pix / IW, gx = x0 - 1 + pix % IW;
        inb[u] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        const size_t g = (((size_t)b * H + gyc) * W + gxc) * C0 + c * EPB;
        va[u] = *(const uint4*)(a + g);
        vb[u] = *(const uint4*)(b2 + g);
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = tid + u * 256;
        float f[EPB], k[EPB];
        Piece<T>::unpack(va[u], f);
        Piece<T>::unpack(vb[u], k);
#pragma unroll
        for (int j = 0; j < EPB; ++j) f[j] += k[j];
        uint4 v = Piece<T>::pack(f);
        if (!inb[u]) v = make_uint4(0, 0, 0, 0);
        if (i < NPIECE) *(uint4*)(tile + (i / CPP) * PS + (i % CPP) * 16) = v;
    }
    __syncthreads();
    const int py = tid / kOutTW, px = tid % kOutTW;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias[o];
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {  // w: [tap][cout][cin], 64 contiguous wave-uniform floats per tap
        const char* tp = tile + ((py + k / 3) * IW + px + k % 3) * PS;
        const float* wk = w + k * COUT * C0;
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            float f[EPB];
            Piece<T>::unpack(*(const uint4*)(tp + c * 16), f);
#pragma unroll
            for (int j = 0; j < EPB; ++j)
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = fmaf(f[j], wk[o * C0 + c * EPB + j], acc[o]);
        }
    }
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) out[(((size_t)b * COUT + o) * H + gy) * W + gx] = acc[o];
    }
}

hipError_t conv_out_launch(int dtype, const void* a, const void* b, const float* w, const float* bias, float* out, int B,
                           int C0, int cout, int H, int W, hipStream_t s) {
    if (cout > 4 || C0 % 8) return hipErrorInvalidValue;
    const int tiles_x = (W + kOutTW - 1) / kOutTW, tiles_y = (H + kOutTH - 1) / kOutTH;
    const size_t lds = (size_t)(kOutTH + 2) * (kOutTW + 2) * (C0 + 1) * 4;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    dim3 grid(tiles_x * tiles_y * B);
    if (C0 == 32 && cout == 2) {
        if (dtype == DT_BF16)
            hipLaunchKernelGGL((conv_out_fast_kernel<__bf16, 32, 2>), grid, dim3(256), 0, s, (const __bf16*)a, (const __bf16*)b,
                               w, bias, out, H, W, tiles_x, tiles_y);
        else
            hipLaunchKernelGGL((conv_out_fast_kernel<float, 32, 2>), grid, dim3(256), 0, s, (const float*)a, (const float*)b, w,
                               bias, out, H, W, tiles_x, tiles_y);
        return hipGetLastError();
    }
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(conv_out_kernel<__bf16>, grid, dim3(256), lds, s, (const __bf16*)a, (const __bf16*)b, w, bias,
                           out, C0, cout, H, W, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL(conv_out_kernel<float>, grid, dim3(256), lds, s, (const float*)a, (const float*)b, w, bias, out,
                           C0, cout, H, W, tiles_x, tiles_y);
    return hipGetLastError();
}

// =====================================================================================================
// GroupNorm finalisation: partial (sum, sumsq) slabs -> per-(sample, channel) scale / shift
//   scale = rstd_g * gamma_c ; shift = beta_c - mean_g * scale     (torch.nn.GroupNorm, biased variance)
// =====================================================================================================
// One 256-thread block per (group, sample): every thread sums a strided share of the (sum, sumsq) pairs in fp64 (8-byte loads,
// up to four issued before the first use), then wave butterflies + one LDS exchange (fixed order) give the group's totals.
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ stats, int nparts, int Cs, int C,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mr_out) {
    __shared__ double rs[4], rq[4];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int GS = C / kGroups, reps = Cs / C;
    const int per_part = reps * GS, total = nparts * per_part;
    const float2* base = (const float2*)stats + (size_t)b * nparts * Cs;
    double s = 0.0, q = 0.0;
    for (int i0 = tid; i0 < total; i0 += 256 * 4) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            v[u] = make_float2(0.f, 0.f);
            if (i < total) {
                const int part = i / per_part, r = i - part * per_part;
                const int rep = r / GS;
                v[u] = base[(size_t)part * Cs + rep * C + g * GS + (r - rep * GS)];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if ((tid & 63) == 0) { rs[tid >> 6] = s; rq[tid >> 6] = q; }
    __syncthreads();
    const double S = (rs[0] + rs[1]) + (rs[2] + rs[3]), Q = (rq[0] + rq[1]) + (rq[2] + rq[3]);
    const double mean = S / count;
    double var = Q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
    if (mr_out && tid == 0) {  // saved for the backward pass: [B][groups][2] = (mean, rstd)
        mr_out[((size_t)b * kGroups + g) * 2 + 0] = mf;
        mr_out[((size_t)b * kGroups + g) * 2 + 1] = rf;
    }
    for (int i = tid; i < GS; i += 256) {
        const int c = g * GS + i;
        const float sc = rf * gamma[c];
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = (beta ? beta[c] : 0.f) - mf * sc;
    }
}

hipError_t gn_finalize_launch(const float* stats, int nparts, int Cs, int C, double count, const float* gamma,
                              const float* beta, float eps, float* scale, float* shift, int B, hipStream_t s, float* mr_out) {
    if (C % kGroups || Cs % C) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(kGroups, B), dim3(256), 0, s, stats, nparts, Cs, C, count, gamma, beta,
                       eps, scale, shift, mr_out);
    return hipGetLastError();
}

// The same from group-format partials (gn_fused.h) -- used where a sample has more than kGnFuseMaxParts partials (long
// spectrograms at the shallow levels), so that consumers need not re-read them per workgroup.  One block per sample.
__global__ void __launch_bounds__(1024) gn_finalize_groups_kernel(const GnIn gn, int C, float* __restrict__ scale,
                                                                  float* __restrict__ shift) {
    __shared__ float scr[16 * kGroups * 2];
    const int b = blockIdx.x, tid = threadIdx.x, bd = blockDim.x;
    GnInLoads ld;
    gn_in_issue(gn, b, tid, bd, ld);
    gn_in_reduce(gn, b, tid, bd, ld, scr);
    __syncthreads();
    for (int c = tid; c < C; c += bd) {
        float m, r;
        gn_in_group(gn, scr, bd >> 6, c / (C / kGroups), &m, &r);
        const float sc = r * gn.gamma[c];
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = fmaf(-m, sc, gn.beta ? gn.beta[c] : 0.f);
    }
}
// nthreads: the block size of the consumer this replaces the in-kernel finalisation of (64 .. 1024, a multiple of 64) -- the
// reduction order, and with it every bit of the result, is a function of (gn.np, nthreads)
hipError_t gn_finalize_groups_launch(const GnIn& gn, int C, float* scale, float* shift, int B, int nthreads, hipStream_t s) {
    if (C % kGroups || nthreads < 64 || nthreads > 1024 || nthreads % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_finalize_groups_kernel, dim3(B), dim3(nthreads), 0, s, gn, C, scale, shift);
    return hipGetLastError();
}

// =====================================================================================================
// residual pass: y = x + (h*scale + shift)   or   y = x + h(fp32)
// =====================================================================================================
constexpr int kResidIters = 16;  // at most: 16-byte pieces per thread
static inline int resid_bd(int cpp) { return (cpp % 3 == 0) ? 192 : 256; }
int resid_threads(int dtype, int C) { return resid_bd(C / (dtype == DT_BF16 ? 8 : 4)); }
// Pieces per thread of the element-wise passes over one sample (resid, tensor_stats, the GroupNorm-backward passes):
// 16 where the sample is large, fewer on the deep levels so that a sample still spreads over >= 64 workgroups -- with 16
// the level-5 tensor (8 192 pieces) was two workgroups per sample, each a chain of 16 dependent load round trips.
// A function of the sample's size only (never of the batch): a sample's partial sums do not depend on the batch it is in.
int resid_iters(int dtype, int HW, int C) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    const int cpp = C / epb;
    const long long pieces = (long long)HW * cpp;
    long long it = pieces / ((long long)resid_bd(cpp) * 64);
    if (it < 1) it = 1;
    if (it > kResidIters) it = kResidIters;
    return (int)it;
}
int resid_nparts(int dtype, int HW, int C) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    const int cpp = C / epb;
    const long long pieces = (long long)HW * cpp;
    const int per_block = resid_bd(cpp) * resid_iters(dtype, HW, C);
    return (int)((pieces + per_block - 1) / per_block);
}

// LDS: [R][C*2] per-row channel sums | [C*2] workgroup totals | [4 waves][8][2] floats (fused GroupNorm input)
template <typename T, bool HF32, bool HSILU = false>
__global__ void __launch_bounds__(256) resid_kernel(const T* x, const void* __restrict__ hv,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    T* y, float* __restrict__ stats, int HW, int C, const GnIn gn, int groups, int iters,
                                                    int nt) {
    constexpr int EPB = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [R][C*2]
    const int tid = threadIdx.x, bd = blockDim.x;
    const int CPP = C / EPB;
    const int c = tid % CPP;
    const int b = blockIdx.y, part = blockIdx.x;
    const long long pieces = (long long)HW * CPP;
    float* const scr = red + (bd / CPP) * C * 2 + C * 2;
    float sc[EPB], sh[EPB], s[EPB], q[EPB];
#pragma unroll
    for (int j = 0; j < EPB; ++j) { s[j] = q[j] = 0.f; sc[j] = 1.f; sh[j] = 0.f; }
    // one 16-byte piece of x and of h per thread and iteration
    constexpr int HN = HF32 ? EPB / 4 : 1;
    const size_t sbase = (size_t)b * HW * C;
    auto load = [&](size_t e, uint4& vx, uint4 (&vh)[HN]) __attribute__((always_inline)) {
        vx = nt ? nt_load16(x + e) : *(const uint4*)(x + e);  // (nt: uniform)
        if constexpr (HF32) {
#pragma unroll
            for (int k = 0; k < HN; ++k) vh[k] = *(const uint4*)((const float*)hv + e + 4 * k);
        } else {
            vh[0] = nt ? nt_load16((const T*)hv + e) : *(const uint4*)((const T*)hv + e);
        }
    };
    auto process = [&](size_t e, const uint4& vx, const uint4 (&vh)[HN]) __attribute__((always_inline)) {
        float fx[EPB], fh[EPB];
        Piece<T>::unpack(vx, fx);
        if constexpr (HF32) {
#pragma unroll
            for (int k = 0; k < HN; ++k) {
                fh[4 * k] = __uint_as_float(vh[k].x); fh[4 * k + 1] = __uint_as_float(vh[k].y);
                fh[4 * k + 2] = __uint_as_float(vh[k].z); fh[4 * k + 3] = __uint_as_float(vh[k].w);
            }
        } else {
            Piece<T>::unpack(vh[0], fh);
        }
#pragma unroll
        for (int j = 0; j < EPB; ++j) fx[j] = fx[j] + fmaf(HSILU ? silu_f(fh[j]) : fh[j], sc[j], sh[j]);
        const uint4 pv = Piece<T>::pack(fx);
        if (nt) nt_store16(y + e, pv);
        else *(uint4*)(y + e) = pv;
        Piece<T>::unpack(pv, fx);
#pragma unroll
        for (int j = 0; j < EPB; ++j) { s[j] += fx[j]; q[j] = fmaf(fx[j], fx[j], q[j]); }
    };
    if (!HF32) {
        if (gn.stats) {  // uniform: finish the GroupNorm of h here (gn_fused.h)
            float gam[EPB], bet[EPB];
            GnInLoads ld;
            gn_in_issue(gn, b, tid, bd, ld);
            gn_in_params<EPB>(gn, c * EPB, gam, bet);
            gn_in_reduce(gn, b, tid, bd, ld, scr);
            __syncthreads();
            gn_in_fold<EPB>(gn, scr, bd >> 6, C, c * EPB, gam, bet, sc, sh);
        } else {
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                sc[j] = scale[(size_t)b * C + c * EPB + j];
                sh[j] = shift[(size_t)b * C + c * EPB + j];
            }
        }
    }
    // four iterations' loads are issued together (unconditionally: out-of-range slots re-read piece 0 and are dropped; a load
    // under a branch is waited for at once, and the loop used to be one load round trip + one store acknowledgement per
    // iteration), then the four are processed and stored.  x may alias y: a thread only ever touches its own pieces.
    const long long pc0 = (long long)part * iters * bd + tid;
    for (int it0 = 0; it0 < iters; it0 += 4) {
        uint4 vx[4], vh[4][HN];
        size_t e[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long pc = pc0 + (long long)(it0 + u) * bd;
            ok[u] = it0 + u < iters && pc < pieces;
            e[u] = sbase + (size_t)(ok[u] ? pc : 0) * EPB;
            load(e[u], vx[u], vh[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) process(e[u], vx[u], vh[u]);
    }
    if (stats) {
        const int R = bd / CPP, row = tid / CPP;
#pragma unroll
        for (int j = 0; j < EPB; ++j) {
            red[(row * C + c * EPB + j) * 2 + 0] = s[j];
            red[(row * C + c * EPB + j) * 2 + 1] = q[j];
        }
        __syncthreads();
        float* const chan = red + R * C * 2;  // the workgroup's per-channel totals (group format)
        for (int i = tid; i < C * 2; i += bd) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < R; ++r) t += red[r * C * 2 + i];
            if (groups) chan[i] = t;
            else stats[(((size_t)b * gridDim.x + part) * C) * 2 + i] = t;
        }
        if (groups) {  // group-format partials (gn_fused.h)
            __syncthreads();
            if (tid < 64) gn_bins_store<1>(chan, 0, C, 0, C, stats + ((size_t)b * gridDim.x + part) * kGnSlab, tid);
        }
    }
}

hipError_t resid_launch(int dtype, const void* x, const void* h, int h_f32, const float* scale, const float* shift,
                        void* y, float* stats, int B, int HW, int C, hipStream_t s, const GnIn* gn, int groups) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    if (C % epb || C % kGroups) return hipErrorInvalidValue;
    const int cpp = C / epb, bd = resid_bd(cpp);
    if (bd % cpp) return hipErrorInvalidValue;
    GnIn g = {};
    if (gn) g = *gn;
    if (g.stats && (h_f32 == 1 || g.np > kGnFuseMaxParts)) return hipErrorInvalidValue;
    dim3 grid(resid_nparts(dtype, HW, C), B);
    const int iters = resid_iters(dtype, HW, C);
    const size_t lds = (size_t)(bd / cpp) * C * 2 * 4 + (size_t)C * 2 * 4 + 4 * kGroups * 2 * 4;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
#define DDIMX_RESID(TT, HF)                                                                                     \
    hipLaunchKernelGGL((resid_kernel<TT, HF>), grid, dim3(bd), lds, s, (const TT*)x, h, scale, shift, (TT*)y, stats, HW, C, g, groups, iters, nt)
    const int nt = nt_streaming((size_t)B * HW * C * (dtype == DT_BF16 ? 2 : 4));
    if (h_f32 == 2) {  // training forward: h holds the pre-activation, y = x + SiLU(h)*scale + shift
        if (dtype == DT_BF16)
            hipLaunchKernelGGL((resid_kernel<__bf16, false, true>), grid, dim3(bd), lds, s, (const __bf16*)x, h, scale, shift, (__bf16*)y, stats, HW, C, g, groups, iters, nt);
        else
            hipLaunchKernelGGL((resid_kernel<float, false, true>), grid, dim3(bd), lds, s, (const float*)x, h, scale, shift, (float*)y, stats, HW, C, g, groups, iters, nt);
    } else if (dtype == DT_BF16) { if (h_f32) DDIMX_RESID(__bf16, true); else DDIMX_RESID(__bf16, false); }
    else { if (h_f32) DDIMX_RESID(float, true); else DDIMX_RESID(float, false); }
#undef DDIMX_RESID
    return hipGetLastError();
}


// ---- per-channel statistics of an NHWC tensor (used when a tensor arrives without producer stats) ----
template <typename T>
__global__ void __launch_bounds__(256) tensor_stats_kernel(const T* __restrict__ x, float* __restrict__ stats, int HW,
                                                           int C, int groups, int iters) {
    constexpr int EPB = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int CPP = C / EPB, c = tid % CPP, b = blockIdx.y, part = blockIdx.x;
    const long long pieces = (long long)HW * CPP;
    float s[EPB], q[EPB];
#pragma unroll
    for (int j = 0; j < EPB; ++j) s[j] = q[j] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const long long pc = ((long long)part * iters + it) * bd + tid;
        if (pc >= pieces) break;
        float f[EPB];
        Piece<T>::unpack(*(const uint4*)(x + (size_t)b * HW * C + (size_t)pc * EPB), f);
#pragma unroll
        for (int j = 0; j < EPB; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
    }
    const int R = bd / CPP, row = tid / CPP;
#pragma unroll
    for (int j = 0; j < EPB; ++j) {
        red[(row * C + c * EPB + j) * 2 + 0] = s[j];
        red[(row * C + c * EPB + j) * 2 + 1] = q[j];
    }
    __syncthreads();
    float* const chan = red + R * C * 2;
    for (int i = tid; i < C * 2; i += bd) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < R; ++r) t += red[r * C * 2 + i];
        if (groups) chan[i] = t;
        else stats[(((size_t)b * gridDim.x + part) * C) * 2 + i] = t;
    }
    if (groups) {
        __syncthreads();
        if (tid < 64) gn_bins_store<1>(chan, 0, C, 0, C, stats + ((size_t)b * gridDim.x + part) * kGnSlab, tid);
    }
}
hipError_t tensor_stats_launch(int dtype, const void* x, float* stats, int B, int HW, int C, hipStream_t s, int groups) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    if (C % epb || (groups && C % kGroups)) return hipErrorInvalidValue;
    const int cpp = C / epb, bd = resid_bd(cpp);
    if (bd % cpp) return hipErrorInvalidValue;
    dim3 grid(resid_nparts(dtype, HW, C), B);
    const size_t lds = (size_t)(bd / cpp) * C * 2 * 4 + (size_t)C * 2 * 4;
    if (dtype == DT_BF16) hipLaunchKernelGGL(tensor_stats_kernel<__bf16>, grid, dim3(bd), lds, s, (const __bf16*)x, stats, HW, C, groups, resid_iters(dtype, HW, C));
    else hipLaunchKernelGGL(tensor_stats_kernel<float>, grid, dim3(bd), lds, s, (const float*)x, stats, HW, C, groups, resid_iters(dtype, HW, C));
    return hipGetLastError();
}

// ---- layout converters (test / boundary helpers) -------------------------------------------------------
template <typename T>
__global__ void to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int C, int HW, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = i % C;
        const long long p = (i / C) % HW, b = i / ((long long)C * HW);
        out[i] = from_f<T>(in[(b * C + c) * HW + p]);
    }
}
template <typename T>
__global__ void from_nhwc_kernel(const T* __restrict__ in, float* __restrict__ out, int C, int HW, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long p = i % HW;
        const int c = (i / HW) % C;
        const long long b = i / ((long long)C * HW);
        out[i] = to_f<T>(in[(b * HW + p) * C + c]);
    }
}
hipError_t to_nhwc_launch(int dtype, const float* in, void* out, int B, int C, int HW, hipStream_t s) {
    const long long n = (long long)B * C * HW;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == DT_BF16) hipLaunchKernelGGL(to_nhwc_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, in, (__bf16*)out, C, HW, n);
    else hipLaunchKernelGGL(to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, s, in, (float*)out, C, HW, n);
    return hipGetLastError();
}
hipError_t from_nhwc_launch(int dtype, const void* in, float* out, int B, int C, int HW, hipStream_t s) {
    const long long n = (long long)B * C * HW;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == DT_BF16) hipLaunchKernelGGL(from_nhwc_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)in, out, C, HW, n);
    else hipLaunchKernelGGL(from_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)in, out, C, HW, n);
    return hipGetLastError();
}

// =====================================================================================================
// small dense layer: one wave per output feature, all batch rows at once (weight-bandwidth bound)
// =====================================================================================================
__global__ void __launch_bounds__(256) linear_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ y, int B, int N, int K, int act, int in_silu) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* wr = W + (size_t)n * K;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            const float4 wv = *(const float4*)(wr + k);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (b0 + r < B) {
                    const size_t row = idx ? (size_t)idx[b0 + r] : (size_t)(b0 + r);
                    float4 xv = *(const float4*)(x + row * K + k);
                    if (in_silu) { xv.x = silu_f(xv.x); xv.y = silu_f(xv.y); xv.z = silu_f(xv.z); xv.w = silu_f(xv.w); }
                    acc[r] = fmaf(xv.x, wv.x, acc[r]); acc[r] = fmaf(xv.y, wv.y, acc[r]);
                    acc[r] = fmaf(xv.z, wv.z, acc[r]); acc[r] = fmaf(xv.w, wv.w, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float t = wave_sum(acc[r]);
            if (lane == 0 && b0 + r < B) {
                const float v = t + bias[n];
                y[(size_t)(b0 + r) * N + n] = act ? silu_f(v) : v;
            }
        }
    }
}

hipError_t linear_rows_launch(const float* x, const int64_t* idx, const float* W, const float* bias, float* y, int B,
                              int N, int K, int act_silu, hipStream_t s, int in_silu) {
    if (K % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(linear_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, idx, W, bias, y, B, N, K, act_silu, in_silu);
    return hipGetLastError();
}

// rows of a precomputed [n_timesteps][E] timestep-embedding table: out[b] = table[t[b]]  (eval mode: the MLP of
// models/diffusion.py:110-120 is a pure function of t, and all rows of a sampling step share one t)
__global__ void __launch_bounds__(256) temb_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ t,
                                                          float* __restrict__ out, int E) {
    const int b = blockIdx.y;
    const float4* src = (const float4*)(table + (size_t)t[b] * E);
    float4* dst = (float4*)(out + (size_t)b * E);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < E / 4; i += gridDim.x * 256) dst[i] = src[i];
}
hipError_t temb_gather_launch(const float* table, const int64_t* t, float* out, int B, int E, hipStream_t s) {
    if (E % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(temb_gather_kernel, dim3((E / 4 + 255) / 256, B), dim3(256), 0, s, table, t, out, E);
    return hipGetLastError();
}

// =====================================================================================================
// LayerNorm over rows (two-pass in registers: mean, then centred variance)
// =====================================================================================================
template <typename TX>
__global__ void __launch_bounds__(256) layernorm_kernel(const TX* __restrict__ x, const float* __restrict__ add,
                                                        int add_rows, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ y, int N, int chunk_rows) {
    __shared__ float red[4];
    __shared__ float bc;
    const int m = blockIdx.x, tid = threadIdx.x;
    const TX* xr = x + (size_t)m * N;
    const float* ar = add ? add + (size_t)(m % add_rows) * N : nullptr;
    // every load of the row -- x, the addend, gamma, beta -- is issued before the first use, unconditionally (clamped index,
    // dropped by select): the kernel is a chain of load round trips and barriers, a load under `if (n < N)` is waited for at once
    float v[8], gam[8], bet[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n0 = tid + i * 256, n = n0 < N ? n0 : N - 1;
        const float xv = to_f<TX>(xr[n]), av = (ar ? ar : gamma)[n];  // (no addend: gamma is read in its place and dropped)
        v[i] = xv + (ar ? av : 0.f);
        gam[i] = gamma[n];
        bet[i] = beta[n];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        if (n >= N) v[i] = 0.f;
        s += v[i];
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bc = (red[0] + red[1] + red[2] + red[3]) / (float)N;
    __syncthreads();
    const float mean = bc;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        if (n < N) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    }
    q = wave_sum(q);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) bc = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
    __syncthreads();
    const float rstd = bc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        if (n < N) {
            // chunk_rows > 0: chunk-major output for fnet_dense_kernel, [sample = m / chunk_rows][n / 4][32 rows][4] (fnet_dense.hip)
            const size_t o = chunk_rows > 0 ? (size_t)(m / chunk_rows) * 32 * N + ((size_t)(n / 4) * 32 + m % chunk_rows) * 4 + n % 4
                                            : (size_t)m * N + n;
            y[o] = (v[i] - mean) * rstd * gam[i] + bet[i];
        }
    }
}

hipError_t layernorm_launch(int x_dtype, const void* x, const float* add, int add_rows, const float* gamma,
                            const float* beta, float eps, float* y, int M, int N, hipStream_t s, int chunk_rows) {
    if (N > 2048 || chunk_rows > 32 || (chunk_rows > 0 && (N % 4 || M % chunk_rows))) return hipErrorInvalidValue;
    if (x_dtype == DT_BF16)
        hipLaunchKernelGGL(layernorm_kernel<__bf16>, dim3(M), dim3(256), 0, s, (const __bf16*)x, add, add_rows, gamma, beta,
                           eps, y, N, chunk_rows);
    else
        hipLaunchKernelGGL(layernorm_kernel<float>, dim3(M), dim3(256), 0, s, (const float*)x, add, add_rows, gamma, beta,
                           eps, y, N, chunk_rows);
    return hipGetLastError();
}

// =====================================================================================================
// sampler step kernels (functions/denoising.py:22-43): scalars come from a device table indexed by a
// device step counter, so a captured graph can be replayed for every step.
// =====================================================================================================
__global__ void step_begin_kernel(const float* __restrict__ coef, const int* __restrict__ step, int64_t* __restrict__ t,
                                  int B, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) t[i] = (int64_t)coef[(size_t)step[0] * stride];
}
__global__ void step_end_kernel(int* step) { step[0] += 1; }

hipError_t step_begin_launch(const float* coef, const int* step, int64_t* t, int B, int stride, hipStream_t s) {
    hipLaunchKernelGGL(step_begin_kernel, dim3((B + 63) / 64), dim3(64), 0, s, coef, step, t, B, stride);
    return hipGetLastError();
}
hipError_t step_end_launch(int* step, hipStream_t s) {
    hipLaunchKernelGGL(step_end_kernel, dim3(1), dim3(1), 0, s, step);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) ddim_update_kernel(float* __restrict__ xt, const float* __restrict__ et,
                                                          const float* __restrict__ noise, float* __restrict__ x0,
                                                          const float* __restrict__ coef, const int* __restrict__ step,
                                                          long long n4) {
    const float* c = coef + (size_t)step[0] * 6;
    const float s1 = c[1], s2 = c[2], s3 = c[3], c2 = c[4], c1 = c[5];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 x = ((const float4*)xt)[i];
        const float4 e = ((const float4*)et)[i];
        float xs[4] = {x.x, x.y, x.z, x.w};
        const float es[4] = {e.x, e.y, e.z, e.w};
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (noise) { const float4 z = ((const float4*)noise)[i]; nz[0] = z.x; nz[1] = z.y; nz[2] = z.z; nz[3] = z.w; }
        float p0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // xt.add_(et, alpha=-sqrt(1-at)).div_(sqrt(at))  -> x0 prediction, in place
            const float v = __fdiv_rn(fmaf(es[j], -s1, xs[j]), s2);
            p0[j] = v;
            // xt.mul_(sqrt(at_next)).add_(et, alpha=c2).add_(noise, alpha=c1)
            float u = fmaf(es[j], c2, __fmul_rn(v, s3));
            if (noise) u = fmaf(nz[j], c1, u);
            xs[j] = u;
        }
        ((float4*)x0)[i] = make_float4(p0[0], p0[1], p0[2], p0[3]);
        ((float4*)xt)[i] = make_float4(xs[0], xs[1], xs[2], xs[3]);
    }
}

hipError_t ddim_update_launch(float* xt, const float* et, const float* noise, float* x0, const float* coef,
                              const int* step, long long n, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const long long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(ddim_update_kernel, dim3(blocks), dim3(256), 0, s, xt, et, noise, x0, coef, step, n4);
    return hipGetLastError();
}


// ---- ddpm_steps update (functions/denoising.py:72-90), one pass: coef row = (t, (1/at).sqrt(), (1/at-1).sqrt(),
// atm1.sqrt()*beta_t, (1-beta_t).sqrt()*(1-atm1), 1-at, mask*exp(0.5*log(beta_t))) built on the host with the
// reference's fp32 tensor arithmetic; every product/sum is rounded separately like the eager ops it replaces.
__global__ void __launch_bounds__(256) ddpm_update_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                                          const float* __restrict__ noise, float* __restrict__ x0,
                                                          float* __restrict__ xn, const float* __restrict__ coef,
                                                          const int* __restrict__ step, long long n) {
    const float* c = coef + (size_t)step[0] * 7;
    const float a0 = c[1], a1 = c[2], m1 = c[3], m2 = c[4], den = c[5], sig = c[6];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float xv = x[i];
        float p = __fsub_rn(__fmul_rn(a0, xv), __fmul_rn(a1, e[i]));
        p = fminf(fmaxf(p, -1.0f), 1.0f);
        x0[i] = p;
        const float mean = __fdiv_rn(__fadd_rn(__fmul_rn(m1, p), __fmul_rn(m2, xv)), den);
        xn[i] = __fadd_rn(mean, __fmul_rn(sig, noise[i]));
    }
}
hipError_t ddpm_update_launch(const float* x, const float* e, const float* noise, float* x0, float* xn, const float* coef,
                              const int* step, long long n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(ddpm_update_kernel, dim3(blocks), dim3(256), 0, s, x, e, noise, x0, xn, coef, step, n);
    return hipGetLastError();
}

// ---- q-sample (functions/losses.py:12-13): x = x0*sqrt(a_t) + e*sqrt(1-a_t), separate fp32 roundings ----
__global__ void __launch_bounds__(256) qsample_kernel(const float* __restrict__ x0, const float* __restrict__ e,
                                                      const float* __restrict__ alphas, const int64_t* __restrict__ t,
                                                      float* __restrict__ x, long long per) {
    const int b = blockIdx.y;
    const float a = alphas[t[b]];
    const float sa = __fsqrt_rn(a), sb = __fsqrt_rn(__fsub_rn(1.0f, a));
    const size_t base = (size_t)b * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256)
        x[base + i] = __fadd_rn(__fmul_rn(x0[base + i], sa), __fmul_rn(e[base + i], sb));
}
hipError_t qsample_launch(const float* x0, const float* e, const float* alphas, const int64_t* t, float* x, int B,
                          long long per, hipStream_t s) {
    const int blocks = (int)((per + 255) / 256 < 1024 ? (per + 255) / 256 : 1024);
    hipLaunchKernelGGL(qsample_kernel, dim3(blocks, B), dim3(256), 0, s, x0, e, alphas, t, x, per);
    return hipGetLastError();
}

// ---- loss (functions/losses.py:15-18): per-sample sum of squared error, then batch mean --------------
constexpr int kSqParts = 64;
int sqerr_nparts() { return kSqParts; }
__global__ void __launch_bounds__(256) sqerr_part_kernel(const float* __restrict__ e, const float* __restrict__ o,
                                                         float* __restrict__ partial, long long per) {
    __shared__ float red[4];
    const int b = blockIdx.y, part = blockIdx.x;
    const long long chunk = (per + kSqParts - 1) / kSqParts;
    const long long lo = part * chunk, hi = (lo + chunk < per) ? lo + chunk : per;
    const size_t base = (size_t)b * per;
    float s = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) { const float d = e[base + i] - o[base + i]; s = fmaf(d, d, s); }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[b * kSqParts + part] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sqerr_final_kernel(const float* __restrict__ partial, float* __restrict__ loss, int B) {
    // one wave: loss[b] = sum of parts; loss[B] = mean over the batch
    const int lane = threadIdx.x;
    float tot = 0.f;
    for (int b = 0; b < B; ++b) {
        const float v = wave_sum(lane < kSqParts ? partial[b * kSqParts + lane] : 0.f);
        if (lane == 0) loss[b] = v;
        tot += v;
    }
    if (lane == 0) loss[B] = tot / (float)B;
}
hipError_t sqerr_launch(const float* e, const float* out, float* partial, float* loss_per, int B, long long per,
                        hipStream_t s) {
    hipLaunchKernelGGL(sqerr_part_kernel, dim3(kSqParts, B), dim3(256), 0, s, e, out, partial, per);
    hipLaunchKernelGGL(sqerr_final_kernel, dim3(1), dim3(64), 0, s, partial, loss_per, B);
    return hipGetLastError();
}

// ---- EMA (models/ema.py:16-23): shadow = (1-mu)*p + mu*shadow over all tensors in one launch ----------
constexpr int kEmaBlock = 4096;
int ema_block_elems() { return kEmaBlock; }
__global__ void __launch_bounds__(256) ema_multi_kernel(const long long* __restrict__ shadow_ptrs,
                                                        const long long* __restrict__ param_ptrs,
                                                        const long long* __restrict__ sizes,
                                                        const int* __restrict__ blk_tensor,
                                                        const long long* __restrict__ blk_off, float c_p, float c_s) {
    const int ti = blk_tensor[blockIdx.x];
    float* sh = (float*)shadow_ptrs[ti];
    const float* p = (const float*)param_ptrs[ti];
    const long long n = sizes[ti], off = blk_off[blockIdx.x];
    for (int i = threadIdx.x; i < kEmaBlock; i += 256) {
        const long long k = off + i;
        if (k < n) sh[k] = __fadd_rn(__fmul_rn(c_p, p[k]), __fmul_rn(c_s, sh[k]));
    }
}
hipError_t ema_multi_launch(const long long* shadow_ptrs, const long long* param_ptrs, const long long* sizes,
                            const int* blk_tensor, const long long* blk_off, int nblocks, float mu, hipStream_t s) {
    const float c_p = (float)(1.0 - (double)mu);
    hipLaunchKernelGGL(ema_multi_kernel, dim3(nblocks), dim3(256), 0, s, shadow_ptrs, param_ptrs, sizes, blk_tensor,
                       blk_off, c_p, mu);
    return hipGetLastError();
}


// =====================================================================================================
// training-step tail (runners/diffusion.py:155-173): multi-tensor kernels over pointer tables, one launch each
// =====================================================================================================
// sum of squares of all gradient tensors: per-block partials (fixed order) then one block finishes:
// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_)
__global__ void __launch_bounds__(256) sqnorm_multi_kernel(const long long* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                           const int* __restrict__ blk_tensor, const long long* __restrict__ blk_off,
                                                           float* __restrict__ partial) {
    __shared__ float red[4];
    const int ti = blk_tensor[blockIdx.x];
    const float* g = (const float*)ptrs[ti];
    const long long n = sizes[ti], off = blk_off[blockIdx.x];
    float s = 0.f;
    for (int i = threadIdx.x; i < kEmaBlock; i += 256) {
        const long long k = off + i;
        if (k < n) s = fmaf(g[k], g[k], s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) sqnorm_final_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                           float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = norm;
        out[1] = coef < 1.0f ? coef : 1.0f;
    }
}
hipError_t grad_norm_multi_launch(const long long* ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                                  int nblocks, float max_norm, float* partial, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sqnorm_multi_kernel, dim3(nblocks), dim3(256), 0, s, ptrs, sizes, blk_tensor, blk_off, partial);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, max_norm, out);
    return hipGetLastError();
}

// decoupled = 2: AdaBelief (Zhuang et al. 2020, weight_decouple, no rectification, no amsgrad): v <- b2 v + (1-b2)(g-m)^2 + eps.
// g' = g * coef[1] in registers (the clip coefficient stays on the device: no host sync), then Adam / AdamW (torch semantics,
// amsgrad off): decoupled: p *= 1 - lr*wd ; else g += wd*p.  m = m + (1-b1)(g - m); v = b2*v + (1-b2) g*g;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  bc1 = 1 - b1^t, bc2s = sqrt(1 - b2^t) from the host.
__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamArgs a) {
    const int ti = a.blk_tensor[blockIdx.x];
    float* p = (float*)a.p[ti];
    const float* g = (const float*)a.g[ti];  // never written: torch.optim leaves p.grad alone (clip / L2 terms stay in registers)
    float* m = (float*)a.m[ti];
    float* v = (float*)a.v[ti];
    const long long n = a.sizes[ti], off = a.blk_off[blockIdx.x];
    const float cc = a.clip ? a.clip[1] : 1.0f;
    const float lr = a.dyn ? a.dyn[0] : a.lr, bc1 = a.dyn ? a.dyn[1] : a.bc1, bc2s = a.dyn ? a.dyn[2] : a.bc2s;
    const float step_size = lr / bc1;
    for (int i = threadIdx.x; i < kEmaBlock; i += 256) {
        const long long k = off + i;
        if (k >= n) continue;
        float gk = __fmul_rn(g[k], cc);
        float pk = p[k];
        if (a.decoupled) pk = __fmul_rn(pk, 1.0f - lr * a.wd);
        else gk = fmaf(a.wd, pk, gk);
        const float mk = fmaf(1.0f - a.b1, __fsub_rn(gk, m[k]), m[k]);
        float vk;
        if (a.decoupled == 2) {  // AdaBelief: the second moment follows (g - m)^2 and absorbs eps every step
            const float r = __fsub_rn(gk, mk);
            vk = __fadd_rn(fmaf(__fmul_rn(r, r), 1.0f - a.b2, __fmul_rn(v[k], a.b2)), a.eps);
        } else {
            vk = fmaf(__fmul_rn(gk, gk), 1.0f - a.b2, __fmul_rn(v[k], a.b2));
        }
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vk), bc2s), a.eps);
        pk = fmaf(-step_size, __fdiv_rn(mk, denom), pk);
        m[k] = mk; v[k] = vk; p[k] = pk;
    }
}
__global__ void __launch_bounds__(256) scale_multi_kernel(const long long* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                          const int* __restrict__ blk_tensor, const long long* __restrict__ blk_off,
                                                          const float* __restrict__ coef) {
    const int ti = blk_tensor[blockIdx.x];
    float* g = (float*)ptrs[ti];
    const long long n = sizes[ti], off = blk_off[blockIdx.x];
    const float c = coef[0];
    if (c == 1.0f) return;  // torch multiplies by the clamped coefficient; x * 1.0f is the identity
    for (int i = threadIdx.x; i < kEmaBlock; i += 256) {
        const long long k = off + i;
        if (k < n) g[k] = __fmul_rn(g[k], c);
    }
}
hipError_t scale_multi_launch(const long long* ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                              int nblocks, const float* coef, hipStream_t s) {
    hipLaunchKernelGGL(scale_multi_kernel, dim3(nblocks), dim3(256), 0, s, ptrs, sizes, blk_tensor, blk_off, coef);
    return hipGetLastError();
}
hipError_t adam_multi_launch(const AdamArgs& a, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================================
// weight packing (fp32 parameters -> internal layouts)
// =====================================================================================================
__global__ void pack_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
hipError_t pack_copy_launch(const float* src, float* dst, long long n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(pack_copy_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

// many small fp32 copies in one launch: the (src, dst, n) triples travel as kernel arguments
__global__ void __launch_bounds__(256) pack_copy_multi_kernel(const PackCopyBatch b) {
    const float* src = b.src[blockIdx.y];
    float* dst = b.dst[blockIdx.y];
    const long long n = b.n[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
hipError_t pack_copy_multi_launch(const PackCopyBatch& b, hipStream_t s) {
    if (b.count < 1) return hipSuccess;
    long long mx = 0;
    for (int i = 0; i < b.count; ++i) if (b.n[i] > mx) mx = b.n[i];
    const int bx = (int)((mx + 255) / 256 < 256 ? (mx + 255) / 256 : 256);
    hipLaunchKernelGGL(pack_copy_multi_kernel, dim3(bx, b.count), dim3(256), 0, s, b);
    return hipGetLastError();
}

template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ w, T* __restrict__ dst, int O, int I, int KK) {
    const long long n = (long long)KK * O * I;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int ci = i % I, co = (i / I) % O, tap = i / ((long long)I * O);
        dst[i] = from_f<T>(w[((size_t)co * I + ci) * KK + tap]);
    }
}
hipError_t pack_conv_launch(int dtype, const float* w, void* dst, int O, int I, int KH, int KW, hipStream_t s) {
    const long long n = (long long)KH * KW * O * I;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dtype == DT_BF16) hipLaunchKernelGGL(pack_conv_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, w, (__bf16*)dst, O, I, KH * KW);
    else hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)dst, O, I, KH * KW);
    return hipGetLastError();
}

// many conv-weight packings in one launch (a training step re-packs every conv after its optimizer step: 143 launches of 4 us
// otherwise).  mode 0: the forward layout of pack_conv_kernel, dst[tap][co][ci] = w[co][ci][tap]; mode 1: the data-gradient
// packing of a 3x3 conv (train_kernels.hip: transposed and flipped), dst[tp][ci][co] = w[co][ci][8 - tp].
template <typename T>
__device__ __forceinline__ void pack_conv_entry(const float* __restrict__ w, T* __restrict__ dst, int O, int I, int KK, int mode) {
    const int n = KK * O * I;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (mode == 0) {
            const int ci = i % I, co = (i / I) % O, tap = i / (I * O);
            dst[i] = from_f<T>(w[((size_t)co * I + ci) * KK + tap]);
        } else {
            const int co = i % O, ci = (i / O) % I, tp = i / (O * I);
            dst[i] = from_f<T>(w[((size_t)co * I + ci) * 9 + (8 - tp)]);
        }
    }
}
__global__ void __launch_bounds__(256) pack_conv_multi_kernel(const PackConvBatch b) {
    const int e = blockIdx.y;
    if (b.f32[e]) pack_conv_entry<float>(b.src[e], (float*)b.dst[e], b.O[e], b.I[e], b.KK[e], b.mode[e]);
    else pack_conv_entry<__bf16>(b.src[e], (__bf16*)b.dst[e], b.O[e], b.I[e], b.KK[e], b.mode[e]);
}
hipError_t pack_conv_multi_launch(const PackConvBatch& b, hipStream_t s) {
    if (b.count < 1) return hipSuccess;
    long long mx = 0;
    for (int i = 0; i < b.count; ++i) {
        const long long n = (long long)b.KK[i] * b.O[i] * b.I[i];
        if (n > mx) mx = n;
    }
    const int bx = (int)((mx + 255) / 256 < 64 ? (mx + 255) / 256 : 64);
    hipLaunchKernelGGL(pack_conv_multi_kernel, dim3(bx, b.count), dim3(256), 0, s, b);
    return hipGetLastError();
}

// conv weight [O][I][KH][KW] fp32 (KK = KH * KW taps, row-major) -> bf16 in MFMA fragment order (conv_wreg.h):
//   dst[step = tap * (I/16) + kg][nb][lane = h * 32 + l31][j]  =  w[co = nb * 32 + l31][ci = kg * 16 + h * 8 + j][tap]
__global__ void pack_conv_frag_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int O, int I, int KK) {
    const int KG = I / 16, NBLK = O / 32;
    const long long n = (long long)KK * O * I;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long long blk = i >> 9;
        const int nb = (int)(blk % NBLK), st = (int)(blk / NBLK);
        const int tap = st / KG, kg = st % KG;
        const int co = nb * 32 + (lane & 31), ci = kg * 16 + (lane >> 5) * 8 + j;
        dst[i] = (__bf16)w[((size_t)co * I + ci) * KK + tap];
    }
}
hipError_t pack_conv_frag_launch(const float* w, void* dst, int O, int I, int KK, hipStream_t s) {
    if (O % 32 || I % 16) return hipErrorInvalidValue;
    const long long n = (long long)KK * O * I;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_conv_frag_kernel, dim3(blocks), dim3(256), 0, s, w, (__bf16*)dst, O, I, KK);
    return hipGetLastError();
}

// packed taps [ntaps][NOUT][CIN] bf16 -> fragment order [ntaps * CIN/16][NOUT/32][64][8] (conv_wreg.h)
__global__ void pack_frag_from_taps_kernel(const __bf16* __restrict__ src, __bf16* __restrict__ dst, int ntaps, int NOUT, int CIN) {
    const int KG = CIN / 16, NBLK = NOUT / 32;
    const long long n = (long long)ntaps * NOUT * CIN;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long long blk = i >> 9;
        const int nb = (int)(blk % NBLK), st = (int)(blk / NBLK);
        const int tap = st / KG, kg = st % KG;
        const int co = nb * 32 + (lane & 31), ci = kg * 16 + (lane >> 5) * 8 + j;
        dst[i] = src[((size_t)tap * NOUT + co) * CIN + ci];
    }
}
hipError_t pack_frag_from_taps_launch(const void* src, void* dst, int ntaps, int NOUT, int CIN, hipStream_t s) {
    if (NOUT % 32 || CIN % 16) return hipErrorInvalidValue;
    const long long n = (long long)ntaps * NOUT * CIN;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_frag_from_taps_kernel, dim3(blocks), dim3(256), 0, s, (const __bf16*)src, (__bf16*)dst, ntaps, NOUT, CIN);
    return hipGetLastError();
}

// ConvTranspose2d(k4,s2,p1) weight [I][O][4][4] -> sub-pixel form [a][tap=(dyi,dx)][vc=b*O+co][ci]:
// output (2py+a, 2px+b) reads input (py+dy-1, px+dx-1) through kernel element kh = 3+a-2dy, kw = 3+b-2dx
// (dy = a+dyi in {a,a+1}; dx in {0,1,2}); combinations whose kw falls outside 0..3 are zero.
template <typename T>
__global__ void pack_convT_kernel(const float* __restrict__ w, T* __restrict__ dst, int I, int O) {
    const long long n = 2LL * 6 * 2 * O * I;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int ci = i % I;
        const int vc = (i / I) % (2 * O);
        const int tap = (i / ((long long)I * 2 * O)) % 6;
        const int a = i / ((long long)I * 2 * O * 6);
        const int b = vc / O, co = vc % O;
        const int dy = a + tap / 3, dx = tap % 3;
        const int kh = 3 + a - 2 * dy, kw = 3 + b - 2 * dx;
        float v = 0.f;
        if (kw >= 0 && kw < 4 && kh >= 0 && kh < 4) v = w[(((size_t)ci * O + co) * 4 + kh) * 4 + kw];
        dst[i] = from_f<T>(v);
    }
}
hipError_t pack_convT_launch(int dtype, const float* w, void* dst, int I, int O, hipStream_t s) {
    const long long n = 2LL * 6 * 2 * O * I;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dtype == DT_BF16) hipLaunchKernelGGL(pack_convT_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, w, (__bf16*)dst, I, O);
    else hipLaunchKernelGGL(pack_convT_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)dst, I, O);
    return hipGetLastError();
}

__global__ void pack_perm_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int C, int Fr) {
    const long long n = (long long)rows * C * Fr;
    const int Wd = C * Fr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int col = i % Wd;
        const long long r = i / Wd;
        const int f = col / C, c = col % C;
        dst[i] = src[r * Wd + (long long)c * Fr + f];
    }
}
hipError_t pack_perm_cols_launch(const float* src, float* dst, int rows, int C, int Fr, hipStream_t s) {
    const long long n = (long long)rows * C * Fr;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_perm_cols_kernel, dim3(blocks), dim3(256), 0, s, src, dst, rows, C, Fr);
    return hipGetLastError();
}
__global__ void pack_perm_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int Fr, int K) {
    const long long n = (long long)C * Fr * K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int k = i % K;
        const int row = i / K;
        const int f = row / C, c = row % C;
        dst[i] = src[((long long)c * Fr + f) * K + k];
    }
}
hipError_t pack_perm_rows_launch(const float* src, float* dst, int C, int Fr, int K, hipStream_t s) {
    const long long n = (long long)C * Fr * K;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_perm_rows_kernel, dim3(blocks), dim3(256), 0, s, src, dst, C, Fr, K);
    return hipGetLastError();
}

}  // namespace ddimx
