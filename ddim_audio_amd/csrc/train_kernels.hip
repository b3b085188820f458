// Backward-pass kernels of the training step that are not MFMA convolutions (gfx950): GroupNorm / SiLU backward
// around the fused convolutions, parameter-gradient reductions, the edge convolutions' gradients.
// The reference obtains all of these from autograd (runners/diffusion.py:150 `loss.backward()`) over
// models/diffusion.py:42-56 (Residual_Block), :189-208 (edge convs), :110-120 (BetaEmbedding).
#include "kernels.h"
#include "train_kernels.h"
#include "wgrad_mfma.h"

namespace ddimx {

hipError_t wgrad_geometry_bf16(int, int, int, WgradGeom*);
hipError_t wgrad_geometry_f32(int, int, int, WgradGeom*);
hipError_t wgrad_launch_bf16(int, int, int, const WgradArgs&, int, hipStream_t);
hipError_t wgrad_launch_f32(int, int, int, const WgradArgs&, int, hipStream_t);
hipError_t wgrad_geometry(int dtype, int mode, int ci, int co, WgradGeom* g) {
    return dtype == DT_BF16 ? wgrad_geometry_bf16(mode, ci, co, g) : wgrad_geometry_f32(mode, ci, co, g);
}
hipError_t wgrad_launch(int dtype, int mode, int ci, int co, const WgradArgs& a, int nsplit, hipStream_t s) {
    return dtype == DT_BF16 ? wgrad_launch_bf16(mode, ci, co, a, nsplit, s) : wgrad_launch_f32(mode, ci, co, a, nsplit, s);
}

// dst[co][ci][tap] = sum_s partial[s][tap][co][ci]
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int ntaps, int co,
                                                           int ci, float* __restrict__ dst) {
    const int n = ntaps * co * ci;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += (double)partial[(size_t)k * n + i];
    const int c = i % ci, o = (i / ci) % co, tap = i / (ci * co);
    dst[((size_t)o * ci + c) * ntaps + tap] = (float)s;
}
hipError_t wgrad_reduce_launch(const float* partial, int nsplit, int ntaps, int co, int ci, float* dst, hipStream_t s) {
    const int n = ntaps * co * ci;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, nsplit, ntaps, co, ci, dst);
    return hipGetLastError();
}

__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
// d/du SiLU(u) = s (1 + u (1 - s)),  s = sigmoid(u)
__device__ __forceinline__ float dsilu_f(float u) { const float s = sigmoid_f(u); return s * fmaf(u, 1.0f - s, 1.0f); }

// =====================================================================================================
// GroupNorm backward, step 1: per-(sample, channel) partial sums  P = sum g' ,  Q = sum g' * v
//   MODE 0 (GroupNorm fed by SiLU(u): GN1, GN2):   g' = g,                          v = SiLU(u)
//   MODE 1 (GroupNorm followed by SiLU: GN0):      g' = g * SiLU'(scale*x + shift), v = x      (u = x)
// same partitioning as tensor_stats / resid (resid_nparts), output [B][nparts][C][2]
// =====================================================================================================
constexpr int kTrIters = 16;
static inline int tr_bd(int cpp) { return (cpp % 3 == 0) ? 192 : 256; }
static inline int tr_nparts(int dtype, int HW, int C) { return resid_nparts(dtype, HW, C); }

template <typename T, int MODE>
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const T* __restrict__ g, const T* __restrict__ u,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ stats, int HW, int C) {
    constexpr int EPB = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int CPP = C / EPB, c = tid % CPP, b = blockIdx.y, part = blockIdx.x;
    const long long pieces = (long long)HW * CPP;
    float sc[EPB], sh[EPB], P[EPB], Q[EPB];
#pragma unroll
    for (int j = 0; j < EPB; ++j) {
        P[j] = Q[j] = 0.f;
        sc[j] = MODE == 1 ? scale[(size_t)b * C + c * EPB + j] : 1.f;
        sh[j] = MODE == 1 ? shift[(size_t)b * C + c * EPB + j] : 0.f;
    }
    for (int it = 0; it < kTrIters; ++it) {
        const long long pc = ((long long)part * kTrIters + it) * bd + tid;
        if (pc >= pieces) break;
        const size_t e = (size_t)b * HW * C + (size_t)pc * EPB;
        float fg[EPB], fu[EPB];
        Piece<T>::unpack(*(const uint4*)(g + e), fg);
        Piece<T>::unpack(*(const uint4*)(u + e), fu);
#pragma unroll
        for (int j = 0; j < EPB; ++j) {
            float gp, v;
            if (MODE == 0) { gp = fg[j]; v = silu_f(fu[j]); }
            else { gp = fg[j] * dsilu_f(fmaf(fu[j], sc[j], sh[j])); v = fu[j]; }
            P[j] += gp;
            Q[j] = fmaf(gp, v, Q[j]);
        }
    }
    const int R = bd / CPP, row = tid / CPP;
#pragma unroll
    for (int j = 0; j < EPB; ++j) {
        red[(row * C + c * EPB + j) * 2 + 0] = P[j];
        red[(row * C + c * EPB + j) * 2 + 1] = Q[j];
    }
    __syncthreads();
    for (int i = tid; i < C * 2; i += bd) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += red[r * C * 2 + i];
        stats[(((size_t)b * gridDim.x + part) * C) * 2 + i] = t;
    }
}
hipError_t gn_bwd_stats_launch(int dtype, int mode, const void* g, const void* u, const float* scale, const float* shift,
                               float* stats, int B, int HW, int C, hipStream_t s) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    if (C % epb) return hipErrorInvalidValue;
    const int cpp = C / epb, bd = tr_bd(cpp);
    if (bd % cpp) return hipErrorInvalidValue;
    dim3 grid(tr_nparts(dtype, HW, C), B);
    const size_t lds = (size_t)(bd / cpp) * C * 2 * 4;
#define DDIMX_L(TT, M) hipLaunchKernelGGL((gn_bwd_stats_kernel<TT, M>), grid, dim3(bd), lds, s, (const TT*)g, (const TT*)u, scale, shift, stats, HW, C)
    if (dtype == DT_BF16) { if (mode) DDIMX_L(__bf16, 1); else DDIMX_L(__bf16, 0); }
    else { if (mode) DDIMX_L(float, 1); else DDIMX_L(float, 0); }
#undef DDIMX_L
    return hipGetLastError();
}

// =====================================================================================================
// GroupNorm backward, step 2: partial sums -> per-(sample, channel) coefficients of
//     d(input of the norm) = ca * g' + cb * v + cc
// and the per-sample parameter-gradient terms  dgamma_b[b][c] = rstd (Q - mean P),  dbeta_b[b][c] = P.
//   S1 = sum_{c in group} gamma_c P_c,  S2 = sum_{c in group} gamma_c rstd (Q_c - mean P_c),  N = elements per group
//   ca = gamma_c rstd,  cb = -rstd^2 S2 / N,  cc = -rstd S1 / N + mean rstd^2 S2 / N
// grid (groups, B)
// =====================================================================================================
__global__ void __launch_bounds__(64) gn_bwd_finalize_kernel(const float* __restrict__ stats, int nparts, int C, double count,
                                                             const float* __restrict__ gamma, const float* __restrict__ mr,
                                                             float* __restrict__ coef /*[B][3][C]*/, float* __restrict__ dgb /*[B][2][C]*/) {
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int GS = C / kGroups;
    const float mean = mr[((size_t)b * kGroups + g) * 2 + 0], rstd = mr[((size_t)b * kGroups + g) * 2 + 1];
    double P = 0.0, Q = 0.0;
    const int c = g * GS + tid;
    if (tid < GS) {
        for (int p = 0; p < nparts; ++p) {
            const float* q = stats + (((size_t)b * nparts + p) * C + c) * 2;
            P += (double)q[0];
            Q += (double)q[1];
        }
    }
    const double gm = tid < GS ? (double)gamma[c] : 0.0;
    const double dg = (double)rstd * (Q - (double)mean * P);
    double s1 = gm * P, s2 = gm * dg;
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (tid < GS) {
        const double r = (double)rstd;
        coef[((size_t)b * 3 + 0) * C + c] = (float)(gm * r);
        coef[((size_t)b * 3 + 1) * C + c] = (float)(-r * r * s2 / count);
        coef[((size_t)b * 3 + 2) * C + c] = (float)(-r * s1 / count + (double)mean * r * r * s2 / count);
        dgb[((size_t)b * 2 + 0) * C + c] = (float)dg;
        dgb[((size_t)b * 2 + 1) * C + c] = (float)P;
    }
}
hipError_t gn_bwd_finalize_launch(const float* stats, int nparts, int C, double count, const float* gamma, const float* mr,
                                  float* coef, float* dgb, int B, hipStream_t s) {
    if (C % kGroups || C / kGroups > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(kGroups, B), dim3(64), 0, s, stats, nparts, C, count, gamma, mr, coef, dgb);
    return hipGetLastError();
}

// dst[c] = sum_b src[b * stride + c]   (fixed order)
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ src, int B, long long stride, int C,
                                                     float* __restrict__ dst) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += (double)src[(size_t)b * stride + c];
    dst[c] = (float)s;
}
hipError_t colsum_launch(const float* src, int B, long long stride, int C, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256), dim3(256), 0, s, src, B, stride, C, dst);
    return hipGetLastError();
}
// dst[b][c] = sum_p src[b][p][c]
__global__ void __launch_bounds__(256) partsum_kernel(const float* __restrict__ src, int nparts, int C, float* __restrict__ dst,
                                                      long long dst_stride) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += (double)src[((size_t)b * nparts + p) * C + c];
    dst[(size_t)b * dst_stride + c] = (float)s;
}
hipError_t partsum_launch(const float* src, int B, int nparts, int C, float* dst, long long dst_stride, hipStream_t s) {
    hipLaunchKernelGGL(partsum_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, src, nparts, C, dst, dst_stride);
    return hipGetLastError();
}

// =====================================================================================================
// GroupNorm backward, step 3 (elementwise):
//   MODE 0:  du = (ca*g + cb*SiLU(u) + cc) * SiLU'(u)                 + per-(sample, part, channel) sums of du
//   MODE 1:  dx = gy + ca*(g*SiLU'(scale*x+shift)) + cb*x + cc [+ extra]          (x = u)
// =====================================================================================================
template <typename T, int MODE>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ u,
                                                           const T* __restrict__ gy, const T* __restrict__ extra,
                                                           const float* __restrict__ coef, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, T* __restrict__ out,
                                                           float* __restrict__ sums, int HW, int C) {
    constexpr int EPB = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int CPP = C / EPB, c = tid % CPP, b = blockIdx.y, part = blockIdx.x;
    const long long pieces = (long long)HW * CPP;
    float ca[EPB], cb[EPB], cc[EPB], sc[EPB], sh[EPB], acc[EPB];
#pragma unroll
    for (int j = 0; j < EPB; ++j) {
        const int ch = c * EPB + j;
        ca[j] = coef[((size_t)b * 3 + 0) * C + ch];
        cb[j] = coef[((size_t)b * 3 + 1) * C + ch];
        cc[j] = coef[((size_t)b * 3 + 2) * C + ch];
        sc[j] = MODE == 1 ? scale[(size_t)b * C + ch] : 1.f;
        sh[j] = MODE == 1 ? shift[(size_t)b * C + ch] : 0.f;
        acc[j] = 0.f;
    }
    for (int it = 0; it < kTrIters; ++it) {
        const long long pc = ((long long)part * kTrIters + it) * bd + tid;
        if (pc >= pieces) break;
        const size_t e = (size_t)b * HW * C + (size_t)pc * EPB;
        float fg[EPB], fu[EPB], fo[EPB];
        Piece<T>::unpack(*(const uint4*)(g + e), fg);
        Piece<T>::unpack(*(const uint4*)(u + e), fu);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                const float sg = sigmoid_f(fu[j]);
                const float ds = fmaf(ca[j], fg[j], fmaf(cb[j], fu[j] * sg, cc[j]));
                fo[j] = ds * (sg * fmaf(fu[j], 1.0f - sg, 1.0f));
            }
        } else {
            float fy[EPB];
            Piece<T>::unpack(*(const uint4*)(gy + e), fy);
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                const float gp = fg[j] * dsilu_f(fmaf(fu[j], sc[j], sh[j]));
                fo[j] = fy[j] + fmaf(ca[j], gp, fmaf(cb[j], fu[j], cc[j]));
            }
            if (extra) {
                float fe[EPB];
                Piece<T>::unpack(*(const uint4*)(extra + e), fe);
#pragma unroll
                for (int j = 0; j < EPB; ++j) fo[j] += fe[j];
            }
        }
        const uint4 pv = Piece<T>::pack(fo);
        *(uint4*)(out + e) = pv;
        if (MODE == 0) {
            Piece<T>::unpack(pv, fo);  // sums of the values as stored (what the weight-gradient kernel will read)
#pragma unroll
            for (int j = 0; j < EPB; ++j) acc[j] += fo[j];
        }
    }
    if (MODE == 0 && sums) {
        const int R = bd / CPP, row = tid / CPP;
#pragma unroll
        for (int j = 0; j < EPB; ++j) red[row * C + c * EPB + j] = acc[j];
        __syncthreads();
        for (int i = tid; i < C; i += bd) {
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += red[r * C + i];
            sums[((size_t)b * gridDim.x + part) * C + i] = t;
        }
    }
}
hipError_t gn_bwd_apply_launch(int dtype, int mode, const void* g, const void* u, const void* gy, const void* extra,
                               const float* coef, const float* scale, const float* shift, void* out, float* sums, int B,
                               int HW, int C, hipStream_t s) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    if (C % epb) return hipErrorInvalidValue;
    const int cpp = C / epb, bd = tr_bd(cpp);
    if (bd % cpp) return hipErrorInvalidValue;
    dim3 grid(tr_nparts(dtype, HW, C), B);
    const size_t lds = (size_t)(bd / cpp) * C * 4;
#define DDIMX_L(TT, M)                                                                                                      \
    hipLaunchKernelGGL((gn_bwd_apply_kernel<TT, M>), grid, dim3(bd), lds, s, (const TT*)g, (const TT*)u, (const TT*)gy, \
                       (const TT*)extra, coef, scale, shift, (TT*)out, sums, HW, C)
    if (dtype == DT_BF16) { if (mode) DDIMX_L(__bf16, 1); else DDIMX_L(__bf16, 0); }
    else { if (mode) DDIMX_L(float, 1); else DDIMX_L(float, 0); }
#undef DDIMX_L
    return hipGetLastError();
}

// =====================================================================================================
// data-gradient weights: the transposed, spatially flipped 3x3 kernel in the forward conv's packed layout
//   dst[tap'][ci][co] = w[co][ci][8 - tap']        (w: Conv2d.weight [O][I][3][3]; dst: [9][I][O] as T)
// =====================================================================================================
template <typename T>
__global__ void pack_conv_dgrad_kernel(const float* __restrict__ w, T* __restrict__ dst, int O, int I) {
    const int n = 9 * O * I;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int co = i % O, ci = (i / O) % I, tp = i / (O * I);
        dst[i] = from_f<T>(w[((size_t)co * I + ci) * 9 + (8 - tp)]);
    }
}
hipError_t pack_conv_dgrad_launch(int dtype, const float* w, void* dst, int O, int I, hipStream_t s) {
    const int n = 9 * O * I;
    const int blocks = (n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256;
    if (dtype == DT_BF16) hipLaunchKernelGGL(pack_conv_dgrad_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, w, (__bf16*)dst, O, I);
    else hipLaunchKernelGGL(pack_conv_dgrad_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)dst, O, I);
    return hipGetLastError();
}

}  // namespace ddimx
