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

// dst[co][ci][tap] = sum_s partial[s][tap][co][ci]; KS threads share an output (splits s = q, q+KS, ...: loads unrolled so
// that eight are in flight per thread), folded in a fixed order through LDS
template <int KS>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int ntaps, int co,
                                                           int ci, float* __restrict__ dst) {
    constexpr int OUT = 256 / KS;
    __shared__ double red[KS][OUT];
    const int n = ntaps * co * ci;
    const int ol = threadIdx.x % OUT, q = threadIdx.x / OUT;
    const int i = blockIdx.x * OUT + ol;
    double s = 0.0;
    if (i < n) {
        int k = q;
        for (; k + 7 * KS < nsplit; k += 8 * KS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + u * KS) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; k < nsplit; k += KS) s += (double)partial[(size_t)k * n + i];
    }
    red[q][ol] = s;
    __syncthreads();
    if (q == 0 && i < n) {
        s = 0.0;
#pragma unroll
        for (int k = 0; k < KS; ++k) s += red[k][ol];
        const int c = i % ci, o = (i / ci) % co, tap = i / (ci * co);
        dst[((size_t)o * ci + c) * ntaps + tap] = (float)s;
    }
}
// The same sums in the same order (split s = q, q + 16, ... per thread, then the 16 partial sums in order), four consecutive
// outputs per thread: a 16-byte load per split instead of four 4-byte ones, 256 contiguous bytes of a slab per workgroup
// instead of 64 (the scalar form moved 19 MB of level-0 slabs in 21 us; round 3: 48 launches per step).  n % 4 == 0.
__global__ void __launch_bounds__(256) wgrad_reduce4_kernel(const float* __restrict__ partial, int nsplit, int ntaps, int co, int ci,
                                                            float* __restrict__ dst) {
    constexpr int KS = 16, OG = 16;  // 16 split groups x 16 output quads = 64 outputs per workgroup
    __shared__ double red[KS][OG * 4];
    const int n = ntaps * co * ci;
    const int og = threadIdx.x % OG, q = threadIdx.x / OG;
    const int i0 = (blockIdx.x * OG + og) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i0 < n) {
        int k = q;
        for (; k + 7 * KS < nsplit; k += 8 * KS) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(partial + (size_t)(k + u * KS) * n + i0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
        }
        for (; k < nsplit; k += KS) {
            const float4 v = *(const float4*)(partial + (size_t)k * n + i0);
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
        }
    }
    red[q][og * 4 + 0] = s0; red[q][og * 4 + 1] = s1; red[q][og * 4 + 2] = s2; red[q][og * 4 + 3] = s3;
    __syncthreads();
    const int ol = threadIdx.x, i = blockIdx.x * OG * 4 + ol;
    if (ol < OG * 4 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < KS; ++k) t += red[k][ol];
        const int c = i % ci, o = (i / ci) % co, tap = i / (ci * co);
        dst[((size_t)o * ci + c) * ntaps + tap] = (float)t;
    }
}
hipError_t wgrad_reduce_launch(const float* partial, int nsplit, int ntaps, int co, int ci, float* dst, hipStream_t s) {
    const int n = ntaps * co * ci;
    if (nsplit >= 64 && n % 4 == 0)  // many thin slabs (levels 0-2): 16 threads per output quad
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((n + 63) / 64), dim3(256), 0, s, partial, nsplit, ntaps, co, ci, dst);
    else if (nsplit >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((n + 15) / 16), dim3(256), 0, s, partial, nsplit, ntaps, co, ci, dst);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((n + 63) / 64), dim3(256), 0, s, partial, nsplit, ntaps, co, ci, dst);
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
static inline int tr_bd(int cpp) { return (cpp % 3 == 0) ? 192 : 256; }
static inline int tr_nparts(int dtype, int HW, int C) { return resid_nparts(dtype, HW, C); }

template <typename T, int MODE>
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const T* __restrict__ g, const T* __restrict__ u,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ stats, int HW, int C, int iters, int nt) {
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
    // four iterations' loads are issued together, unconditionally (out-of-range slots re-read piece 0 and are dropped): one
    // load round trip per four iterations instead of one per iteration
    const long long pc0 = (long long)part * iters * bd + tid;
    const size_t sbase = (size_t)b * HW * C;
    for (int it0 = 0; it0 < iters; it0 += 4) {
        uint4 vg[4], vu[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long pc = pc0 + (long long)(it0 + k) * bd;
            ok[k] = it0 + k < iters && pc < pieces;
            const size_t e = sbase + (size_t)(ok[k] ? pc : 0) * EPB;
            vg[k] = nt ? nt_load16(g + e) : *(const uint4*)(g + e);
            vu[k] = nt ? nt_load16(u + e) : *(const uint4*)(u + e);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            float fg[EPB], fu[EPB];
            Piece<T>::unpack(vg[k], fg);
            Piece<T>::unpack(vu[k], fu);
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                float gp, v;
                if (MODE == 0) { gp = fg[j]; v = silu_f(fu[j]); }
                else { gp = fg[j] * dsilu_f(fmaf(fu[j], sc[j], sh[j])); v = fu[j]; }
                P[j] += gp;
                Q[j] = fmaf(gp, v, Q[j]);
            }
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
#define DDIMX_L(TT, M) hipLaunchKernelGGL((gn_bwd_stats_kernel<TT, M>), grid, dim3(bd), lds, s, (const TT*)g, (const TT*)u, scale, shift, stats, HW, C, resid_iters(dtype, HW, C), nt_streaming((size_t)B * HW * C * (dtype == DT_BF16 ? 2 : 4)))
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
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const float* __restrict__ stats, int nparts, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mr,
                                                              float* __restrict__ coef /*[B][3][C]*/, float* __restrict__ dgb /*[B][2][C]*/) {
    __shared__ double rp[8][32], rq[8][32];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int GS = C / kGroups;  // <= 32
    const float mean = mr[((size_t)b * kGroups + g) * 2 + 0], rstd = mr[((size_t)b * kGroups + g) * 2 + 1];
    const int ch = tid & 31, pl = tid >> 5;  // 8 part-lanes per channel
    const int c = g * GS + ch;
    double P = 0.0, Q = 0.0;
    {   // eight partials per thread in flight, unconditionally (clamped index, dropped by select), added in the original order: the
        // loop with a run-time trip count was one load round trip per partial, and this kernel is nothing else
        const int cs = ch < GS ? c : g * GS;
        for (int p0 = pl; p0 < nparts; p0 += 64) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + 8 * u;
                v[u] = *(const float2*)(stats + (((size_t)b * nparts + (p < nparts ? p : nparts - 1)) * C + cs) * 2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = ch < GS && p0 + 8 * u < nparts;
                P += in ? (double)v[u].x : 0.0;
                Q += in ? (double)v[u].y : 0.0;
            }
        }
    }
    rp[pl][ch] = P; rq[pl][ch] = Q;
    __syncthreads();
    if (tid >= 64) return;
    P = Q = 0.0;
    if (tid < GS) {
        for (int k = 0; k < 8; ++k) { P += rp[k][tid]; Q += rq[k][tid]; }
    }
    const int cc = g * GS + tid;
    const double gm = tid < GS ? (double)gamma[cc] : 0.0;
    const double dg = (double)rstd * (Q - (double)mean * P);
    double s1 = gm * P, s2 = gm * dg;
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (tid < GS) {
        const double r = (double)rstd;
        coef[((size_t)b * 3 + 0) * C + cc] = (float)(gm * r);
        coef[((size_t)b * 3 + 1) * C + cc] = (float)(-r * r * s2 / count);
        coef[((size_t)b * 3 + 2) * C + cc] = (float)(-r * s1 / count + (double)mean * r * r * s2 / count);
        dgb[((size_t)b * 2 + 0) * C + cc] = (float)dg;
        dgb[((size_t)b * 2 + 1) * C + cc] = (float)P;
    }
}
hipError_t gn_bwd_finalize_launch(const float* stats, int nparts, int C, double count, const float* gamma, const float* mr,
                                  float* coef, float* dgb, int B, hipStream_t s) {
    if (C % kGroups || C / kGroups > 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(kGroups, B), dim3(256), 0, s, stats, nparts, C, count, gamma, mr, coef, dgb);
    return hipGetLastError();
}

// dst[c] = sum_b src[b * stride + c]: 32 columns x 8 row-lanes per block, lanes folded in a fixed order
// 16 columns x 16 row slices per block; eight loads in flight per thread
__device__ __forceinline__ double colsum_slice(const float* __restrict__ src, int B, long long stride, int c, int rl) {
    double s = 0.0;
    int b = rl;
    for (; b + 7 * 16 < B; b += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + u * 16) * stride + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < B; b += 16) s += (double)src[(size_t)b * stride + c];
    return s;
}
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ src, int B, long long stride, int C,
                                                     float* __restrict__ dst) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    red[rl][cl] = c < C ? colsum_slice(src, B, stride, c, rl) : 0.0;
    __syncthreads();
    if (rl == 0 && c < C) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        dst[c] = (float)s;
    }
}
hipError_t colsum_launch(const float* src, int B, long long stride, int C, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 15) / 16), dim3(256), 0, s, src, B, stride, C, dst);
    return hipGetLastError();
}
// the same for up to kMax (src, dst) pairs in one launch (the entries travel as kernel arguments): the backward defers
// its per-block parameter-gradient batch sums and flushes them together
__global__ void __launch_bounds__(256) colsum_multi_kernel(const ColsumBatch q) {
    __shared__ double red[8][32];
    const int e = blockIdx.y;
    const int C = q.C[e], B = q.B[e];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    if (blockIdx.x * 32 >= C) return;  // uniform
    const float* src = q.src[e];
    const long long stride = q.stride[e];
    double s = 0.0;
    if (c < C)
        for (int b = rl; b < B; b += 8) s += (double)src[(size_t)b * stride + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        s = 0.0;
        for (int k = 0; k < 8; ++k) s += red[k][cl];
        q.dst[e][c] = (float)s;
    }
}
hipError_t colsum_multi_launch(const ColsumBatch& q, hipStream_t s) {
    if (q.count < 1) return hipSuccess;
    int mx = 0;
    for (int i = 0; i < q.count; ++i) if (q.C[i] > mx) mx = q.C[i];
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((mx + 31) / 32, q.count), dim3(256), 0, s, q);
    return hipGetLastError();
}
// dst[b][c] = sum_p src[((b*nparts + p)*C + c) * src_step]
__global__ void __launch_bounds__(256) partsum_kernel(const float* __restrict__ src, int nparts, int C, float* __restrict__ dst,
                                                      long long dst_stride, int src_step) {
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
    double s = 0.0;
    {   // eight partials per thread in flight (see gn_bwd_finalize_kernel), same order of addition
        const int cs = c < C ? c : C - 1;
        for (int p0 = rl; p0 < nparts; p0 += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + 8 * u;
                v[u] = src[(((size_t)b * nparts + (p < nparts ? p : nparts - 1)) * C + cs) * src_step];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (c < C && p0 + 8 * u < nparts) ? (double)v[u] : 0.0;
        }
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        s = 0.0;
        for (int k = 0; k < 8; ++k) s += red[k][cl];
        dst[(size_t)b * dst_stride + c] = (float)s;
    }
}
__global__ void __launch_bounds__(256) partsum_multi_kernel(const PartsumBatch q) {
    __shared__ double red[8][32];
    const int e = blockIdx.z;
    const int C = q.C[e], nparts = q.nparts[e];
    if ((int)blockIdx.y >= q.B[e] || (int)blockIdx.x * 32 >= C) return;  // uniform
    const float* __restrict__ src = q.src[e];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
    double s = 0.0;
    {   // partsum_kernel's loop, word for word: the sums carry the same bits
        const int cs = c < C ? c : C - 1;
        for (int p0 = rl; p0 < nparts; p0 += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + 8 * u;
                v[u] = src[((size_t)b * nparts + (p < nparts ? p : nparts - 1)) * C + cs];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (c < C && p0 + 8 * u < nparts) ? (double)v[u] : 0.0;
        }
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        s = 0.0;
        for (int k = 0; k < 8; ++k) s += red[k][cl];
        q.dst[e][(size_t)b * q.dst_stride[e] + c] = (float)s;
    }
}
hipError_t partsum_multi_launch(const PartsumBatch& q, hipStream_t s) {
    if (q.count < 1) return hipSuccess;
    int cm = 0, bm = 0;
    for (int i = 0; i < q.count; ++i) { if (q.C[i] > cm) cm = q.C[i]; if (q.B[i] > bm) bm = q.B[i]; }
    hipLaunchKernelGGL(partsum_multi_kernel, dim3((cm + 31) / 32, bm, q.count), dim3(256), 0, s, q);
    return hipGetLastError();
}
hipError_t partsum_launch(const float* src, int B, int nparts, int C, float* dst, long long dst_stride, hipStream_t s,
                          int src_step) {
    hipLaunchKernelGGL(partsum_kernel, dim3((C + 31) / 32, B), dim3(256), 0, s, src, nparts, C, dst, dst_stride, src_step);
    return hipGetLastError();
}

// =====================================================================================================
// GroupNorm backward, step 3 (elementwise):
//   MODE 0:  du = (ca*g + cb*SiLU(u) + cc) * SiLU'(u)                 + per-(sample, part, channel) sums of du
//   MODE 1:  dx = gy + ca*(g*SiLU'(scale*x+shift)) + cb*x + cc [+ extra]          (x = u)
//            + (nstats != null) the first statistics pass of the block that reads dx as ITS dy: P = sum dx, Q = sum dx * SiLU(nu),
//              nu = that block's saved u2 -- same partition, same per-thread order of additions and the same rounded dx values as
//              gn_bwd_stats_kernel<T, 0> over (dx, nu), so the slabs are bit-identical to that pass and it need not run
// =====================================================================================================
template <typename T, int MODE>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ u,
                                                           const T* __restrict__ gy, const T* __restrict__ extra,
                                                           const float* __restrict__ coef, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, T* __restrict__ out,
                                                           float* __restrict__ sums, int HW, int C, int iters,
                                                           const T* __restrict__ nu, float* __restrict__ nstats, int nt) {
    constexpr int EPB = Piece<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int CPP = C / EPB, c = tid % CPP, b = blockIdx.y, part = blockIdx.x;
    const long long pieces = (long long)HW * CPP;
    float ca[EPB], cb[EPB], cc[EPB], sc[EPB], sh[EPB], acc[EPB];
    float nP[MODE == 1 ? EPB : 1], nQ[MODE == 1 ? EPB : 1];
    const bool chain = MODE == 1 && nstats != nullptr;  // uniform
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < EPB; ++j) nP[j] = nQ[j] = 0.f;
    }
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
    // NIT iterations' loads are issued together, unconditionally (out-of-range slots re-read piece 0 and are dropped; an
    // absent `extra` reads g in its place): one load round trip per NIT iterations instead of one per iteration
    constexpr int NIT = MODE == 0 ? 4 : 2;
    const long long pc0 = (long long)part * iters * bd + tid;
    const size_t sbase = (size_t)b * HW * C;
    const T* const pex = (MODE == 1 && extra) ? extra : g;
    for (int it0 = 0; it0 < iters; it0 += NIT) {
        uint4 vg[NIT], vu[NIT], vy[MODE == 1 ? NIT : 1], ve[MODE == 1 ? NIT : 1], vn[MODE == 1 ? NIT : 1];
        size_t e[NIT];
        bool ok[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const long long pc = pc0 + (long long)(it0 + k) * bd;
            ok[k] = it0 + k < iters && pc < pieces;
            e[k] = sbase + (size_t)(ok[k] ? pc : 0) * EPB;
            if (nt) {  // (uniform) streamed once: keep the lines out of the caches' replacement order
                vg[k] = nt_load16(g + e[k]);
                vu[k] = nt_load16(u + e[k]);
            } else {
                vg[k] = *(const uint4*)(g + e[k]);
                vu[k] = *(const uint4*)(u + e[k]);
            }
            if (MODE == 1) {
                vy[k] = nt ? nt_load16(gy + e[k]) : *(const uint4*)(gy + e[k]);
                ve[k] = nt ? nt_load16(pex + e[k]) : *(const uint4*)(pex + e[k]);
                if (chain) vn[k] = nt ? nt_load16(nu + e[k]) : *(const uint4*)(nu + e[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (!ok[k]) continue;
            float fg[EPB], fu[EPB], fo[EPB];
            Piece<T>::unpack(vg[k], fg);
            Piece<T>::unpack(vu[k], fu);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < EPB; ++j) {
                    const float sg = sigmoid_f(fu[j]);
                    const float ds = fmaf(ca[j], fg[j], fmaf(cb[j], fu[j] * sg, cc[j]));
                    fo[j] = ds * (sg * fmaf(fu[j], 1.0f - sg, 1.0f));
                }
            } else {
                float fy[EPB], fe[EPB];
                Piece<T>::unpack(vy[k], fy);
                Piece<T>::unpack(ve[k], fe);
#pragma unroll
                for (int j = 0; j < EPB; ++j) {
                    const float gp = fg[j] * dsilu_f(fmaf(fu[j], sc[j], sh[j]));
                    fo[j] = fy[j] + fmaf(ca[j], gp, fmaf(cb[j], fu[j], cc[j]));
                    if (extra) fo[j] += fe[j];
                }
            }
            const uint4 pv = Piece<T>::pack(fo);
            if (nt) nt_store16(out + e[k], pv);
            else *(uint4*)(out + e[k]) = pv;
            if (MODE == 1 && chain) {
                float fn[EPB];
                Piece<T>::unpack(pv, fo);  // the values as stored: what the separate pass would read
                Piece<T>::unpack(vn[k], fn);
#pragma unroll
                for (int j = 0; j < EPB; ++j) {
                    nP[j] += fo[j];
                    nQ[j] = fmaf(fo[j], silu_f(fn[j]), nQ[j]);
                }
            }
            if (MODE == 0) {
                Piece<T>::unpack(pv, fo);  // sums of the values as stored (what the weight-gradient kernel will read)
#pragma unroll
                for (int j = 0; j < EPB; ++j) acc[j] += fo[j];
            }
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
    if (MODE == 1 && chain) {  // exactly gn_bwd_stats_kernel's reduction
        const int R = bd / CPP, row = tid / CPP;
#pragma unroll
        for (int j = 0; j < EPB; ++j) {
            red[(row * C + c * EPB + j) * 2 + 0] = nP[j];
            red[(row * C + c * EPB + j) * 2 + 1] = nQ[j];
        }
        __syncthreads();
        for (int i = tid; i < C * 2; i += bd) {
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += red[r * C * 2 + i];
            nstats[(((size_t)b * gridDim.x + part) * C) * 2 + i] = t;
        }
    }
}
hipError_t gn_bwd_apply_launch(int dtype, int mode, const void* g, const void* u, const void* gy, const void* extra,
                               const float* coef, const float* scale, const float* shift, void* out, float* sums, int B,
                               int HW, int C, hipStream_t s, const void* nu, float* nstats) {
    const int epb = dtype == DT_BF16 ? 8 : 4;
    if (C % epb) return hipErrorInvalidValue;
    const int cpp = C / epb, bd = tr_bd(cpp);
    if (bd % cpp) return hipErrorInvalidValue;
    dim3 grid(tr_nparts(dtype, HW, C), B);
    if (nstats && (mode != 1 || !nu)) return hipErrorInvalidValue;
    const int nt = nt_streaming((size_t)B * HW * C * (dtype == DT_BF16 ? 2 : 4));
    const size_t lds = (size_t)(bd / cpp) * C * 4 * (nstats ? 2 : 1);
#define DDIMX_L(TT, M)                                                                                                      \
    hipLaunchKernelGGL((gn_bwd_apply_kernel<TT, M>), grid, dim3(bd), lds, s, (const TT*)g, (const TT*)u, (const TT*)gy, \
                       (const TT*)extra, coef, scale, shift, (TT*)out, sums, HW, C, resid_iters(dtype, HW, C), (const TT*)nu, nstats, nt)
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


// =====================================================================================================
// FNet bottleneck, training (models/diffusion.py:123-167 + transformers modeling_fnet.py:138-279)
// =====================================================================================================
// Dropout masks are a pure function of (seed, stream, element index) so the backward regenerates them instead of
// storing them.  (The reference draws them from torch's global RNG; only the distribution can be matched.)
__device__ __forceinline__ float dropout_keep(unsigned long long seed, unsigned stream, unsigned long long e, unsigned thresh,
                                              float inv_keep) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)stream + 1) + e * 0xD1342543DE82EF95ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) >= thresh ? inv_keep : 0.f;
}
static inline unsigned drop_thresh(float p) { return p <= 0.f ? 0u : (unsigned)((double)p * 4294967296.0); }

// seed_ctr (nullable, device): added to the seed when the launch RUNS -- a training step replayed from a hipGraph keeps its
// per-call mask stream by bumping that counter between replays, where an eager step passes a new seed by value
__global__ void __launch_bounds__(256) dropout_apply_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n,
                                                            unsigned long long seed, const unsigned long long* __restrict__ seed_ctr,
                                                            unsigned stream, unsigned thresh, float inv_keep) {
    if (seed_ctr) seed += *seed_ctr;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
        dst[i] = src[i] * dropout_keep(seed, stream, (unsigned long long)i, thresh, inv_keep);
}
hipError_t dropout_apply_launch(const float* src, float* dst, long long n, float p, unsigned long long seed, unsigned stream,
                                hipStream_t s, const unsigned long long* seed_ctr) {
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n, seed, seed_ctr, stream, drop_thresh(p),
                       1.0f / (1.0f - p));
    return hipGetLastError();
}

// y = LN(drop(x) + add[m % add_rows]) * gamma + beta; keeps the pre-norm row (sum_out, nullable) and (mean, rstd)
template <typename TX>
__global__ void __launch_bounds__(256) ln_train_kernel(const TX* __restrict__ x, const float* __restrict__ add, int add_rows,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ y, float* __restrict__ sum_out,
                                                       float* __restrict__ stat, int N, unsigned long long seed,
                                                       const unsigned long long* __restrict__ seed_ctr, unsigned stream,
                                                       unsigned thresh, float inv_keep) {
    __shared__ float red[4];
    __shared__ float bc;
    const int m = blockIdx.x, tid = threadIdx.x;
    if (seed_ctr) seed += *seed_ctr;
    const TX* xr = x + (size_t)m * N;
    const float* ar = add ? add + (size_t)(m % add_rows) * N : nullptr;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        v[i] = 0.f;
        if (n < N) {
            float xv = to_f<TX>(xr[n]);
            if (thresh) xv *= dropout_keep(seed, stream, (unsigned long long)m * N + n, thresh, inv_keep);
            v[i] = xv + (ar ? ar[n] : 0.f);
            s += v[i];
            if (sum_out) sum_out[(size_t)m * N + n] = v[i];
        }
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
    if (tid == 0) {
        bc = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
        stat[(size_t)m * 2 + 0] = mean;
        stat[(size_t)m * 2 + 1] = bc;
    }
    __syncthreads();
    const float rstd = bc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        if (n < N) y[(size_t)m * N + n] = (v[i] - mean) * rstd * gamma[n] + beta[n];
    }
}
hipError_t ln_train_launch(int x_dtype, const void* x, const float* add, int add_rows, const float* gamma, const float* beta,
                           float eps, float* y, float* sum_out, float* stat, int M, int N, float p, unsigned long long seed,
                           unsigned stream, hipStream_t s, const unsigned long long* seed_ctr) {
    if (N > 2048) return hipErrorInvalidValue;
    const unsigned th = drop_thresh(p);
    const float ik = 1.0f / (1.0f - p);
    if (x_dtype == DT_BF16)
        hipLaunchKernelGGL(ln_train_kernel<__bf16>, dim3(M), dim3(256), 0, s, (const __bf16*)x, add, add_rows, gamma, beta, eps, y,
                           sum_out, stat, N, seed, seed_ctr, stream, th, ik);
    else
        hipLaunchKernelGGL(ln_train_kernel<float>, dim3(M), dim3(256), 0, s, (const float*)x, add, add_rows, gamma, beta, eps, y,
                           sum_out, stat, N, seed, seed_ctr, stream, th, ik);
    return hipGetLastError();
}

// LayerNorm backward.  x: the pre-norm rows (fp32 or, for the embedding norm, TX + add rows); stat: (mean, rstd).
//   dx = rstd (gamma dy - mean_n(gamma dy) - xhat mean_n(gamma dy xhat)),  dgamma = sum_m dy xhat,  dbeta = sum_m dy
// A block walks kLnRows rows and keeps the per-column parameter sums in registers: partial [nblocks][2][N].
constexpr int kLnRows = 8;
template <typename TX>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ add, int add_rows, const float* __restrict__ stat,
                                                     const float* __restrict__ gamma, float* __restrict__ dx,
                                                     float* __restrict__ partial, int M, int N) {
    __shared__ float r1[4], r2[4];
    __shared__ float b1, b2;
    const int tid = threadIdx.x;
    float gm[8], dgam[8], dbet[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        gm[i] = n < N ? gamma[n] : 0.f;
        dgam[i] = dbet[i] = 0.f;
    }
    for (int rr = 0; rr < kLnRows; ++rr) {
        const int m = blockIdx.x * kLnRows + rr;
        if (m >= M) break;  // uniform
        const float mean = stat[(size_t)m * 2], rstd = stat[(size_t)m * 2 + 1];
        const float* ar = add ? add + (size_t)(m % add_rows) * N : nullptr;
        float g[8], xh[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid + i * 256;
            g[i] = xh[i] = 0.f;
            if (n < N) {
                const float d = dy[(size_t)m * N + n];
                xh[i] = (to_f<TX>(x[(size_t)m * N + n]) + (ar ? ar[n] : 0.f) - mean) * rstd;
                g[i] = d * gm[i];
                s1 += g[i];
                s2 = fmaf(g[i], xh[i], s2);
                dgam[i] = fmaf(d, xh[i], dgam[i]);
                dbet[i] += d;
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        __syncthreads();
        if ((tid & 63) == 0) { r1[tid >> 6] = s1; r2[tid >> 6] = s2; }
        __syncthreads();
        if (tid == 0) { b1 = (r1[0] + r1[1] + r1[2] + r1[3]) / (float)N; b2 = (r2[0] + r2[1] + r2[2] + r2[3]) / (float)N; }
        __syncthreads();
        const float m1 = b1, m2 = b2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid + i * 256;
            if (n < N) dx[(size_t)m * N + n] = rstd * (g[i] - m1 - xh[i] * m2);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        if (n < N) {
            partial[((size_t)blockIdx.x * 2 + 0) * N + n] = dgam[i];
            partial[((size_t)blockIdx.x * 2 + 1) * N + n] = dbet[i];
        }
    }
}
int ln_bwd_nblocks(int M) { return (M + kLnRows - 1) / kLnRows; }
hipError_t ln_bwd_launch(int x_dtype, const float* dy, const void* x, const float* add, int add_rows, const float* stat,
                         const float* gamma, float* dx, float* partial, float* dgamma, float* dbeta, int M, int N, hipStream_t s) {
    if (N > 2048) return hipErrorInvalidValue;
    const int nb = ln_bwd_nblocks(M);
    if (x_dtype == DT_BF16)
        hipLaunchKernelGGL(ln_bwd_kernel<__bf16>, dim3(nb), dim3(256), 0, s, dy, (const __bf16*)x, add, add_rows, stat, gamma, dx, partial, M, N);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, dy, (const float*)x, add, add_rows, stat, gamma, dx, partial, M, N);
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 15) / 16), dim3(256), 0, s, partial, nb, 2ll * N, N, dgamma);
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 15) / 16), dim3(256), 0, s, partial + N, nb, 2ll * N, N, dbeta);
    return hipGetLastError();
}

// gelu_new and its derivative (transformers activations.py:59-66)
__device__ __forceinline__ float dgelu_new_f(float v) {
    const float k = 0.7978845608028654f, a = 0.044715f;
    const float t = tanhf(k * (v + a * v * v * v));
    return 0.5f * (1.0f + t) + 0.5f * v * (1.0f - t * t) * k * (1.0f + 3.0f * a * v * v);
}
// mode 0: dst = gelu_new(src);  mode 1: dst = src * gelu_new'(aux)
__global__ void __launch_bounds__(256) gelu_kernel(const float* __restrict__ src, const float* __restrict__ aux,
                                                   float* __restrict__ dst, long long n, int mode) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
        dst[i] = mode ? src[i] * dgelu_new_f(aux[i]) : gelu_new_f(src[i]);
}
hipError_t gelu_launch(const float* src, const float* aux, float* dst, long long n, int mode, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gelu_kernel, dim3(blocks), dim3(256), 0, s, src, aux, dst, n, mode);
    return hipGetLastError();
}

// dst[c][r] = f(src[r][c]);  f = identity or gelu_new   (32x32 tiles through LDS)
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C,
                                                        int act) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) { v = src[(size_t)r * C + c]; if (act) v = gelu_new_f(v); }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < R && c < C) dst[(size_t)c * R + r] = tile[tx][j];
    }
}
hipError_t transpose_launch(const float* src, float* dst, int R, int C, int act, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, dst, R, C, act);
    return hipGetLastError();
}

template <typename T>
__global__ void __launch_bounds__(256) cast_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) dst[i] = to_f<T>(src[i]);
}
hipError_t cast_f32_launch(int dtype, const void* src, float* dst, long long n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == DT_BF16) hipLaunchKernelGGL(cast_f32_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)src, dst, n);
    else hipLaunchKernelGGL(cast_f32_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, dst, n);
    return hipGetLastError();
}

// =====================================================================================================
// timestep-embedding MLP backward (models/diffusion.py:110-120): tiny dense layers, batch rows <= a few dozen
// =====================================================================================================
// dW[n][k] = sum_b dy[b][n] * f(x[row(b)][k]),  db[n] = sum_b dy[b][n];  f = SiLU when x_silu (x holds pre-activations)
__global__ void __launch_bounds__(256) linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const int64_t* __restrict__ idx, float* __restrict__ dW,
                                                           float* __restrict__ db, int B, int N, int K, int x_silu) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < K) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const size_t row = idx ? (size_t)idx[b] : (size_t)b;
            float xv = x[row * K + k];
            if (x_silu) xv = silu_f(xv);
            acc = fmaf(dy[(size_t)b * N + n], xv, acc);
        }
        dW[(size_t)n * K + k] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float t = 0.f;
        for (int b = 0; b < B; ++b) t += dy[(size_t)b * N + n];
        db[n] = t;
    }
}
// dx[b][k] = (sum_n dy[b][n] W[n][k]) * SiLU'(xpre[b][k]); block = 16 k x 16 n-lanes
__global__ void __launch_bounds__(256) linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                           const float* __restrict__ xpre, float* __restrict__ dx, int N, int K) {
    __shared__ float red[16][17];
    const int kl = threadIdx.x & 15, nl = threadIdx.x >> 4;
    const int b = blockIdx.y, k = blockIdx.x * 16 + kl;
    float acc = 0.f;
    if (k < K)
        for (int n = nl; n < N; n += 16) acc = fmaf(dy[(size_t)b * N + n], W[(size_t)n * K + k], acc);
    red[nl][kl] = acc;
    __syncthreads();
    if (nl == 0 && k < K) {
        acc = 0.f;
        for (int j = 0; j < 16; ++j) acc += red[j][kl];
        dx[(size_t)b * K + k] = acc * dsilu_f(xpre[(size_t)b * K + k]);
    }
}
hipError_t linear_bwd_w_launch(const float* dy, const float* x, const int64_t* idx, float* dW, float* db, int B, int N, int K,
                               int x_silu, hipStream_t s) {
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((K + 255) / 256, N), dim3(256), 0, s, dy, x, idx, dW, db, B, N, K, x_silu);
    return hipGetLastError();
}
hipError_t linear_bwd_x_launch(const float* dy, const float* W, const float* xpre, float* dx, int B, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((K + 15) / 16, B), dim3(256), 0, s, dy, W, xpre, dx, N, K);
    return hipGetLastError();
}

// =====================================================================================================
// edge convolutions (models/diffusion.py:189-208): Conv2d(cio -> C0) at the input, Conv2d(C0 -> cio) at the output
// =====================================================================================================
// data gradient of the output conv: ds[b][y][x][c] = sum_{k,o} d_eps[b][o][y-ky+1][x-kx+1] * w[k][o][c]   (w: packed [9][cout][C0])
// (d_eps NCHW fp32, ds NHWC T; it is the gradient of BOTH summands of `x + hidden[0]`, :284)
template <typename T>
__global__ void __launch_bounds__(256) conv_out_bwd_data_kernel(const float* __restrict__ de, const float* __restrict__ w,
                                                                T* __restrict__ ds, int C0, int cout, int H, int W) {
    constexpr int EPB = Piece<T>::N;
    extern __shared__ float wl[];  // [k][o][c]: the forward's packed layout, copied as is
    for (int i = threadIdx.x; i < 9 * cout * C0; i += 256) wl[i] = w[i];
    __syncthreads();
    const int CPP = C0 / EPB;
    const long long pieces = (long long)H * W * CPP;
    const int b = blockIdx.y;
    for (long long pc = blockIdx.x * 256ll + threadIdx.x; pc < pieces; pc += gridDim.x * 256ll) {
        const int c0 = (int)(pc % CPP) * EPB;
        const long long pix = pc / CPP;
        const int y = (int)(pix / W), x = (int)(pix % W);
        float acc[EPB];
#pragma unroll
        for (int j = 0; j < EPB; ++j) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y - k / 3 + 1, xx = x - k % 3 + 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const size_t off = ok ? (size_t)yy * W + xx : 0;
            for (int o = 0; o < cout; ++o) {
                const float dv = de[((size_t)b * cout + o) * H * W + off];
                const float d = ok ? dv : 0.f;
                const float* wp = wl + (k * cout + o) * C0 + c0;
#pragma unroll
                for (int j = 0; j < EPB; ++j) acc[j] = fmaf(d, wp[j], acc[j]);
            }
        }
        *(uint4*)(ds + ((size_t)b * H * W + pix) * C0 + c0) = Piece<T>::pack(acc);
    }
}
// Fast path (C0 = 32, cout = 2): every thread keeps the 9 x 2 x EPB weights of its channel piece in registers and walks
// pixels; the 18 d_eps values of a pixel are shared by the piece lanes (same address: one transaction).
template <typename T>
__global__ void __launch_bounds__(256) conv_out_bwd_data_reg_kernel(const float* __restrict__ de, const float* __restrict__ w,
                                                                    T* __restrict__ ds, int H, int W) {
    constexpr int C0 = 32, COUT = 2, EPB = Piece<T>::N, PPB = C0 / EPB, PIX = 256 / PPB;
    const int j = threadIdx.x % PPB, pl = threadIdx.x / PPB, b = blockIdx.y;
    float wr[9 * COUT][EPB];
#pragma unroll
    for (int ko = 0; ko < 9 * COUT; ++ko)
#pragma unroll
        for (int e = 0; e < EPB; ++e) wr[ko][e] = w[ko * C0 + j * EPB + e];
    const long long HW = (long long)H * W;
    const float* d0 = de + (size_t)b * COUT * HW;
    for (long long pix = (long long)blockIdx.x * PIX + pl; pix < HW; pix += (long long)gridDim.x * PIX) {
        const int y = (int)(pix / W), x = (int)(pix % W);
        float dv[9 * COUT];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y - k / 3 + 1, xx = x - k % 3 + 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const size_t off = ok ? (size_t)yy * W + xx : 0;
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float v = d0[(size_t)o * HW + off];
                dv[k * COUT + o] = ok ? v : 0.f;
            }
        }
        float acc[EPB];
#pragma unroll
        for (int e = 0; e < EPB; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ko = 0; ko < 9 * COUT; ++ko)
#pragma unroll
            for (int e = 0; e < EPB; ++e) acc[e] = fmaf(dv[ko], wr[ko][e], acc[e]);
        *(uint4*)(ds + ((size_t)b * HW + pix) * C0 + j * EPB) = Piece<T>::pack(acc);
    }
}
hipError_t conv_out_bwd_data_launch(int dtype, const float* d_eps, const float* w, void* ds, int B, int C0, int cout, int H,
                                    int W, hipStream_t s) {
    if (C0 == 32 && cout == 2) {
        const long long HW = (long long)H * W;
        const int pixb = dtype == DT_BF16 ? 64 : 32;
        const long long want = (HW + pixb * 8 - 1) / (pixb * 8);  // ~8 pixels per thread
        dim3 grid((unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want)), B);
        if (dtype == DT_BF16) hipLaunchKernelGGL(conv_out_bwd_data_reg_kernel<__bf16>, grid, dim3(256), 0, s, d_eps, w, (__bf16*)ds, H, W);
        else hipLaunchKernelGGL(conv_out_bwd_data_reg_kernel<float>, grid, dim3(256), 0, s, d_eps, w, (float*)ds, H, W);
        return hipGetLastError();
    }
    const int epb = dtype == DT_BF16 ? 8 : 4;
    if (C0 % epb) return hipErrorInvalidValue;
    const long long pieces = (long long)H * W * (C0 / epb);
    dim3 grid((unsigned)((pieces + 255) / 256 < 4096 ? (pieces + 255) / 256 : 4096), B);
    const size_t lds = (size_t)9 * cout * C0 * 4;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(conv_out_bwd_data_kernel<__bf16>, grid, dim3(256), lds, s, d_eps, w, (__bf16*)ds, C0, cout, H, W);
    else
        hipLaunchKernelGGL(conv_out_bwd_data_kernel<float>, grid, dim3(256), lds, s, d_eps, w, (float*)ds, C0, cout, H, W);
    return hipGetLastError();
}

// weight gradients of both edge convs as one correlation:
//   R[kk][i][c] = sum_{b,y,x} G[b][y][x][c] * S[b][i][y + ky - 1][x + kx - 1]     (zero padding of S)
// G = g1 (+ g2): NHWC T with C channels; S: NCHW fp32 with NI <= 4 planes.
//   input conv : G = d(hidden[0]), S = x       -> dW_in[c][i][kk]  = R[kk][i][c],     db_in[c]  = sumG[c]
//   output conv: G = x + hidden[0], S = d_eps  -> dW_out[i][c][kk] = R[8 - kk][i][c], db_out[i] = sumS[i]
// persistent blocks over 16x16 tiles; partial [nblocks][9*NI*C + C + NI]; edge_wgrad_reduce maps to the layouts.
constexpr int kEdgeT = 16;
template <typename T>
__global__ void __launch_bounds__(288) edge_wgrad_kernel(const T* __restrict__ g1, const T* __restrict__ g2,
                                                         const float* __restrict__ S, float* __restrict__ partial, int C, int NI,
                                                         int H, int W, int tiles_x, int tiles_y, int total_tiles) {
    extern __shared__ float sm[];
    float* Gt = sm;                                   // [256][C + 1]
    float* St = sm + 256 * (C + 1);                   // [NI][18][18]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int items = 9 * C;
    float acc[2][4];  // up to 2 items per thread (items <= 2 * blockDim), NI <= 4
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[q][i] = 0.f;
    float sumg = 0.f, sums = 0.f;
    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        const int b = t / (tiles_x * tiles_y), tt = t % (tiles_x * tiles_y);
        const int y0 = (tt / tiles_x) * kEdgeT, x0 = (tt % tiles_x) * kEdgeT;
        __syncthreads();
        for (int i = tid; i < 256 * C; i += nthr) {
            const int c = i % C, p = i / C;
            const int y = y0 + p / kEdgeT, x = x0 + p % kEdgeT;
            float v = 0.f;
            if (y < H && x < W) {
                const size_t e = (((size_t)b * H + y) * W + x) * C + c;
                v = to_f<T>(g1[e]);
                if (g2) v += to_f<T>(g2[e]);
            }
            Gt[p * (C + 1) + c] = v;
        }
        for (int i = tid; i < NI * 18 * 18; i += nthr) {
            const int xx = i % 18, yy = (i / 18) % 18, pl = i / 324;
            const int y = y0 + yy - 1, x = x0 + xx - 1;
            St[i] = (y >= 0 && y < H && x >= 0 && x < W) ? S[(((size_t)b * NI + pl) * H + y) * W + x] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int it = tid + q * nthr;
            if (it >= items) break;
            const int c = it % C, kk = it / C;
            const int ky = kk / 3, kx = kk % 3;
            for (int p = 0; p < 256; ++p) {
                const float gv = Gt[p * (C + 1) + c];
                const int so = (p / kEdgeT + ky) * 18 + p % kEdgeT + kx;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < NI) acc[q][i] = fmaf(gv, St[i * 324 + so], acc[q][i]);
            }
        }
        if (tid < C) {
            for (int p = 0; p < 256; ++p) sumg += Gt[p * (C + 1) + tid];
        } else if (tid - C < NI) {
            const int pl = tid - C;
            for (int p = 0; p < 256; ++p) sums += St[pl * 324 + (p / kEdgeT + 1) * 18 + p % kEdgeT + 1];
        }
    }
    float* out = partial + (size_t)blockIdx.x * (9 * NI * C + C + NI);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int it = tid + q * nthr;
        if (it >= items) break;
        const int c = it % C, kk = it / C;
        for (int i = 0; i < NI; ++i) out[(kk * NI + i) * C + c] = acc[q][i];
    }
    if (tid < C) out[9 * NI * C + tid] = sumg;
    else if (tid - C < NI) out[9 * NI * C + C + tid - C] = sums;
}
// Fast path for the reference shape (C = 32 channels, NI = 2 planes).  A block owns a strip of PIXB columns x `rows`
// rows of one sample; thread = (pixel column, 16-byte channel piece): G is read with one 16-byte load per pixel, the
// 3x3xNI neighbourhood of S comes from an LDS tile (staged 32 rows at a time), and each thread keeps EPB x 9 x NI
// accumulators in registers for the whole strip.  The pixel lanes are folded at the end (wave shuffles, then LDS).
constexpr int kEdgeChunk = 32;  // rows of S staged per LDS tile
template <typename T>
__global__ void __launch_bounds__(256) edge_wgrad_strip_kernel(const T* __restrict__ g1, const T* __restrict__ g2,
                                                               const float* __restrict__ S, float* __restrict__ partial, int H,
                                                               int W, int rows, int sx, int sy) {
    constexpr int C = 32, NI = 2, EPB = Piece<T>::N, PPB = C / EPB, PIXB = 256 / PPB, SW = PIXB + 2;
    constexpr int NACC = EPB * 9 * NI;
    __shared__ float St[NI][kEdgeChunk + 2][SW];
    __shared__ float red[4][PPB][NACC + EPB + NI];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = tid % PPB, p = tid / PPB;
    const int strip = blockIdx.x;
    const int b = strip / (sx * sy), x0 = (strip % sx) * PIXB, y0 = ((strip / sx) % sy) * rows;
    const int y1 = y0 + rows < H ? y0 + rows : H;
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    float sumg[EPB], sums[NI];
#pragma unroll
    for (int e = 0; e < EPB; ++e) sumg[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) sums[i] = 0.f;
    const int x = x0 + p;
    const bool xok = x < W;
    for (int yc = y0; yc < y1; yc += kEdgeChunk) {
        const int nr = y1 - yc < kEdgeChunk ? y1 - yc : kEdgeChunk;
        __syncthreads();
        for (int i = tid; i < NI * (kEdgeChunk + 2) * SW; i += 256) {
            const int cx = i % SW, ry = (i / SW) % (kEdgeChunk + 2), pl = i / (SW * (kEdgeChunk + 2));
            const int gy = yc - 1 + ry, gx = x0 - 1 + cx;
            const bool ok = ry < nr + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const float v = S[(((size_t)b * NI + pl) * H + (ok ? gy : 0)) * W + (ok ? gx : 0)];
            St[pl][ry][cx] = ok ? v : 0.f;
        }
        __syncthreads();
        // G pieces are requested one row ahead of their use (clamped address, masked after the load)
        const size_t ebase = (((size_t)b * H + yc) * W + (xok ? x : x0)) * C + j * EPB;
        const size_t rstride = (size_t)W * C;
        uint4 n1 = *(const uint4*)(g1 + ebase), n2 = make_uint4(0, 0, 0, 0);
        if (g2) n2 = *(const uint4*)(g2 + ebase);
#pragma unroll 1
        for (int r = 0; r < nr; ++r) {
            const uint4 c1 = n1, c2 = n2;
            const size_t en = ebase + (size_t)(r + 1 < nr ? r + 1 : r) * rstride;
            n1 = *(const uint4*)(g1 + en);
            if (g2) n2 = *(const uint4*)(g2 + en);
            float g[EPB];
            Piece<T>::unpack(c1, g);
            if (g2) {
                float g2v[EPB];
                Piece<T>::unpack(c2, g2v);
#pragma unroll
                for (int q = 0; q < EPB; ++q) g[q] += g2v[q];
            }
            if (!xok) {
#pragma unroll
                for (int q = 0; q < EPB; ++q) g[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < EPB; ++q) sumg[q] += g[q];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (xok) sums[i] += St[i][r + 1][p + 1];
#pragma unroll
                for (int kk = 0; kk < 9; ++kk) {
                    const float sv = St[i][r + kk / 3][p + kk % 3];
#pragma unroll
                    for (int q = 0; q < EPB; ++q) acc[(q * 9 + kk) * NI + i] = fmaf(g[q], sv, acc[(q * 9 + kk) * NI + i]);
                }
            }
        }
    }
    // fold the pixel lanes of a wave (lanes with equal j: xor over the lane bits above log2(PPB)), then the 4 waves
    auto fold = [&](float v) __attribute__((always_inline)) -> float {
#pragma unroll
        for (int o = PPB; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = fold(acc[k]);
#pragma unroll
    for (int q = 0; q < EPB; ++q) sumg[q] = fold(sumg[q]);
#pragma unroll
    for (int i = 0; i < NI; ++i) sums[i] = fold(sums[i]);
    __syncthreads();
    if (lane < PPB) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) red[wave][lane][k] = acc[k];
#pragma unroll
        for (int q = 0; q < EPB; ++q) red[wave][lane][NACC + q] = sumg[q];
#pragma unroll
        for (int i = 0; i < NI; ++i) red[wave][lane][NACC + EPB + i] = sums[i];
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * (9 * NI * C + C + NI);
    for (int k = tid; k < PPB * (NACC + EPB); k += 256) {
        const int jj = k / (NACC + EPB), r = k % (NACC + EPB);
        const float v = (red[0][jj][r] + red[1][jj][r]) + (red[2][jj][r] + red[3][jj][r]);
        if (r < NACC) {
            const int i = r % NI, kk = (r / NI) % 9, q = r / (NI * 9);
            out[(kk * NI + i) * C + jj * EPB + q] = v;
        } else {
            out[9 * NI * C + jj * EPB + (r - NACC)] = v;
        }
    }
    if (tid < NI)  // every piece lane of a pixel added the same S values: take piece 0
        out[9 * NI * C + C + tid] = (red[0][0][NACC + EPB + tid] + red[1][0][NACC + EPB + tid]) +
                                    (red[2][0][NACC + EPB + tid] + red[3][0][NACC + EPB + tid]);
}

// mode 0 (input conv): dW[c][i][kk], db[c] = sumG;  mode 1 (output conv): dW[i][c][kk] = R[8-kk], db[i] = sumS
__global__ void __launch_bounds__(256) edge_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks, int C, int NI,
                                                                int mode, float* __restrict__ dW, float* __restrict__ db) {
    __shared__ double red[4][64];
    const int per = 9 * NI * C + C + NI;
    const int o64 = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o64;
    double s = 0.0;
    if (i < per) {  // (same order of addition as the one-load-per-iteration loop this was: eight loads in flight)
        int k = q;
        for (; k + 28 < nblocks; k += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + 4 * u) * per + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; k < nblocks; k += 4) s += (double)partial[(size_t)k * per + i];
    }
    red[q][o64] = s;
    __syncthreads();
    if (q != 0 || i >= per) return;
    s = (red[0][o64] + red[1][o64]) + (red[2][o64] + red[3][o64]);
    if (i < 9 * NI * C) {
        const int c = i % C, pl = (i / C) % NI, kk = i / (C * NI);
        if (mode == 0) dW[((size_t)c * NI + pl) * 9 + kk] = (float)s;
        else dW[((size_t)pl * C + c) * 9 + (8 - kk)] = (float)s;
    } else if (i < 9 * NI * C + C) {
        if (mode == 0) db[i - 9 * NI * C] = (float)s;
    } else if (mode == 1) {
        db[i - 9 * NI * C - C] = (float)s;
    }
}
constexpr int kStripRows = 128;
static inline bool edge_fast(int C, int NI) { return C == 32 && NI == 2; }
static inline int edge_pixb(int dtype) { return dtype == DT_BF16 ? 64 : 32; }
int edge_wgrad_nblocks(int dtype, int B, int C, int NI, int H, int W) {
    if (edge_fast(C, NI)) {
        const int pb = edge_pixb(dtype);
        return (int)((long long)B * ((W + pb - 1) / pb) * ((H + kStripRows - 1) / kStripRows));
    }
    const long long t = (long long)B * ((H + kEdgeT - 1) / kEdgeT) * ((W + kEdgeT - 1) / kEdgeT);
    return (int)(t < 1024 ? t : 1024);
}
size_t edge_wgrad_partial_floats(int dtype, int B, int C, int NI, int H, int W) {
    return (size_t)edge_wgrad_nblocks(dtype, B, C, NI, H, W) * (9 * NI * C + C + NI);
}
hipError_t edge_wgrad_launch(int dtype, int mode, const void* g1, const void* g2, const float* S, float* partial, float* dW,
                             float* db, int B, int C, int NI, int H, int W, hipStream_t s) {
    if (NI > 4 || 9 * C > 2 * 288 || C + NI > 288) return hipErrorInvalidValue;
    const int nb = edge_wgrad_nblocks(dtype, B, C, NI, H, W);
    if (edge_fast(C, NI)) {
        const int pb = edge_pixb(dtype);
        const int sx = (W + pb - 1) / pb, sy = (H + kStripRows - 1) / kStripRows;
        if (dtype == DT_BF16)
            hipLaunchKernelGGL(edge_wgrad_strip_kernel<__bf16>, dim3(nb), dim3(256), 0, s, (const __bf16*)g1, (const __bf16*)g2, S, partial,
                               H, W, kStripRows, sx, sy);
        else
            hipLaunchKernelGGL(edge_wgrad_strip_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)g1, (const float*)g2, S, partial,
                               H, W, kStripRows, sx, sy);
    } else {
        const int tx = (W + kEdgeT - 1) / kEdgeT, ty = (H + kEdgeT - 1) / kEdgeT;
        const size_t lds = (size_t)(256 * (C + 1) + NI * 324) * 4;
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        if (dtype == DT_BF16)
            hipLaunchKernelGGL(edge_wgrad_kernel<__bf16>, dim3(nb), dim3(288), lds, s, (const __bf16*)g1, (const __bf16*)g2, S, partial,
                               C, NI, H, W, tx, ty, B * tx * ty);
        else
            hipLaunchKernelGGL(edge_wgrad_kernel<float>, dim3(nb), dim3(288), lds, s, (const float*)g1, (const float*)g2, S, partial, C,
                               NI, H, W, tx, ty, B * tx * ty);
    }
    const int per = 9 * NI * C + C + NI;
    hipLaunchKernelGGL(edge_wgrad_reduce_kernel, dim3((per + 63) / 64), dim3(256), 0, s, partial, nb, C, NI, mode, dW, db);
    return hipGetLastError();
}

// d(out)[b] = 2 * g[b] * (out[b] - e[b])   (functions/losses.py:18: per-sample sum of squares; g = upstream gradient per sample)
// with_mean: g has B + 1 entries, the last one the upstream gradient of the batch MEAN (the loss vector's [B] entry): + g[B] / B per sample
__global__ void __launch_bounds__(256) sqerr_bwd_kernel(const float* __restrict__ e, const float* __restrict__ o,
                                                        const float* __restrict__ g, float* __restrict__ d, long long per, int with_mean) {
    const int b = blockIdx.y;
    const float c = 2.0f * (g[b] + (with_mean ? g[gridDim.y] / (float)gridDim.y : 0.f));
    const size_t base = (size_t)b * per;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < per; i += gridDim.x * 256ll) d[base + i] = c * (o[base + i] - e[base + i]);
}
hipError_t sqerr_bwd_launch(const float* e, const float* out, const float* g, float* d, int B, long long per, hipStream_t s, int with_mean) {
    const int blocks = (int)((per + 255) / 256 < 1024 ? (per + 255) / 256 : 1024);
    hipLaunchKernelGGL(sqerr_bwd_kernel, dim3(blocks, B), dim3(256), 0, s, e, out, g, d, per, with_mean);
    return hipGetLastError();
}

}  // namespace ddimx
