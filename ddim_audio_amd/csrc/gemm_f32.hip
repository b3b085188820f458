// fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) for the FNet bottleneck:
//   C[z][M][N] (+)= A[z][M][K] * B[z][N][K]^T  (+bias[n]) (gelu_new) (+resid[m][n])
// Both operands are K-contiguous ("NT"), which is how torch Linear weights ([N][K]) and token
// matrices ([M][K]) are laid out, and how the DFT-as-GEMM factors are arranged (fnet in api.cpp).
// 64x64 workgroup tile, 4 waves (2x2) of one 32x32 MFMA tile each, BK = 32, register-staged
// double-buffered LDS.  Arbitrary M, N, K (zero-filled edges); the FNet is ~1 % of the FLOPs.
#include "kernels.h"

namespace ddimx {

constexpr int GBM = 64, GBN = 64, GBK = 32, GLS = GBK + 4;  // LDS row stride 36 floats = 9 slots (odd)

__device__ __forceinline__ float4 gemm_ld4(const float* base, int row, int rows, int k, int K, int ld, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
        const float* p = base + (size_t)row * ld + k;
        if (vec && k + 3 < K) {
            v = *(const float4*)p;
        } else {
            if (k < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
            if (k + 3 < K) v.w = p[3];
        }
    }
    return v;
}

__global__ void __launch_bounds__(256) gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float sA[2][GBM * GLS];
    __shared__ __attribute__((aligned(16))) float sB[2][GBN * GLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN, z = blockIdx.z;
    const float* A = g.A + (size_t)z * g.sA;
    const float* B = g.B + (size_t)z * g.sB;
    float* C = g.C + (size_t)z * g.sC;
    const bool va = (g.lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    const bool vb = (g.ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);

    // staging: 64 rows x 8 float4 per operand = 512 float4 -> 2 per thread
    float4 ra[2], rb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = tid + i * 256, row = pc >> 3, kq = (pc & 7) * 4;
            ra[i] = gemm_ld4(A, m0 + row, g.M, k0 + kq, g.K, g.lda, va);
            rb[i] = gemm_ld4(B, n0 + row, g.N, k0 + kq, g.K, g.ldb, vb);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = tid + i * 256, row = pc >> 3, kq = (pc & 7) * 4;
            *(float4*)&sA[buf][row * GLS + kq] = ra[i];
            *(float4*)&sB[buf][row * GLS + kq] = rb[i];
        }
    };
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nk = (g.K + GBK - 1) / GBK;
    load(0);
    store(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) load((kc + 1) * GBK);
        const float* pa = &sA[kc & 1][(wm * 32 + l31) * GLS + h * 4];
        const float* pb = &sB[kc & 1][(wn * 32 + l31) * GLS + h * 4];
#pragma unroll
        for (int kg = 0; kg < GBK / 8; ++kg) {
            const float4 a = *(const float4*)(pa + kg * 8);
            const float4 b = *(const float4*)(pb + kg * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        if (kc + 1 < nk) store((kc + 1) & 1);
        __syncthreads();
    }
    // D[row m][col n]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + l31;
    if (n < g.N) {
        const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < g.M) {
                const size_t o = (size_t)m * g.ldc + n;
                float v = acc[r];
                if (g.accumulate) v += C[o];
                v += bv;
                if (g.act == 1) v = gelu_new_f(v);
                if (g.resid) v += g.resid[(size_t)z * g.sC + o];
                C[o] = v;
            }
        }
    }
}

hipError_t gemm_f32_launch(const GemmArgs& g, hipStream_t s) {
    dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, g.batch > 0 ? g.batch : 1);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, g);
    return hipGetLastError();
}

}  // namespace ddimx
