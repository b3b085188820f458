// "NT" GEMM for the FNet bottleneck:  C[z][M][N] (+)= A[z][M][K] * B[z][N][K]^T  (+bias) (gelu_new) (+resid)
// Both operands are K-contiguous fp32 in memory (torch Linear weights [N][K], token matrices [M][K],
// DFT-as-GEMM factors).  Two MFMA paths share the structure (64x64 workgroup tile, 4 waves 2x2 of one
// 32x32 tile, register-staged double-buffered LDS):
//   PREC_F32  : v_mfma_f32_32x32x2_f32, exact fp32 FMA chain (parity mode, and always for the DFT factors)
//   PREC_BF16 : operands rounded to bf16 while they are staged into LDS, v_mfma_f32_32x32x16_bf16 with
//               fp32 accumulation (performance mode, dense-weight GEMMs only)
// Skinny shapes (M = B*S is small) are split along K over blockIdx.z; partial tiles go to a workspace and a
// second kernel sums them in a fixed order and applies the epilogue (deterministic, no atomics).
#include "kernels.h"

namespace ddimx {

constexpr int GBM = 64, GBN = 64;

template <int PREC> struct GemmTraits;
template <> struct GemmTraits<0> {  // fp32
    static constexpr int BK = 32, ROWB = BK * 4 + 16;  // 144 B = 9 slots (odd)
};
template <> struct GemmTraits<1> {  // bf16
    static constexpr int BK = 64, ROWB = BK * 2 + 16;  // 144 B
};

__device__ __forceinline__ float4 gemm_ld4(const float* base, int row, int rows, int k, int kend, int ld, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
        const float* p = base + (size_t)row * ld + k;
        if (vec && k + 3 < kend) {
            v = *(const float4*)p;
        } else {
            if (k < kend) v.x = p[0];
            if (k + 1 < kend) v.y = p[1];
            if (k + 2 < kend) v.z = p[2];
            if (k + 3 < kend) v.w = p[3];
        }
    }
    return v;
}

template <int PREC>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const GemmArgs g) {
    typedef GemmTraits<PREC> TR;
    constexpr int BK = TR::BK, ROWB = TR::ROWB;
    constexpr int F4 = BK / 4;          // float4 loads per row per chunk
    constexpr int LPT = GBM * F4 / 256;  // float4 loads per thread per operand (2 for fp32, 4 for bf16)
    __shared__ __attribute__((aligned(16))) char sA[2][GBM * ROWB];
    __shared__ __attribute__((aligned(16))) char sB[2][GBN * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const int z = blockIdx.z / g.splitk, ks = blockIdx.z % g.splitk;
    const float* A = g.A + (size_t)z * g.sA;
    const float* B = g.B + (size_t)z * g.sB;
    const bool va = (g.lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    const bool vb = (g.ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
    // this block's K range (whole chunks)
    const int nchunks_all = (g.K + BK - 1) / BK;
    const int cper = (nchunks_all + g.splitk - 1) / g.splitk;
    const int c0 = ks * cper;
    const int c1 = (c0 + cper < nchunks_all) ? c0 + cper : nchunks_all;
    const int nk = c1 - c0;

    float4 ra[LPT], rb[LPT];
    auto load = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;
            ra[i] = gemm_ld4(A, m0 + row, g.M, k0 + kq, g.K, g.lda, va);
            rb[i] = gemm_ld4(B, n0 + row, g.N, k0 + kq, g.K, g.ldb, vb);
        }
    };
    auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;
            if constexpr (PREC == 0) {
                *(float4*)(sA[buf] + row * ROWB + kq * 4) = ra[i];
                *(float4*)(sB[buf] + row * ROWB + kq * 4) = rb[i];
            } else {
                *(uint2*)(sA[buf] + row * ROWB + kq * 2) =
                    make_uint2(Piece<__bf16>::pk(ra[i].x, ra[i].y), Piece<__bf16>::pk(ra[i].z, ra[i].w));
                *(uint2*)(sB[buf] + row * ROWB + kq * 2) =
                    make_uint2(Piece<__bf16>::pk(rb[i].x, rb[i].y), Piece<__bf16>::pk(rb[i].z, rb[i].w));
            }
        }
    };
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (nk > 0) {
        load(c0 * BK);
        store(0);
    }
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) load((c0 + kc + 1) * BK);
        const char* pa = sA[kc & 1] + (wm * 32 + l31) * ROWB + h * 16;
        const char* pb = sB[kc & 1] + (wn * 32 + l31) * ROWB + h * 16;
#pragma unroll
        for (int kg = 0; kg < (BK * (PREC ? 2 : 4)) / 32; ++kg) {  // 32-byte k-groups
            const uint4 a = *(const uint4*)(pa + kg * 32);
            const uint4 b = *(const uint4*)(pb + kg * 32);
            if constexpr (PREC == 0) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                              __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
            }
        }
        if (kc + 1 < nk) store((kc + 1) & 1);
        __syncthreads();
    }
    // D[row m][col n]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + l31;
    if (n >= g.N) return;
    if (g.splitk > 1) {
        float* P = g.partial + ((size_t)(z * g.splitk + ks) * g.M) * g.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < g.M) P[(size_t)m * g.N + n] = acc[r];
        }
        return;
    }
    float* C = g.C + (size_t)z * g.sC;
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

__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(const GemmArgs g) {
    const long long total = (long long)g.batch * g.M * g.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = i % g.N;
        const long long mz = i / g.N;
        const int m = mz % g.M, z = mz / g.M;
        float v = 0.f;
        for (int s = 0; s < g.splitk; ++s) v += g.partial[((size_t)(z * g.splitk + s) * g.M + m) * g.N + n];
        const size_t o = (size_t)z * g.sC + (size_t)m * g.ldc + n;
        if (g.accumulate) v += g.C[o];
        if (g.bias) v += g.bias[n];
        if (g.act == 1) v = gelu_new_f(v);
        if (g.resid) v += g.resid[o];
        g.C[o] = v;
    }
}

int gemm_pick_splitk(int M, int N, int K, int batch, int bf16) {
    const int bk = bf16 ? 64 : 32;
    const int tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN) * (batch > 0 ? batch : 1);
    const int chunks = (K + bk - 1) / bk;
    if (tiles > 128) return 1;  // the kernel is latency-bound per K-chunk: up to ~1 tile per 2 CUs a split still pays
    int s = 1;
    while (s < 8 && tiles * s * 2 <= 512 && chunks / (s * 2) >= 2) s *= 2;
    return s;
}


// split-K reduce fused with the row LayerNorm that follows the GEMM in the FNet (output.LayerNorm after the FFN):
// one block per output row: v = sum_s partial + bias (+resid); out = LN(v) * gamma + beta.  N <= 2048.
__global__ void __launch_bounds__(256) gemm_reduce_ln_kernel(const GemmArgs g, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ float bc;
    const int m = blockIdx.x, tid = threadIdx.x;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        v[i] = 0.f;
        if (n < g.N) {
            float t = 0.f;
            for (int k = 0; k < g.splitk; ++k) t += g.partial[((size_t)k * g.M + m) * g.N + n];
            if (g.bias) t += g.bias[n];
            if (g.resid) t += g.resid[(size_t)m * g.ldc + n];
            v[i] = t;
            s += t;
        }
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bc = (red[0] + red[1] + red[2] + red[3]) / (float)g.N;
    __syncthreads();
    const float mean = bc;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        if (n < g.N) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    }
    q = wave_sum(q);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) bc = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)g.N + eps);
    __syncthreads();
    const float rstd = bc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = tid + i * 256;
        if (n < g.N) out[(size_t)m * g.N + n] = (v[i] - mean) * rstd * gamma[n] + beta[n];
    }
}

// GEMM (always through the partial workspace, batch 1) followed by the fused reduce + bias + resid + LayerNorm
hipError_t gemm_ln_launch(GemmArgs g, const float* gamma, const float* beta, float eps, float* out, hipStream_t s) {
    if (g.N > 2048 || g.batch > 1 || !g.partial) return hipErrorInvalidValue;
    g.batch = 1;
    if (g.splitk < 1) g.splitk = 1;
    const int sk = g.splitk;
    dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, sk);
    GemmArgs k = g;  // splitk > 1: the GEMM kernel writes raw partial tiles [ks][M][N] by itself
    if (sk == 1) {   // no split: plain store of the raw product into partial[0]
        k.C = g.partial; k.ldc = g.N; k.bias = nullptr; k.resid = nullptr; k.act = 0; k.accumulate = 0;
    }
    if (g.bf16) hipLaunchKernelGGL(gemm_nt_kernel<1>, grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL(gemm_nt_kernel<0>, grid, dim3(256), 0, s, k);
    hipLaunchKernelGGL(gemm_reduce_ln_kernel, dim3(g.M), dim3(256), 0, s, g, gamma, beta, eps, out);
    return hipGetLastError();
}

hipError_t gemm_launch(GemmArgs g, hipStream_t s) {
    if (g.batch < 1) g.batch = 1;
    if (g.splitk < 1 || !g.partial) g.splitk = 1;
    dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, g.batch * g.splitk);
    if (g.bf16)
        hipLaunchKernelGGL(gemm_nt_kernel<1>, grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<0>, grid, dim3(256), 0, s, g);
    if (g.splitk > 1) {
        const long long total = (long long)g.batch * g.M * g.N;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}


// =====================================================================================================
// FNet Fourier mixing in one launch per layer:  Z[b] = Re(FFT2(X[b])) + X[b]   (transformers modeling_fnet.py:138-166,
// `torch.fft.fftn(x, dim=(1, 2)).real`; models/diffusion.py:158-166 applies it 12 times per forward)
//   stage 1 (exact fp32 MFMA):  Ut[2h'+p][s] = sum_h D[2h'+p][h] X[b][s][h]        (p = 0: cos, 1: sin rows of the hidden DFT)
//   stage 2 (fp32 FMA chains):  Z[b][s'][h'] = sum_s C_S[s'][s] Ut[2h'][s] - S_S[s'][s] Ut[2h'+1][s]  + X[b][s'][h']
// A workgroup owns one sample and 32 output frequencies h' (64 rows of Ut): the sequence transform only needs those rows, so
// Ut never leaves LDS.  Replaces two GEMM launches + a split-K reduction (26.5 -> ~10 us per layer at B = 8, S = 32).
// The K loop of stage 1 is latency-bound, hence the long chunks (128 k, 4 per tile) and the K halves split over the waves.
// Supported: S <= 32 (T <= 1024 for the 6-level network), S % 8 == 0, hid % 128 == 0; otherwise the caller falls back to the
// two-GEMM path.
// =====================================================================================================
constexpr int MIX_BK = 128, MIX_ROWB = MIX_BK * 4 + 16;  // 528 B rows: 33 slots of 16 B (odd)
struct MixSmem {
    static constexpr int A_BYTES = 64 * MIX_ROWB, B_BYTES = 32 * MIX_ROWB;  // B: up to 32 token rows
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static size_t bytes(int S) { return (size_t)2 * STAGE + (size_t)S * 2 * S * 4; }
};
__global__ void __launch_bounds__(256) fnet_mix_kernel(const float* __restrict__ dft_hidden /*[2hid][hid]*/,
                                                       const float* __restrict__ dft_seq /*[S][2S]*/,
                                                       const float* __restrict__ X /*[B][S][hid]*/, float* __restrict__ Z, int S,
                                                       int hid) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* const sAB = sm;                                      // 2 stages x (A 64 rows | B 32 rows)
    float* const dsl = (float*)(sm + 2 * MixSmem::STAGE);      // dft_seq copy [S][2S]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, kh = wave >> 1;                   // row half of the 64 Ut rows, k half of every chunk
    const int m0 = blockIdx.x * 64, b = blockIdx.y;
    const float* A = dft_hidden + (size_t)m0 * hid;
    const float* Xb = X + (size_t)b * S * hid;
    for (int i = tid; i < S * 2 * S; i += 256) dsl[i] = dft_seq[i];

    constexpr int F4 = MIX_BK / 4;                              // float4 per row per chunk
    constexpr int NA = 64 * F4 / 256, NB = 32 * F4 / 256;
    // two chunks of loads are kept in flight in registers (the K loop is latency-bound): set 1 = chunk kc+1, set 2 = kc+2
    f32x4_t ra1[NA], rb1[NB], ra2[NA], rb2[NB];  // native vectors: HIP's float4 struct arrays were left in scratch memory
#define DDIMX_MIX_LOAD(RA, RB, K0)                                                                                     \
    do {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                               \
            const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;                                           \
            RA[i] = *(const f32x4_t*)(A + (size_t)row * hid + (K0) + kq);                                               \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                               \
            const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;                                           \
            RB[i] = row < S ? *(const f32x4_t*)(Xb + (size_t)row * hid + (K0) + kq) : (f32x4_t)(0.f);                  \
        }                                                                                                              \
    } while (0)
#define DDIMX_MIX_STORE(BUF, RA, RB)                                                                                   \
    do {                                                                                                               \
        char* pa_ = sAB + (BUF) * MixSmem::STAGE;                                                                      \
        char* pb_ = pa_ + MixSmem::A_BYTES;                                                                            \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                               \
            const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;                                           \
            *(f32x4_t*)(pa_ + row * MIX_ROWB + kq * 4) = RA[i];                                                         \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                               \
            const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;                                           \
            *(f32x4_t*)(pb_ + row * MIX_ROWB + kq * 4) = RB[i];                                                         \
        }                                                                                                              \
    } while (0)
    f32x16_t acc0;  // this wave's 32 Ut rows x 32 token columns, over its half of every k chunk
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
    const int nk = hid / MIX_BK;
    DDIMX_MIX_LOAD(ra1, rb1, 0);
    if (nk > 1) DDIMX_MIX_LOAD(ra2, rb2, MIX_BK);
    DDIMX_MIX_STORE(0, ra1, rb1);
#pragma unroll
    for (int i = 0; i < NA; ++i) ra1[i] = ra2[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) rb1[i] = rb2[i];
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 2 < nk) DDIMX_MIX_LOAD(ra2, rb2, (kc + 2) * MIX_BK);
        const char* pa = sAB + (kc & 1) * MixSmem::STAGE + (wm * 32 + l31) * MIX_ROWB + h * 16;
        const char* pb = sAB + (kc & 1) * MixSmem::STAGE + MixSmem::A_BYTES + l31 * MIX_ROWB + h * 16;
#pragma unroll
        for (int kg = 0; kg < MIX_BK / 16; ++kg) {             // this wave's half of the chunk: 8 groups of 8 floats
            const int off = (kh * (MIX_BK / 16) + kg) * 32;
            const uint4 a = *(const uint4*)(pa + off);
            const uint4 b0 = *(const uint4*)(pb + off);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b0.x), acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b0.y), acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b0.z), acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b0.w), acc0, 0, 0, 0);
        }
        if (kc + 1 < nk) {
            DDIMX_MIX_STORE((kc + 1) & 1, ra1, rb1);  // chunk kc+1 was requested an iteration ago
#pragma unroll
            for (int i = 0; i < NA; ++i) ra1[i] = ra2[i];
#pragma unroll
            for (int i = 0; i < NB; ++i) rb1[i] = rb2[i];
        }
        __syncthreads();
    }
    // Ut -> LDS, de-interleaved: utc[j][s] = Ut[2j][s] (cos rows), uts[j][s] = Ut[2j+1][s]; row stride 65 floats.
    // The k halves are summed in a fixed order (kh = 0 writes, barrier, kh = 1 adds).
    float* const utc = (float*)sAB;
    float* const uts = utc + 32 * 65;
    auto put = [&](bool add) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // Ut row inside the block (D layout of the MFMA)
            float* dst = ((m & 1) ? uts : utc) + (m >> 1) * 65;
            if (l31 < S) dst[l31] = add ? dst[l31] + acc0[r] : acc0[r];
        }
    };
    if (kh == 0) put(false);
    __syncthreads();
    if (kh == 1) put(true);
    __syncthreads();
    // stage 2: thread = (output frequency j = tid % 32, rows s' = tid / 32 + 8 i); dft_seq row = [cos | -sin]
    const int j = tid & 31, s0 = tid >> 5;
    const float* uc = utc + j * 65;
    const float* us = uts + j * 65;
    for (int sp = s0; sp < S; sp += 8) {
        const float* dr = dsl + (size_t)sp * 2 * S;
        float a = 0.f;
        for (int s2 = 0; s2 < S; ++s2) a = fmaf(dr[s2], uc[s2], a);
        for (int s2 = 0; s2 < S; ++s2) a = fmaf(dr[S + s2], us[s2], a);
        const size_t o = ((size_t)b * S + sp) * hid + blockIdx.x * 32 + j;
        Z[o] = a + X[o];
    }
}
#undef DDIMX_MIX_LOAD
#undef DDIMX_MIX_STORE
bool fnet_mix_supported(int S, int hid) { return S >= 8 && S <= 32 && S % 8 == 0 && hid % MIX_BK == 0 && hid % 32 == 0; }
hipError_t fnet_mix_launch(const float* dft_hidden, const float* dft_seq, const float* X, float* Z, int B, int S, int hid,
                           hipStream_t s) {
    if (!fnet_mix_supported(S, hid)) return hipErrorInvalidValue;
    static bool attr_done = false;
    const size_t lds = MixSmem::bytes(32);
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)fnet_mix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(fnet_mix_kernel, dim3(2 * hid / 64, B), dim3(256), MixSmem::bytes(S), s, dft_hidden, dft_seq, X, Z, S, hid);
    return hipGetLastError();
}

}  // namespace ddimx
