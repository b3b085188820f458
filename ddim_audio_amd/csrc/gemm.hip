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

    // Interior tiles of aligned operands with K a multiple of the chunk (every dense GEMM of the FNet): plain 16-byte loads, no
    // branch around them -- a load under `if (row < rows)` is waited for at once, which made the eight loads of a chunk a chain
    // of round trips and kept the prefetch below from overlapping the MFMAs.  (uniform per workgroup)
    const bool full = va && vb && m0 + GBM <= g.M && n0 + GBN <= g.N && g.K % BK == 0;
    float4 ra[LPT], rb[LPT];
    auto load = [&](int k0) __attribute__((always_inline)) {
        if (full) {
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;
                ra[i] = *(const float4*)(A + (size_t)(m0 + row) * g.lda + k0 + kq);
                rb[i] = *(const float4*)(B + (size_t)(n0 + row) * g.ldb + k0 + kq);
            }
        } else {
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;
                ra[i] = gemm_ld4(A, m0 + row, g.M, k0 + kq, g.K, g.lda, va);
                rb[i] = gemm_ld4(B, n0 + row, g.N, k0 + kq, g.K, g.ldb, vb);
            }
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
    auto mma = [&](int buf) __attribute__((always_inline)) {
        const char* pa = sA[buf] + (wm * 32 + l31) * ROWB + h * 16;
        const char* pb = sB[buf] + (wn * 32 + l31) * ROWB + h * 16;
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
    };

    constexpr int NPRE = 4;
    if (full && nk >= 1 && nk <= NPRE) {
        // Short K ranges (the split-K slices of the FNet GEMMs are 2-4 chunks): ALL chunks are requested at once -- the kernel is
        // otherwise one load round trip per chunk, with a microsecond of arithmetic in total.  Chunks past the range re-read the
        // last one (uniform clamp, no branch around a load) and are not multiplied.
        float4 qa[NPRE][LPT], qb[NPRE][LPT];
#pragma unroll
        for (int c = 0; c < NPRE; ++c) {
            const int k0 = (c0 + (c < nk ? c : nk - 1)) * BK;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const int pc = tid + i * 256, row = pc / F4, kq = (pc % F4) * 4;
                qa[c][i] = *(const float4*)(A + (size_t)(m0 + row) * g.lda + k0 + kq);
                qb[c][i] = *(const float4*)(B + (size_t)(n0 + row) * g.ldb + k0 + kq);
            }
        }
#pragma unroll
        for (int c = 0; c < NPRE; ++c) {
            if (c < nk) {  // uniform
#pragma unroll
                for (int i = 0; i < LPT; ++i) { ra[i] = qa[c][i]; rb[i] = qb[c][i]; }
                store(c & 1);
                __syncthreads();  // (the other buffer's MFMAs finished before the previous barrier)
                mma(c & 1);
            }
        }
    } else {
        if (nk > 0) {
            load(c0 * BK);
            store(0);
        }
        __syncthreads();
        for (int kc = 0; kc < nk; ++kc) {
            if (kc + 1 < nk) load((c0 + kc + 1) * BK);
            mma(kc & 1);
            if (kc + 1 < nk) store((kc + 1) & 1);
            __syncthreads();
        }
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

// (All loads of an element are issued before the first use: the K slices through clamped indices -- at most 8 slices, host
// cap kMaxSplitK --, absent operands through a valid dummy address, dropped by select.  A loop `for s < splitk: v += load` with a
// run-time trip count was one load round trip per slice, in a kernel that is nothing but that chain.)
__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(const GemmArgs g) {
    const long long total = (long long)g.batch * g.M * g.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = i % g.N;
        const long long mz = i / g.N;
        const int m = mz % g.M, z = mz / g.M;
        const size_t o = (size_t)z * g.sC + (size_t)m * g.ldc + n;
        float p[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int sc = s < g.splitk ? s : g.splitk - 1;
            p[s] = g.partial[((size_t)(z * g.splitk + sc) * g.M + m) * g.N + n];
        }
        const float cv = g.C[o];
        const float bv = (g.bias ? g.bias : g.partial)[n];
        const float rv = (g.resid ? g.resid : g.C)[o];
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) v += s < g.splitk ? p[s] : 0.f;
        for (int s = 8; s < g.splitk; ++s) v += g.partial[((size_t)(z * g.splitk + s) * g.M + m) * g.N + n];
        if (g.accumulate) v += cv;
        if (g.bias) v += bv;
        if (g.act == 1) v = gelu_new_f(v);
        if (g.resid) v += rv;
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
    // rows of the FNet are 512 wide: elements tid and tid + 256 are loaded -- every K slice, bias, residual, gamma, beta -- before
    // the first use, unconditionally (clamped indices, dropped by select); wider rows take the general loop for the rest
    float v[8];
    float s = 0.f;
    float gam[2], bet[2];
    {
        float p[2][8], bv[2], rv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n0 = tid + i * 256, n = n0 < g.N ? n0 : g.N - 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) p[i][k] = g.partial[((size_t)(k < g.splitk ? k : g.splitk - 1) * g.M + m) * g.N + n];
            bv[i] = (g.bias ? g.bias : gamma)[n];
            rv[i] = (g.resid ? g.resid + (size_t)m * g.ldc : gamma)[n];
            gam[i] = gamma[n];
            bet[i] = beta[n];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = tid + i * 256;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += k < g.splitk ? p[i][k] : 0.f;
            for (int k = 8; k < g.splitk; ++k) t += g.partial[((size_t)k * g.M + m) * g.N + (n < g.N ? n : g.N - 1)];
            if (g.bias) t += bv[i];
            if (g.resid) t += rv[i];
            v[i] = n < g.N ? t : 0.f;
            s += v[i];
        }
    }
#pragma unroll
    for (int i = 2; i < 8; ++i) {
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
        if (n < g.N) out[(size_t)m * g.N + n] = (v[i] - mean) * rstd * (i < 2 ? gam[i] : gamma[n]) + (i < 2 ? bet[i] : beta[n]);
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
// ---- Re(FFT2(X)) + X in one launch -------------------------------------------------------------------------------------
// Z[b] = C_S (X[b] C_H) - S_S (X[b] S_H) + X[b]   (S <= 32 tokens, hid % 64 == 0).
// Stage 1 (hidden DFT, exact fp32 MFMA): a workgroup owns 32 rows of the interleaved table D_H (16 output frequencies j:
// row 2j = cos_j, 2j+1 = sin_j) and one sample; its four waves split K = hid into quarters.  v_mfma_f32_32x32x2_f32 takes one
// k per lane half, and any pairing of k with (lane half, step) is valid as long as A and B use the same one, so every lane
// loads its operands STRAIGHT FROM GLOBAL MEMORY in MFMA layout (16-byte loads, all issued before the first MFMA): lane
// (r = lane % 32, h = lane / 32) holds D_H[m0 + r][k0 + 8g + 4h + i] and X[b][r][same k].  No LDS staging, no K loop, no barrier
// before the MFMAs: the kernel is one load round plus hid/8 dependent MFMAs per wave.
// Stage 2 (sequence DFT): the four K-quarter partials are summed in a fixed order through LDS, then
// Z[s'][j] = sum_s C_S[s'][s] Ut_cos[j][s] - S_S[s'][s] Ut_sin[j][s] + X[s'][j] as fp32 FMA chains.
constexpr int MIX_ROWS = 32;  // D_H rows per workgroup
struct MixSmem {
    static size_t bytes(int S) { return (size_t)(4 * MIX_ROWS * 33 + S * 2 * S) * 4; }
};
__global__ void __launch_bounds__(256) fnet_mix_kernel(const float* __restrict__ dft_hidden /*[2hid][hid]*/,
                                                       const float* __restrict__ dft_seq /*[S][2S]*/,
                                                       const float* __restrict__ X /*[B][S][hid]*/, float* __restrict__ Z, int S,
                                                       int hid, float* __restrict__ zc, float* __restrict__ zstats) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    float* const part = (float*)sm;                            // [4 K-quarters][32 rows][33]
    float* const dsl = part + 4 * MIX_ROWS * 33;               // dft_seq copy [S][2S]
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * MIX_ROWS, b = blockIdx.y;
    const int kq_len = hid / 4, k0 = kq * kq_len;
    const float* arow = dft_hidden + (size_t)(m0 + l31) * hid + k0 + 4 * h;
    const float* brow = X + ((size_t)b * S + (l31 < S ? l31 : 0)) * hid + k0 + 4 * h;
    const bool bvalid = l31 < S;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // everything the later stages read from memory is requested now, ahead of the MFMA block: the sequence-DFT table (staged to
    // LDS after the MFMAs) and the residual rows of this thread's outputs.  S <= 32: at most 8 table values and 2 outputs per thread.
    const int j = tid & 15, s0 = tid >> 4;  // stage 2: output frequency j, rows s0 + 16 i
    float dpre[8], xres[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = tid + u * 256;
        dpre[u] = dft_seq[i < S * 2 * S ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sp = s0 + 16 * u;
        xres[u] = X[((size_t)b * S + (sp < S ? sp : 0)) * hid + blockIdx.x * (MIX_ROWS / 2) + j];
    }
    constexpr int G = 16;  // 8-float groups per pass: 16 x (A + B) x 4 registers in flight
    for (int g0 = 0; g0 < kq_len / 8; g0 += G) {
        f32x4_t ra[G], rb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            ra[g] = *(const f32x4_t*)(arow + (g0 + g) * 8);
            rb[g] = *(const f32x4_t*)(brow + (g0 + g) * 8);  // rows past S read row 0 and are dropped: no branch around a load
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (!bvalid) rb[g] = (f32x4_t)(0.f);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][0], rb[g][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][1], rb[g][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][2], rb[g][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][3], rb[g][3], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = tid + u * 256;
        if (i < S * 2 * S) dsl[i] = dpre[u];
    }
    // this wave's partial Ut[32 rows][tokens] -> its LDS slab (D layout of the MFMA: row = (r & 3) + 8 (r >> 2) + 4 h, col = l31)
    float* const mine = part + kq * MIX_ROWS * 33;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = acc[r];
    __syncthreads();
    // fixed-order sum of the four K quarters, in place in slab 0
    for (int i = tid; i < MIX_ROWS * 32; i += 256) {
        const int o = (i >> 5) * 33 + (i & 31);
        part[o] = ((part[o] + part[MIX_ROWS * 33 + o]) + part[2 * MIX_ROWS * 33 + o]) + part[3 * MIX_ROWS * 33 + o];
    }
    __syncthreads();
    // stage 2: thread = (output frequency j = tid % 16, rows s' = tid / 16 + 16 u); dft_seq row = [cos | -sin]
    const float* uc = part + (2 * j) * 33;
    const float* us = part + (2 * j + 1) * 33;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sp = s0 + 16 * u;
        if (sp >= S) break;  // (whole 16-lane groups, and S % 8 == 0: whole waves)
        const float* dr = dsl + (size_t)sp * 2 * S;
        float a = 0.f;
        for (int s2 = 0; s2 < S; ++s2) a = fmaf(dr[s2], uc[s2], a);
        for (int s2 = 0; s2 < S; ++s2) a = fmaf(dr[S + s2], us[s2], a);
        const size_t o = ((size_t)b * S + sp) * hid + blockIdx.x * (MIX_ROWS / 2) + j;
        const float z = a + xres[u];
        if (Z) Z[o] = z;
        if (zc) {  // (uniform) the chunk-major copy and the LayerNorm statistics of row sp over this workgroup's 16 features
            const int k = blockIdx.x * (MIX_ROWS / 2) + j;
            zc[(size_t)b * 32 * hid + ((size_t)(k / 4) * 32 + sp) * 4 + k % 4] = z;
            const float sm = group_sum<16>(z);
            const float d = z - sm * (1.0f / 16.0f);
            const float m2 = group_sum<16>(d * d);
            const int p = blockIdx.x;  // part = column block; [B][parts / 2][32 rows][2 x (sum, m2)]
            if (j == 0) *(float2*)(zstats + (size_t)b * gridDim.x * 64 + ((size_t)(p / 2) * 32 + sp) * 4 + (p % 2) * 2) = make_float2(sm, m2);
        }
    }
}
bool fnet_mix_supported(int S, int hid) { return S >= 8 && S <= 32 && S % 8 == 0 && hid % 512 == 0; }
hipError_t fnet_mix_launch(const float* dft_hidden, const float* dft_seq, const float* X, float* Z, int B, int S, int hid,
                           hipStream_t s, float* zc, float* zstats) {
    if (!fnet_mix_supported(S, hid) || (zc == nullptr) != (zstats == nullptr) || (!Z && !zc)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fnet_mix_kernel, dim3(2 * hid / MIX_ROWS, B), dim3(256), MixSmem::bytes(S), s, dft_hidden, dft_seq, X, Z, S, hid,
                       zc, zstats);
    return hipGetLastError();
}

}  // namespace ddimx
