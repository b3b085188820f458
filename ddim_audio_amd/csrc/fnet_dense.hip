// Dense layers of the FNet bottleneck at sequence lengths S <= 32 (T <= 1024 for the six-level network), inference walk.
//
//   out[b][t][n] = act( sum_k W[n][k] * xf(X[b][t][k]) + bias[n] ) (+ LN(R)[b][t][n])        (modeling_fnet.py:226-279,
//                                                                                               models/diffusion.py:131-167)
//
// A FNet layer was a chain of six latency-bound launches (mix, LayerNorm, GEMM, split-K reduce, GEMM, reduce + LayerNorm): with
// 256 token rows at a batch of 8 there is a microsecond of arithmetic per launch and nothing to hide its round trips behind.
// This kernel removes the launches that only re-shape data between the GEMMs:
//   * no split-K workspace and no reduce launch: a workgroup owns 32*WF output features of ONE sample (<= 32 tokens); its
//     waves split K among themselves and their partial tiles meet in LDS, summed in a fixed order;
//   * no LayerNorm launch in front of the layer: the row statistics arrive as per-part (sum, centred sum of squares) pairs
//     written by the PRODUCER of the rows (fnet_mix_kernel, or this kernel's `ostats`), every consumer lane folds its row's
//     parts (Chan's combination: no cancellation) and normalises the token operand while it converts it; gamma is folded
//     into the weight matrix and beta into the bias when the weights are packed (W' = W diag(gamma), b' = b + W beta);
//   * the residual LN(R) of the output LayerNorm's input is recomputed from R and the same statistics in the epilogue.
//
// Operands go STRAIGHT FROM GLOBAL MEMORY INTO MFMA LAYOUT -- A = 32 weight rows, B = the sample's tokens, lane (r = lane % 32,
// h = lane / 32) holds 8 (bf16) or 4 (fp32) consecutive k of row r per step -- and every matrix is STORED in the order its
// reader wants.  The first version read row-major matrices: a wave load then touches 32 rows = 32 cache lines for 32 useful
// bytes each, the texture addresser looks up one line per clock, and the kernel was nothing but that queue (in-kernel stamps,
// tools/dbg/fnet_dense_bench.hip: 8 400 of 21 800 cycles until the loads were merely ISSUED; 10 240 line look-ups per
// workgroup).  So:
//   * weights are packed in FRAGMENT order (fnet_fold_kernel): [32-row block][k step][lane][16 bytes] -- one wave load is one
//     contiguous KiB (8 lines);
//   * token matrices between the kernels of a layer are CHUNK-MAJOR: [sample][k / 4][32 rows][4 fp32] (or [k / 8][32][8 bf16]):
//     the 32 lanes of a half wave read 512 contiguous bytes; producers (fnet_mix_kernel, this kernel's epilogue) write that
//     order directly; rows >= S of a block are never written and are dropped by select when read;
//   * row statistics likewise: [sample][part / 2][32 rows][(sum, m2) x 2].
// PREC 1: v_mfma_f32_32x32x16_bf16 on weights pre-rounded to bf16 at pack time and tokens rounded here (the same values the
// staged GEMM multiplied); PREC 0: v_mfma_f32_32x32x2_f32, exact fp32 (parity mode and the reference's mixed mode).
// Every choice depends on the sample (S, K, N) only: a sample's result is bit-identical alone or in any batch.
#include "kernels.h"

#ifdef DDIMX_FD_STAMP  // tools/dbg/fnet_dense_bench.hip only: phase stamps of wave 0 of every workgroup
__device__ unsigned long long* fd_stamps = nullptr;
#define FD_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (fd_stamps && threadIdx.x == 0) fd_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FD_STAMP(k)
#endif

namespace ddimx {

namespace {

constexpr int kPitch = 36;  // floats per token row of a partial tile in LDS (144 B = 9 sixteen-byte slots: conflict-free b128)

// mean / rstd of one row from parts of `n_part` elements each; the calling lane holds the parts q (< cnt) it loaded as
// (sum, m2) pairs; G lanes (common.h group_sum) share the row.  All lanes of a group end with the same bits.
template <int G, int NQ>
__device__ __forceinline__ void fold_row_stats(const float (&ps)[NQ], const float (&pm)[NQ], int cnt, float n_part, float n_row,
                                               float eps, float* mean_out, float* rstd_out) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) s += q < cnt ? ps[q] : 0.f;
    s = group_sum<G>(s);
    const float mean = s / n_row;
    float m2 = 0.f;
    const float inv_n = 1.0f / n_part;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float d = ps[q] * inv_n - mean;
        m2 += q < cnt ? fmaf(n_part * d, d, pm[q]) : 0.f;
    }
    m2 = group_sum<G>(m2);
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf(m2 / n_row + eps);
}

// gelu_new (transformers activations.py:59-66): 0.5 v (1 + tanh(u)) = v * sigmoid(2u), u = sqrt(2/pi) (v + 0.044715 v^3);
// branch-free (tanhf is a libm call with range branches), v_exp_f32 / v_rcp_f32 are ~1 ulp
__device__ __forceinline__ float gelu_new_fast(float v) {
    const float u = 0.7978845608028654f * fmaf(0.044715f * v * v, v, v);
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u));
}

__device__ __forceinline__ bf16x8_t to_bf16x8(const uint4& lo, const uint4& hi, float a, float c, bool valid) {
    float f[8];
    Piece<float>::unpack(lo, f);
    Piece<float>::unpack(hi, f + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = valid ? fmaf(f[i], a, c) : 0.f;  // (select: rows >= S of a chunk block are never written)
    return __builtin_bit_cast(bf16x8_t, Piece<__bf16>::pack(f));
}

// element offset of (row r, column k) inside one sample's chunk-major block of CH-element chunks
template <int CH> __device__ __forceinline__ size_t chunk_off(int r, int k) { return ((size_t)(k / CH) * 32 + r) * CH + k % CH; }

}  // namespace

// grid (N / (32 WF), B), block 64 * WF * KS.
//   XL: token operand layout, 0 row-major fp32 [B*S][K] (the projection's input), 1 chunk-major (fp32 chunks of 4; TXB: bf16 of 8)
//   XF 1: the token operand is (x - mean_row) * rstd_row (statistics from xstats)
//   GP = MFMA steps (PREC 1: 16 k each; PREC 0: 8 k each) whose loads are in flight together; (K / KS) / kstep % GP == 0
//   RES: + LayerNorm(R) residual
template <int PREC, bool TXB, bool TOB, int WF, int KS, int XF, int XL, int GP, bool RES>
__global__ void __launch_bounds__(64 * WF * KS) fnet_dense_kernel(const FnetDenseArgs a) {
    static_assert(PREC == 1 || (!TXB && !TOB), "the exact path multiplies and stores fp32");
    static_assert(XF == 0 || !TXB, "normalised rows are fp32");
    static_assert(XL == 1 || !TXB, "row-major tokens are fp32");
    constexpr int LPT = 8 * WF;  // lanes that share a token row in the epilogue
    __shared__ __attribute__((aligned(16))) float part[KS * WF * 32 * kPitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int fb = wave % WF, ks = wave / WF;
    const int b = blockIdx.y, S = a.S, K = a.K, N = a.N;
    const int nb = blockIdx.x * WF + fb;  // 32-row block of W
    const int kw = K / KS, kb = ks * kw;
    const bool tvalid = l31 < S;
    FD_STAMP(0);

    // ---- everything the epilogue reads from memory is requested first (thread = one token, four consecutive features)
    const int et = tid / LPT, ef4 = tid % LPT;  // (meaningful for tid < 32 * LPT)
    const bool ethread = tid < 32 * LPT;
    const int etc = ethread ? et : 0;  // (any row of the block is a valid address)
    const int en = blockIdx.x * WF * 32 + ef4 * 4;
    const float4 ebias = *(const float4*)(a.bias + en);
    float4 er = make_float4(0.f, 0.f, 0.f, 0.f), eg = er, ebt = er;
    float rps[4], rpm[4];
    int rcnt = 0;
    if constexpr (RES) {  // R and its statistics are chunk-major (no run-time branch around loads: hipcc drains them at the merge)
        er = *(const float4*)(a.R + (size_t)b * 32 * N + chunk_off<4>(etc, en));
        eg = *(const float4*)(a.rgamma + en);
        ebt = *(const float4*)(a.rbeta + en);
        rcnt = a.rnp / LPT;  // parts per lane, 1 .. 4 (host-checked)
        const float* sp = a.rstats + (size_t)b * a.rnp * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = ef4 * rcnt + (q < rcnt ? q : 0);
            const float2 v = *(const float2*)(sp + ((size_t)(p / 2) * 32 + etc) * 4 + (p % 2) * 2);
            rps[q] = v.x; rpm[q] = v.y;
        }
    }
    // ---- row statistics of the token operand: lane (r, h) folds half of its row's parts, the halves meet by one swap
    float xa = 1.f, xc = 0.f;
    float xps[16], xpm[16];
    if constexpr (XF == 1) {
        const int nq = a.xnp / 4;  // float4 (two parts) per lane, <= 8 (host-checked)
        const float* sp = a.xstats + (size_t)b * a.xnp * 64 + ((size_t)h * nq * 32 + l31) * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = *(const float4*)(sp + (size_t)(q < nq ? q : 0) * 128);
            xps[2 * q] = v.x; xpm[2 * q] = v.y; xps[2 * q + 1] = v.z; xpm[2 * q + 1] = v.w;
        }
    }

    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if constexpr (PREC == 1) {
        const int nsteps = kw / 16, g_first = kb / 16;
        // fragment-order weights: [nb][K/16][64 lanes][8 bf16]
        const uint4* wp = (const uint4*)a.W + ((size_t)nb * (K / 16) + g_first) * 64 + lane;
        // tokens: chunk-major bf16 [K/8][32][8]: step g, lane (r, h) -> chunk 2g + h; chunk-major fp32 [K/4][32][4]: chunks
        // 4g + 2h + j; row-major fp32: 8 floats at k = 16 g + 8 h
        const char* xp;
        size_t xstep;  // bytes per step
        if constexpr (XL == 0) {
            xp = (const char*)((const float*)a.X + ((size_t)b * S + (tvalid ? l31 : 0)) * K + kb + 8 * h);
            xstep = 64;
        } else if constexpr (TXB) {
            xp = (const char*)((const __bf16*)a.X + (size_t)b * 32 * K + ((size_t)(2 * g_first + h) * 32 + l31) * 8);
            xstep = 2 * 32 * 16;
        } else {
            xp = (const char*)((const float*)a.X + (size_t)b * 32 * K + ((size_t)(4 * g_first + 2 * h) * 32 + l31) * 4);
            xstep = 4 * 32 * 16;
        }
        constexpr size_t x2 = XL == 0 ? 16 : 32 * 16;  // second 16 bytes of an fp32 operand
        for (int g0 = 0; g0 < nsteps; g0 += GP) {
            uint4 wq[GP], xq[GP][TXB ? 1 : 2];
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                wq[g] = wp[(size_t)(g0 + g) * 64];
                xq[g][0] = *(const uint4*)(xp + (size_t)(g0 + g) * xstep);
                if constexpr (!TXB) xq[g][1] = *(const uint4*)(xp + (size_t)(g0 + g) * xstep + x2);
            }
            __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the loads between the MFMAs, three or four in flight)
            FD_STAMP(1);
            if constexpr (XF == 1) {
                if (g0 == 0) {  // (the statistics were requested before the operands: they are here first)
                    float mean, rstd;
                    fold_row_stats<2, 16>(xps, xpm, a.xnp / 2, (float)a.xn, (float)(a.xnp * a.xn), a.eps, &mean, &rstd);
                    xa = rstd;
                    xc = -mean * rstd;
                }
            }
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                bf16x8_t bv;
                if constexpr (TXB) {
                    uint4 t = xq[g][0];
                    if (!tvalid) t = make_uint4(0, 0, 0, 0);
                    bv = __builtin_bit_cast(bf16x8_t, t);
                } else {
                    bv = to_bf16x8(xq[g][0], xq[g][1], xa, xc, tvalid);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wq[g]), bv, acc, 0, 0, 0);
            }
        }
    } else {
        const int ngroups = kw / 8, g_first = kb / 8;
        // fragment-order weights: [nb][K/8][64 lanes][4 fp32]; tokens chunk-major: group g', lane (r, h) -> chunk 2g' + h
        const f32x4_t* wp = (const f32x4_t*)a.W + ((size_t)nb * (K / 8) + g_first) * 64 + lane;
        const char* xp;
        size_t xstep;
        if constexpr (XL == 0) {
            xp = (const char*)((const float*)a.X + ((size_t)b * S + (tvalid ? l31 : 0)) * K + kb + 4 * h);
            xstep = 32;
        } else {
            xp = (const char*)((const float*)a.X + (size_t)b * 32 * K + ((size_t)(2 * g_first + h) * 32 + l31) * 4);
            xstep = 2 * 32 * 16;
        }
        for (int g0 = 0; g0 < ngroups; g0 += GP) {
            f32x4_t ra[GP], rb[GP];
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                ra[g] = wp[(size_t)(g0 + g) * 64];
                rb[g] = *(const f32x4_t*)(xp + (size_t)(g0 + g) * xstep);
            }
            __builtin_amdgcn_sched_barrier(0);
            FD_STAMP(1);
            if constexpr (XF == 1) {
                if (g0 == 0) {
                    float mean, rstd;
                    fold_row_stats<2, 16>(xps, xpm, a.xnp / 2, (float)a.xn, (float)(a.xnp * a.xn), a.eps, &mean, &rstd);
                    xa = rstd;
                    xc = -mean * rstd;
                }
            }
#pragma unroll
            for (int g = 0; g < GP; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[g][i] = tvalid ? fmaf(rb[g][i], xa, xc) : 0.f;
            }
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][0], rb[g][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][1], rb[g][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][2], rb[g][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][3], rb[g][3], acc, 0, 0, 0);
            }
        }
    }

    FD_STAMP(2);
    // ---- this wave's partial tile D[feature][token] -> LDS as [token][feature]: register quad q = features 8q + 4h .. + 3
    {
        float* mine = part + ((ks * WF + fb) * 32 + l31) * kPitch + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4*)(mine + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();
    FD_STAMP(3);
    if (!ethread) return;  // (whole waves: 32 * LPT is a multiple of 64; no barrier follows)

    // ---- epilogue: fixed-order sum of the K slices, bias, activation, residual LayerNorm, store, row statistics
    const float* src = part + ((ef4 / 8) * 32 + et) * kPitch + (ef4 % 8) * 4;
    float4 ssum = *(const float4*)src;
#pragma unroll
    for (int k2 = 1; k2 < KS; ++k2) {
        const float4 p = *(const float4*)(src + k2 * WF * 32 * kPitch);
        ssum.x += p.x; ssum.y += p.y; ssum.z += p.z; ssum.w += p.w;
    }
    float v[4] = {ssum.x + ebias.x, ssum.y + ebias.y, ssum.z + ebias.z, ssum.w + ebias.w};
    if (a.act == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_new_fast(v[i]);
    }
    if constexpr (RES) {
        float mean, rstd;
        fold_row_stats<LPT, 4>(rps, rpm, rcnt, (float)a.rn, (float)(a.rnp * a.rn), a.eps, &mean, &rstd);
        const float rr[4] = {er.x, er.y, er.z, er.w}, gg[4] = {eg.x, eg.y, eg.z, eg.w}, bb[4] = {ebt.x, ebt.y, ebt.z, ebt.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += fmaf((rr[i] - mean) * rstd, gg[i], bb[i]);
    }
    const bool evalid = et < S;
    if (evalid) {
        if constexpr (TOB) {  // (chunk-major only: chunks of 8 bf16)
            __bf16* o = (__bf16*)a.out + (size_t)b * 32 * N + chunk_off<8>(et, en);
            *(uint2*)o = make_uint2(Piece<__bf16>::pk(v[0], v[1]), Piece<__bf16>::pk(v[2], v[3]));
        } else {
            float* o = a.out_chunk ? (float*)a.out + (size_t)b * 32 * N + chunk_off<4>(et, en)
                                   : (float*)a.out + ((size_t)b * S + et) * N + en;
            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (a.ostats) {
        const float s = group_sum<LPT>((v[0] + v[1]) + (v[2] + v[3]));
        const float m = s / (float)(32 * WF);
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[i] - m; m2 = fmaf(d, d, m2); }
        m2 = group_sum<LPT>(m2);
        if (ef4 == 0 && evalid) {
            const int p = blockIdx.x;
            *(float2*)(a.ostats + (size_t)b * gridDim.x * 64 + ((size_t)(p / 2) * 32 + et) * 4 + (p % 2) * 2) = make_float2(s, m2);
        }
    }
    FD_STAMP(4);
}

bool fnet_dense_supported(int S, int K, int N) { return S >= 1 && S <= 32 && K % 512 == 0 && N % 64 == 0 && K >= 512; }

#ifndef DDIMX_FD_STAMP
// Shapes of the path: K = hid (512) with N = inter / width ("wide": 32 features x 4 K-quarters per workgroup), and
// K = inter / width (2048) with N = hid ("deep": 32 features x 8 K-eighths) -- tools/dbg/fnet_dense_bench.hip.
hipError_t fnet_dense_launch(const FnetDenseArgs& a, int B, int bf16, hipStream_t s) {
    if (!fnet_dense_supported(a.S, a.K, a.N)) return hipErrorInvalidValue;
    if (a.xstats && (a.xnp > 32 || a.xnp % 4 || a.xn < 1 || !a.x_chunk)) return hipErrorInvalidValue;
    const bool deep = a.K >= 4 * a.N || a.K > 1024;
    const int WF = 1, KS = deep ? 8 : 4;
    if (a.R && (a.rnp % (8 * WF) || a.rnp / (8 * WF) > 4 || a.rnp / (8 * WF) < 1)) return hipErrorInvalidValue;
    if (a.ostats && (a.N / (32 * WF)) % 2) return hipErrorInvalidValue;
    if ((a.x_bf16 && !a.x_chunk) || (a.out_bf16 && !a.out_chunk)) return hipErrorInvalidValue;
    const int kw = a.K / KS;
    dim3 grid(a.N / (32 * WF), B), block(64 * WF * KS);
#define DDIMX_FD(PREC, TXB, TOB, WF_, KS_, XF, XL, GP, RES)                                                           \
    do {                                                                                                              \
        if ((kw / (PREC ? 16 : 8)) % GP) return hipErrorInvalidValue;                                                 \
        hipLaunchKernelGGL((fnet_dense_kernel<PREC, TXB, TOB, WF_, KS_, XF, XL, GP, RES>), grid, block, 0, s, a);     \
        return hipGetLastError();                                                                                     \
    } while (0)
    const bool xf = a.xstats != nullptr, res = a.R != nullptr;
    if (bf16) {
        if (!deep) {  // normalised chunk-major fp32 tokens, K = 512: 8 steps per wave; no residual
            if (!xf || a.x_bf16 || res) return hipErrorInvalidValue;
            if (a.out_bf16) DDIMX_FD(1, false, true, 1, 4, 1, 1, 8, false);
            else DDIMX_FD(1, false, false, 1, 4, 1, 1, 8, false);
        } else {
            if (a.out_bf16 || xf) return hipErrorInvalidValue;
            if (a.x_bf16) { if (res) DDIMX_FD(1, true, false, 1, 8, 0, 1, 16, true); else DDIMX_FD(1, true, false, 1, 8, 0, 1, 16, false); }
            else if (!a.x_chunk && !res) DDIMX_FD(1, false, false, 1, 8, 0, 0, 8, false);
            else if (!res) DDIMX_FD(1, false, false, 1, 8, 0, 1, 8, false);  // chunk-major fp32 tokens (the projection)
            else return hipErrorInvalidValue;
        }
    } else {
        if (a.out_bf16 || a.x_bf16) return hipErrorInvalidValue;
        if (!deep) { if (!xf || res) return hipErrorInvalidValue; DDIMX_FD(0, false, false, 1, 4, 1, 1, 16, false); }
        else {
            if (xf) return hipErrorInvalidValue;
            if (a.x_chunk) { if (res) DDIMX_FD(0, false, false, 1, 8, 0, 1, 16, true); else DDIMX_FD(0, false, false, 1, 8, 0, 1, 16, false); }
            else if (!res) DDIMX_FD(0, false, false, 1, 8, 0, 0, 16, false);
            else return hipErrorInvalidValue;
        }
    }
#undef DDIMX_FD
    return hipErrorInvalidValue;
}
#endif

// ---- Fourier mixing of the path:  Z = Re(FFT2(X)) + X with X = LayerNorm(V) taken on the fly ------------------------------------
// fnet_mix_kernel (gemm.hip) restated for the layouts above, absorbing the previous layer's output LayerNorm (one launch less per
// layer).  With X = N diag(gamma) + 1 beta^T (N = the normalised rows of V):
//   X D_H^T = N (D_H diag(gamma))^T + 1 (D_H beta)^T        -> the hidden DFT runs on N against a PER-LAYER table with gamma folded
//                                                              in (fragment order, built at pack time by fnet_table_kernel);
//   the constant rows (D_H beta) survive the sequence DFT only at s' = 0 (sum_s cos(2 pi s s'/S) = S delta(s'), the sine sums
//   vanish): S * bc[j] is added to row 0, bc = C_H beta (pack time);
//   the residual X[s'][j] = N[s'][j] gamma[j] + beta[j] is formed per output element from V and the row statistics.
// Stage 1: A = 32 table rows in fragment order (one contiguous KiB per wave load), B = V chunk-major, normalised in registers.
// Stage 2 as fnet_mix_kernel; Z leaves chunk-major with its row statistics.  NORM = false: X = V as is (layer 0: the projection).
struct FnetMixSmem {
    static size_t bytes(int S) { return (size_t)(4 * 32 * 36 + S * 2 * S) * 4; }
};
template <bool NORM>
__global__ void __launch_bounds__(256) fnet_mix2_kernel(const FnetMixArgs a) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    constexpr int MP = 36;                 // floats per row of a partial slab (144 B: the sequence DFT reads it 16 bytes at a time)
    float* const part = (float*)sm;        // [4 K-quarters][32 rows][MP]
    float* const dsl = part + 4 * 32 * MP;  // dft_seq copy [S][2S]
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int S = a.S, hid = a.hid, b = blockIdx.y;
    const bool tvalid = l31 < S;
    const int ngroups = hid / 4 / 8, g_first = kq * ngroups;  // 8-k groups of this wave's K quarter
    const float* vb = a.V + (size_t)b * 32 * hid;
    // ---- stage-2 operands of this thread: output frequency j, rows s0 + 16 u
    const int j = tid & 15, s0 = tid >> 4, k = blockIdx.x * 16 + j;
    float dpre[8], xres[2], rs[2], rm[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = tid + u * 256;
        dpre[u] = a.dft_seq[i < S * 2 * S ? i : 0];
    }
    float gk = 1.f, bk = 0.f, bck = 0.f;
    if constexpr (NORM) { gk = a.gamma[k]; bk = a.beta[k]; bck = a.bc[k]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sp = s0 + 16 * u;  // (< 32: a valid address in the block whatever S)
        xres[u] = vb[((size_t)(k / 4) * 32 + sp) * 4 + k % 4];
        if constexpr (NORM) {  // part j of row sp (16 parts of hid / 16 elements)
            const float2 v = *(const float2*)(a.vstats + (size_t)b * 16 * 64 + ((size_t)(j / 2) * 32 + sp) * 4 + (j % 2) * 2);
            rs[u] = v.x; rm[u] = v.y;
        }
    }
    // ---- statistics of the MFMA operand's row (lane (r, h): 8 of the 16 parts)
    float xps[8], xpm[8];
    if constexpr (NORM) {
        const float* sp = a.vstats + (size_t)b * 16 * 64 + ((size_t)h * 4 * 32 + l31) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *(const float4*)(sp + (size_t)q * 128);
            xps[2 * q] = v.x; xpm[2 * q] = v.y; xps[2 * q + 1] = v.z; xpm[2 * q + 1] = v.w;
        }
    }
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const f32x4_t* wp = (const f32x4_t*)a.tab + ((size_t)blockIdx.x * (hid / 8) + g_first) * 64 + lane;
    const char* xp = (const char*)(vb + ((size_t)(2 * g_first + h) * 32 + l31) * 4);
    constexpr int GP = 16;
    float xa = 1.f, xc = 0.f;
    for (int g0 = 0; g0 < ngroups; g0 += GP) {
        f32x4_t ra[GP], rb[GP];
#pragma unroll
        for (int g = 0; g < GP; ++g) {
            ra[g] = wp[(size_t)(g0 + g) * 64];
            rb[g] = *(const f32x4_t*)(xp + (size_t)(g0 + g) * 2 * 32 * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NORM) {
            if (g0 == 0) {
                float mean, rstd;
                fold_row_stats<2, 8>(xps, xpm, 8, (float)(hid / 16), (float)hid, a.eps, &mean, &rstd);
                xa = rstd;
                xc = -mean * rstd;
            }
        }
#pragma unroll
        for (int g = 0; g < GP; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[g][i] = tvalid ? fmaf(rb[g][i], xa, xc) : 0.f;
        }
#pragma unroll
        for (int g = 0; g < GP; ++g) {
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
    // this wave's partial Ut[32 table rows][tokens] -> its LDS slab (D layout: row = (r & 3) + 8 (r >> 2) + 4 h, col = l31)
    float* const mine = part + kq * 32 * MP;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * h) * MP + l31] = acc[r];
    __syncthreads();
    {   // fixed-order sum of the four K quarters, in place in slab 0: thread = (row, four tokens)
        const int o = (tid >> 3) * MP + (tid & 7) * 4;
        const float4 p0 = *(const float4*)(part + o), p1 = *(const float4*)(part + 32 * MP + o);
        const float4 p2 = *(const float4*)(part + 2 * 32 * MP + o), p3 = *(const float4*)(part + 3 * 32 * MP + o);
        *(float4*)(part + o) = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                           ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
    }
    __syncthreads();
    const float* uc = part + (2 * j) * MP;
    const float* us = part + (2 * j + 1) * MP;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sp = s0 + 16 * u;
        if (sp >= S) break;  // (whole 16-lane groups, and S % 8 == 0: whole waves)
        const float* dr = dsl + (size_t)sp * 2 * S;
        float t = 0.f;
        for (int s4 = 0; s4 < S; s4 += 4) {  // (same order of addition as the scalar loop; S % 8 == 0)
            const float4 d = *(const float4*)(dr + s4), v = *(const float4*)(uc + s4);
            t = fmaf(d.x, v.x, t); t = fmaf(d.y, v.y, t); t = fmaf(d.z, v.z, t); t = fmaf(d.w, v.w, t);
        }
        for (int s4 = 0; s4 < S; s4 += 4) {
            const float4 d = *(const float4*)(dr + S + s4), v = *(const float4*)(us + s4);
            t = fmaf(d.x, v.x, t); t = fmaf(d.y, v.y, t); t = fmaf(d.z, v.z, t); t = fmaf(d.w, v.w, t);
        }
        float x = xres[u];
        if constexpr (NORM) {
            const float n_part = (float)(hid / 16);
            const float mean = group_sum<16>(rs[u]) / (float)hid;
            const float d = rs[u] / n_part - mean;
            const float m2 = group_sum<16>(fmaf(n_part * d, d, rm[u]));
            const float rstd = 1.0f / sqrtf(m2 / (float)hid + a.eps);
            x = fmaf((x - mean) * rstd, gk, bk);
            if (sp == 0) t = fmaf((float)S, bck, t);
        }
        const float z = t + x;
        a.zc[(size_t)b * 32 * hid + ((size_t)(k / 4) * 32 + sp) * 4 + k % 4] = z;
        const float sm2 = group_sum<16>(z);
        const float dz = z - sm2 * (1.0f / 16.0f);
        const float m2z = group_sum<16>(dz * dz);
        const int p = blockIdx.x;
        if (j == 0) *(float2*)(a.zstats + (size_t)b * gridDim.x * 64 + ((size_t)(p / 2) * 32 + sp) * 4 + (p % 2) * 2) = make_float2(sm2, m2z);
    }
}
#ifndef DDIMX_FD_STAMP
hipError_t fnet_mix2_launch(const FnetMixArgs& a, int B, hipStream_t s) {
    if (a.hid != 512 || a.S < 8 || a.S > 32 || a.S % 8) return hipErrorInvalidValue;  // (16 statistics parts, one pass of 16 groups)
    const dim3 grid(a.hid / 16, B), block(256);
    if (a.vstats)
        hipLaunchKernelGGL(fnet_mix2_kernel<true>, grid, block, FnetMixSmem::bytes(a.S), s, a);
    else
        hipLaunchKernelGGL(fnet_mix2_kernel<false>, grid, block, FnetMixSmem::bytes(a.S), s, a);
    return hipGetLastError();
}
#endif

// Per-layer hidden-DFT table in fragment order: row 2j = cos(2 pi j h / H) gamma[h], row 2j + 1 = sin(2 pi j h / H) gamma[h]
// (gamma null: 1), as [row / 32][h / 8][lane = 32 hh + row % 32][4], h = 8 g + 4 hh + i (argument reduced exactly mod H, fp64
// trigonometry rounded once, like model.py::_dft_tables); bc[j] = sum_h cos(2 pi j h / H) beta[h] (beta null: not written),
// fixed-order fp64 tree.  One workgroup per table row.
__global__ void __launch_bounds__(256) fnet_table_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ tab, float* __restrict__ bc, int H) {
    __shared__ double red[4];
    const int row = blockIdx.x, jf = row >> 1, p = row & 1, tid = threadIdx.x;
    double s = 0.0;
    for (int hcol = tid; hcol < H; hcol += 256) {
        const int m = (int)(((long long)jf * hcol) % H);
        const double ang = 2.0 * 3.14159265358979323846 * (double)m / (double)H;
        const double cv = cos(ang), tv = p ? sin(ang) : cv;
        const int g = hcol / 8, hh = (hcol % 8) / 4, i = hcol % 4;
        tab[(((size_t)(row / 32) * (H / 8) + g) * 64 + hh * 32 + row % 32) * 4 + i] = (float)tv * (gamma ? gamma[hcol] : 1.0f);
        if (beta && p == 0) s += (double)(float)cv * (double)beta[hcol];
    }
    if (!beta || p) return;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bc[jf] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
hipError_t fnet_table_launch(const float* gamma, const float* beta, float* tab, float* bc, int H, hipStream_t s) {
    if (H % 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fnet_table_kernel, dim3(2 * H), dim3(256), 0, s, gamma, beta, tab, bc, H);
    return hipGetLastError();
}

// ---- weight packing for the path ------------------------------------------------------------------------------------------
// Wf = W diag(gamma) (gamma null: W) in FRAGMENT order -- bf16: [n / 32][k / 16][lane = 32 h + n % 32][8], k = 16 g + 8 h + i;
// fp32: [n / 32][k / 8][lane][4], k = 8 g + 4 h + i --; bf[n] = bias[n] + sum_k W[n][k] beta[k] (beta null: not written).
// One workgroup per row n; the beta sum is a fixed-order tree.
template <typename TW>
__global__ void __launch_bounds__(256) fnet_fold_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ bias,
                                                        TW* __restrict__ Wf, float* __restrict__ bf, int K) {
    constexpr int E = 16 / sizeof(TW);  // elements per lane and step
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = W[(size_t)n * K + k];
        const int g = k / (2 * E), hh = (k % (2 * E)) / E, i = k % E;
        Wf[(((size_t)(n / 32) * (K / (2 * E)) + g) * 64 + hh * 32 + n % 32) * E + i] = from_f<TW>(gamma ? w * gamma[k] : w);
        if (beta) s = fmaf(w, beta[k], s);
    }
    if (!beta) return;
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bf[n] = bias[n] + ((red[0] + red[1]) + (red[2] + red[3]));
}
hipError_t fnet_fold_launch(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, int wf_bf16,
                            float* bf, int N, int K, hipStream_t s) {
    if (N % 32 || K % 16) return hipErrorInvalidValue;
    if (wf_bf16)
        hipLaunchKernelGGL(fnet_fold_kernel<__bf16>, dim3(N), dim3(256), 0, s, W, gamma, beta, bias, (__bf16*)Wf, bf, K);
    else
        hipLaunchKernelGGL(fnet_fold_kernel<float>, dim3(N), dim3(256), 0, s, W, gamma, beta, bias, (float*)Wf, bf, K);
    return hipGetLastError();
}

}  // namespace ddimx
