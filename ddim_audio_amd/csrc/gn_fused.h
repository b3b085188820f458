// GroupNorm statistics without a launch of their own (inference walk).
//
// Producers (conv_mfma_kernel, resid_kernel, conv_in, tensor_stats) can write their per-workgroup partials already folded
// to the 8 groups:   gstats[B][np][kGnSlab]  fp32, the first 16 floats = [kGroups][2] (sum, sumsq), the rest zero padding:
// one whole 128-byte line per workgroup.  (64-byte slabs made two workgroups -- usually on different XCDs -- write halves of one
// line: the partial-line writes held every workgroup's retirement back by about a microsecond.)
// Consumers (conv_mfma_kernel's halo prologue, resid_kernel) then finish the normalisation themselves: every workgroup sums
// its sample's np partials in fp64 (np <= kGnFuseMaxParts, a few KiB of L2 hits issued before the weight / halo loads) and folds
// mean / rstd with gamma / beta into the per-channel (scale, shift) it keeps in registers.  Same arithmetic as
// gn_finalize_kernel (fp64 sums, biased variance, rstd = 1 / sqrt(var + eps)); the summation order is fixed by (np, block
// size) alone, so a sample's result does not depend on the batch it is in.  Nothing here crosses workgroups inside one
// launch: partials are read by the NEXT kernel on the stream, so no fences or atomics are needed.
#pragma once
#include "common.h"

namespace ddimx {

constexpr int kGnSlab = 32;           // floats per partial slab (one 128-byte line)
constexpr int kGnFuseMaxParts = 256;  // above this a separate gn_finalize_groups launch is cheaper than every workgroup re-reading

struct GnIn {
    const float* stats;  // [B][np][kGnSlab] group partials of the tensor being normalised; null: not used
    const float* gamma;  // [C]
    const float* beta;   // [C] or null
    double inv_count;    // 1 / elements per (sample, group)
    float eps;
    int np;
};

// Phase 1, all threads of the block (nthreads % 64 == 0): thread -> (group = tid % 8, slice = tid / 8) sums every
// (nthreads / 8)-th partial.  Split in two so that the caller can put its own loads between the issue and the first use:
// gn_in_issue starts the first eight loads, gn_in_reduce adds them (and any further rounds), combines the lanes of equal group by
// butterflies and leaves the wave's (sum, sumsq) per group in scr[wave][8][2] (LDS floats).  The caller's barrier follows.
// (Loads are unconditional with a clamped index and masked when added: a load under `p < np ? load : 0` is issued inside an
// exec-masked branch and waited for at once, which turned eight loads into a chain of round trips.)
struct GnInLoads { float2 v[8]; };
__device__ __forceinline__ void gn_in_issue(const GnIn& gi, int b, int tid, int nthreads, GnInLoads& L) {
    const int g = tid & 7, slice = tid >> 3, nslice = nthreads >> 3;
    const float2* base = (const float2*)gi.stats + (size_t)b * gi.np * (kGnSlab / 2) + g;
    const int last = gi.np - 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = slice + u * nslice;
        L.v[u] = base[(size_t)(p < last ? p : last) * (kGnSlab / 2)];
    }
}
__device__ __forceinline__ void gn_in_reduce(const GnIn& gi, int b, int tid, int nthreads, const GnInLoads& L, float* scr) {
    const int g = tid & 7, slice = tid >> 3, nslice = nthreads >> 3;
    const float2* base = (const float2*)gi.stats + (size_t)b * gi.np * (kGnSlab / 2) + g;
    const int last = gi.np - 1;
    // fp32 up to the per-wave totals (<= 8 x 8 partials per thread and wave: the partials themselves are fp32 sums of thousands of
    // elements, a few more fp32 additions lose nothing), fp64 for the cross-wave sum and the variance (gn_in_group)
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const bool in = slice + u * nslice <= last;
        s += in ? L.v[u].x : 0.f;
        q += in ? L.v[u].y : 0.f;
    }
    for (int p0 = slice + nslice * 8; p0 <= last; p0 += nslice * 8) {  // more than 8 partials per thread: further rounds
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * nslice;
            v[u] = base[(size_t)(p < last ? p : last) * (kGnSlab / 2)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = p0 + u * nslice <= last;
            s += in ? v[u].x : 0.f;
            q += in ? v[u].y : 0.f;
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if ((tid & 63) < 8) *(float2*)(scr + ((tid >> 6) * kGroups + g) * 2) = make_float2(s, q);
}
// Phase 2 (after the barrier): mean and rstd of group g.  The waves' totals are added and the variance formed in fp64
// (cancellation in E[x^2] - mean^2); the reciprocal square root in fp32 with one Newton step (a double-precision divide and
// square root cost a microsecond in every workgroup's prologue and buy nothing at fp32 output precision).
__device__ __forceinline__ void gn_in_group(const GnIn& gi, const float* scr, int nwaves, int g, float* mean, float* rstd) {
    double S = 0.0, Q = 0.0;
    for (int w = 0; w < nwaves; ++w) {
        const float2 t = *(const float2*)(scr + (w * kGroups + g) * 2);
        S += (double)t.x; Q += (double)t.y;
    }
    const double m = S * gi.inv_count;
    double var = Q * gi.inv_count - m * m;
    if (var < 0.0) var = 0.0;
    const float ve = (float)(var + (double)gi.eps);
    float r = __builtin_amdgcn_rsqf(ve);
    r = r * (1.5f - 0.5f * ve * r * r);
    *mean = (float)m;
    *rstd = r;
}
// gamma / beta of N consecutive channels (N = 4 or 8, c0 % N == 0: 16-byte loads), loaded early -- before the barrier -- so
// that their latency hides behind the reduction.  beta == null (uniform): zeros.
template <int N>
__device__ __forceinline__ void gn_in_params(const GnIn& gi, int c0, float* gam, float* bet) {
#pragma unroll
    for (int j = 0; j < N; j += 4) {
        const float4 t = *(const float4*)(gi.gamma + c0 + j);
        gam[j] = t.x; gam[j + 1] = t.y; gam[j + 2] = t.z; gam[j + 3] = t.w;
    }
    if (gi.beta) {
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            const float4 t = *(const float4*)(gi.beta + c0 + j);
            bet[j] = t.x; bet[j + 1] = t.y; bet[j + 2] = t.z; bet[j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) bet[j] = 0.f;
    }
}
// Folded (scale, shift) of N consecutive channels c0 .. c0+N-1 (N <= 8 and c0 % N == 0: they span at most two groups).
template <int N>
__device__ __forceinline__ void gn_in_fold(const GnIn& gi, const float* scr, int nwaves, int C, int c0, const float* gam,
                                           const float* bet, float* sc, float* sh) {
    const int GS = C / kGroups;
    const int g_lo = c0 / GS, g_hi = (c0 + N - 1) / GS;
    float m_lo, r_lo, m_hi, r_hi;
    gn_in_group(gi, scr, nwaves, g_lo, &m_lo, &r_lo);
    m_hi = m_lo; r_hi = r_lo;
    if (g_hi != g_lo) gn_in_group(gi, scr, nwaves, g_hi, &m_hi, &r_hi);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const bool hi = (c0 + j) / GS != g_lo;
        const float m = hi ? m_hi : m_lo, r = hi ? r_hi : r_lo;
        const float s = r * gam[j];
        sc[j] = s;
        sh[j] = fmaf(-m, s, bet[j]);
    }
}

// Producer side: fold this workgroup's per-row channel sums rows[r][k][2] (LDS; r < NROWS with row stride `rstride` floats;
// virtual channel ch0 + k, real channel (ch0 + k) % C) into the 8 group bins and store them as one slab dst[kGnSlab].  Wave 0
// only (lane = threadIdx % 64), after the barrier that completed `rows`.  Lane -> (bin = lane / 4, quarter = lane % 4):
// fixed-order sums, two butterflies; lanes 0-31 then write the whole line (16 bins + 16 zeros).  All LDS reads are
// unconditional (clamped index, masked value) so that they pipeline instead of forming a chain of LDS round trips.
template <int NROWS>
__device__ __forceinline__ void gn_bins_store(const float* rows, int rstride, int n_ch, int ch0, int C, float* dst, int lane) {
    const int GS = C / kGroups;
    const int bin = lane >> 2, qt = lane & 3, g = bin >> 1, sq = bin & 1;
    const int nii = (GS + 3) >> 2;
    float t = 0.f;
    // virtual channels of group g: rep * C + g*GS + i; this workgroup holds [ch0, ch0 + n_ch)
    for (int rep0 = 0; rep0 * C < ch0 + n_ch; ++rep0) {
#pragma unroll 2
        for (int ii = 0; ii < nii; ++ii) {
            const int i = qt + 4 * ii;
            const int k = rep0 * C + g * GS + i - ch0;
            const bool in = i < GS && k >= 0 && k < n_ch;
            const int kc = in ? k : 0;
            float u = 0.f;
#pragma unroll
            for (int r = 0; r < NROWS; ++r) u += rows[r * rstride + kc * 2 + sq];
            t += in ? u : 0.f;
        }
    }
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    const float v = __shfl(t, (lane & 15) * 4, 64);
    if (lane < kGnSlab) dst[lane] = lane < 2 * kGroups ? v : 0.f;
}

}  // namespace ddimx
