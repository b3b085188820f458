// 3x3 convolution whose input GroupNorm is FOLDED INTO THE WEIGHTS (inference, bf16, resident weights: gfx950).
//
// The second conv of Residual_Block normalises its input with a plain per-channel affine, no activation in between
// (models/diffusion.py:49-51: conv1(GN1(h)), h = SiLU(conv0(..) + temb) already activated by the producer):
//     y = s_c * h + t_c ,   out = conv_W(y) + bias            (s, t: the folded GroupNorm of THIS sample)
// Convolution is linear, so the affine moves into the operands that are small and per workgroup instead of being applied to
// every input element (and every halo element again) in registers:
//     out[p][co] = sum_taps sum_ci (W[co][tap][ci] * s_ci) * h[p + tap][ci]  +  sum_{taps inside the image at p} U[tap][co]  + bias[co]
//     W' = bf16(W * s)   scaled once per workgroup in LDS (a workgroup only ever works on one sample),
//     U[tap][co] = sum_ci W'[co][tap][ci] * (t_ci / s_ci)   (fp32, from the ROUNDED W', so that W' h + U = W' (h + t/s) holds exactly:
//                  the rounding of W' then acts on the normalised value, not on the un-centred one),
// with the zero padding of the NORMALISED tensor (conv2d pads after GroupNorm) reproduced by leaving out the taps that fall
// outside the image: nine per-pixel border classes (row class x column class), one addend vector each.
// What that buys on the HBM-bound level (C = 32: 144 FLOP/B): the input halo needs no arithmetic at all, so it goes global -> LDS
// by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPRs, out-of-range pixels arrive as zeros through the buffer bounds check,
// measured tools/dbg/lds_dma_oob.hip) into a double buffer, one whole tile ahead of the MFMAs; the kernel keeps the vector
// ALU for the epilogue only (the register-transform kernel was VALU-issue bound: DESIGN section 4).
//
// Tile loop of a workgroup (persistent over `tiles_per_wg` consecutive tiles of one sample), three barriers per tile:
//   top:  s_waitcnt vmcnt(<stores of the previous tile>) -> this wave's share of halo(t) has landed; barrier
//         issue the LDS-DMA of halo(t+1) into the other buffer (it held the previous tile's output tile: free since the barrier)
//   MFMA over the 9 taps from halo[cur] and the resident W'                      ; barrier A (all reads of halo[cur] returned)
//   epilogue 1: accumulators + addend(class) -> SiLU -> bf16 -> output tile, overlaid on halo[cur]     ; barrier B
//   epilogue 2: whole NHWC pixel rows leave with 16-byte stores; GroupNorm statistics of the values as stored.
#pragma once
#include "conv_mfma.h"

namespace ddimx {

struct FoldArgs {
    const void* in;         // [B][H][W][C] bf16
    const void* w;          // [9][C][C] bf16 ([tap][cout][cin], ddimx_pack_conv)
    const float* bias;      // [C] or null
    const float* chan_add;  // per-sample per-cout vector or null
    int chan_add_stride;
    const float* in_scale;  // [B][C] folded GroupNorm (gn.stats == null)
    const float* in_shift;
    GnIn gn;                // gn.stats != null: the input's group partials, finished in the prologue (gn_fused.h)
    void* out;              // [B][H][W][C] bf16
    float* stats;           // as ConvArgs::stats
    int stats_groups_c;
    int act;                // 0 none, 1 SiLU
    int B, H, W;
    int tiles_x, tiles_y, tiles_per_wg, wgs_per_sample;
    unsigned long long* stamps;  // diagnostic builds only (-DDDIMX_STAMP), layout as conv_mfma_kernel's
    int stagger;            // units of 256 cycles: see the tile loop
};

template <int C_, int TH_, int TW_, int WM_, int WN_>
struct FoldCfg {
    static constexpr int C = C_, TH = TH_, TW = TW_, WM = WM_, WN = WN_;
    static constexpr int ES = 2, EPB = 8;
    static constexpr int NWAVES = WM * WN, NTHREADS = 64 * NWAVES;
    static constexpr int P = TH * TW;
    static constexpr int MT = P / (32 * WM), NT = C / (32 * WN);
    static constexpr int IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    static constexpr int CPP = C / EPB;            // 16-byte pieces per pixel
    static constexpr int PSLOT = CPP + 1;          // + one pad piece: pixel stride odd in 16-byte slots (conflict-free b128 reads)
    static constexpr int PSTRIDE = PSLOT * 16;
    static constexpr int HALO_PIECES = NPIX * PSLOT;
    static constexpr int HALO_DMA = (HALO_PIECES + 63) / 64;   // LDS-DMA instructions (1 KiB each) per halo
    static constexpr int HALO_BYTES = HALO_DMA * 1024;
    static constexpr int HDPW = (HALO_DMA + NWAVES - 1) / NWAVES;  // per wave; every wave issues the same number (counted waits)
    static constexpr int WROW = C * ES + 16, WROWP = CPP + 1;
    static constexpr int W_PIECES = 9 * C * WROWP;
    static constexpr int W_DMA = (W_PIECES + 63) / 64;
    static constexpr int WBUF_BYTES = W_DMA * 1024;
    static constexpr int WDPW = (W_DMA + NWAVES - 1) / NWAVES;
    static constexpr int OSTRIDE = C * ES + 16;
    static constexpr int OUT_BYTES = P * OSTRIDE;
    static constexpr int KG = C * ES / 32;
    static constexpr int OPP = C / EPB, OLPP = next_pow2(OPP);
    static constexpr int STEP = NTHREADS / OLPP, NPASS = (P + STEP - 1) / STEP;
    static constexpr int SMALL_FLOATS = 9 * C + 3 * C + NWAVES * kGroups * 2;   // addv9, addv, sv, rv, gnscr
    static constexpr int LDS_BYTES = WBUF_BYTES + 2 * HALO_BYTES + 1024 + SMALL_FLOATS * 4;
    static constexpr int RED_BYTES = NWAVES * C * 2 * 4;
    static_assert(P % (32 * WM) == 0 && MT >= 1 && C % (32 * WN) == 0 && NT >= 1, "tile must split into 32x32 MFMA blocks");
    static_assert((PSLOT % 2) == 1, "pixel stride must be odd in 16-byte slots");
    static_assert(OUT_BYTES <= HALO_BYTES, "the output tile overlays one halo buffer");
    static_assert(9 * C * 4 <= HALO_BYTES, "U scratch overlays the second halo buffer");
    static_assert(RED_BYTES <= WBUF_BYTES, "statistics scratch overlays the weights");
    static_assert(P % STEP == 0, "whole passes in epilogue 2");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(TW == 32 || TW == 16 || TW == 8, "TW");
    static_assert(OPP == OLPP, "every thread stores in epilogue 2 (the counted vmcnt at the top of a tile relies on it)");
    static_assert(WN == 1 && (MT * 32 * OLPP) % 64 == 0 && NPASS == MT * 32 * OLPP / 64, "wave-local epilogue 2");
};

// LDS-DMA through a buffer resource: lane i's 16 bytes at (voff + soff) land at lds_addr + 16 * i; offsets outside the resource
// write zeros.  Inline asm: invisible to hipcc's waitcnt pass (count it yourself), m0 saved and restored in the statement.
__device__ __forceinline__ void lds_dma16_buf(const u32x4_t& rs, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rs), "s"(lds_addr), "s"(soff)
        : "memory");
}

template <class F>
__global__ void __launch_bounds__(F::NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) conv3_fold_kernel(const FoldArgs a) {
    typedef __bf16 T;
    constexpr int C = F::C, ES = 2, EPB = 8, NP = 4, TW = F::TW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const wbuf = smem;
    char* const halo0 = wbuf + F::WBUF_BYTES;
    char* const dummy = halo0 + 2 * F::HALO_BYTES;
    float* const addv9 = (float*)(dummy + 1024);   // [9][C]: addend per border class (class 4 = interior)
    float* const addv = addv9 + 9 * C;             // [C] bias + per-sample channel vector
    float* const sv = addv + C;                    // [C] folded GroupNorm scale s
    float* const rv = sv + C;                      // [C] t / s
    float* const gnscr = rv + C;
    float* const ubuf = (float*)(halo0 + F::HALO_BYTES);  // [9][C] (prologue only; the second halo buffer is still unused)

    DDIMX_STAMP_ENTRY
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % F::WM, wn = wave / F::WM;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware workgroup order (as conv_mfma_kernel): neighbouring tile ranges share an L2
    int lwg;
    {
        const int nwg = gridDim.x, x8 = blockIdx.x & 7, i8 = blockIdx.x >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        lwg = (x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q) + i8;
    }
    const int wg = lwg % a.wgs_per_sample;
    const int bs = lwg / a.wgs_per_sample;
    const int ntile_s = a.tiles_x * a.tiles_y;
    const int t_begin = wg * a.tiles_per_wg;
    const int t_end = (t_begin + a.tiles_per_wg < ntile_s) ? t_begin + a.tiles_per_wg : ntile_s;

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned wbuf_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(wbuf));
    const unsigned halo_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(halo0));
    const unsigned dummy_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(dummy));

    // ---- input halo by LDS-DMA ---------------------------------------------------------------------------------------
    // resource over THIS sample's [H][W][C] tensor: offsets outside it (rows above / below the image) read zeros by themselves;
    // pixels left / right of the image wrap into the neighbouring row and are masked per lane (border tiles only)
    u32x4_t rs;
    {
        const uint64_t p = (uint64_t)((const T*)a.in + (size_t)bs * a.H * a.W * C);
        rs[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
        rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        rs[2] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)a.H * a.W * C * ES));
        rs[3] = 0x00020000u;
    }
    unsigned hrel[F::HDPW];  // tile-invariant per-lane offsets relative to the halo origin (interior tiles add one scalar)
#pragma unroll
    for (int j = 0; j < F::HDPW; ++j) {
        const int idx = (wave + j * F::NWAVES) * 64 + lane;
        const int pix = idx / F::PSLOT, pc = idx % F::PSLOT;
        const bool real = pix < F::NPIX && pc < F::CPP;
        hrel[j] = real ? (unsigned)((((pix / F::IW) * a.W + pix % F::IW) * C + pc * EPB) * ES) : 0x80000000u;
    }
    // All HDPW LDS-DMAs of a wave's share in ONE statement: m0 is saved once, steps by NWAVES KiB from slot to slot, and is
    // restored once (5 scalar instructions + a nop per DMA otherwise: the tile loop is issue-bound).  The wave's last slot may lie
    // past the halo (HALO_DMA is not a multiple of NWAVES): that one goes to `last_lds` (the dummy KiB).
    auto dma_share = [&](const unsigned (&voff)[F::HDPW], unsigned soff, unsigned first_lds, unsigned last_lds) __attribute__((always_inline)) {
        static_assert(F::HDPW == 7, "the asm block below is written out for 7 DMAs per wave");
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %10\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %1, %8, %9 offen lds\n\t"
            "s_add_u32 m0, m0, %12\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %2, %8, %9 offen lds\n\t"
            "s_add_u32 m0, m0, %12\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %3, %8, %9 offen lds\n\t"
            "s_add_u32 m0, m0, %12\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %4, %8, %9 offen lds\n\t"
            "s_add_u32 m0, m0, %12\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %5, %8, %9 offen lds\n\t"
            "s_add_u32 m0, m0, %12\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %6, %8, %9 offen lds\n\t"
            "s_mov_b32 m0, %11\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %7, %8, %9 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "s"(rs), "s"(soff),
              "s"(first_lds), "s"(last_lds), "i"(F::NWAVES * 1024)
            : "memory", "scc");
    };
    auto halo_dma = [&](int ty, int tx, int buf) __attribute__((always_inline)) {
        const int hy0 = ty * F::TH - 1, hx0 = tx * TW - 1;
        const bool interior = hy0 >= 0 && hx0 >= 0 && hy0 + F::IH <= a.H && hx0 + F::IW <= a.W;  // uniform
        const unsigned dst0 = halo_lds + (unsigned)buf * F::HALO_BYTES;
        const unsigned first = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)wave_u * 1024u);
        constexpr int QL = (F::HDPW - 1) * F::NWAVES;  // slot of a wave's last DMA, minus the wave index
        const unsigned last = __builtin_amdgcn_readfirstlane(wave_u + QL < F::HALO_DMA ? dst0 + (unsigned)(wave_u + QL) * 1024u : dummy_lds);
        if (interior) {
            dma_share(hrel, __builtin_amdgcn_readfirstlane((unsigned)((hy0 * a.W + hx0) * C * ES)), first, last);
        } else {
            unsigned vo[F::HDPW];
#pragma unroll
            for (int j = 0; j < F::HDPW; ++j) {
                const int idx = (wave + j * F::NWAVES) * 64 + lane;
                const int pix = idx / F::PSLOT, pc = idx % F::PSLOT;
                const int gy = hy0 + pix / F::IW, gx = hx0 + pix % F::IW;
                const bool ok = pix < F::NPIX && pc < F::CPP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                vo[j] = ok ? (unsigned)(((gy * a.W + gx) * C + pc * EPB) * ES) : 0x80000000u;
            }
            dma_share(vo, 0u, first, last);
        }
    };

    // ---- prologue: every load issued before the first wait ---------------------------------------------------------------
    const bool gn_fused = a.gn.stats != nullptr;  // uniform
    float add_b = 0.f, add_c = 0.f;
    {
        const float* pb = a.bias ? a.bias : (const float*)a.w;
        const float* pc = a.chan_add ? a.chan_add + (size_t)bs * a.chan_add_stride : (const float*)a.w;
        const int ic = tid < C ? tid : C - 1;
        add_b = pb[ic];
        add_c = pc[ic];
    }
    GnInLoads gn_ld;
    if (gn_fused) gn_in_issue(a.gn, bs, tid, F::NTHREADS, gn_ld);
    f32x2_t sc[NP], sh[NP];
    const bool fold_thread = tid < F::CPP;  // these threads fold 8 channels each
    {
        const int c0 = (fold_thread ? tid : 0) * EPB;
        const float* psc = gn_fused ? a.gn.gamma + c0 : a.in_scale + (size_t)bs * C + c0;
        const float* psh = gn_fused ? (a.gn.beta ? a.gn.beta : a.gn.gamma) + c0 : a.in_shift + (size_t)bs * C + c0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            sc[j] = *(const f32x2_t*)(psc + 2 * j);
            sh[j] = *(const f32x2_t*)(psh + 2 * j);
        }
    }
    // weights: one resident chunk, all taps (global_load_lds; lanes on row padding re-read the base)
#pragma unroll
    for (int j = 0; j < F::WDPW; ++j) {
        const int q = wave_u + j * F::NWAVES;
        const int idx = (wave + j * F::NWAVES) * 64 + lane;
        const int row = idx / F::WROWP, pc = idx % F::WROWP;
        const bool real = idx < F::W_PIECES && pc < F::CPP;
        const char* src = (const char*)a.w + (real ? (size_t)(row * C + pc * EPB) * ES : 0);
        lds_dma16(src, __builtin_amdgcn_readfirstlane(q < F::W_DMA ? wbuf_lds + q * 1024 : dummy_lds));
    }
    int ty = t_begin / a.tiles_x, tx = t_begin % a.tiles_x;  // the tile walk is consecutive: no division per tile
    halo_dma(ty, tx, 0);
    if (tid < C) addv[tid] = (a.bias ? add_b : 0.f) + (a.chan_add ? add_c : 0.f);
    if (gn_fused) {
        gn_in_reduce(a.gn, bs, tid, F::NTHREADS, gn_ld, gnscr);
        __syncthreads();
    }
    if (fold_thread) {
        float fs[EPB], fh[EPB];
        if (gn_fused) {
            float gam[EPB], bet[EPB];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                gam[2 * j] = sc[j].x; gam[2 * j + 1] = sc[j].y;
                bet[2 * j] = a.gn.beta ? sh[j].x : 0.f; bet[2 * j + 1] = a.gn.beta ? sh[j].y : 0.f;
            }
            gn_in_fold<EPB>(a.gn, gnscr, F::NWAVES, C, tid * EPB, gam, bet, fs, fh);
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) { fs[2 * j] = sc[j].x; fs[2 * j + 1] = sc[j].y; fh[2 * j] = sh[j].x; fh[2 * j + 1] = sh[j].y; }
        }
#pragma unroll
        for (int j = 0; j < EPB; ++j) {
            // a channel whose scale is (next to) zero keeps a tiny one: W' = W * 1e-12 and t / 1e-12 stay far inside the bf16 /
            // fp32 exponent range, W' (h + t/s) = 1e-12 W h + W t (1 +- 2^-9): the constant channel the reference computes
            float s = fs[j];
            if (fabsf(s) < 1e-12f) s = s < 0.f ? -1e-12f : 1e-12f;  // (a NaN scale stays a NaN)
            sv[tid * EPB + j] = s;
            rv[tid * EPB + j] = fh[j] / s;
        }
    }
    // the weights have landed (this wave's WDPW DMAs are older than its HDPW halo DMAs, which stay in flight across the weight
    // passes below; LDS-DMA is invisible to the compiler: raw barriers, or its fence would drain them)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(F::HDPW) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // W' = bf16(W * s), in place
#pragma unroll 1
    for (int p = tid; p < 9 * C * F::CPP; p += F::NTHREADS) {
        const int row = p / F::CPP, pc = p % F::CPP;
        uint4* wp = (uint4*)(wbuf + row * F::WROW + pc * 16);
        f32x2_t f[NP];
        Pairs<T>::unpack(*wp, f);
        const float4 s0 = *(const float4*)(sv + pc * EPB), s1 = *(const float4*)(sv + pc * EPB + 4);
        f[0] *= (f32x2_t){s0.x, s0.y}; f[1] *= (f32x2_t){s0.z, s0.w};
        f[2] *= (f32x2_t){s1.x, s1.y}; f[3] *= (f32x2_t){s1.z, s1.w};
        *wp = Pairs<T>::pack(f);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // U[tap][co] = sum_ci W'[tap][co][ci] * (t/s)[ci]
#pragma unroll 1
    for (int p = tid; p < 9 * C; p += F::NTHREADS) {
        float u = 0.f;
#pragma unroll
        for (int pc = 0; pc < F::CPP; ++pc) {
            f32x2_t f[NP];
            Pairs<T>::unpack(*(const uint4*)(wbuf + p * F::WROW + pc * 16), f);
            const float4 r0 = *(const float4*)(rv + pc * EPB), r1 = *(const float4*)(rv + pc * EPB + 4);
            u = fmaf(f[0].x, r0.x, u); u = fmaf(f[0].y, r0.y, u); u = fmaf(f[1].x, r0.z, u); u = fmaf(f[1].y, r0.w, u);
            u = fmaf(f[2].x, r1.x, u); u = fmaf(f[2].y, r1.y, u); u = fmaf(f[3].x, r1.z, u); u = fmaf(f[3].y, r1.w, u);
        }
        ubuf[p] = u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // addend per border class: class = 3 * rowclass + colclass; rowclass 0 = first image row (taps dy = 0 fall outside),
    // 2 = last image row (dy = 2 outside), 1 = neither; columns alike
#pragma unroll 1
    for (int p = tid; p < 9 * C; p += F::NTHREADS) {
        const int cls = p / C, co = p % C, rc = cls / 3, cc = cls % 3;
        float u = addv[co];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const bool in = !(rc == 0 && dy == 0) && !(rc == 2 && dy == 2) && !(cc == 0 && dx == 0) && !(cc == 2 && dx == 2);
                const float v = ubuf[(dy * 3 + dx) * C + co];
                u += in ? v : 0.f;
            }
        addv9[p] = u;
    }
    // (visible after the first tile's barriers; ubuf is free again before the first DMA into the second halo buffer: the top
    // barrier of the first tile follows)

    // ---- per-lane operand offsets ------------------------------------------------------------------------------------------
    int pixoff[F::MT];
#pragma unroll
    for (int m = 0; m < F::MT; ++m) {
        const int p = (wm * F::MT + m) * 32 + l31;
        pixoff[m] = ((p / TW) * F::IW + p % TW) * F::PSTRIDE + h * 16;
    }
    const int woff = (wn * F::NT * 32 + l31) * F::WROW + h * 16;
    const int oc = lane % F::OLPP, oslot_w = lane / F::OLPP;  // epilogue 2: lane -> (16-byte piece, pixel) inside the wave's own pixels
    const bool ovalid = true;
    f32x2_t st_s[NP], st_q[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
    const unsigned out_bytes = (unsigned)((size_t)a.H * a.W * C * ES);
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc((T*)a.out + (size_t)bs * a.H * a.W * C, out_bytes);

    // The CU's two workgroups start together and -- with no memory wait left in the tile loop -- would stay in lock-step for their
    // whole life: both in the MFMA phase (one matrix pipe per SIMD, shared), then both in the vector epilogue (one VALU, shared),
    // every phase twice as long and nothing overlapped.  The workgroup that was placed second on its CU (LDS base != 0) therefore
    // starts its tile loop half a tile later; from then on one's MFMAs run beside the other's epilogue.
    if (a.stagger > 0) {  // uniform
        unsigned la;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
        if (la & 0xffu) {
#pragma unroll 1
            for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(4);
        }
    }
    int cur = 0;
    DDIMX_STAMP_DECL
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int y0 = ty * F::TH, x0 = tx * TW;
        char* const halo = halo0 + cur * F::HALO_BYTES;
        DDIMX_STAMP_AT(9);
        // top: all but this wave's NPASS youngest vector-memory operations (the previous tile's stores) are done, i.e. its share
        // of halo(t) has landed; lgkmcnt(0): its LDS reads of the previous output tile have returned
        if (t == t_begin) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the first halo has no stores behind it)
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(F::NPASS) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DDIMX_STAMP_AT(8);
        int nty = ty, ntx = tx + 1;
        if (ntx == a.tiles_x) { ntx = 0; ++nty; }
        if (t + 1 < t_end) halo_dma(nty, ntx, cur ^ 1);
        DDIMX_STAMP_AT(0);

        f32x16_t acc[F::NT][F::MT];
#pragma unroll
        for (int n = 0; n < F::NT; ++n)
#pragma unroll
            for (int m = 0; m < F::MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
        // operand fragments of tap k+1 are in flight while the MFMAs of tap k run (two register sets; the scheduling barriers keep
        // hipcc from sinking the loads back down to their use, which serialises every MFMA behind an LDS round trip)
        uint4 bq[2][F::KG][F::MT], aq[2][F::KG][F::NT];
        auto frag_load = [&](int tap, int st) __attribute__((always_inline)) {
            const int hoff = ((tap / 3) * F::IW + tap % 3) * F::PSTRIDE;
#pragma unroll
            for (int kg = 0; kg < F::KG; ++kg) {
#pragma unroll
                for (int m = 0; m < F::MT; ++m) bq[st][kg][m] = *(const uint4*)(halo + pixoff[m] + hoff + kg * 32);
#pragma unroll
                for (int n = 0; n < F::NT; ++n) aq[st][kg][n] = *(const uint4*)(wbuf + woff + (tap * C + n * 32) * F::WROW + kg * 32);
            }
        };
        frag_load(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) frag_load(tap + 1, (tap + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kg = 0; kg < F::KG; ++kg)
#pragma unroll
                for (int n = 0; n < F::NT; ++n)
#pragma unroll
                    for (int m = 0; m < F::MT; ++m) Mma<T>::run(aq[tap & 1][kg][n], bq[tap & 1][kg][m], acc[n][m]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // barrier A: every wave's reads of halo[cur] have RETURNED (the waits in front of the last MFMAs can sink below a raw
        // barrier: wait here explicitly) -- the output tile overlays it next
        DDIMX_STAMP_AT(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DDIMX_STAMP_AT(2);

        // ---- epilogue 1: accumulators + addend(class) -> SiLU -> bf16 -> output tile [pixel][cout] ----------------------------
        char* const otile = halo;
        const bool border = y0 == 0 || x0 == 0 || y0 + F::TH >= a.H || x0 + TW >= a.W;  // uniform
        auto epi1 = [&](auto act_tag, auto border_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr bool BORDER = decltype(border_tag)::value;
            // every addend of the tile first: an LDS read behind an LDS write of the same phase cannot be moved up by the compiler
            // (it cannot prove the two apart) and each would cost a full LDS round trip in front of its quad
            float4 av[BORDER ? F::MT : 1][F::NT][4];
#pragma unroll
            for (int m = 0; m < (BORDER ? F::MT : 1); ++m) {
                int cls = 4;
                if (BORDER) {
                    const int p = (wm * F::MT + m) * 32 + l31;
                    const int vy = y0 + p / TW, vx = x0 + p % TW;
                    cls = (vy == 0 ? 0 : (vy == a.H - 1 ? 2 : 1)) * 3 + (vx == 0 ? 0 : (vx == a.W - 1 ? 2 : 1));
                }
#pragma unroll
                for (int n = 0; n < F::NT; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        av[m][n][q] = *(const float4*)(addv9 + cls * C + (wn * F::NT + n) * 32 + q * 8 + h * 4);
            }
#pragma unroll
            for (int m = 0; m < F::MT; ++m) {
                const int p = (wm * F::MT + m) * 32 + l31;
#pragma unroll
                for (int n = 0; n < F::NT; ++n) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = (wn * F::NT + n) * 32 + q * 8 + h * 4;
                        const float4 ad = av[BORDER ? m : 0][n][q];
                        f32x2_t v01 = {acc[n][m][q * 4 + 0], acc[n][m][q * 4 + 1]};
                        f32x2_t v23 = {acc[n][m][q * 4 + 2], acc[n][m][q * 4 + 3]};
                        v01 += (f32x2_t){ad.x, ad.y};
                        v23 += (f32x2_t){ad.z, ad.w};
                        if (ACT) { v01 = silu2(v01); v23 = silu2(v23); }
                        *(uint2*)(otile + p * F::OSTRIDE + cl * ES) =
                            make_uint2(Piece<__bf16>::pk(v01.x, v01.y), Piece<__bf16>::pk(v23.x, v23.y));
                    }
                }
            }
        };
        if (a.act) { if (border) epi1(std::integral_constant<int, 1>(), std::true_type()); else epi1(std::integral_constant<int, 1>(), std::false_type()); }
        else { if (border) epi1(std::integral_constant<int, 0>(), std::true_type()); else epi1(std::integral_constant<int, 0>(), std::false_type()); }
        DDIMX_STAMP_AT(3);
        // ---- epilogue 2: whole pixel rows leave with 16-byte stores; statistics of the values as stored.  WN == 1: a wave's MFMA
        // blocks hold ALL channels of its pixels, so it re-reads only what it wrote itself (LDS operations of one wave execute in
        // order): no workgroup barrier between the two epilogues.
        {
            const unsigned cbase = (unsigned)(oc * 16);
#pragma unroll
            for (int k = 0; k < F::NPASS; ++k) {
                const int p = wm * (F::MT * 32) + oslot_w + k * (64 / F::OLPP);
                const int vy = y0 + p / TW, vx = x0 + p % TW;
                const uint4 v = *(const uint4*)(otile + p * F::OSTRIDE + oc * 16);
                buf_store16(out_rsrc, (unsigned)((vy * a.W + vx) * C * ES) + cbase, v);
                f32x2_t f[NP];
                Pairs<T>::unpack(v, f);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    st_s[j] += f[j];
                    st_q[j] = fma2(f[j], f[j], st_q[j]);
                }
            }
        }
        cur ^= 1;
        ty = nty; tx = ntx;
        DDIMX_STAMP_AT(6);
    }
    DDIMX_STAMP_AT(10);

    // ---- statistics: one partial per workgroup (as conv_mfma_kernel) -----------------------------------------------------
    if (a.stats) {  // uniform
        float* const red = (float*)smem;
#pragma unroll
        for (int o = F::OLPP; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                st_s[j].x += __shfl_xor(st_s[j].x, o, 64);
                st_s[j].y += __shfl_xor(st_s[j].y, o, 64);
                st_q[j].x += __shfl_xor(st_q[j].x, o, 64);
                st_q[j].y += __shfl_xor(st_q[j].y, o, 64);
            }
        }
        __syncthreads();  // everyone is done with the weights / output tile
        if (lane < F::OLPP && ovalid) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                red[(wave * C + oc * EPB + 2 * j) * 2 + 0] = st_s[j].x;
                red[(wave * C + oc * EPB + 2 * j) * 2 + 1] = st_q[j].x;
                red[(wave * C + oc * EPB + 2 * j + 1) * 2 + 0] = st_s[j].y;
                red[(wave * C + oc * EPB + 2 * j + 1) * 2 + 1] = st_q[j].y;
            }
        }
        __syncthreads();
        const int nparts = a.wgs_per_sample;
        if (a.stats_groups_c) {
            if (wave == 0)
                gn_bins_store<F::NWAVES>(red, C * 2, C, 0, a.stats_groups_c, a.stats + ((size_t)bs * nparts + wg) * kGnSlab, lane);
        } else {
            for (int i = tid; i < C * 2; i += F::NTHREADS) {
                float tt = 0.f;
#pragma unroll
                for (int w = 0; w < F::NWAVES; ++w) tt += red[w * C * 2 + i];
                a.stats[((size_t)bs * nparts + wg) * C * 2 + i] = tt;
            }
        }
    }
    DDIMX_STAMP_AT(11);
    DDIMX_STAMP_FLUSH();
}

template <class F>
hipError_t launch_fold_cfg(const FoldArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_fold_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3_fold_kernel<F>, dim3(a.wgs_per_sample * a.B), dim3(F::NTHREADS), F::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// conv_inst_bf16_fold.hip: tile geometry of the folded kernel for C channels (hipErrorInvalidValue: none), and its launch
struct FoldGeom { int th, tw, lds_bytes, nthreads; };
hipError_t fold_geometry(int C, FoldGeom* g);
hipError_t fold_launch(int C, const FoldArgs& a, hipStream_t stream);

}  // namespace ddimx
