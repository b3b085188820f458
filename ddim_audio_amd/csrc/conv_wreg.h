// 3x3 convolution with the WEIGHTS STREAMED STRAIGHT INTO REGISTERS (inference walk, bf16, C >= 64: gfx950).
//
// conv_mfma_kernel stages its weights through LDS.  From C = 64 up they do not fit beside the halo, so every workgroup
// re-streams all 9 * C * C * 2 bytes of them through a three-stage LDS-DMA ring for EVERY tile of 256 pixels or fewer -- 73 KB at
// C = 64, 166 KB at C = 96, 1.2 MB at C = 256 -- and LDS-DMA moves about 25 GB/s per CU (MI355X_MICROARCH.md, ldsdma-fill): the
// level-1..5 convolutions were bound by that stream (one 20 KB chunk per 1.2-1.4 us at C = 96: 45 us for 17 us of MFMA work), with
// a workgroup barrier and 3-5 DMA issues per chunk on top.
// Here the A operand never touches LDS: the weights are packed once per weight set in MFMA FRAGMENT ORDER
//     wf[step = tap * KG + kg][n-block][lane = h * 32 + l31][8]   =   W[cout = nb * 32 + l31][tap][cin = kg * 16 + h * 8 + j]
// so that the fragment of one (tap, 16-channel group, 32-cout block) is ONE fully coalesced 1 KiB wave load, served by L2 (the
// whole level's weights stay resident there).  A wave owns one 32-cout block (NT = 1) and MT >= 1 blocks of 32 pixels, so each
// fragment feeds MT MFMAs; fragments are loaded D steps ahead into a rolling register ring that simply wraps into the next tile
// (the weights do not depend on the tile), B operands come from the transformed halo tile in LDS one step ahead.  No weight ring,
// no chunk barriers, no DMA bookkeeping: the MFMA loop of a tile is 9 * C / 16 straight-line steps.
// LDS holds the halo / output tile only, so two or three workgroups share a CU where one did.
// Everything around the MFMA loop is conv_mfma_kernel's: GroupNorm affine (+ SiLU) applied while the halo is staged, the
// consumer-side GroupNorm finalisation (gn_fused.h), + bias / + timestep embedding, SiLU, whole-row stores, group statistics.
// Whole tiles only (the host falls back to conv_mfma_kernel for ragged images).
#pragma once
#include "conv_mfma.h"

namespace ddimx {

struct WregArgs {
    const void* in;         // [B][H][W][C] bf16
    const void* wf;         // fragment-order weights (pack_conv_frag_launch / pack_frag_from_taps_launch; UP4: [2 classes] of them)
    const void* skip;       // UP4: tensor of the output's shape added in the epilogue, or null
    const float* bias;      // [C] or null
    const float* chan_add;  // per-sample per-cout vector or null
    int chan_add_stride;
    const float* in_scale;  // [B][C] folded GroupNorm (xf != XF_NONE, gn.stats == null)
    const float* in_shift;
    GnIn gn;
    void* out;              // [B][H][W][C] bf16
    float* stats;
    int stats_groups_c;
    int xf, act;            // XF_NONE / XF_AFFINE / XF_AFFINE_SILU; act 0 / 1 (SiLU)
    int B, H, W;            // the INPUT image (Downsample: the output is H/2 x W/2)
    int tiles_x, tiles_y, tiles_per_wg, wgs_per_sample;  // tiles of the OUTPUT image
    unsigned long long* stamps;  // diagnostic builds only (-DDDIMX_STAMP)
    int dbg;                     // diagnostic builds only
};

template <int CIN_, int COUT_, int MODE_, int TH_, int TW_, int WM_, int WN_, int D_, int NS_ = 1, int KS_ = 1>
struct WregCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, MODE = MODE_, TH = TH_, TW = TW_, WM = WM_, WN = WN_, D = D_;
    // CONV3: 3x3 stride 1 (Residual_Block); DOWN4: 4x4 stride 2 (Downsample); UP4: ConvTranspose2d 4x4 stride 2 (Upsample) as
    // sub-pixel convolutions exactly as conv_mfma_kernel does them: grid.z = output-row parity, COUT = 2 x the real output
    // channels (the two column parities side by side: one input pixel -> 2 * Cprev contiguous output elements), 2 x 3 taps of the
    // input rows vy + cls .. + 1, the skip tensor added in the epilogue (models/diffusion.py:59-67,284).
    static constexpr int NTAPS = MODE == DOWN4 ? 16 : (MODE == UP4 ? 6 : 9), TAPW = MODE == DOWN4 ? 4 : 3, SXY = MODE == DOWN4 ? 2 : 1;
    static constexpr int NCLS = MODE == UP4 ? 2 : 1;
    static constexpr int NS = NS_;      // output-channel splits: grid.y workgroups share a pixel tile (the latency-bound deep levels
                                        // have too few pixels to fill 256 CUs otherwise; each stages the small halo for itself)
    static constexpr int NB = COUT / NS;   // output channels of one workgroup
    static constexpr int ES = 2, EPB = 8;
    static constexpr int KS = KS_;      // K splits: KS wave groups share a (pixel block, cout block) and take NSTEP / KS steps each; the deep
                                        // levels are a dependent chain of 72-144 steps per wave on an otherwise empty chip (0.5-1.1
                                        // resident waves per SIMD, profiles/r03/final/step_pmc.txt); the partial accumulators meet in LDS
    static constexpr int NWAVES = WM * WN * KS, NTHREADS = 64 * NWAVES;
    static constexpr int P = TH * TW;
    static constexpr int MT = P / (32 * WM), NT = NB / (32 * WN);
    static constexpr int KACC_BYTES = (KS - 1) * WM * WN * MT * 16 * 64 * 4;  // partial accumulators of the wave groups ks > 0
    static constexpr int KG = CIN / 16, NSTEP = NTAPS * KG, NBLK = COUT / 32, NSL = NSTEP / KS;
    static constexpr int IH = SXY * TH + 2, IW = SXY * TW + 2, NPIX = IH * IW;
    static constexpr int PSTRIDE = CIN * ES + 16;
    static constexpr int ROWRAW = IW * PSTRIDE;
    static constexpr int ROWRES = TW == 8 ? 128 : 0;  // (row stride mod 256 B) wanted when a 32-pixel block spans rows
    static constexpr int ROWSTRIDE = (TW >= 32 || SXY == 2) ? ROWRAW : ROWRAW + ((ROWRES - ROWRAW % 256) + 256) % 256;
    static constexpr int HALO_BYTES = IH * ROWSTRIDE;
    static constexpr int OSTRIDE = NB * ES + 16;
    static constexpr int OUT_BYTES = P * OSTRIDE;
    static constexpr int HO_BYTES = HALO_BYTES > OUT_BYTES ? HALO_BYTES : OUT_BYTES;  // the output tile overlays the halo
    static constexpr int ADD_BYTES = NB * 4;
    static constexpr int GN_BYTES = NWAVES * kGroups * 2 * 4;
    static constexpr int RED_BYTES = NWAVES * NB * 2 * 4;
    static constexpr int LDS_RAW = ADD_BYTES + HO_BYTES + KACC_BYTES + GN_BYTES + 256;  // + 256 B sink of the weight warm-up touches
    static constexpr int LDS_BYTES = LDS_RAW > RED_BYTES ? LDS_RAW : RED_BYTES;
    static constexpr int CPP = CIN / EPB, LPP = next_pow2(CPP);
    static constexpr int PPP = NTHREADS / LPP, HPT = (NPIX + PPP - 1) / PPP;
    static constexpr int OPP = NB / EPB, OLPP = next_pow2(OPP);
    static constexpr int STEP = NTHREADS / OLPP, NPASS = (P + STEP - 1) / STEP;
    static constexpr int WG_PER_CU = (160 * 1024) / LDS_BYTES < 1 ? 1 : (160 * 1024) / LDS_BYTES;
    // waves per SIMD the LDS footprint allows (at most 3 asked of the register allocator: <= 168 registers per lane)
    static constexpr int W_LDS = (WG_PER_CU * NWAVES) / 4;
    // (asking for three -- two six-wave workgroups per CU -- was measured: 65 spilled registers at C = 96, 41 -> 68 us per launch)
    static constexpr int MINW = W_LDS < 1 ? 1 : (W_LDS > 2 ? 2 : W_LDS);
    static_assert(NT == 1, "a wave owns one 32-cout block: every weight fragment is loaded by exactly WM waves");
    static_assert(P % (32 * WM) == 0 && MT >= 1, "pixel tile must split into 32-pixel MFMA blocks");
    static_assert(TW == 8 || TW == 16 || TW == 32, "TW");
    static_assert((PSTRIDE / 16) % 2 == 1, "pixel stride must be odd in 16-byte slots");
    static_assert(NSTEP % KS == 0 && D >= 2 && D <= NSL && NSL % D == 0, "prefetch depth (the register ring wraps into the next tile: D must divide the step count of a wave)");
    static_assert(NTHREADS % LPP == 0 && NTHREADS % OLPP == 0 && OLPP <= 64 && NTHREADS <= 1024, "thread maps");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(HPT <= 24, "halo pieces per thread (all of a tile's loads are kept in registers)");
    static_assert(COUT % NS == 0 && NB % 32 == 0 && CIN % 16 == 0, "cout split");
};

template <class F>
__global__ void __launch_bounds__(F::NTHREADS, F::MINW) conv3_wreg_kernel(const WregArgs a) {
    typedef __bf16 T;
    constexpr int CIN = F::CIN, COUT = F::COUT, ES = 2, EPB = 8, NP = 4, TW = F::TW, SXY = F::SXY;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const addv = (float*)smem;
    char* const halo = smem + F::ADD_BYTES;
    char* const otile = halo;
    float* const gnscr = (float*)(smem + F::LDS_RAW - F::GN_BYTES - 256);
    float* const kacc = (float*)(smem + F::ADD_BYTES + F::HO_BYTES);
    char* const sink = smem + F::LDS_RAW - 256;

    DDIMX_STAMP_ENTRY
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % F::WM, wn = (wave / F::WM) % F::WN, ks = wave / (F::WM * F::WN);
    const int l31 = lane & 31, h = lane >> 5;

    int lwg;  // XCD-aware order (as conv_mfma_kernel)
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
    const int cout0 = blockIdx.y * F::NB;
    const int cls = blockIdx.z;  // UP4: output-row parity; 0 otherwise

    // ---- weight fragments: step s of this wave's cout block = one coalesced 16-byte-per-lane load -------------------------
    // (buffer load: resource in SGPRs, ONE per-lane offset register, the step as scalar offset -- 64-bit per-step addresses would be
    // hoisted out of the tile loop by the compiler, two registers per step, and spill)
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc((const char*)a.wf + (size_t)cls * (F::NTAPS * CIN * COUT * ES), (unsigned)(F::NTAPS * CIN * COUT * ES));
    const unsigned wlane = (unsigned)(((cout0 / 32 + wn) * 64 + lane) * 16);
    auto wfrag = [&](int s) __attribute__((always_inline)) -> uint4 {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane, s * (F::NBLK * 1024), 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    // L2 warm-up: inside the step every conv's weights were last read one U-Net evaluation ago and have long left the XCD's L2
    // (18 MB of deep-level weights per step); the fragment stream then misses at every step and runs at (miss latency / D) per step
    // (14 -> 24 us per launch at C = 256).  So the launch first TOUCHES every 128-byte line of the weight tensor once per XCD, all
    // misses in flight together, under the GroupNorm reduction and the halo staging: hardware deals workgroups round-robin over the
    // 8 XCDs (observed, speed only -- a wrong guess only warms the wrong L2), so workgroup i takes slice i / 8 of the lines of "its"
    // XCD, one dword per lane.  The touches are LDS-DMA into a 256-byte sink: no destination registers, nothing ever waits for
    // them except, in order, the first fragment load.
    {
        u32x4_t wrs;
        const uint64_t wp = (uint64_t)a.wf + (uint64_t)cls * (F::NTAPS * CIN * COUT * ES);
        wrs[0] = __builtin_amdgcn_readfirstlane((unsigned)wp);
        wrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(wp >> 32));
        wrs[2] = (unsigned)(F::NTAPS * CIN * COUT * ES);
        wrs[3] = 0x00020000u;
        const unsigned sink_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(sink));
        constexpr unsigned NL = (unsigned)F::NTAPS * CIN * COUT * ES / 128u;
        const unsigned nsl = (gridDim.x * gridDim.y + 7u) >> 3;  // workgroups per XCD of ONE class
        const unsigned slice = (blockIdx.y * gridDim.x + blockIdx.x) >> 3;  // (per row-parity class: its own weight tensor)
#pragma unroll 1
        for (unsigned k = 0; (k * F::NWAVES * nsl) * 64u < NL; ++k) {
            const unsigned line = ((k * F::NWAVES + (unsigned)wave) * nsl + slice) * 64u + (unsigned)lane;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "buffer_load_dword %1, %2, 0 offen lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(line < NL ? line * 128u : 0x80000000u), "s"(wrs), "s"(sink_lds)
                : "memory");
        }
    }
    uint4 aw[F::D];
#pragma unroll
    for (int d = 0; d < F::D; ++d) aw[d] = wfrag(ks * F::NSL + d);

    // ---- halo staging (register transform, as conv_mfma_kernel's general path) ----------------------------------------------
    const int hc = tid % F::LPP, hslot = tid / F::LPP;
    const bool hvalid = hc < F::CPP;
    f32x2_t sc[NP], sh[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
    const unsigned in_bytes = (unsigned)((size_t)a.H * a.W * CIN * ES);  // a.H x a.W: the INPUT image
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc((const T*)a.in + (size_t)bs * a.H * a.W * CIN, in_bytes);
    auto piece_xf = [&](auto xf_tag, uint4 v, bool ok) __attribute__((always_inline)) -> uint4 {
        constexpr int XF = decltype(xf_tag)::value;
        if (XF != XF_NONE) {
            f32x2_t f[NP];
            Pairs<T>::unpack(v, f);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                f[j] = fma2(f[j], sc[j], sh[j]);
                if (XF == XF_AFFINE_SILU) f[j] = silu2(f[j]);
            }
            const uint4 tv = Pairs<T>::pack(f);
            v.x = ok ? tv.x : 0u; v.y = ok ? tv.y : 0u; v.z = ok ? tv.z : 0u; v.w = ok ? tv.w : 0u;
        }
        return v;
    };
    // all pieces of a tile in flight at once (one memory round trip per tile, not one per group of pieces)
    uint4 hreg[F::HPT];
    unsigned hok = 0;
    auto halo_issue = [&](int ty, int tx) __attribute__((always_inline)) {
        const int hy0 = ty * F::TH * SXY - 1, hx0 = tx * TW * SXY - 1;
        hok = 0;
#pragma unroll
        for (int i = 0; i < F::HPT; ++i) {
            const int pix = i * F::PPP + hslot;
            const int gy = hy0 + pix / F::IW, gx = hx0 + pix % F::IW;
            const bool ok = hvalid && pix < F::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            hreg[i] = buf_load16(in_rsrc, ok ? (unsigned)(((gy * a.W + gx) * CIN + hc * EPB) * ES) : kOOB);
            hok |= ok ? (1u << i) : 0u;
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // keep the loads here (hipcc sinks them to their use otherwise)
    };
    auto halo_commit_xf = [&](auto xf_tag) __attribute__((always_inline)) {
        if (!hvalid) return;
#pragma unroll
        for (int i = 0; i < F::HPT; ++i) {
            const int pix = i * F::PPP + hslot;
            if (pix < F::NPIX)
                *(uint4*)(halo + (pix / F::IW) * F::ROWSTRIDE + (pix % F::IW) * F::PSTRIDE + hc * 16) = piece_xf(xf_tag, hreg[i], (hok >> i) & 1u);
        }
    };
    auto halo_commit = [&]() __attribute__((always_inline)) {
        if (a.xf == XF_AFFINE_SILU) halo_commit_xf(std::integral_constant<int, XF_AFFINE_SILU>());
        else if (a.xf == XF_AFFINE) halo_commit_xf(std::integral_constant<int, XF_AFFINE>());
        else halo_commit_xf(std::integral_constant<int, XF_NONE>());
    };

    // ---- per-lane operand offsets / epilogue constants ----------------------------------------------------------------------
    int pixoff[F::MT];
#pragma unroll
    for (int m = 0; m < F::MT; ++m) {
        const int p = (wm * F::MT + m) * 32 + l31;
        pixoff[m] = ((p / TW) * SXY + (F::MODE == UP4 ? cls : 0)) * F::ROWSTRIDE + (p % TW) * SXY * F::PSTRIDE + h * 16;
    }
    const int oc = tid % F::OLPP, oslot = tid / F::OLPP;
    const bool ovalid = oc < F::OPP;
    f32x2_t st_s[NP], st_q[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
    // output image: Ho rows of Wo "virtual" pixels x COUT (UP4: 2 * H rows, each virtual pixel = two real ones side by side)
    const int Ho = F::MODE == UP4 ? 2 * a.H : a.H / SXY, Wo = a.W / SXY;
    const unsigned out_bytes = (unsigned)((size_t)Ho * Wo * COUT * ES);
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc((T*)a.out + (size_t)bs * Ho * Wo * COUT, out_bytes);
    const __amdgpu_buffer_rsrc_t skip_rsrc = make_rsrc(a.skip ? (const T*)a.skip + (size_t)bs * Ho * Wo * COUT : (const T*)a.out, a.skip ? out_bytes : 0u);

    // ---- prologue (as conv_mfma_kernel): addend, GroupNorm input, first halo -------------------------------------------------
    constexpr int AIT = (F::NB + F::NTHREADS - 1) / F::NTHREADS;
    float add_b[AIT], add_c[AIT];
    {
        const float* pb = a.bias ? a.bias + cout0 : (const float*)a.wf;
        const float* pc = a.chan_add ? a.chan_add + (size_t)bs * a.chan_add_stride + cout0 : (const float*)a.wf;
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = tid + k * F::NTHREADS;
            const int ic = i < F::NB ? i : F::NB - 1;
            add_b[k] = pb[ic];
            add_c[k] = pc[ic];
        }
    }
    const bool gn_fused = a.xf != XF_NONE && a.gn.stats != nullptr;  // uniform
    GnInLoads gn_ld;
    if (gn_fused) gn_in_issue(a.gn, bs, tid, F::NTHREADS, gn_ld);
    int ty = t_begin / a.tiles_x, tx = t_begin % a.tiles_x;
    if (a.xf != XF_NONE && hvalid) {
        const float* psc = gn_fused ? a.gn.gamma + hc * EPB : a.in_scale + (size_t)bs * CIN + hc * EPB;
        const float* psh = gn_fused ? (a.gn.beta ? a.gn.beta : a.gn.gamma) + hc * EPB : a.in_shift + (size_t)bs * CIN + hc * EPB;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            sc[j] = *(const f32x2_t*)(psc + 2 * j);
            sh[j] = *(const f32x2_t*)(psh + 2 * j);
        }
    }
    halo_issue(ty, tx);  // the first tile's halo overlaps the GroupNorm reduction below
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int i = tid + k * F::NTHREADS;
        if (i < F::NB) addv[i] = (a.bias ? add_b[k] : 0.f) + (a.chan_add ? add_c[k] : 0.f);
    }
    if (gn_fused) {
        gn_in_reduce(a.gn, bs, tid, F::NTHREADS, gn_ld, gnscr);
        __syncthreads();
        if (hvalid) {
            float gam[EPB], bet[EPB], fs[EPB], fh[EPB];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                gam[2 * j] = sc[j].x; gam[2 * j + 1] = sc[j].y;
                bet[2 * j] = a.gn.beta ? sh[j].x : 0.f; bet[2 * j + 1] = a.gn.beta ? sh[j].y : 0.f;
            }
            gn_in_fold<EPB>(a.gn, gnscr, F::NWAVES, CIN, hc * EPB, gam, bet, fs, fh);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                sc[j].x = fs[2 * j]; sc[j].y = fs[2 * j + 1];
                sh[j].x = fh[2 * j]; sh[j].y = fh[2 * j + 1];
            }
        }
    }
    halo_commit();
    __syncthreads();
    DDIMX_STAMP_DECL

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int y0 = ty * F::TH, x0 = tx * TW;
        DDIMX_STAMP_AT(0);
        f32x16_t acc[F::MT];
#pragma unroll
        for (int m = 0; m < F::MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

        // ---- MFMA loop: step s = (tap, 16-channel group).  A = aw[s % D] (loaded D steps ago, possibly during the previous tile),
        // B = bq[s & 1] (LDS, loaded one step ago); the scheduling barriers keep hipcc from sinking the loads down to their use.
        uint4 bq[2][F::MT];
        auto run_steps = [&](auto ks_tag) __attribute__((always_inline)) {  // this wave's NSL steps (offsets are compile-time per KS group)
            constexpr int S0 = decltype(ks_tag)::value * F::NSL;
            {
                constexpr int tap = S0 / F::KG, kg = S0 % F::KG;
                constexpr int hoff = (tap / F::TAPW) * F::ROWSTRIDE + (tap % F::TAPW) * F::PSTRIDE + kg * 32;
#pragma unroll
                for (int m = 0; m < F::MT; ++m) bq[0][m] = *(const uint4*)(halo + pixoff[m] + hoff);
            }
#pragma unroll
            for (int s = 0; s < F::NSL; ++s) {
                if (s + 1 < F::NSL) {
                    const int tap = (S0 + s + 1) / F::KG, kg = (S0 + s + 1) % F::KG;
                    const int hoff = (tap / F::TAPW) * F::ROWSTRIDE + (tap % F::TAPW) * F::PSTRIDE + kg * 32;
#pragma unroll
                    for (int m = 0; m < F::MT; ++m) bq[(s + 1) & 1][m] = *(const uint4*)(halo + pixoff[m] + hoff);
                }
                __builtin_amdgcn_sched_barrier(0);
                const uint4 af = aw[s % F::D];
#pragma unroll
                for (int m = 0; m < F::MT; ++m) Mma<T>::run(af, bq[s & 1][m], acc[m]);
                aw[s % F::D] = wfrag(S0 + (s + F::D) % F::NSL);  // wraps into the next tile: the weights do not depend on the tile
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (F::KS == 1) {
            run_steps(std::integral_constant<int, 0>());
        } else {
            static_assert(F::KS == 2, "K split");
            if (ks == 0) run_steps(std::integral_constant<int, 0>()); else run_steps(std::integral_constant<int, 1>());
        }
        DDIMX_STAMP_AT(1);
        if constexpr (F::KS > 1) {  // the partial accumulators of the wave groups ks > 0 go through LDS (a region of their own)
            if (ks > 0) {
                float* dstp = kacc + (((ks - 1) * F::WN + wn) * F::WM + wm) * (F::MT * 16 * 64) + lane * 4;
#pragma unroll
                for (int m = 0; m < F::MT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *(float4*)(dstp + (m * 4 + q) * 256) = make_float4(acc[m][q * 4 + 0], acc[m][q * 4 + 1], acc[m][q * 4 + 2], acc[m][q * 4 + 3]);
            }
        }
        __syncthreads();  // barrier A: every wave is done reading this tile's halo (and the partial accumulators are in LDS)
        DDIMX_STAMP_AT(2);
        if constexpr (F::KS > 1) {
            if (ks == 0) {
#pragma unroll
                for (int k2 = 1; k2 < F::KS; ++k2) {
                    const float* srcp = kacc + (((k2 - 1) * F::WN + wn) * F::WM + wm) * (F::MT * 16 * 64) + lane * 4;
#pragma unroll
                    for (int m = 0; m < F::MT; ++m)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 pv = *(const float4*)(srcp + (m * 4 + q) * 256);
                            acc[m][q * 4 + 0] += pv.x; acc[m][q * 4 + 1] += pv.y; acc[m][q * 4 + 2] += pv.z; acc[m][q * 4 + 3] += pv.w;
                        }
                }
            }
        }

        // ---- epilogue 1: accumulators + addend -> SiLU -> bf16 -> output tile [pixel][cout] (overlays the halo) ------------------
        auto epi1 = [&](auto act_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
            float4 av[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = *(const float4*)(addv + wn * 32 + q * 8 + h * 4);
#pragma unroll
            for (int m = 0; m < F::MT; ++m) {
                const int p = (wm * F::MT + m) * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wn * 32 + q * 8 + h * 4;
                    f32x2_t v01 = {acc[m][q * 4 + 0], acc[m][q * 4 + 1]};
                    f32x2_t v23 = {acc[m][q * 4 + 2], acc[m][q * 4 + 3]};
                    v01 += (f32x2_t){av[q].x, av[q].y};
                    v23 += (f32x2_t){av[q].z, av[q].w};
                    if (ACT) { v01 = silu2(v01); v23 = silu2(v23); }
                    *(uint2*)(otile + p * F::OSTRIDE + cl * ES) = make_uint2(Piece<__bf16>::pk(v01.x, v01.y), Piece<__bf16>::pk(v23.x, v23.y));
                }
            }
        };
        if (F::KS == 1 || ks == 0) { if (a.act) epi1(std::integral_constant<int, 1>()); else epi1(std::integral_constant<int, 0>()); }
        DDIMX_STAMP_AT(3);
        __syncthreads();  // barrier B: output tile complete
        DDIMX_STAMP_AT(5);

        // ---- epilogue 2: whole pixel rows leave with 16-byte stores (+ skip tensor); statistics of the values as stored ------------
        auto epi2 = [&](auto skip_tag) __attribute__((always_inline)) {
            constexpr bool SKIP = decltype(skip_tag)::value;
            if (!ovalid) return;
            const unsigned cbase = (unsigned)(cout0 * ES + oc * 16);
            unsigned offs[F::NPASS];
            uint4 skv[SKIP ? F::NPASS : 1];
#pragma unroll
            for (int k = 0; k < F::NPASS; ++k) {  // all skip loads of the tile in flight before the first store
                const int p = oslot + k * F::STEP;
                const int vy = y0 + p / TW, vx = x0 + p % TW;
                const bool in = F::P % F::STEP == 0 || p < F::P;
                const int oy = F::MODE == UP4 ? 2 * vy + cls : vy;
                offs[k] = in ? (unsigned)((oy * Wo + vx) * COUT * ES) + cbase : kOOB;
                if (SKIP) skv[k] = buf_load16(skip_rsrc, offs[k]);
            }
#pragma unroll
            for (int k = 0; k < F::NPASS; ++k) {
                const int p = oslot + k * F::STEP;
                if (F::P % F::STEP != 0 && p >= F::P) break;
                uint4 v = *(const uint4*)(otile + p * F::OSTRIDE + oc * 16);
                f32x2_t f[NP];
                Pairs<T>::unpack(v, f);
                if (SKIP) {
                    f32x2_t kk[NP];
                    Pairs<T>::unpack(skv[k], kk);
#pragma unroll
                    for (int j = 0; j < NP; ++j) f[j] += kk[j];
                    v = Pairs<T>::pack(f);
                    Pairs<T>::unpack(v, f);  // statistics of the values as stored
                }
                buf_store16(out_rsrc, offs[k], v);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    st_s[j] += f[j];
                    st_q[j] = fma2(f[j], f[j], st_q[j]);
                }
            }
        };
        if (a.skip) epi2(std::true_type()); else epi2(std::false_type());
        DDIMX_STAMP_AT(6);
        if (++tx == a.tiles_x) { tx = 0; ++ty; }
        if (t + 1 < t_end) {
            halo_issue(ty, tx);
            __syncthreads();  // barrier C: output tile fully read before the halo region is overwritten
            DDIMX_STAMP_AT(7);
            halo_commit();
            DDIMX_STAMP_AT(8);
            __syncthreads();  // barrier D
            DDIMX_STAMP_AT(9);
        }
    }
    DDIMX_STAMP_AT(10);

    // ---- statistics: one partial per workgroup (as conv_mfma_kernel) --------------------------------------------------------------
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
        __syncthreads();
        if (lane < F::OLPP && ovalid) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                red[(wave * F::NB + oc * EPB + 2 * j) * 2 + 0] = st_s[j].x;
                red[(wave * F::NB + oc * EPB + 2 * j) * 2 + 1] = st_q[j].x;
                red[(wave * F::NB + oc * EPB + 2 * j + 1) * 2 + 0] = st_s[j].y;
                red[(wave * F::NB + oc * EPB + 2 * j + 1) * 2 + 1] = st_q[j].y;
            }
        }
        __syncthreads();
        const int nparts = a.wgs_per_sample * F::NCLS;
        const int part = wg * F::NCLS + cls;
        if (a.stats_groups_c) {
            if (wave == 0)
                gn_bins_store<F::NWAVES>(red, F::NB * 2, F::NB, cout0, a.stats_groups_c,
                                         a.stats + (((size_t)bs * nparts + part) * F::NS + blockIdx.y) * kGnSlab, lane);
        } else {
            for (int i = tid; i < F::NB * 2; i += F::NTHREADS) {
                float tt = 0.f;
#pragma unroll
                for (int w = 0; w < F::NWAVES; ++w) tt += red[w * F::NB * 2 + i];
                a.stats[(((size_t)bs * nparts + part) * COUT + cout0) * 2 + i] = tt;
            }
        }
    }
    DDIMX_STAMP_AT(11);
    DDIMX_STAMP_FLUSH();
}

template <class F>
hipError_t launch_wreg_cfg(const WregArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_wreg_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3_wreg_kernel<F>, dim3(a.wgs_per_sample * a.B, F::NS, F::NCLS), dim3(F::NTHREADS), F::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// conv_inst_bf16_wreg.hip
struct WregGeom { int th, tw, lds_bytes, nthreads, nsplit; };
// packed tap layout [ntaps][NOUT][CIN] bf16 (ddimx_pack_conv / one row-parity class of ddimx_pack_convT) -> fragment order
hipError_t pack_frag_from_taps_launch(const void* src, void* dst, int ntaps, int NOUT, int CIN, hipStream_t s);
hipError_t wreg_geometry(int mode, int cin, int cout, WregGeom* g);
hipError_t wreg_launch(int mode, int cin, int cout, const WregArgs& a, hipStream_t stream);
// weights [O][I][KH][KW] fp32 -> bf16 fragment order [KH*KW * I/16][O/32][64][8]  (kernels.hip)
hipError_t pack_conv_frag_launch(const float* w, void* dst, int O, int I, int KK, hipStream_t s);

}  // namespace ddimx
