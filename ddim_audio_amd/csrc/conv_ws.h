// 3x3 convolution of the Residual_Block with SPECIALISED WAVES (inference walk, bf16, C = 32 / 64 / 96: gfx950).
//
// conv_mfma_kernel and conv3_wreg_kernel run every tile as a sequence of phases that all waves of the workgroup execute
// together -- halo load (a memory round trip), transform + LDS write, MFMA loop, epilogue to LDS, stores -- separated by
// barriers; the matrix pipe is busy 18-28 % of the time (rocprofv3 PMC, profiles/r03/conv_pmc_final.txt), the second workgroup of
// the CU overlaps little of it (a wave parks at every barrier and at the halo wait), and the weight stream through `vmcnt` makes
// any halo load that is put into the MFMA loop be waited for D steps later (the counter is in order).
// Here the roles are split between the waves of ONE workgroup per CU:
//   * MFMA waves (WM x C/32) only multiply: B operands from the transformed halo of tile t in LDS, A operands from the
//     fragment-order weights (conv_wreg.h: resident in registers at C = 32, a rolling register ring from C = 64 up), then the
//     first epilogue (+ bias / + embedding, SiLU, bf16, output tile in LDS);
//   * LOADER waves (NL) never touch the matrix pipe: while tile t multiplies they drain the output tile of t - 1 to memory
//     (whole pixel rows, statistics of the values as stored), transform the halo of tile t + 1 -- requested one tile period
//     earlier -- into the OTHER halo buffer (GroupNorm affine, SiLU) and request the halo of tile t + 2.
// Two barriers per tile, both roles reach every one of them:
//     A(t): the MFMA waves are done reading halo[t & 1]; halo[(t + 1) & 1] is complete; the output tile of t - 1 has left
//     B(t): the output tile of t is complete
// so per tile the critical path is max(MFMA loop, loader work) + first epilogue, and the loads, the element-wise work of the
// loaders and the matrix work overlap inside the workgroup instead of relying on a neighbour.
// A workgroup walks TPW consecutive tiles of one sample (sample size only: bit-identical alone or in any batch) and is alone on
// its CU (LDS: two halos + one output tile); 32 workgroups per sample at T = 1024 = one round of 256 CUs at a batch of 8, and
// the statistics of a sample are 32 partial slabs instead of 128-256 (the consumers' prologues shrink with them).
// Interface: WregArgs (conv_wreg.h), whole tiles only.
#pragma once
#include "conv_wreg.h"

namespace ddimx {

template <int C_, int TH_, int TW_, int WM_, int NL_, int D_, int TPW_>
struct WsCfg {
    static constexpr int C = C_, CIN = C_, COUT = C_, TH = TH_, TW = TW_, WM = WM_, WN = C_ / 32, NL = NL_, D = D_, TPW = TPW_;
    static constexpr int NMW = WM * WN, NWAVES = NMW + NL, NTHREADS = 64 * NWAVES, NLT = 64 * NL;
    static constexpr int ES = 2, EPB = 8;
    static constexpr int P = TH * TW;
    static constexpr int MT = P / (32 * WM);
    static constexpr int KG = C / 16, NSTEP = 9 * KG, NBLK = C / 32;
    static constexpr int IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    static constexpr int PSTRIDE = C * ES + 16;
    static constexpr int ROWRAW = IW * PSTRIDE;
    static constexpr int ROWRES = TW == 8 ? 128 : 0;
    static constexpr int ROWSTRIDE = TW >= 32 ? ROWRAW : ROWRAW + ((ROWRES - ROWRAW % 256) + 256) % 256;
    static constexpr int HALO_BYTES = IH * ROWSTRIDE;
    static constexpr int OSTRIDE = C * ES + 16;
    static constexpr int OUT_BYTES = P * OSTRIDE;
    static constexpr int ADD_BYTES = C * 4;
    static constexpr int GN_BYTES = NWAVES * kGroups * 2 * 4;
    static constexpr int RED_BYTES = WM * C * 2 * 4;
    static constexpr int LDS_BYTES = ADD_BYTES + 2 * HALO_BYTES + OUT_BYTES + GN_BYTES + 256;  // + sink of the weight warm-up touches
    static constexpr int CPP = C / EPB, LPP = next_pow2(CPP);
    static constexpr int PPP = NLT / LPP, HPT = (NPIX + PPP - 1) / PPP;
    static constexpr int OPP = C / EPB, OLPP = next_pow2(OPP);
    static constexpr int STEP = NLT / OLPP, NPASS = (P + STEP - 1) / STEP;
    static_assert(C % 32 == 0 && P % (32 * WM) == 0 && MT >= 1, "tile must split into 32-pixel MFMA blocks");
    static_assert(TW == 8 || TW == 16 || TW == 32, "TW");
    static_assert((PSTRIDE / 16) % 2 == 1, "pixel stride must be odd in 16-byte slots");
    static_assert(D >= 2 && D <= NSTEP && NSTEP % D == 0, "fragment ring depth");
    static_assert(NLT % LPP == 0 && NLT % OLPP == 0 && OLPP <= 64 && NTHREADS <= 1024, "thread maps");
    static_assert(LDS_BYTES <= 160 * 1024 && RED_BYTES <= ADD_BYTES + HALO_BYTES, "LDS budget");
    static_assert(HPT <= 24, "halo pieces per loader thread (all of a tile's loads are kept in registers)");
};

template <class F>
__global__ void __launch_bounds__(F::NTHREADS, (F::NWAVES + 3) / 4) conv3_ws_kernel(const WregArgs a) {
    typedef __bf16 T;
    constexpr int C = F::C, ES = 2, EPB = 8, NP = 4, TW = F::TW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const addv = (float*)smem;
    char* const halo0 = smem + F::ADD_BYTES;
    char* const otile = halo0 + 2 * F::HALO_BYTES;
    float* const gnscr = (float*)(otile + F::OUT_BYTES);
    char* const sink = (char*)gnscr + F::GN_BYTES;

    DDIMX_STAMP_ENTRY
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the role branches are s_cbranch, not exec masks)
    const bool mrole = wave < F::NMW;
    const int wm = wave % F::WM, wn = wave / F::WM;   // MFMA waves: pixel-block group, 32-cout block
    const int ltid = tid - 64 * F::NMW;               // loader threads: 0 .. NLT - 1
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

    // ---- weight fragments (MFMA waves): step s of this wave's cout block = one coalesced 16-byte-per-lane load ----------------
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(a.wf, (unsigned)(9 * C * C * ES));
    const unsigned wlane = (unsigned)(((mrole ? wn : 0) * 64 + lane) * 16);
    auto wfrag = [&](int s) __attribute__((always_inline)) -> uint4 {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane, s * (F::NBLK * 1024), 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    if constexpr (F::D < F::NSTEP) {  // streamed weights: L2 warm-up touches as conv3_wreg_kernel (all waves, sliced per XCD)
        u32x4_t wrs;
        const uint64_t wp = (uint64_t)a.wf;
        wrs[0] = __builtin_amdgcn_readfirstlane((unsigned)wp);
        wrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(wp >> 32));
        wrs[2] = (unsigned)(9 * C * C * ES);
        wrs[3] = 0x00020000u;
        const unsigned sink_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(sink));
        constexpr unsigned NLN = (unsigned)9 * C * C * ES / 128u;
        const unsigned nsl = (gridDim.x + 7u) >> 3;
        const unsigned slice = blockIdx.x >> 3;
#pragma unroll 1
        for (unsigned k = 0; (k * F::NWAVES * nsl) * 64u < NLN; ++k) {
            const unsigned line = ((k * F::NWAVES + (unsigned)wave) * nsl + slice) * 64u + (unsigned)lane;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "buffer_load_dword %1, %2, 0 offen lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(line < NLN ? line * 128u : 0x80000000u), "s"(wrs), "s"(sink_lds)
                : "memory");
        }
    }
    uint4 aw[F::D];
    if (mrole) {
#pragma unroll
        for (int d = 0; d < F::D; ++d) aw[d] = wfrag(d);
    }

    // ---- loader state: halo staging (register transform) and the drain of the output tile ----------------------------------------
    const int hc = ltid % F::LPP, hslot = ltid / F::LPP;
    const bool hvalid = !mrole && hc < F::CPP;
    f32x2_t sc[NP], sh[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
    const unsigned in_bytes = (unsigned)((size_t)a.H * a.W * C * ES);
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc((const T*)a.in + (size_t)bs * a.H * a.W * C, in_bytes);
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
    uint4 hreg[F::HPT];
    unsigned hok = 0;
    // piece i of this thread = halo pixel i * PPP + hslot: its byte offset relative to the halo's first pixel is the same for every
    // tile (the address arithmetic of ten pieces -- constant divisions, bounds -- cost 1 400 cycles per tile when it was redone per
    // tile, tools/ws_stamps.py); interior tiles add the tile's base as the scalar offset of the buffer load
    unsigned hrel[F::HPT];
    unsigned hin = 0;  // pieces that exist (thread has a channel slice, pixel inside the halo)
#pragma unroll
    for (int i = 0; i < F::HPT; ++i) {
        const int pix = i * F::PPP + hslot;
        const bool in = hvalid && pix < F::NPIX;
        hrel[i] = in ? (unsigned)((((pix / F::IW) * a.W + pix % F::IW) * C + hc * EPB) * ES) : 0x80000000u;
        hin |= in ? (1u << i) : 0u;
    }
    // The halo loads are inline asm with a counted wait of their own (halo_wait): hipcc's waitcnt pass puts the drain's stores --
    // issued between a tile's loads (one iteration earlier) and their first use -- in front of the loads and ends the commit with
    // s_waitcnt vmcnt(0), i.e. every tile waited for its predecessor's output to be WRITTEN (commit 3 250 -> 4 900 cycles per tile,
    // tools/ws_stamps.py).  vmcnt counts in order: with the loads older than the NPASS stores, vmcnt(NPASS) is exact.
    u32x4_t in_rs;
    {
        const uint64_t ip = (uint64_t)((const T*)a.in + (size_t)bs * a.H * a.W * C);
        in_rs[0] = __builtin_amdgcn_readfirstlane((unsigned)ip);
        in_rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(ip >> 32));
        in_rs[2] = in_bytes;
        in_rs[3] = 0x00020000u;
    }
    auto halo_issue = [&](int ty, int tx) __attribute__((always_inline)) {
        const int hy0 = ty * F::TH - 1, hx0 = tx * TW - 1;
        if (hy0 >= 0 && hy0 + F::IH <= a.H && hx0 >= 0 && hx0 + F::IW <= a.W) {  // (uniform) interior tile
            const int base = (hy0 * a.W + hx0) * C * ES;
            hok = hin;
#pragma unroll
            for (int i = 0; i < F::HPT; ++i) {
                u32x4_t v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(hrel[i]), "s"(in_rs), "s"(base) : "memory");
                hreg[i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        } else {
            hok = 0;
            unsigned off[F::HPT];
#pragma unroll
            for (int i = 0; i < F::HPT; ++i) {
                const int pix = i * F::PPP + hslot;
                const int gy = hy0 + pix / F::IW, gx = hx0 + pix % F::IW;
                const bool ok = hvalid && pix < F::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                off[i] = ok ? (unsigned)(((gy * a.W + gx) * C + hc * EPB) * ES) : kOOB;
                hok |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < F::HPT; ++i) {
                u32x4_t v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off[i]), "s"(in_rs) : "memory");
                hreg[i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    // wait until at most `younger` memory operations issued after the halo loads are outstanding (0 or NPASS), then tie the
    // registers to the wait so that no use moves above it
    auto halo_wait = [&](bool stores_behind) __attribute__((always_inline)) {
        if (stores_behind && F::OLPP == F::OPP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F::NPASS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < F::HPT; ++i) {
            u32x4_t v = {hreg[i].x, hreg[i].y, hreg[i].z, hreg[i].w};
            asm volatile("" : "+v"(v));
            hreg[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto halo_commit_xf = [&](auto xf_tag, char* hb) __attribute__((always_inline)) {
        if (!hvalid) return;
#pragma unroll
        for (int i = 0; i < F::HPT; ++i) {
            const int pix = i * F::PPP + hslot;
            if (pix < F::NPIX)
                *(uint4*)(hb + (pix / F::IW) * F::ROWSTRIDE + (pix % F::IW) * F::PSTRIDE + hc * 16) = piece_xf(xf_tag, hreg[i], (hok >> i) & 1u);
        }
    };
    auto halo_commit = [&](char* hb) __attribute__((always_inline)) {
        if (a.xf == XF_AFFINE_SILU) halo_commit_xf(std::integral_constant<int, XF_AFFINE_SILU>(), hb);
        else if (a.xf == XF_AFFINE) halo_commit_xf(std::integral_constant<int, XF_AFFINE>(), hb);
        else halo_commit_xf(std::integral_constant<int, XF_NONE>(), hb);
    };
    const int oc = ltid % F::OLPP, oslot = ltid / F::OLPP;
    const bool ovalid = !mrole && oc < F::OPP;
    const unsigned out_bytes = (unsigned)((size_t)a.H * a.W * C * ES);
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc((T*)a.out + (size_t)bs * a.H * a.W * C, out_bytes);
    // drain of one output tile (loaders): whole pixel rows leave with 16-byte stores; all LDS reads of the tile first
    auto drain = [&](int y0, int x0) __attribute__((always_inline)) {
        if (!ovalid) return;
        const int base = ((y0 * a.W + x0) * C) * ES;
        uint4 v[F::NPASS];
#pragma unroll
        for (int k = 0; k < F::NPASS; ++k) {
            const int p = oslot + k * F::STEP;
            v[k] = *(const uint4*)(otile + (F::P % F::STEP == 0 || p < F::P ? p : 0) * F::OSTRIDE + oc * 16);
        }
#pragma unroll
        for (int k = 0; k < F::NPASS; ++k) {
            const int p = oslot + k * F::STEP;
            const bool in = F::P % F::STEP == 0 || p < F::P;
            const u32x4_t t = {v[k].x, v[k].y, v[k].z, v[k].w};
            __builtin_amdgcn_raw_buffer_store_b128(t, out_rsrc, in ? (unsigned)((((p / TW) * a.W + p % TW) * C) * ES + oc * 16) : 0x80000000u, base, 0);
        }
    };

    // ---- MFMA state -----------------------------------------------------------------------------------------------------------------
    int pixoff[F::MT];
#pragma unroll
    for (int m = 0; m < F::MT; ++m) {
        const int p = (wm * F::MT + m) * 32 + l31;
        pixoff[m] = (p / TW) * F::ROWSTRIDE + (p % TW) * F::PSTRIDE + h * 16;
    }

    // ---- prologue: addend, GroupNorm input (all threads reduce the partials), first halo (loaders) ---------------------------------
    constexpr int AIT = (C + F::NTHREADS - 1) / F::NTHREADS;
    float add_b[AIT], add_c[AIT];
    {
        const float* pb = a.bias ? a.bias : (const float*)a.wf;
        const float* pc = a.chan_add ? a.chan_add + (size_t)bs * a.chan_add_stride : (const float*)a.wf;
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = tid + k * F::NTHREADS;
            const int ic = i < C ? i : C - 1;
            add_b[k] = pb[ic];
            add_c[k] = pc[ic];
        }
    }
    const bool gn_fused = a.xf != XF_NONE && a.gn.stats != nullptr;  // uniform
    GnInLoads gn_ld;
    if (gn_fused) gn_in_issue(a.gn, bs, tid, F::NTHREADS, gn_ld);
    int ty = t_begin / a.tiles_x, tx = t_begin % a.tiles_x;  // the tile the loaders fetch next
    if (a.xf != XF_NONE && hvalid) {
        const float* psc = gn_fused ? a.gn.gamma + hc * EPB : a.in_scale + (size_t)bs * C + hc * EPB;
        const float* psh = gn_fused ? (a.gn.beta ? a.gn.beta : a.gn.gamma) + hc * EPB : a.in_shift + (size_t)bs * C + hc * EPB;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            sc[j] = *(const f32x2_t*)(psc + 2 * j);
            sh[j] = *(const f32x2_t*)(psh + 2 * j);
        }
    }
    if (!mrole) halo_issue(ty, tx);
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int i = tid + k * F::NTHREADS;
        if (i < C) addv[i] = (a.bias ? add_b[k] : 0.f) + (a.chan_add ? add_c[k] : 0.f);
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
            gn_in_fold<EPB>(a.gn, gnscr, F::NWAVES, C, hc * EPB, gam, bet, fs, fh);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                sc[j].x = fs[2 * j]; sc[j].y = fs[2 * j + 1];
                sh[j].x = fh[2 * j]; sh[j].y = fh[2 * j + 1];
            }
        }
    }
    if (!mrole) {
        halo_wait(false);
        halo_commit(halo0);
        if (++tx == a.tiles_x) { tx = 0; ++ty; }
        if (t_begin + 1 < t_end) halo_issue(ty, tx);  // the second tile's halo is in flight from here on
    }
    __syncthreads();  // halo[0] and the addend are complete
    DDIMX_STAMP_DECL

    // Each role runs its own copy of the tile loop (disjoint register sets: the accumulators and weight fragments of the MFMA waves
    // do not add to the halo pieces and statistics of the loaders); both execute exactly two barriers per tile, then two more
    // around the statistics hand-over.
    if (mrole) {
        f32x2_t st_s[8], st_q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
#pragma unroll 1
        for (int t = t_begin; t < t_end; ++t) {
            char* const hb = halo0 + ((t - t_begin) & 1) * F::HALO_BYTES;
            DDIMX_STAMP_AT(0);
            // ---- MFMA loop of tile t (as conv3_wreg_kernel): A = aw[s % D], B = bq[s & 1] from the halo, one step ahead
            f32x16_t acc[F::MT];
#pragma unroll
            for (int m = 0; m < F::MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            uint4 bq[2][F::MT];
#pragma unroll
            for (int m = 0; m < F::MT; ++m) bq[0][m] = *(const uint4*)(hb + pixoff[m]);
#ifdef DDIMX_STAMP  // diagnostic build only: which part of the MFMA loop slows the loaders down (results are wrong with either bit)
            const bool dbg_no_mfma = a.dbg & 1, dbg_no_lds = a.dbg & 2;
#endif
#pragma unroll
            for (int s = 0; s < F::NSTEP; ++s) {
                if (s + 1 < F::NSTEP) {
                    const int tap = (s + 1) / F::KG, kg = (s + 1) % F::KG;
                    const int hoff = (tap / 3) * F::ROWSTRIDE + (tap % 3) * F::PSTRIDE + kg * 32;
#ifdef DDIMX_STAMP
                    if (!dbg_no_lds)
#endif
#pragma unroll
                    for (int m = 0; m < F::MT; ++m) bq[(s + 1) & 1][m] = *(const uint4*)(hb + pixoff[m] + hoff);
                }
                __builtin_amdgcn_sched_barrier(0);
                const uint4 af = aw[s % F::D];
#ifdef DDIMX_STAMP
                if (!dbg_no_mfma)
#endif
#pragma unroll
                for (int m = 0; m < F::MT; ++m) Mma<T>::run(af, bq[s & 1][m], acc[m]);
#ifdef DDIMX_STAMP
                if (!(a.dbg & 4))
#endif
                if constexpr (F::D < F::NSTEP) aw[s % F::D] = wfrag((s + F::D) % F::NSTEP);  // wraps into the next tile
                __builtin_amdgcn_sched_barrier(0);
            }
            DDIMX_STAMP_AT(1);
            __syncthreads();  // A(t)
            DDIMX_STAMP_AT(2);
            // ---- first epilogue: accumulators + addend -> SiLU -> bf16 -> output tile [pixel][cout]; statistics of the values as
            // stored, per channel, in this lane's registers (16 channels: quad q = channels 8 q + 4 h .. + 3)
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
                        const uint32_t u01 = Piece<__bf16>::pk(v01.x, v01.y), u23 = Piece<__bf16>::pk(v23.x, v23.y);
                        *(uint2*)(otile + p * F::OSTRIDE + cl * ES) = make_uint2(u01, u23);
                        const f32x2_t r01 = Pairs<T>::up(u01), r23 = Pairs<T>::up(u23);
                        st_s[2 * q] += r01; st_s[2 * q + 1] += r23;
                        st_q[2 * q] = fma2(r01, r01, st_q[2 * q]); st_q[2 * q + 1] = fma2(r23, r23, st_q[2 * q + 1]);
                    }
                }
            };
            if (a.act) epi1(std::integral_constant<int, 1>()); else epi1(std::integral_constant<int, 0>());
            DDIMX_STAMP_AT(3);
            __syncthreads();  // B(t)
            DDIMX_STAMP_AT(4);
        }
        // ---- statistics: one partial per workgroup.  Each lane's 16 channel sums are folded over the 32 lanes of its half (pixels),
        // the WM waves of a cout block meet in the freed addend / halo region (every MFMA loop and first epilogue ended before B
        // of the last tile; the loaders' last drain reads the output tile only)
        if (a.stats) {  // uniform
            float* const red = (float*)smem;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                st_s[j].x = group_sum<16>(st_s[j].x); st_s[j].y = group_sum<16>(st_s[j].y);
                st_q[j].x = group_sum<16>(st_q[j].x); st_q[j].y = group_sum<16>(st_q[j].y);
                st_s[j].x += __shfl_xor(st_s[j].x, 16, 64); st_s[j].y += __shfl_xor(st_s[j].y, 16, 64);
                st_q[j].x += __shfl_xor(st_q[j].x, 16, 64); st_q[j].y += __shfl_xor(st_q[j].y, 16, 64);
            }
            if (l31 == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // pair j = quad j / 2, channels 8 (j / 2) + 4 h + 2 (j % 2) .. + 1
                    const int ch = wn * 32 + (j >> 1) * 8 + h * 4 + (j & 1) * 2;
                    *(float4*)(red + (wm * C + ch) * 2) = make_float4(st_s[j].x, st_q[j].x, st_s[j].y, st_q[j].y);
                }
            }
            __syncthreads();
            if (a.stats_groups_c) {
                if (wave == 0)
                    gn_bins_store<F::WM>(red, C * 2, C, 0, a.stats_groups_c, a.stats + ((size_t)bs * a.wgs_per_sample + wg) * kGnSlab, lane);
            } else {
                for (int i = tid; i < C * 2; i += 64 * F::NMW) {
                    float tt = 0.f;
#pragma unroll
                    for (int w = 0; w < F::WM; ++w) tt += red[w * C * 2 + i];
                    a.stats[(((size_t)bs * a.wgs_per_sample + wg) * C) * 2 + i] = tt;
                }
            }
        }
        DDIMX_STAMP_FLUSH();
    } else {
        // (ty, tx) = tile t + 1, whose halo was requested one iteration ago (a whole tile period to land: with the request
        // placed in the same iteration as its use the loaders spent a memory round trip per tile waiting, 98 us per launch
        // at C = 32); (py, px) = the tile whose output is drained next, one behind the MFMA waves
        int py = t_begin / a.tiles_x, px = t_begin % a.tiles_x;
#pragma unroll 1
        for (int t = t_begin; t < t_end; ++t) {
            DDIMX_STAMP_AT(5);
            if (t > t_begin) {
                drain(py * F::TH, px * TW);  // the output tile of t - 1
                if (++px == a.tiles_x) { px = 0; ++py; }
            }
            DDIMX_STAMP_AT(6);
            if (t + 1 < t_end) {
                halo_wait(t > t_begin);  // (the drain above issued NPASS stores behind the loads, except in the first iteration)
                DDIMX_STAMP_AT(11);
                halo_commit(halo0 + (((t - t_begin) & 1) ^ 1) * F::HALO_BYTES);  // tile t + 1
                DDIMX_STAMP_AT(7);
                if (++tx == a.tiles_x) { tx = 0; ++ty; }
                if (t + 2 < t_end) halo_issue(ty, tx);  // tile t + 2
            }
            DDIMX_STAMP_AT(8);
            __syncthreads();  // A(t)
            DDIMX_STAMP_AT(9);
            __syncthreads();  // B(t)
            DDIMX_STAMP_AT(10);
        }
        drain(py * F::TH, px * TW);  // the last tile's output
        if (a.stats) __syncthreads();  // (uniform) the MFMA waves' statistics hand-over
        DDIMX_STAMP_FLUSH();
    }
}

template <class F>
hipError_t launch_ws_cfg(const WregArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_ws_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3_ws_kernel<F>, dim3(a.wgs_per_sample * a.B), dim3(F::NTHREADS), F::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// conv_inst_bf16_ws.hip
struct WsGeom { int th, tw, lds_bytes, nthreads, tiles_per_wg; };
hipError_t ws_geometry(int c, WsGeom* g);
hipError_t ws_launch(int c, const WregArgs& a, hipStream_t stream);

}  // namespace ddimx
