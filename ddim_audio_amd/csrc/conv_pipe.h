// Residual_Block 3x3 convolution, SOFTWARE-PIPELINED INSIDE EACH WAVE (inference walk, bf16, C = 32 / 64, whole tiles: gfx950).
//
// conv_mfma_kernel / conv3_wreg_kernel run a tile as a sequence of phases -- stage halo, multiply, first epilogue, second
// epilogue -- with workgroup barriers between them: on a SIMD the matrix time and the vector time ADD (MFMA busy 19-28 %, VALU
// 41-47 %, a wave issues 41 % of its life: profiles/r03/conv_pmc_final.txt), and moving phases to other waves does not help
// (conv_ws.h, round 3).  What does overlap is a wave's OWN vector instructions in the issue gaps of its OWN MFMAs
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost": an MFMA holds the vector issue for 8 of its 32 cycles).  So here
// every wave runs ONE straight-line instruction stream per tile in which each MFMA step carries a few vector instructions of
// two other pipeline stages:
//
//     MFMA   tile t, block m          reads halo buffer t & 1 (LDS), weights from registers
//     XF     tile t + 1               raw halo pieces (registers, loaded a tile earlier) -> GroupNorm affine (+ SiLU) -> bf16 ->
//                                     halo buffer (t + 1) & 1; the freed registers at once take the loads of tile t + 2
//     EPI    block m - 1              accumulators (+ bias / embedding, folded into the accumulator's initial value) -> SiLU ->
//                                     group statistics -> bf16 -> WAVE-LOCAL transposition through 2.5 KB of LDS -> whole 16-byte
//                                     pieces of NHWC pixel rows to global memory
//
// * Weights are RESIDENT IN REGISTERS: a wave owns one 32-cout block, its 9 * C / 16 MFMA A-fragments (72 registers at C = 32, 144
//   at C = 64; fragment order, conv_wreg.h) are loaded once per workgroup.  No weight traffic, no weight ring in the loop.
// * m-outer order: a block of 32 pixels x 32 couts is finished in 9 * C / 16 consecutive MFMAs, so only two accumulators live
//   (this block's and the previous one's, which the epilogue is still reading).
// * ONE workgroup barrier per tile (the halo double buffer changes hands); the epilogue needs none (a wave transposes only what
//   it produced), and the barrier is a bare s_barrier behind `s_waitcnt lgkmcnt(0)`: global loads and stores stay in flight
//   across it (a __syncthreads() would drain them -- a memory round trip per tile).
// * Everything is compile-time scheduled: the tile body is NS_TILE = MT * 9 * C / 16 steps, each {B-operand read two steps
//   ahead, one MFMA, its share of XF and EPI instructions} between scheduling barriers; scalar f32 arithmetic only (packed f32
//   instructions beside MFMAs are an anti-lever, same guide table).
// * One workgroup per CU (LDS: two halo buffers).  Statistics: per-lane (sum, sum of squares) of the four 4-channel quads of the
//   lane -- each quad lies inside one GroupNorm group for C >= 32 -- of the fp32 values BEFORE the bf16 rounding (the rounding
//   error of a group's mean is 2^-9 / sqrt(N) relative, N >= 10^5; saves an unpack per element), folded to the 8 groups at the
//   end: one 128-byte slab per workgroup (gn_fused.h).  The partition depends on the sample's size only.
#pragma once
#include "conv_wreg.h"

namespace ddimx {

template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

template <int C_, int TH_, int WM_, int XF_, int MINW_ = 1, int DBG_ = 0>
struct PipeCfg {
    static constexpr int C = C_, TH = TH_, TW = 32, WM = WM_, XF = XF_, MINW = MINW_;
    static constexpr int DBG = DBG_;  // diagnostic builds only: 1 = no MFMAs, 2 = no XF stage, 4 = no EPI stage (timing of the rest; results are garbage)
    static constexpr int WN = C / 32;                    // one 32-cout block per wave
    static constexpr int NBLK = C / 32;
    static constexpr int NWAVES = WM * WN, NTHREADS = 64 * NWAVES;
    static constexpr int MT = TH / WM;                   // 32-pixel blocks (= tile rows) per wave and tile
    static constexpr int KG = C / 16, NSTEP = 9 * KG;    // MFMA steps per block
    static constexpr int NS_TILE = MT * NSTEP;
    static constexpr int IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    static constexpr int PSTRIDE = C * 2 + 16, ROWSTRIDE = IW * PSTRIDE, HALO_BYTES = IH * ROWSTRIDE;
    static constexpr int LPP = C / 8;                    // 16-byte pieces (= threads) per halo pixel
    static constexpr int PPP = NTHREADS / LPP;           // halo pixels per pass
    static constexpr int HPT = (NPIX + PPP - 1) / PPP;   // pieces per thread and tile
    static constexpr int NXE = HPT * 8;                  // elements a thread transforms per tile
    static constexpr int TR_STRIDE = 80, TR_BYTES = 32 * TR_STRIDE;  // per-wave transposition buffer: 32 pixels x (64 B + pad)
    static constexpr int DUMP_BYTES = NTHREADS * 16;     // where the pieces beyond the halo go (one slot per thread)
    static constexpr int GN_BYTES = NWAVES * kGroups * 2 * 4;
    static constexpr int OFF_TR = 2 * HALO_BYTES, OFF_DUMP = OFF_TR + NWAVES * TR_BYTES, OFF_GN = OFF_DUMP + DUMP_BYTES;
    static constexpr int LDS_BYTES = OFF_GN + GN_BYTES + 256;
#ifndef DDIMX_PIPE_PD
#define DDIMX_PIPE_PD 2
#endif
    static constexpr int PD = DDIMX_PIPE_PD, NBQ = PD + 1;   // B operands are read PD steps ahead into a ring of NBQ
    // Work units of the two filler stages.  A wave issues one instruction per ~4.75 cycles of any kind (transcendentals 8.3); five
    // scalar vector instructions hide in the wave's own MFMA (32.6 cycles), an MFMA costs the stream ~9 (tools/dbg/mfma_valu_overlap.hip,
    // profiles/r04/mfma_valu_overlap_microbench.txt).  A transcendental result needs one independent instruction before its consumer
    // (else hipcc pads with s_nop): every unit is cut in two phases EP steps apart, phase b of one unit sharing its step with phase a
    // of the next, XF and EPI units sharing steps, so that the scheduler always has independent chains to interleave.
    //   XF unit u = (piece u / 4, word u % 4): two elements -> one packed word (a: unpack, affine, exp; b: rcp, product, pack; the
    //   piece's last unit also zeroes / writes / re-loads);   XPB units per block at block steps E0 + j * EP (+ EP)
    //   EPI unit e = (quad e / 2, half e % 2) of the PREVIOUS block: two accumulator elements -> one packed word (a: exp; b: rcp,
    //   product, statistics, pack; the quad's second unit writes 8 bytes to the transposition buffer) at block steps E0 + e * EP
    //   (+ EP); the read-back at ER, the two global stores at ES.
    static constexpr int NXU = 4 * HPT, XPB = (NXU + MT - 1) / MT;
    static constexpr int E0 = 2, EP = (NSTEP - 6) / (XPB > 8 ? XPB : 8) > 0 ? (NSTEP - 6) / (XPB > 8 ? XPB : 8) : 1;
    static constexpr int ER = E0 + 8 * EP + 2, ES = ER + 3;   // (phase b of a unit runs EP steps after its phase a: beside phase a of the next)
    static constexpr int xstep(int u) { return (u / XPB) * NSTEP + E0 + (u % XPB) * EP; }
    static_assert(C == 32 || C == 64, "widths whose weight fragments fit the register file next to everything else");
    static_assert(TH % WM == 0 && MT % 2 == 0, "blocks per wave must be even (two alternating accumulators across tiles)");
    static_assert(NTHREADS % LPP == 0 && NTHREADS <= 1024, "thread map");
    static_assert(HPT <= 16, "validity masks are 16 bits");
    static_assert(HALO_BYTES + 16 <= 65536, "LDS immediates");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(ER < ES && ES < NSTEP && E0 + XPB * EP < NSTEP, "filler schedule");
};

template <class F>
__global__ void __launch_bounds__(F::NTHREADS, F::MINW) conv3_pipe_kernel(const WregArgs a) {
    constexpr int C = F::C, XF = F::XF, HPT = F::HPT, NSTEP = F::NSTEP, MT = F::MT;
    constexpr float kNegLog2e = -1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const trb = smem + F::OFF_TR;
    char* const dump = smem + F::OFF_DUMP;
    float* const gnscr = (float*)(smem + F::OFF_GN);

    DDIMX_STAMP_ENTRY
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % F::WM, wn = wave / F::WM;
    const int l31 = lane & 31, h = lane >> 5;

    int lwg;  // XCD-aware order (as conv_mfma_kernel): neighbouring tile ranges share an L2
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
    const int W_ = a.W;

    // ---- weights: every fragment of this wave's cout block, once ---------------------------------------------------------------
    uint4 wfr[NSTEP];
    {
        const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(a.wf, (unsigned)(9 * C * C * 2));
        const unsigned wlane = (unsigned)((wn * 64 + lane) * 16);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane, s * (F::NBLK * 1024), 0);
            wfr[s] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
    // ---- accumulator initial value: bias + this sample's timestep-embedding slice of the lane's 16 couts ------------------------
    f32x16_t addvec;
    {
        const float* pb = a.bias ? a.bias + wn * 32 + h * 4 : (const float*)a.wf;
        const float* pc = a.chan_add ? a.chan_add + (size_t)bs * a.chan_add_stride + wn * 32 + h * 4 : (const float*)a.wf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 vb = *(const float4*)(pb + (a.bias ? q * 8 : 0));
            const float4 vc = *(const float4*)(pc + (a.chan_add ? q * 8 : 0));
            addvec[q * 4 + 0] = (a.bias ? vb.x : 0.f) + (a.chan_add ? vc.x : 0.f);
            addvec[q * 4 + 1] = (a.bias ? vb.y : 0.f) + (a.chan_add ? vc.y : 0.f);
            addvec[q * 4 + 2] = (a.bias ? vb.z : 0.f) + (a.chan_add ? vc.z : 0.f);
            addvec[q * 4 + 3] = (a.bias ? vb.w : 0.f) + (a.chan_add ? vc.w : 0.f);
        }
    }

    // ---- halo pieces of this thread (tile independent) -------------------------------------------------------------------------
    const int hc = tid % F::LPP, hslot = tid / F::LPP;
    unsigned hrel[HPT];   // global byte offset relative to the halo origin
    int hlds[HPT];        // LDS byte offset inside a halo buffer (pieces beyond the halo: this thread's dump slot)
    unsigned m_tb = 0, m_lr = 0;  // bit k: piece k lies in the top / left halo row / column; bit 16 + k: bottom / right
#pragma unroll
    for (int k = 0; k < HPT; ++k) {
        const int pix = k * F::PPP + hslot;
        const bool in = pix < F::NPIX;
        const int iy = pix / F::IW, ix = pix % F::IW;
        hrel[k] = in ? (unsigned)(((iy * W_ + ix) * C + hc * 8) * 2) : 0x80000000u;
        hlds[k] = in ? iy * F::ROWSTRIDE + ix * F::PSTRIDE + hc * 16 : 0;
        if (in && iy == 0) m_tb |= 1u << k;
        if (in && iy == F::IH - 1) m_tb |= 0x10000u << k;
        if (in && ix == 0) m_lr |= 1u << k;
        if (in && ix == F::IW - 1) m_lr |= 0x10000u << k;
    }
    // only a thread's LAST piece can lie beyond the halo; it is transformed like the others and written to the thread's dump slot
    const bool last_in = (HPT - 1) * F::PPP + hslot < F::NPIX;
    char* const my_dump = dump + tid * 16;
    const unsigned in_bytes = (unsigned)((size_t)a.H * W_ * C * 2);
    const char* const in_ptr = (const char*)a.in + (size_t)bs * in_bytes;
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc(in_ptr, in_bytes);
    const __amdgpu_buffer_rsrc_t null_rsrc = make_rsrc(in_ptr, 0u);
    const unsigned out_bytes = in_bytes;
    char* const out_ptr = (char*)a.out + (size_t)bs * out_bytes;
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(out_ptr, out_bytes);
    const __amdgpu_buffer_rsrc_t out_null = make_rsrc(out_ptr, 0u);

    // halo origin of tile (ty, tx) as a byte offset (mod 2^32: rows / columns outside the image wrap out of the buffer's range or
    // onto a neighbouring pixel -- either way the piece is zeroed by its validity bit after the transform)
    auto halo_base = [&](int ty, int tx) __attribute__((always_inline)) -> unsigned {
        return (unsigned)(((ty * F::TH - 1) * W_ + tx * F::TW - 1) * (C * 2));
    };
    auto halo_inval = [&](int ty, int tx) __attribute__((always_inline)) -> unsigned {
        const unsigned sel_tb = (ty == 0 ? 0xFFFFu : 0u) | (ty == a.tiles_y - 1 ? 0xFFFF0000u : 0u);
        const unsigned sel_lr = (tx == 0 ? 0xFFFFu : 0u) | (tx == a.tiles_x - 1 ? 0xFFFF0000u : 0u);
        const unsigned x = (m_tb & sel_tb) | (m_lr & sel_lr);
        return (x | (x >> 16)) & 0xFFFFu;
    };

    // ---- GroupNorm input: folded (scale, shift) of this thread's 8 channels -----------------------------------------------------------
    float sc[8], sh[8];
    {
        const bool gn_fused = a.gn.stats != nullptr;  // uniform
        GnInLoads gn_ld;
        if (gn_fused) gn_in_issue(a.gn, bs, tid, F::NTHREADS, gn_ld);
        const float* psc = gn_fused ? a.gn.gamma + hc * 8 : a.in_scale + (size_t)bs * C + hc * 8;
        const float* psh = gn_fused ? (a.gn.beta ? a.gn.beta : a.gn.gamma) + hc * 8 : a.in_shift + (size_t)bs * C + hc * 8;
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const float4 u = *(const float4*)(psc + j), v = *(const float4*)(psh + j);
            sc[j] = u.x; sc[j + 1] = u.y; sc[j + 2] = u.z; sc[j + 3] = u.w;
            sh[j] = v.x; sh[j + 1] = v.y; sh[j + 2] = v.z; sh[j + 3] = v.w;
        }
        if (gn_fused) {
            gn_in_reduce(a.gn, bs, tid, F::NTHREADS, gn_ld, gnscr);
            __syncthreads();
            float bet[8], fs[8], fh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bet[j] = a.gn.beta ? sh[j] : 0.f;
            gn_in_fold<8>(a.gn, gnscr, F::NWAVES, C, hc * 8, sc, bet, fs, fh);
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = fs[j]; sh[j] = fh[j]; }
        }
    }

    // ---- pipeline stage XF: one element / one piece --------------------------------------------------------------------------------
    uint4 hreg[HPT];
    unsigned xw[4];
    auto word_of = [&](auto k_, auto w_) __attribute__((always_inline)) -> unsigned {
        constexpr int k = decltype(k_)::value, w = decltype(w_)::value;
        return w == 0 ? hreg[k].x : (w == 1 ? hreg[k].y : (w == 2 ? hreg[k].z : hreg[k].w));
    };
    // two elements of piece k (word p: channels 2p, 2p + 1 of the piece) -> one packed word, in two phases (held in xy / xe[p & 1]).
    // SCALAR f32 arithmetic on purpose: v_pk_*_f32 does not run beside the wave's MFMA (tools/dbg/mfma_valu_overlap.hip: one
    // v_pk_fma_f32 in an MFMA gap costs the whole gap, +25 cycles, while up to five v_fma_f32 / v_cvt_pk / one v_exp are free)
    float xy[2][2], xe[2][2];
    auto xf_a = [&](auto k_, auto p_) __attribute__((always_inline)) {
        constexpr int p = decltype(p_)::value;
        const unsigned w0 = word_of(k_, p_);
        const float x0 = __uint_as_float(w0 << 16), x1 = __uint_as_float(w0 & 0xffff0000u);
        const float y0 = fmaf(x0, sc[2 * p], sh[2 * p]), y1 = fmaf(x1, sc[2 * p + 1], sh[2 * p + 1]);
        xy[p & 1][0] = y0; xy[p & 1][1] = y1;
        if constexpr (XF == XF_AFFINE_SILU) {
            xe[p & 1][0] = __builtin_amdgcn_exp2f(y0 * kNegLog2e);
            xe[p & 1][1] = __builtin_amdgcn_exp2f(y1 * kNegLog2e);
        }
    };
    auto xf_b = [&](auto k_, auto p_) __attribute__((always_inline)) {
        constexpr int p = decltype(p_)::value;
        float y0 = xy[p & 1][0], y1 = xy[p & 1][1];
        if constexpr (XF == XF_AFFINE_SILU) {
            y0 *= __builtin_amdgcn_rcpf(xe[p & 1][0] + 1.0f);
            y1 *= __builtin_amdgcn_rcpf(xe[p & 1][1] + 1.0f);
        }
        xw[p] = Piece<__bf16>::pk(y0, y1);
    };
    // zero what lies outside the image (AFTER the transform: the conv pads the normalised tensor), one LDS write; the piece's
    // registers then take the same piece of the tile after next
    auto xf_finish = [&](auto k_, char* hdst, unsigned inval, __amdgpu_buffer_rsrc_t nrsrc, unsigned nbase) __attribute__((always_inline)) {
        constexpr int k = decltype(k_)::value;
        const bool bad = (inval >> k) & 1u;
        uint4 v;
        v.x = bad ? 0u : xw[0];
        v.y = bad ? 0u : xw[1];
        v.z = bad ? 0u : xw[2];
        v.w = bad ? 0u : xw[3];
        char* dst = hdst + hlds[k];
        if constexpr (k == HPT - 1) dst = last_in ? dst : my_dump;
        *(uint4*)dst = v;
        hreg[k] = buf_load16(nrsrc, nbase + hrel[k]);
    };

    // ---- pipeline stage EPI: one accumulator element; the block's store ---------------------------------------------------------------
    float st_s[4], st_q[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { st_s[q] = 0.f; st_q[q] = 0.f; }
    char* const tr_w = trb + wave * F::TR_BYTES + l31 * F::TR_STRIDE + h * 8;           // + q * 16
    const char* const tr_r = trb + wave * F::TR_BYTES + (lane >> 2) * F::TR_STRIDE + (lane & 3) * 16;  // + k * 16 * TR_STRIDE
    const unsigned st_lane = (unsigned)((lane >> 2) * (C * 2) + wn * 64 + (lane & 3) * 16);       // + k * 16 * C * 2
    uint4 trv[2];
    unsigned ow[2];
    float ev[2][2], ee[2][2];
    auto epi_a = [&](const f32x16_t& acc, auto e_) __attribute__((always_inline)) {
        constexpr int e = decltype(e_)::value, q = e / 2, i2 = e % 2;
        const float v0 = acc[q * 4 + 2 * i2], v1 = acc[q * 4 + 2 * i2 + 1];
        ev[i2][0] = v0; ev[i2][1] = v1;
        ee[i2][0] = __builtin_amdgcn_exp2f(v0 * kNegLog2e);
        ee[i2][1] = __builtin_amdgcn_exp2f(v1 * kNegLog2e);
    };
    auto epi_b = [&](auto e_) __attribute__((always_inline)) {
        constexpr int e = decltype(e_)::value, q = e / 2, i2 = e % 2;
        const float v0 = ev[i2][0] * __builtin_amdgcn_rcpf(ee[i2][0] + 1.0f);
        const float v1 = ev[i2][1] * __builtin_amdgcn_rcpf(ee[i2][1] + 1.0f);
        st_s[q] += v0; st_q[q] = fmaf(v0, v0, st_q[q]);
        st_s[q] += v1; st_q[q] = fmaf(v1, v1, st_q[q]);
        ow[i2] = Piece<__bf16>::pk(v0, v1);
        if constexpr (i2 == 1) *(uint2*)(tr_w + q * 16) = make_uint2(ow[0], ow[1]);
    };
    auto epi_readback = [&]() __attribute__((always_inline)) {
        trv[0] = *(const uint4*)(tr_r);
        trv[1] = *(const uint4*)(tr_r + 16 * F::TR_STRIDE);
    };
    auto epi_store = [&](__amdgpu_buffer_rsrc_t orsrc, unsigned rowbase) __attribute__((always_inline)) {
        const u32x4_t t0 = {trv[0].x, trv[0].y, trv[0].z, trv[0].w}, t1 = {trv[1].x, trv[1].y, trv[1].z, trv[1].w};
        const unsigned rb = __builtin_amdgcn_readfirstlane(rowbase);  // (wave-uniform; a VGPR here costs a waterfall loop per store)
        __builtin_amdgcn_raw_buffer_store_b128(t0, orsrc, st_lane, rb, 0);
        __builtin_amdgcn_raw_buffer_store_b128(t1, orsrc, st_lane + 16 * C * 2, rb, 0);
    };

    // ---- prologue: tile 0 transformed outright, tile 1 requested ------------------------------------------------------------------------
    int ty = t_begin / a.tiles_x, tx = t_begin % a.tiles_x;   // current tile
    int ty2 = ty, tx2 = tx;                                   // the tile whose loads are issued next
    auto advance = [&](int& y, int& x) __attribute__((always_inline)) { if (++x == a.tiles_x) { x = 0; ++y; } };
    {
        const unsigned b0 = halo_base(ty2, tx2);
#pragma unroll
        for (int k = 0; k < HPT; ++k) hreg[k] = buf_load16(in_rsrc, b0 + hrel[k]);
    }
    unsigned inval_a = halo_inval(ty2, tx2);  // validity of the pieces now in hreg
    advance(ty2, tx2);
    {
        const bool has1 = t_begin + 1 < t_end;
        const __amdgpu_buffer_rsrc_t r1 = has1 ? in_rsrc : null_rsrc;
        const unsigned b1 = halo_base(ty2, tx2);
        const unsigned inval_b = halo_inval(ty2, tx2);
        static_for<0, HPT>([&](auto k_) {
            static_for<0, 4>([&](auto p_) { xf_a(k_, p_); xf_b(k_, p_); });
            xf_finish(k_, halo, inval_a, r1, b1);
        });
        inval_a = inval_b;
        advance(ty2, tx2);
    }
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }  // acc1 = the "previous block" of the first tile: SiLU(0) = 0 adds nothing
    unsigned prev_rowbase = 0;                     // output row of the previous tile's last block (first tile: stores dropped)
    __amdgpu_buffer_rsrc_t prev_rsrc = out_null;
    const int lanebase = wm * MT * F::ROWSTRIDE + l31 * F::PSTRIDE + h * 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DDIMX_STAMP_DECL

    // ---- the tile body -------------------------------------------------------------------------------------------------------------------
    auto body = [&](auto xfn_, int cur, int t) __attribute__((always_inline)) {
        constexpr bool XFN = decltype(xfn_)::value;  // a next tile exists: stage XF runs
        // the two copies of the body share the previous block's epilogue arithmetic: left alone, LLVM hoists it (16 v_mul + 16 v_exp) in
        // front of the branch that picks the copy -- out of the MFMA gaps it was scheduled into.  An empty asm that "redefines" the
        // accumulator, with a different text per copy, keeps each copy's arithmetic where it was written.
        if constexpr (XFN) asm volatile("; tile body, a next tile exists" : "+v"(acc1));
        else asm volatile("; tile body, last tile of the workgroup" : "+v"(acc1));
        const char* const hsrc = halo + cur * F::HALO_BYTES + lanebase;
        char* const hdst = halo + (cur ^ 1) * F::HALO_BYTES;
        const bool has2 = t + 2 < t_end;
        const __amdgpu_buffer_rsrc_t r2 = has2 ? in_rsrc : null_rsrc;
        const unsigned b2 = halo_base(ty2, tx2);
        const unsigned inval_b = halo_inval(ty2, tx2);
        const unsigned rowbase = (unsigned)(((ty * F::TH + wm * MT) * W_ + tx * F::TW) * (C * 2));  // this wave's first output row
        uint4 bq[F::NBQ];
        auto lds_b = [&](auto g_) __attribute__((always_inline)) -> uint4 {
            constexpr int G = decltype(g_)::value, m = G / NSTEP, s = G % NSTEP, tap = s / F::KG, kg = s % F::KG;
            constexpr int off = (m + tap / 3) * F::ROWSTRIDE + (tap % 3) * F::PSTRIDE + kg * 32;
            return *(const uint4*)(hsrc + off);
        };
        static_for<0, F::PD>([&](auto g_) { bq[decltype(g_)::value % F::NBQ] = lds_b(g_); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, F::NS_TILE>([&](auto g_) {
            constexpr int G = decltype(g_)::value, m = G / NSTEP, s = G % NSTEP;
            if constexpr (G + F::PD < F::NS_TILE) bq[(G + F::PD) % F::NBQ] = lds_b(std::integral_constant<int, G + F::PD>());
            {
                f32x16_t& acc = (m & 1) ? acc1 : acc0;
                const bf16x8_t av = __builtin_bit_cast(bf16x8_t, wfr[s]), bv = __builtin_bit_cast(bf16x8_t, bq[G % F::NBQ]);
                if constexpr (F::DBG & 1) { if constexpr (s == 0) acc = addvec; asm volatile("" :: "v"(av), "v"(bv)); }
                else if constexpr (s == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, addvec, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
            }
            // XF: units of the next tile's halo (phase b of the previous unit first: its inputs are a step old)
            if constexpr (XFN && !(F::DBG & 2)) {
                static_for<0, F::NXU>([&](auto u_) {
                    constexpr int u = decltype(u_)::value;
                    if constexpr (F::xstep(u) + F::EP == G) {
                        xf_b(std::integral_constant<int, u / 4>(), std::integral_constant<int, u % 4>());
                        if constexpr (u % 4 == 3) xf_finish(std::integral_constant<int, u / 4>(), hdst, inval_a, r2, b2);
                    }
                });
                static_for<0, F::NXU>([&](auto u_) {
                    constexpr int u = decltype(u_)::value;
                    if constexpr (F::xstep(u) == G) xf_a(std::integral_constant<int, u / 4>(), std::integral_constant<int, u % 4>());
                });
            }
            // EPI: the previous block (block MT - 1 of the previous tile for m = 0)
            if constexpr (!(F::DBG & 4)) {
                const f32x16_t& pacc = (m & 1) ? acc0 : acc1;  // (m - 1) & 1, and (MT - 1) & 1 = 1 for m = 0
                static_for<0, 8>([&](auto e_) {
                    if constexpr (s == F::E0 + (decltype(e_)::value + 1) * F::EP) epi_b(e_);
                });
                static_for<0, 8>([&](auto e_) {
                    if constexpr (s == F::E0 + decltype(e_)::value * F::EP) epi_a(pacc, e_);
                });
                if constexpr (s == F::ER) epi_readback();
                if constexpr (s == F::ES) {
                    if constexpr (m == 0) epi_store(prev_rsrc, prev_rowbase);
                    else epi_store(out_rsrc, rowbase + (unsigned)((m - 1) * W_ * (C * 2)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        inval_a = inval_b;
        prev_rowbase = rowbase + (unsigned)((MT - 1) * W_ * (C * 2));
        prev_rsrc = out_rsrc;
    };

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) body(std::true_type(), cur, t);
        else body(std::false_type(), cur, t);
        advance(ty, tx);
        advance(ty2, tx2);
        DDIMX_STAMP_AT(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's halo reads have returned, its halo writes have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DDIMX_STAMP_AT(1);
    }
    // ---- drain: the last block's epilogue ----------------------------------------------------------------------------------------------
    static_for<0, 8>([&](auto e_) { epi_a(acc1, e_); epi_b(e_); });
    epi_readback();
    epi_store(prev_rsrc, prev_rowbase);
    DDIMX_STAMP_AT(2);

    // ---- statistics: one group-format slab per workgroup (gn_fused.h) --------------------------------------------------------------------
    if (a.stats) {  // uniform
        float* const red = (float*)smem;  // [wave][h][q][2]; overlays the halo (every wave is past its last LDS read: barrier above)
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                st_s[q] += __shfl_xor(st_s[q], o, 64);
                st_q[q] += __shfl_xor(st_q[q], o, 64);
            }
        }
        if (l31 == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(float2*)(red + ((wave * 2 + h) * 4 + q) * 2) = make_float2(st_s[q], st_q[q]);
        }
        __syncthreads();
        if (tid < kGnSlab) {
            // bin = group * 2 + (0: sum, 1: sum of squares).  Quad (wn, q, h) covers channels wn * 32 + q * 8 + h * 4 .. + 3.
            float tot = 0.f;
            if (tid < 2 * kGroups) {
                const int g = tid >> 1, sq = tid & 1;
                constexpr int GS = C / kGroups;
#pragma unroll
                for (int w = 0; w < F::NWAVES; ++w)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ch = (w / F::WM) * 32 + q * 8 + hh * 4;
                            const float v = red[((w * 2 + hh) * 4 + q) * 2 + sq];
                            tot += (ch / GS == g) ? v : 0.f;
                        }
            }
            a.stats[((size_t)bs * a.wgs_per_sample + wg) * kGnSlab + tid] = tot;
        }
    }
    DDIMX_STAMP_AT(3);
    DDIMX_STAMP_FLUSH();
}

template <class F>
hipError_t launch_pipe_cfg(const WregArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_pipe_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3_pipe_kernel<F>, dim3(a.wgs_per_sample * a.B), dim3(F::NTHREADS), F::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// conv_inst_bf16_pipe.hip
struct PipeGeom { int th, tw, lds_bytes, nthreads; };
hipError_t pipe_geometry(int c, PipeGeom* g);
hipError_t pipe_launch(int c, int xf, const WregArgs& a, hipStream_t stream);

}  // namespace ddimx
