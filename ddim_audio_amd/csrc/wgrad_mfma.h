// Weight-gradient convolution on MFMA for the U-Net's NHWC activations (gfx950) -- the backward twin of
// conv_mfma.h for the parameters of Residual_Block / Downsample / Upsample (reference models/diffusion.py:12-78;
// the reference gets these gradients from autograd, runners/diffusion.py:150).
//
//   dW[co][tap][ci] = sum over (sample, pixel) of  du[pixel][co] * a[pixel * S + tap - 1][ci]
//
//   CONV3  S = 1, 3x3 taps : du = gradient of the conv output, a = the conv input (re-derived from the saved
//                            tensor with the same GroupNorm-affine / SiLU transform the forward prologue applied)
//   DOWN4  S = 2, 4x4 taps : Downsample (du = output gradient [H/2][W/2][CO], a = input [H][W][CI]) and, with the
//                            roles swapped, Upsample (du = the layer INPUT [H][W][C], a = the OUTPUT gradient
//                            [2H][2W][Cprev]): both yield dst[du-channel][a-channel][4][4], which is exactly the
//                            memory order of Conv2d.weight [Cout][Cin][4][4] and ConvTranspose2d.weight [Cin][Cout][4][4].
//
// The contraction runs over pixels, which are the SLOW index of NHWC, so both MFMA operands are "k-strided".
// bf16: tiles stay [pixel][32 channels] in LDS (64-byte rows, written with plain 16-byte stores) and are read with
// ds_read_b64_tr_b16, the CDNA4 transposing LDS read: 4 pixels x 16 channels per 16-lane group, conflict-free on
// 64-byte rows.  fp32 (parity mode): v_mfma_f32_32x32x2_f32 takes ONE k per lane, so plain ds_read_b32 of
// [pixel][32 floats] rows is already the operand layout.
//
// Work decomposition: grid.y enumerates (co-block group, ci-block group); grid.x splits the (sample, tile) range;
// a workgroup walks its tiles, every wave owning one 32x32 (co, ci) block pair for TPW taps (accumulators stay in
// registers for the whole walk: 9 or 8 tiles of 16 VGPRs), K-split over the waves when there is only one pair.
// Each workgroup writes ONE partial [tap][co][ci] slab; wgrad_reduce sums the slabs in a fixed order (deterministic)
// into the parameter-gradient layout.
#pragma once
#include <type_traits>

#include "conv_mfma.h"

namespace ddimx {

struct WgradArgs {
    const void* a;         // [B][Ha][Wa][CI]
    const void* du;        // [B][Hd][Wd][CO]
    const float* a_scale;  // [B][CI] (xf != XF_NONE)
    const float* a_shift;
    int xf;                // transform of `a` while it is staged (XF_* of conv_mfma.h)
    float* partial;        // [gridDim.x][NTAPS][CO][CI]
    int B, Ha, Wa, Hd, Wd;
    int tiles_x, tiles_y, total_tiles, tiles_per_wg;
};

template <typename T, int CI_, int CO_, int MODE_>
struct WgCfg {
    typedef T elem;
    static constexpr int CI = CI_, CO = CO_, MODE = MODE_;
    static constexpr int ES = sizeof(T), EPB = 16 / ES;
    static constexpr int NBO = CO / 32, NBI = CI / 32;
    static constexpr int NCO = NBO <= 3 ? NBO : 2;
    static constexpr int NCI = NCO == 3 ? 1 : (NBI % 2 == 0 ? 2 : 1);
    // taps are split over waves: DOWN4 halves (8 + 8); CONV3 with a single block pair thirds (3 + 3 + 3: 48 accumulator
    // registers per wave instead of 144, so several workgroups share a CU and overlap staging with MFMAs)
    static constexpr int TS = MODE == DOWN4 ? 2 : (NBO * NBI == 1 ? 3 : 1);
    static constexpr int NTAPS = MODE == DOWN4 ? 16 : 9;
    static constexpr int TAPW = MODE == DOWN4 ? 4 : 3;
    static constexpr int TPW = NTAPS / TS;                     // taps (accumulator tiles) per wave
    static constexpr int BASE = NCO * NCI * TS;
    static constexpr int KSUB = BASE >= 3 ? 1 : 4 / BASE;      // K-split over waves when there are few block pairs
    static constexpr int NW = BASE * KSUB, NTHREADS = 64 * NW;
    static constexpr int S = MODE == DOWN4 ? 2 : 1;
    static constexpr int TW = 16, TH = MODE == DOWN4 ? 4 : 8, P = TH * TW;
    static constexpr int IH = TH * S + 2, IW = TW * S + 2, NPIX = IH * IW;
    static constexpr int ROW = 32 * ES;                        // bytes per pixel per 32-channel block
    static constexpr int PPB = 32 / EPB;                       // 16-byte pieces per pixel per block
    static constexpr int SS_BYTES = NCI * 32 * 2 * 4;
    static constexpr int DU_BYTES = NCO * P * ROW, HALO_BYTES = NCI * NPIX * ROW;
    static constexpr int RED_BYTES = KSUB > 1 ? BASE * TPW * 4096 : 0;
    static constexpr int LDS_RAW = SS_BYTES + DU_BYTES + HALO_BYTES;
    static constexpr int LDS_BYTES = LDS_RAW > RED_BYTES ? LDS_RAW : RED_BYTES;
    static constexpr int GRID_Y = (NBO / NCO) * (NBI / NCI);
    // waves per SIMD the register allocator must leave room for
    static constexpr int MINW = 1;
    static_assert(CI % 32 == 0 && CO % 32 == 0, "channel counts must be multiples of 32");
    static_assert(NBO % NCO == 0 && NBI % NCI == 0, "block grouping");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(NTHREADS <= 1024, "");
};

template <class C>
__global__ void __launch_bounds__(C::NTHREADS, C::MINW) wgrad_mfma_kernel(const WgradArgs a) {
    typedef typename C::elem T;
    constexpr int ES = C::ES, EPB = C::EPB, ROW = C::ROW, PPB = C::PPB, S = C::S;
    constexpr int NP = EPB / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const ss = (float*)smem;                  // [NCI*32][2] scale, shift of the current sample
    char* const du_l = smem + C::SS_BYTES;           // [NCO][P][32]
    char* const ha_l = du_l + C::DU_BYTES;           // [NCI][NPIX][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave % C::KSUB;
    const int ts = (wave / C::KSUB) % C::TS;
    const int co_i = (wave / (C::KSUB * C::TS)) % C::NCO;
    const int ci_i = wave / (C::KSUB * C::TS * C::NCO);
    const int cob0 = (blockIdx.y % (C::NBO / C::NCO)) * C::NCO;  // first 32-channel block of this workgroup
    const int cib0 = (blockIdx.y / (C::NBO / C::NCO)) * C::NCI;

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = t_begin + a.tiles_per_wg < a.total_tiles ? t_begin + a.tiles_per_wg : a.total_tiles;
    const int tiles_s = a.tiles_x * a.tiles_y;

    f32x16_t acc[C::TPW];
#pragma unroll
    for (int t = 0; t < C::TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-lane operand addressing (tile independent)
    // bf16: transposing read -- lane 4q+p of a 16-lane group supplies (row q, columns 4p..4p+3) of a 4-pixel x
    // 16-channel block and receives channel (lane % 16) of the 4 pixels; groups 0/1 = channels 0-15/16-31 at
    // k 0-7, groups 2/3 = the same at k 8-15.  fp32: lane = (channel lane % 32, k lane / 32).
    const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int tr_k = 8 * (g16 >> 1) + q4;                 // + 4j for the second read
    const int tr_col = (16 * (g16 & 1) + 4 * p4) * 2;     // byte offset inside the 64-byte row
    const int f_m = lane & 31, f_k = lane >> 5;

    // ---- staging.  bf16: the global loads of tile t+1 are issued right after tile t has been committed to LDS, so they
    // are in flight while tile t is multiplied (register prefetch: NDU + NHA 16-byte pieces per thread).  fp32 (parity
    // mode, twice the pieces) loads and commits a tile in place.
    constexpr int NPC_DU = C::NCO * C::P * PPB, NPC_H = C::NCI * C::NPIX * PPB;
    constexpr int NDU = (NPC_DU + C::NTHREADS - 1) / C::NTHREADS, NHA = (NPC_H + C::NTHREADS - 1) / C::NTHREADS;
    // prefetch only where accumulators + prefetch registers leave two workgroups per CU (measured: it costs the 9-tap,
    // 144-accumulator configurations their second workgroup and more than it gains)
    constexpr bool PREF = ES == 2 && C::TPW * 16 + (NDU + NHA) * 4 <= 176;
    static_assert(NHA <= 32, "halo validity mask");
    uint4 pdu[PREF ? NDU : 1], pha[PREF ? NHA : 1];
    unsigned pok = 0;
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int b = t / tiles_s, tt = t % tiles_s;
        const int y0 = (tt / a.tiles_x) * C::TH, x0 = (tt % a.tiles_x) * C::TW;
        const __amdgpu_buffer_rsrc_t du_rsrc =
            make_rsrc((const T*)a.du + (size_t)b * a.Hd * a.Wd * C::CO, (unsigned)((size_t)a.Hd * a.Wd * C::CO * ES));
        const __amdgpu_buffer_rsrc_t a_rsrc =
            make_rsrc((const T*)a.a + (size_t)b * a.Ha * a.Wa * C::CI, (unsigned)((size_t)a.Ha * a.Wa * C::CI * ES));
#pragma unroll
        for (int i = 0; i < NDU; ++i) {  // du tile: [blk][pixel][32 channels], zeros outside the image
            const int idx = tid + i * C::NTHREADS;
            const int j = idx % PPB, pix = (idx / PPB) % C::P, blk = idx / (PPB * C::P);
            const int y = y0 + pix / C::TW, x = x0 + pix % C::TW;
            const bool ok = idx < NPC_DU && y < a.Hd && x < a.Wd;
            pdu[i] = buf_load16(du_rsrc, ok ? (unsigned)(((y * a.Wd + x) * C::CO + (cob0 + blk) * 32 + j * EPB) * ES) : kOOB);
        }
        pok = 0;
#pragma unroll
        for (int i = 0; i < NHA; ++i) {  // halo of `a`
            const int idx = tid + i * C::NTHREADS;
            const int j = idx % PPB, pix = (idx / PPB) % C::NPIX, blk = idx / (PPB * C::NPIX);
            const int gy = y0 * S - 1 + pix / C::IW, gx = x0 * S - 1 + pix % C::IW;
            const bool ok = idx < NPC_H && gy >= 0 && gy < a.Ha && gx >= 0 && gx < a.Wa;
            pha[i] = buf_load16(a_rsrc, ok ? (unsigned)(((gy * a.Wa + gx) * C::CI + (cib0 + blk) * 32 + j * EPB) * ES) : kOOB);
            pok |= ok ? 1u << i : 0u;
        }
        if constexpr (PREF) {  // keep the loads here: without this LLVM sinks them below the MFMA block, next to their use
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // registers -> LDS; the transform of `a` (padding is zero AFTER the transform) happens here
    auto commit = [&](auto xf_tag) __attribute__((always_inline)) {
        constexpr int XF = decltype(xf_tag)::value;
#pragma unroll
        for (int i = 0; i < NDU; ++i) {
            const int idx = tid + i * C::NTHREADS;
            if (idx < NPC_DU) *(uint4*)(du_l + (size_t)idx * 16) = pdu[i];  // idx order == LDS order
        }
#pragma unroll
        for (int i = 0; i < NHA; ++i) {
            const int idx = tid + i * C::NTHREADS;
            if (idx >= NPC_H) continue;
            uint4 w = pha[i];
            if (XF != XF_NONE) {
                const bool ok = (pok >> i) & 1u;
                const int j = idx % PPB, blk = idx / (PPB * C::NPIX);
                const float* sp = ss + (blk * 32 + j * EPB) * 2;
                f32x2_t f[NP];
                Pairs<T>::unpack(w, f);
#pragma unroll
                for (int e = 0; e < NP; ++e) {
                    const f32x2_t sc = {sp[4 * e + 0], sp[4 * e + 2]}, sh = {sp[4 * e + 1], sp[4 * e + 3]};
                    if (XF == XF_SILU_AFFINE) f[e] = silu2(f[e]);
                    f[e] = fma2(f[e], sc, sh);
                    if (XF == XF_AFFINE_SILU) f[e] = silu2(f[e]);
                }
                const uint4 tv = Pairs<T>::pack(f);
                w.x = ok ? tv.x : 0u; w.y = ok ? tv.y : 0u; w.z = ok ? tv.z : 0u; w.w = ok ? tv.w : 0u;
            }
            *(uint4*)(ha_l + (size_t)idx * 16) = w;
        }
    };

    // fp32: load + commit four pieces at a time (keeps the register footprint of the parity build small)
    auto stage_in_place = [&](auto xf_tag, int t) __attribute__((always_inline)) {
        constexpr int XF = decltype(xf_tag)::value;
        const int b = t / tiles_s, tt = t % tiles_s;
        const int y0 = (tt / a.tiles_x) * C::TH, x0 = (tt % a.tiles_x) * C::TW;
        const __amdgpu_buffer_rsrc_t du_rsrc =
            make_rsrc((const T*)a.du + (size_t)b * a.Hd * a.Wd * C::CO, (unsigned)((size_t)a.Hd * a.Wd * C::CO * ES));
        const __amdgpu_buffer_rsrc_t a_rsrc =
            make_rsrc((const T*)a.a + (size_t)b * a.Ha * a.Wa * C::CI, (unsigned)((size_t)a.Ha * a.Wa * C::CI * ES));
#pragma unroll 1
        for (int i0 = tid; i0 < NPC_DU; i0 += 4 * C::NTHREADS) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * C::NTHREADS;
                const int j = idx % PPB, pix = (idx / PPB) % C::P, blk = idx / (PPB * C::P);
                const int y = y0 + pix / C::TW, x = x0 + pix % C::TW;
                const bool ok = idx < NPC_DU && y < a.Hd && x < a.Wd;
                v[u] = buf_load16(du_rsrc, ok ? (unsigned)(((y * a.Wd + x) * C::CO + (cob0 + blk) * 32 + j * EPB) * ES) : kOOB);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * C::NTHREADS;
                if (idx < NPC_DU) *(uint4*)(du_l + (size_t)idx * 16) = v[u];
            }
        }
#pragma unroll 1
        for (int i0 = tid; i0 < NPC_H; i0 += 4 * C::NTHREADS) {
            uint4 v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * C::NTHREADS;
                const int j = idx % PPB, pix = (idx / PPB) % C::NPIX, blk = idx / (PPB * C::NPIX);
                const int gy = y0 * S - 1 + pix / C::IW, gx = x0 * S - 1 + pix % C::IW;
                ok[u] = idx < NPC_H && gy >= 0 && gy < a.Ha && gx >= 0 && gx < a.Wa;
                v[u] = buf_load16(a_rsrc, ok[u] ? (unsigned)(((gy * a.Wa + gx) * C::CI + (cib0 + blk) * 32 + j * EPB) * ES) : kOOB);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * C::NTHREADS;
                if (idx >= NPC_H) continue;
                uint4 w = v[u];
                if (XF != XF_NONE) {
                    const int j = idx % PPB, blk = idx / (PPB * C::NPIX);
                    const float* sp = ss + (blk * 32 + j * EPB) * 2;
                    f32x2_t f[NP];
                    Pairs<T>::unpack(w, f);
#pragma unroll
                    for (int e = 0; e < NP; ++e) {
                        const f32x2_t sc = {sp[4 * e + 0], sp[4 * e + 2]}, sh = {sp[4 * e + 1], sp[4 * e + 3]};
                        if (XF == XF_SILU_AFFINE) f[e] = silu2(f[e]);
                        f[e] = fma2(f[e], sc, sh);
                        if (XF == XF_AFFINE_SILU) f[e] = silu2(f[e]);
                    }
                    const uint4 tv = Pairs<T>::pack(f);
                    w.x = ok[u] ? tv.x : 0u; w.y = ok[u] ? tv.y : 0u; w.z = ok[u] ? tv.z : 0u; w.w = ok[u] ? tv.w : 0u;
                }
                *(uint4*)(ha_l + (size_t)idx * 16) = w;
            }
        }
    };

    int cur_b = -1;
    if constexpr (PREF) {
        if (t_begin < t_end) issue(t_begin);
    }
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int b = t / tiles_s;
        __syncthreads();  // every wave is done reading the previous tile
        if (b != cur_b && a.xf != XF_NONE) {  // uniform
            for (int i = tid; i < C::NCI * 32; i += C::NTHREADS) {
                ss[i * 2 + 0] = a.a_scale[(size_t)b * C::CI + cib0 * 32 + i];
                ss[i * 2 + 1] = a.a_shift[(size_t)b * C::CI + cib0 * 32 + i];
            }
            __syncthreads();
        }
        cur_b = b;
        auto stage = [&](auto xf_tag) __attribute__((always_inline)) {
            if constexpr (PREF) commit(xf_tag); else stage_in_place(xf_tag, t);
        };
        if (a.xf == XF_AFFINE_SILU) stage(std::integral_constant<int, XF_AFFINE_SILU>());
        else if (a.xf == XF_SILU_AFFINE) stage(std::integral_constant<int, XF_SILU_AFFINE>());
        else if (a.xf == XF_AFFINE) stage(std::integral_constant<int, XF_AFFINE>());
        else stage(std::integral_constant<int, XF_NONE>());
        __syncthreads();
        if constexpr (PREF) {
            if (t + 1 < t_end) issue(t + 1);
        }

        // ---- MFMA: one k-step = one tile row of 16 pixels ----
        const char* du_b = du_l + co_i * (C::P * ROW);
        const char* ha_b = ha_l + ci_i * (C::NPIX * ROW);
#pragma unroll 1
        for (int r = ks; r < C::TH; r += C::KSUB) {
            if constexpr (ES == 2) {
                typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
                typedef __attribute__((address_space(3))) bf16x4_t lds_bf4_t;
                auto tr_read = [&](const char* p) __attribute__((always_inline)) -> bf16x4_t {
                    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_t*)(__attribute__((address_space(3))) char*)const_cast<char*>(p));
                };
                const char* ap = du_b + (r * C::TW + tr_k) * ROW + tr_col;
                const bf16x4_t a0 = tr_read(ap), a1 = tr_read(ap + 4 * ROW);
                const bf16x8_t af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int tp = 0; tp < C::TPW; ++tp) {
                    const int tap = ts * C::TPW + tp;
                    const int dy = tap / C::TAPW, dx = tap % C::TAPW;
                    const char* bp = ha_b + ((r * S + dy) * C::IW + tr_k * S + dx) * ROW + tr_col;
                    const bf16x4_t b0 = tr_read(bp), b1 = tr_read(bp + 4 * S * ROW);
                    const bf16x8_t bfr = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tp], 0, 0, 0);
                }
            } else {
#pragma unroll 2
                for (int k2 = 0; k2 < 8; ++k2) {
                    const int k = 2 * k2 + f_k;
                    const float av = *(const float*)(du_b + (r * C::TW + k) * ROW + f_m * 4);
#pragma unroll
                    for (int tp = 0; tp < C::TPW; ++tp) {
                        const int tap = ts * C::TPW + tp;
                        const int dy = tap / C::TAPW, dx = tap % C::TAPW;
                        const float bv = *(const float*)(ha_b + ((r * S + dy) * C::IW + k * S + dx) * ROW + f_m * 4);
                        acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tp], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- K-split waves fold their accumulators into the ks == 0 wave, one round per split ----
    if constexpr (C::KSUB > 1) {
        float* const red = (float*)smem;
        const int base_i = wave / C::KSUB;
#pragma unroll 1
        for (int s = 1; s < C::KSUB; ++s) {
            __syncthreads();
            if (ks == s) {
#pragma unroll
                for (int t = 0; t < C::TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((base_i * C::TPW + t) * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int t = 0; t < C::TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += red[((base_i * C::TPW + t) * 16 + r) * 64 + lane];
            }
        }
    }
    if (ks == 0) {
        float* const dst = a.partial + (size_t)blockIdx.x * C::NTAPS * C::CO * C::CI;
#pragma unroll
        for (int tp = 0; tp < C::TPW; ++tp) {
            const int tap = ts * C::TPW + tp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (cob0 + co_i) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                const int ci = (cib0 + ci_i) * 32 + (lane & 31);
                dst[((size_t)tap * C::CO + co) * C::CI + ci] = acc[tp][r];
            }
        }
    }
}

template <class C>
hipError_t launch_wgrad_cfg(const WgradArgs& a, int nsplit, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_mfma_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(wgrad_mfma_kernel<C>, dim3(nsplit, C::GRID_Y), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return hipGetLastError();
}

struct WgradGeom {
    int th, tw, ntaps, grid_y;
};
// implemented in wgrad_inst_*.hip; ci = channels of `a`, co = channels of `du`
hipError_t wgrad_geometry(int dtype, int mode, int ci, int co, WgradGeom* g);
hipError_t wgrad_launch(int dtype, int mode, int ci, int co, const WgradArgs& a, int nsplit, hipStream_t s);
// dst[co][ci][tap] (+)= sum_s partial[s][tap][co][ci]   (fixed order; dst is the fp32 parameter-gradient tensor)
hipError_t wgrad_reduce_launch(const float* partial, int nsplit, int ntaps, int co, int ci, float* dst, hipStream_t s);

}  // namespace ddimx
