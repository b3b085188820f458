// Fused implicit-GEMM convolution on MFMA for the U-Net's NHWC activations (gfx950).
//
// One kernel template covers the three convolution shapes of the reference U-Net:
//   CONV3  3x3 stride 1 pad 1      -- the two convs of Residual_Block (models/diffusion.py:28-40)
//   DOWN4  4x4 stride 2 pad 1      -- Downsample (models/diffusion.py:70-78)
//   UP4    ConvTranspose2d 4x4 stride 2 pad 1 as sub-pixel convs -- Upsample (models/diffusion.py:59-67)
// and fuses what surrounds them in the block (models/diffusion.py:42-56):
//   prologue : GroupNorm affine (+SiLU) applied while the input halo tile is staged into LDS
//              (the zero padding is applied AFTER the transform, like conv2d on the normalised tensor)
//   epilogue : + bias, + per-sample channel vector (timestep embedding), SiLU, + skip tensor,
//              per-channel sum / sum-of-squares partials for the NEXT GroupNorm (deterministic slabs).
//
// Work decomposition: a workgroup owns TH x TW output pixels (one sample) x NB output channels.
// The transformed input halo tile stays resident in LDS for all taps; weights stream through LDS in
// chunks (TPC taps x KC input channels), register-staged one chunk ahead of the MFMAs.
// MFMA orientation: A = weights (rows = cout), B = pixels (cols = pixel), so the accumulator holds,
// per lane, 4 consecutive couts of one pixel per register quad -> packed 8/16-byte LDS writes in the
// epilogue, then fully coalesced 16-byte global stores of whole NHWC pixel rows.
//
// T = __bf16 : v_mfma_f32_32x32x16_bf16 (fp32 accumulate);  T = float : v_mfma_f32_32x32x2_f32
// (exact fp32 FMA chain; the parity mode).
#pragma once
#include "common.h"

namespace ddimx {

enum ConvMode { CONV3 = 0, DOWN4 = 1, UP4 = 2 };
enum { XF_NONE = 0, XF_AFFINE = 1, XF_AFFINE_SILU = 2 };

struct ConvArgs {
    const void* in;         // [B][Hin][Win][CIN]
    const void* w;          // [classes][taps][NOUT][CIN]   (classes = 2 for UP4, else 1)
    const float* bias;      // [NOUT] or null
    const float* chan_add;  // per-sample per-cout vector (base + b*chan_add_stride), or null
    const float* in_scale;  // [B][CIN] folded GroupNorm scale (xf != XF_NONE)
    const float* in_shift;  // [B][CIN]
    const void* skip;       // same shape as out, added in the epilogue, or null
    void* out;              // [B][Hout][Wout][COUT]
    float* stats;           // [B][nparts][NOUT][2] partial (sum, sumsq) or null
    int chan_add_stride;
    int xf;                 // XF_*
    int act;                // 0 none, 1 SiLU
    int B, Hin, Win;
    int Hv, Wv;             // virtual output grid (UP4: = input grid; else = output grid)
    int tiles_x, tiles_y;
};

template <typename T, int CIN_, int NOUT_, int NB_, int MODE_, int TH_, int TW_, int WM_, int WN_, int KC_, int TPC_>
struct ConvCfg {
    typedef T elem;
    static constexpr int CIN = CIN_, NOUT = NOUT_, NB = NB_, MODE = MODE_, TH = TH_, TW = TW_, WM = WM_, WN = WN_,
                         KC = KC_, TPC = TPC_;
    static constexpr int ES = sizeof(T);
    static constexpr int EPB = 16 / ES;
    static constexpr int NWAVES = WM * WN, NTHREADS = 64 * NWAVES;
    static constexpr int P = TH * TW;
    static constexpr int MT = P / (32 * WM);
    static constexpr int NT = NB / (32 * WN);
    static constexpr int NTAPS = MODE == CONV3 ? 9 : (MODE == DOWN4 ? 16 : 6);
    static constexpr int TAPW = MODE == DOWN4 ? 4 : 3;  // taps per kernel row
    static constexpr int SXY = MODE == DOWN4 ? 2 : 1;
    static constexpr int IH = MODE == DOWN4 ? 2 * TH + 2 : TH + 2;
    static constexpr int IW = MODE == DOWN4 ? 2 * TW + 2 : TW + 2;
    static constexpr int PSTRIDE = CIN * ES + 16;  // odd number of 16-B slots -> conflict-free b128 reads
    static constexpr int ROWRAW = IW * PSTRIDE;
    static constexpr int ROWRES = TW == 8 ? 128 : 0;  // wanted (row stride mod 256 B) when a 32-lane block spans rows
    static constexpr int ROWSTRIDE =
        (TW >= 32 || SXY == 2) ? ROWRAW : ROWRAW + ((ROWRES - ROWRAW % 256) + 256) % 256;
    static constexpr int HALO_BYTES = IH * ROWSTRIDE;
    static constexpr int WROW = KC * ES + 16;
    static constexpr int WCHUNK_BYTES = TPC * NB * WROW;
    static constexpr int KSPLIT = CIN / KC;
    static constexpr int NCHUNKS = (NTAPS / TPC) * KSPLIT;
    static constexpr int NWBUF = NCHUNKS > 1 ? 2 : 1;
    static constexpr int OSTRIDE = NB * ES + 16;
    static constexpr int OUT_BYTES = P * OSTRIDE;
    static constexpr int MAIN_BYTES = HALO_BYTES + NWBUF * WCHUNK_BYTES;
    static constexpr int RED_BYTES = NWAVES * NB * 2 * 4;  // per-wave per-channel (sum, sumsq)
    static constexpr int LDS_BYTES = (MAIN_BYTES > OUT_BYTES + RED_BYTES ? MAIN_BYTES : OUT_BYTES + RED_BYTES);
    static constexpr int KG = KC * ES / 32;  // 32-byte k-groups per tap per chunk
    static constexpr int WPIECES = TPC * NB * (KC / EPB);
    static constexpr int WPT = (WPIECES + NTHREADS - 1) / NTHREADS;  // weight pieces per thread per chunk
    static constexpr int CPP = CIN / EPB;       // 16-B pieces per input pixel
    static constexpr int LPP = next_pow2(CPP);  // lanes cooperating on one input pixel
    static constexpr int OPP = NB / EPB;        // 16-B pieces per output pixel (this WG's channels)
    static constexpr int OLPP = next_pow2(OPP);

    static_assert(P % (32 * WM) == 0 && MT >= 1, "pixel tile must split into 32-pixel MFMA blocks");
    static_assert(NB % (32 * WN) == 0 && NT >= 1, "cout block must split into 32-row MFMA blocks");
    static_assert(NOUT % NB == 0, "NB must divide NOUT");
    static_assert(TW == 8 || TW == 16 || TW == 32 || TW == 64, "TW must be 8/16/32/64");
    static_assert(CIN % KC == 0 && (KC * ES) % 32 == 0, "KC must divide CIN and cover whole k-groups");
    static_assert(NTAPS % TPC == 0, "TPC must divide the tap count");
    static_assert(TPC == 1 || KC == CIN, "multi-tap chunks need the full channel range");
    static_assert((PSTRIDE / 16) % 2 == 1 && (WROW / 16) % 2 == 1, "strides must be odd in 16-B slots");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget (160 KiB per CU)");
    static_assert(LPP <= NTHREADS && OLPP <= 64, "piece lanes");
    static_assert(NTHREADS % LPP == 0 && NTHREADS % OLPP == 0, "");
};

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                     c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        // 4 k-pairs per 16-byte piece: lane half h holds channel 8g+4h+i for step i, in both operands
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};


template <class C>
__device__ __forceinline__ void conv_w_load(uint4 (&wreg)[C::WPT], const typename C::elem* wbase, int cout0, int tid,
                                            int ch) {
    const int tap0 = (ch / C::KSPLIT) * C::TPC;
    const int kc0 = (ch % C::KSPLIT) * C::KC;
    constexpr int PPR = C::KC / C::EPB;  // 16-B pieces per weight row
#pragma unroll
    for (int i = 0; i < C::WPT; ++i) {
        const int pc = tid + i * C::NTHREADS;
        wreg[i] = make_uint4(0, 0, 0, 0);
        if (pc < C::WPIECES) {
            const int j = pc % PPR, row = (pc / PPR) % C::NB, tp = pc / (PPR * C::NB);
            wreg[i] = *(const uint4*)(wbase + ((size_t)(tap0 + tp) * C::NOUT + cout0 + row) * C::CIN + kc0 + j * C::EPB);
        }
    }
}
template <class C>
__device__ __forceinline__ void conv_w_store(const uint4 (&wreg)[C::WPT], char* dst, int tid) {
    constexpr int PPR = C::KC / C::EPB;
#pragma unroll
    for (int i = 0; i < C::WPT; ++i) {
        const int pc = tid + i * C::NTHREADS;
        if (pc < C::WPIECES) {
            const int j = pc % PPR, row = (pc / PPR) % C::NB, tp = pc / (PPR * C::NB);
            *(uint4*)(dst + (tp * C::NB + row) * C::WROW + j * 16) = wreg[i];
        }
    }
}

template <class C>
__global__ void __launch_bounds__(C::NTHREADS) conv_mfma_kernel(const ConvArgs a) {
    typedef typename C::elem T;
    constexpr int ES = C::ES, EPB = C::EPB, CIN = C::CIN, NB = C::NB, NOUT = C::NOUT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wbuf = smem + C::HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % C::WM, wn = wave / C::WM;
    const int l31 = lane & 31, h = lane >> 5;

    const int tx = blockIdx.x % a.tiles_x;
    const int ty = (blockIdx.x / a.tiles_x) % a.tiles_y;
    const int bs = blockIdx.x / (a.tiles_x * a.tiles_y);
    const int cout0 = blockIdx.y * NB;
    const int cls = blockIdx.z;  // UP4 row-parity class; 0 otherwise
    const int y0 = ty * C::TH, x0 = tx * C::TW;
    const int hy0 = y0 * C::SXY - 1, hx0 = x0 * C::SXY - 1;

    const T* const wbase = (const T*)a.w + (size_t)cls * C::NTAPS * NOUT * CIN;

    // ---- weight chunk staging (global -> registers -> LDS), one chunk ahead of the MFMAs ----------
    uint4 wreg[C::WPT];
#define W_LOAD(ch) conv_w_load<C>(wreg, wbase, cout0, tid, (ch))
#define W_STORE(buf) conv_w_store<C>(wreg, wbuf + (buf) * C::WCHUNK_BYTES, tid)
    W_LOAD(0);

    // ---- stage the input halo tile, applying the GroupNorm affine (+SiLU) on the way ---------------
    {
        constexpr int LPP = C::LPP, PPP = C::NTHREADS / LPP, NPIX = C::IH * C::IW;
        const int c = tid % LPP, pslot = tid / LPP;
        const bool cvalid = c < C::CPP;
        float sc[EPB], sh[EPB];
        if (a.xf != XF_NONE && cvalid) {
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                sc[j] = a.in_scale[(size_t)bs * CIN + c * EPB + j];
                sh[j] = a.in_shift[(size_t)bs * CIN + c * EPB + j];
            }
        }
        const T* const inb = (const T*)a.in + (size_t)bs * a.Hin * a.Win * CIN + c * EPB;
        constexpr int UNR = 4;
        for (int base = 0; base < NPIX; base += PPP * UNR) {
            uint4 v[UNR];
            int off[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int pix = base + u * PPP + pslot;
                const int iy = pix / C::IW, ix = pix % C::IW;
                const int gy = hy0 + iy, gx = hx0 + ix;
                const bool inr = cvalid && pix < NPIX;
                ok[u] = inr && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
                off[u] = inr ? iy * C::ROWSTRIDE + ix * C::PSTRIDE + c * 16 : -1;
                v[u] = make_uint4(0, 0, 0, 0);
                if (ok[u]) v[u] = *(const uint4*)(inb + ((size_t)gy * a.Win + gx) * CIN);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (off[u] < 0) continue;
                if (ok[u] && a.xf != XF_NONE) {
                    float f[EPB];
                    Piece<T>::unpack(v[u], f);
#pragma unroll
                    for (int j = 0; j < EPB; ++j) {
                        float t = fmaf(f[j], sc[j], sh[j]);
                        f[j] = (a.xf == XF_AFFINE_SILU) ? silu_f(t) : t;
                    }
                    v[u] = Piece<T>::pack(f);
                }
                *(uint4*)(halo + off[u]) = v[u];
            }
        }
    }
    W_STORE(0);
    __syncthreads();

    // ---- main loop ---------------------------------------------------------------------------------
    f32x16_t acc[C::NT][C::MT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n)
#pragma unroll
        for (int m = 0; m < C::MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

    int pixoff[C::MT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
        const int p = (wm * C::MT + m) * 32 + l31;
        const int py = p / C::TW, px = p % C::TW;
        pixoff[m] = py * C::SXY * C::ROWSTRIDE + px * C::SXY * C::PSTRIDE + h * 16;
    }
    const int woff = (wn * C::NT * 32 + l31) * C::WROW + h * 16;

    for (int ch = 0; ch < C::NCHUNKS; ++ch) {
        if (ch + 1 < C::NCHUNKS) W_LOAD(ch + 1);
        const char* wb = wbuf + (ch & 1) * C::WCHUNK_BYTES + woff;
        const int tap0 = (ch / C::KSPLIT) * C::TPC;
        const int kc0 = (ch % C::KSPLIT) * C::KC;
#pragma unroll
        for (int tp = 0; tp < C::TPC; ++tp) {
            const int tap = tap0 + tp;
            const int dy = (C::MODE == UP4 ? cls : 0) + tap / C::TAPW, dx = tap % C::TAPW;
            const int hoff = dy * C::ROWSTRIDE + dx * C::PSTRIDE + kc0 * ES;
#pragma unroll
            for (int kg = 0; kg < C::KG; ++kg) {
                uint4 bf[C::MT], af[C::NT];
#pragma unroll
                for (int m = 0; m < C::MT; ++m) bf[m] = *(const uint4*)(halo + pixoff[m] + hoff + kg * 32);
#pragma unroll
                for (int n = 0; n < C::NT; ++n)
                    af[n] = *(const uint4*)(wb + (tp * NB + n * 32) * C::WROW + kg * 32);
#pragma unroll
                for (int n = 0; n < C::NT; ++n)
#pragma unroll
                    for (int m = 0; m < C::MT; ++m) Mma<T>::run(af[n], bf[m], acc[n][m]);
            }
        }
        if (ch + 1 < C::NCHUNKS) W_STORE((ch + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue 1: accumulators -> (+bias, +chan_add, act) -> LDS out tile [pixel][cout] ---------
    char* const otile = smem;  // overlays halo/weights: every wave is past the final barrier
    float* const red = (float*)(smem + C::OUT_BYTES);
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = (wn * C::NT + n) * 32 + q * 8 + h * 4;  // local cout of this register quad
            float add[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = 0.f;
                if (a.bias) v += a.bias[cout0 + cl + i];
                if (a.chan_add) v += a.chan_add[(size_t)bs * a.chan_add_stride + cout0 + cl + i];
                add[i] = v;
            }
#pragma unroll
            for (int m = 0; m < C::MT; ++m) {
                const int p = (wm * C::MT + m) * 32 + l31;
                float f[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[n][m][q * 4 + i] + add[i];
                    f[i] = a.act ? silu_f(v) : v;
                }
                char* dst = otile + p * C::OSTRIDE + cl * ES;
                if constexpr (ES == 4) {
                    *(uint4*)dst = Piece<float>::pack(f);
                } else {
                    *(uint2*)dst = make_uint2(Piece<__bf16>::pk(f[0], f[1]), Piece<__bf16>::pk(f[2], f[3]));
                }
            }
        }
    }
    __syncthreads();

    // ---- epilogue 2: coalesced stores of whole pixel rows, + skip, per-channel stats ----------------
    {
        constexpr int OLPP = C::OLPP, PPP = C::NTHREADS / OLPP;
        const int c = tid % OLPP, pslot = tid / OLPP;
        const bool cvalid = c < C::OPP;
        float s[EPB], q2[EPB];
#pragma unroll
        for (int j = 0; j < EPB; ++j) s[j] = q2[j] = 0.f;
        const int Hout = (C::MODE == UP4) ? 2 * a.Hv : a.Hv;
        const size_t rowlen = (size_t)a.Wv * NOUT;  // elements per output row (UP4: Wv*2*Cprev = Wout*Cprev)
        for (int p = pslot; p < C::P; p += PPP) {
            const int py = p / C::TW, px = p % C::TW;
            const int vy = y0 + py, vx = x0 + px;
            if (!cvalid || vy >= a.Hv || vx >= a.Wv) continue;
            const int oy = (C::MODE == UP4) ? 2 * vy + cls : vy;
            const size_t g = ((size_t)bs * Hout + oy) * rowlen + (size_t)vx * NOUT + cout0 + c * EPB;
            uint4 v = *(const uint4*)(otile + p * C::OSTRIDE + c * 16);
            float f[EPB];
            Piece<T>::unpack(v, f);
            if (a.skip) {
                float k[EPB];
                Piece<T>::unpack(*(const uint4*)((const T*)a.skip + g), k);
#pragma unroll
                for (int j = 0; j < EPB; ++j) f[j] += k[j];
                v = Piece<T>::pack(f);
                Piece<T>::unpack(v, f);  // statistics of the values as stored
            }
            *(uint4*)((T*)a.out + g) = v;
#pragma unroll
            for (int j = 0; j < EPB; ++j) {
                s[j] += f[j];
                q2[j] = fmaf(f[j], f[j], q2[j]);
            }
        }
        if (a.stats) {  // uniform branch
            // lanes with equal c inside a wave: xor-reduce over the pixel-slot bits of the lane id
#pragma unroll
            for (int o = OLPP; o < 64; o <<= 1) {
#pragma unroll
                for (int j = 0; j < EPB; ++j) {
                    s[j] += __shfl_xor(s[j], o, 64);
                    q2[j] += __shfl_xor(q2[j], o, 64);
                }
            }
            if (lane < OLPP && cvalid) {
#pragma unroll
                for (int j = 0; j < EPB; ++j) {
                    red[(wave * NB + c * EPB + j) * 2 + 0] = s[j];
                    red[(wave * NB + c * EPB + j) * 2 + 1] = q2[j];
                }
            }
            __syncthreads();
            const int nparts = a.tiles_x * a.tiles_y * gridDim.z;
            const int part = ((blockIdx.x % (a.tiles_x * a.tiles_y)) * gridDim.z + cls);
            for (int i = tid; i < NB * 2; i += C::NTHREADS) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < C::NWAVES; ++w) t += red[w * NB * 2 + i];
                a.stats[(((size_t)bs * nparts + part) * NOUT + cout0) * 2 + i] = t;
            }
        }
    }
}

#undef W_LOAD
#undef W_STORE

template <class C>
hipError_t launch_conv_cfg(const ConvArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<C>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid(a.tiles_x * a.tiles_y * a.B, C::NOUT / C::NB, C::MODE == UP4 ? 2 : 1);
    hipLaunchKernelGGL(conv_mfma_kernel<C>, grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// Host-side description of one configuration (tile geometry the caller needs for grids and slabs).
struct ConvGeom {
    int th, tw, nb, nout, classes;
};

// Implemented in conv_inst_*.hip: fills geometry / launches for (dtype, mode, cin, cout).
// ``cout`` is the real output channel count (UP4: Cprev); returns hipErrorInvalidValue if unsupported.
hipError_t conv_geometry(int dtype, int mode, int cin, int cout, ConvGeom* g);
hipError_t conv_launch(int dtype, int mode, int cin, int cout, ConvArgs& a, hipStream_t stream);

}  // namespace ddimx
