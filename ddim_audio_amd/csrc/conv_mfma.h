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
// Work decomposition: a PERSISTENT workgroup walks `tiles_per_wg` consecutive tiles of one sample; a tile is
// TH x TW output pixels x NB output channels.  While the MFMAs of tile i run, the halo of tile i+1 is already
// in flight into registers (global_load_dwordx4, waited for only after the epilogue of tile i), so HBM latency
// hides under compute.  The transformed halo tile stays resident in LDS for all taps; weights stay resident in
// LDS for the whole workgroup when all taps fit in one chunk, else they stream through LDS in chunks (TPC taps
// x KC input channels), register-staged one chunk ahead of the MFMAs.  GroupNorm statistics accumulate in
// registers across the workgroup's tiles: one deterministic partial per workgroup.
// MFMA orientation: A = weights (rows = cout), B = pixels (cols = pixel), so the accumulator holds,
// per lane, 4 consecutive couts of one pixel per register quad -> packed 8/16-byte LDS writes in the
// epilogue, then fully coalesced 16-byte global stores of whole NHWC pixel rows.
//
// T = __bf16 : v_mfma_f32_32x32x16_bf16 (fp32 accumulate);  T = float : v_mfma_f32_32x32x2_f32
// (exact fp32 FMA chain; the parity mode).
#pragma once
#include <type_traits>

#include "common.h"
#include "gn_fused.h"

namespace ddimx {

enum ConvMode { CONV3 = 0, DOWN4 = 1, UP4 = 2 };
// input transforms applied while the halo is staged: y = x*scale+shift, SiLU(x*scale+shift), SiLU(x)*scale+shift
enum { XF_NONE = 0, XF_AFFINE = 1, XF_AFFINE_SILU = 2, XF_SILU_AFFINE = 3 };

struct ConvArgs {
    const void* in;         // [B][Hin][Win][CIN]
    const void* w;          // [classes][taps][NOUT][CIN]   (classes = 2 for UP4, else 1)
    const float* bias;      // [NOUT] or null
    const float* chan_add;  // per-sample per-cout vector (base + b*chan_add_stride), or null
    const float* in_scale;  // [B][CIN] folded GroupNorm scale (xf != XF_NONE)
    const float* in_shift;  // [B][CIN]
    const void* skip;       // same shape as out, added in the epilogue, or null
    void* out;              // [B][Hout][Wout][COUT]
    float* stats;           // [B][nparts][NOUT][2] partial (sum, sumsq) or null; with stats_groups_c > 0 instead
                            // [B][nparts * NOUT/NB][kGroups][2]: the workgroup's partials folded to the 8 groups (gn_fused.h)
    // Backward-statistics kernels only (ConvCfg::BWD, data-gradient convs of the training step): the GroupNorm-backward partial
    // sums of the tensor this conv writes, taken in its epilogue instead of by a pass of their own (train_kernels.hip,
    // gn_bwd_stats).  With f = the value as stored and a = aux at the same position:
    //   bwd_mode 1 (GroupNorm fed by SiLU(a)):        P += f,                       Q += f * SiLU(a)
    //   bwd_mode 2 (GroupNorm followed by SiLU):      g = f * SiLU'(a*sc + sh), P += g, Q += g * a     (sc / sh = aux_scale / aux_shift)
    // stats then holds (P, Q) per channel where the forward kernels hold (sum, sumsq).
    const void* aux;        // same shape as out
    const float* aux_scale; // [B][NOUT] (bwd_mode 2)
    const float* aux_shift;
    int bwd_mode;
    int stats_groups_c;     // 0, or the real channel count of the output (NOUT / column classes)
    GnIn gn;                // gn.stats != null (and xf != XF_NONE): scale / shift come from the input's group partials,
                            // finished by every workgroup in its prologue, instead of in_scale / in_shift
    int chan_add_stride;
    int xf;                 // XF_*
    int act;                // 0 none, 1 SiLU, 2 store the pre-activation but take the statistics of SiLU(value) (training)
    int B, Hin, Win;
    int Hv, Wv;             // virtual output grid (UP4: = input grid; else = output grid)
    int tiles_x, tiles_y;
    int tiles_per_wg;       // consecutive tiles (x-major) walked by one workgroup
    int wgs_per_sample;     // ceil(tiles_x*tiles_y / tiles_per_wg); grid.x = B * wgs_per_sample
    unsigned long long* stamps;  // diagnostic builds only (-DDDIMX_STAMP): [grid.x][16] per-phase cycle sums
};

template <typename T, int CIN_, int NOUT_, int NB_, int MODE_, int TH_, int TW_, int WM_, int WN_, int KC_, int TPC_, int OVL_ = 0, int BWD_ = 0>
struct ConvCfg {
    typedef T elem;
    static constexpr int CIN = CIN_, NOUT = NOUT_, NB = NB_, MODE = MODE_, TH = TH_, TW = TW_, WM = WM_, WN = WN_,
                         KC = KC_, TPC = TPC_;
    static constexpr bool BWD = BWD_ != 0;  // separate instantiation with the GroupNorm-backward statistics epilogue
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
    // streaming weights: 3-stage LDS-DMA ring (global_load_lds, 1 KiB per wave-instruction, every wave issues
    // the same number DMA_PER_WAVE so a counted vmcnt is exact); resident weights: one register-staged chunk
    static constexpr int WROWP = KC / EPB + 1;                        // 16-B pieces per padded LDS row
    static constexpr int CHUNK_PIECES = TPC * NB * WROWP;
    static constexpr int DMA_INSTR = (CHUNK_PIECES + 63) / 64;
    static constexpr int DMA_PER_WAVE = (DMA_INSTR + NWAVES - 1) / NWAVES;
    static constexpr int WSTAGE_BYTES = DMA_INSTR * 1024;  // surplus DMA slots of the last round go to a dummy KiB
    static constexpr int NSTAGE = 3;
    static constexpr int WBUF_BYTES = (NCHUNKS > 1 ? NSTAGE : 1) * WSTAGE_BYTES + 1024;
    static constexpr int RED_BYTES = NWAVES * NB * 2 * 4;  // per-wave per-channel (sum, sumsq); overlays wbuf/halo at the end
    static constexpr bool RESIDENT_W = NCHUNKS == 1;       // all taps in one chunk: loaded once per workgroup
    // the epilogue's output tile gets its own LDS region when everything fits in 80 KiB (2 workgroups per CU);
    // otherwise it overlays the halo region (two more barriers per tile)
    // OVL_: 0 = separate when it fits, 1 = force overlay (smaller LDS, more workgroups per CU), 2 = force separate
    static constexpr bool SEPARATE_OUT = OVL_ == 2 || (OVL_ == 0 && WBUF_BYTES + HALO_BYTES + OUT_BYTES <= 80 * 1024);
    static constexpr int HO_BYTES = SEPARATE_OUT ? HALO_BYTES + OUT_BYTES : (HALO_BYTES > OUT_BYTES ? HALO_BYTES : OUT_BYTES);
    static constexpr int ADD_BYTES = NB * 4;  // per-cout epilogue addend (bias + timestep embedding) of this sample
    static constexpr int GN_BYTES = NWAVES * kGroups * 2 * 4;  // per-wave (sum, sumsq) per group (gn_fused.h); behind the rest
    static constexpr int LDS_RAW = ADD_BYTES + WBUF_BYTES + HO_BYTES + GN_BYTES;
    static constexpr int LDS_BYTES = LDS_RAW > RED_BYTES ? LDS_RAW : RED_BYTES;
    static constexpr int KG = KC * ES / 32;  // 32-byte k-groups per tap per chunk
    static constexpr int WPIECES = TPC * NB * (KC / EPB);
    static constexpr int WPT = (WPIECES + NTHREADS - 1) / NTHREADS;  // weight pieces per thread per chunk
    static constexpr int CPP = CIN / EPB;       // 16-B pieces per input pixel
    static constexpr int LPP = next_pow2(CPP);  // lanes cooperating on one input pixel
    static constexpr int OPP = NB / EPB;        // 16-B pieces per output pixel (this WG's channels)
    static constexpr int OLPP = next_pow2(OPP);
    static constexpr int NPIX = IH * IW;        // halo pixels
    static constexpr int PPP = NTHREADS / LPP;  // halo pixels staged per pass
    static constexpr int HPT = (NPIX + PPP - 1) / PPP;  // halo pieces per thread (prefetch registers)

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
    // prefetch the next halo into registers only where it fits beside the accumulators (the HBM-bound, small-C
    // configurations); elsewhere the halo is staged synchronously after the epilogue
    // per-piece offsets are hoisted out of the tile loop (2 registers per piece) when there are few pieces
    static constexpr bool HOIST = HPT <= 16;
    static constexpr bool PREFETCH = HOIST && HPT * 6 + NT * MT * 16 <= 100;
    static constexpr int HREGS = (PREFETCH || HOIST) ? HPT : 4;
    static constexpr int HOFF = HOIST ? HPT : 1;
    // occupancy the LDS footprint allows, as waves per SIMD (2nd __launch_bounds__ argument): keeps the register
    // allocator from trading occupancy for scheduling freedom
    static constexpr int WG_PER_CU = (160 * 1024) / LDS_BYTES < 1 ? 1 : (160 * 1024) / LDS_BYTES;
    static constexpr int W_LDS = (WG_PER_CU * NWAVES + 3) / 4 > 8 ? 8 : (WG_PER_CU * NWAVES + 3) / 4;
    static constexpr int REG_EST = NT * MT * 16 + HREGS * 4 + WPT * 4 + (NT + MT) * 4 + 72;
    static constexpr int W_REG = 512 / ((REG_EST + 7) / 8 * 8) < 1 ? 1 : 512 / ((REG_EST + 7) / 8 * 8);
    static constexpr int MINW_RAW = W_LDS < W_REG ? W_LDS : W_REG;
    // an 8-wave workgroup occupies 2 waves per SIMD: 3 is not a useful target
    static constexpr int MINW = (NWAVES == 8 && MINW_RAW == 3) ? 2 : MINW_RAW;
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


// In-kernel phase stamps (diagnostic build libddimx_stamp.so only; the product library never executes one).
#ifdef DDIMX_STAMP
// per wave: [0..11] cycle sums per phase of the tile loop (s_memtime), [12] s_memrealtime (100 MHz) at kernel entry, [13] at the
// start of the tile loop (= end of the prologue), [14] at exit, [15] HW_ID | XCC_ID << 32 (which CU / SIMD / XCD ran the wave)
#define DDIMX_STAMP_ENTRY unsigned long long st_rt0 = 0; { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0) :: "memory"); }
#define DDIMX_STAMP_DECL unsigned long long st_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, st_last = 0, st_rt1 = 0; { asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1), "=s"(st_last) :: "memory"); }
#define DDIMX_STAMP_AT(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[k] += t_ - st_last; st_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define DDIMX_STAMP_FLUSH() do { if (a.stamps && (threadIdx.x & 63) == 0) { unsigned long long rt2_; unsigned hw_, xcc_; asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt2_), "=s"(hw_), "=s"(xcc_) :: "memory"); unsigned long long* d_ = a.stamps + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 16; for (int k_ = 0; k_ < 12; ++k_) d_[k_] = st_acc[k_]; d_[12] = st_rt0; d_[13] = st_rt1; d_[14] = rt2_; d_[15] = (unsigned long long)hw_ | ((unsigned long long)xcc_ << 32); } } while (0)
#else
#define DDIMX_STAMP_ENTRY
#define DDIMX_STAMP_DECL
#define DDIMX_STAMP_AT(k)
#define DDIMX_STAMP_FLUSH()
#endif

template <class C>
__global__ void __launch_bounds__(C::NTHREADS, C::MINW) conv_mfma_kernel(const ConvArgs a) {
    typedef typename C::elem T;
    constexpr int ES = C::ES, EPB = C::EPB, CIN = C::CIN, NB = C::NB, NOUT = C::NOUT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const addv = (float*)smem;
    char* const wbuf = smem + C::ADD_BYTES;
    char* const halo = wbuf + C::WBUF_BYTES;
    char* const otile = C::SEPARATE_OUT ? halo + C::HALO_BYTES : halo;
    float* const gnscr = (float*)(smem + C::LDS_RAW - C::GN_BYTES);

    DDIMX_STAMP_ENTRY
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % C::WM, wn = wave / C::WM;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware order (speed only, never correctness): hardware deals consecutive blockIdx round-robin over the 8
    // XCDs, each with its own L2.  Give every XCD a contiguous range of logical workgroups so that workgroups of
    // neighbouring tile ranges -- which re-read each other's halo rows -- share an L2 (bijective for any grid size).
    int lwg;
    {
        const int nwg = gridDim.x, x8 = blockIdx.x & 7, i8 = blockIdx.x >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        lwg = (x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q) + i8;
    }
    const int wg = lwg % a.wgs_per_sample;
    const int bs = lwg / a.wgs_per_sample;
    const int cout0 = blockIdx.y * NB;
    const int cls = blockIdx.z;  // UP4 row-parity class; 0 otherwise
    const int ntile_s = a.tiles_x * a.tiles_y;
    const int t_begin = wg * a.tiles_per_wg;
    const int t_end = (t_begin + a.tiles_per_wg < ntile_s) ? t_begin + a.tiles_per_wg : ntile_s;

    const T* const wbase = (const T*)a.w + (size_t)cls * C::NTAPS * NOUT * CIN;

    // ---- weights: LDS-DMA, no VGPRs spent on staging.  Resident (all taps in one chunk): one chunk, issued once in the
    // prologue.  Streaming: a 3-stage ring that runs continuously across tiles -- while chunk g is multiplied, chunk g+1 is
    // landing and chunk g+2 is being issued.  Per-lane source offsets (relative to the chunk base) are tile- and chunk-
    // invariant and hoisted here; the LDS destination of DMA instruction q is stage_base + q * 1 KiB (wave-uniform),
    // lanes that map to row padding re-read the chunk base.
    unsigned wrel[C::DMA_PER_WAVE];
    {
#pragma unroll
        for (int j = 0; j < C::DMA_PER_WAVE; ++j) {
            const int idx = (wave + j * C::NWAVES) * 64 + lane;
            const int tp = idx / (NB * C::WROWP), rem = idx % (NB * C::WROWP);
            const int row = rem / C::WROWP, pc = rem % C::WROWP;
            const bool real = idx < C::CHUNK_PIECES && pc < C::WROWP - 1;
            wrel[j] = real ? (unsigned)(((tp * NOUT + row) * CIN + pc * EPB) * ES) : 0u;
        }
    }
    // wave-uniform LDS byte address of this wave's first DMA slot (readfirstlane: threadIdx-derived values are
    // not provably uniform to the compiler)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned wbuf_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(wbuf) + (unsigned)wave_u * 1024u);
    const unsigned wdummy_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(wbuf) + (unsigned)(C::WBUF_BYTES - 1024));
    auto dma_issue = [&](int ch, int stage) __attribute__((always_inline)) {
        {
            const int tap0 = (ch / C::KSPLIT) * C::TPC, kc0 = (ch % C::KSPLIT) * C::KC;
            const char* cbase = (const char*)(wbase + ((size_t)tap0 * NOUT + cout0) * CIN + kc0);
            const unsigned dst0 = wbuf_lds + stage * C::WSTAGE_BYTES;
#pragma unroll
            for (int j = 0; j < C::DMA_PER_WAVE; ++j) {
                // every wave issues DMA_PER_WAVE instructions (exact counted waits); slots past the chunk land in
                // the dummy KiB behind the ring
                const bool real_slot = (j + 1) * C::NWAVES <= C::DMA_INSTR || wave_u + j * C::NWAVES < C::DMA_INSTR;
                lds_dma16(cbase + wrel[j], __builtin_amdgcn_readfirstlane(real_slot ? dst0 + j * C::NWAVES * 1024 : wdummy_lds));
            }
        }
    };

    // ---- halo staging: per-thread constants (one thread = one 16-byte channel piece of a pixel slot) -----
    const int hc = tid % C::LPP, hslot = tid / C::LPP;
    const bool hvalid = hc < C::CPP;
    constexpr int NP = EPB / 2;  // float2 pairs per 16-byte piece
    f32x2_t sc[NP], sh[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
    // per-sample buffer resources (wave-uniform: built from kernel arguments and blockIdx only)
    const unsigned in_bytes = (unsigned)((size_t)a.Hin * a.Win * CIN * ES);
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc((const T*)a.in + (size_t)bs * a.Hin * a.Win * CIN, in_bytes);

    uint4 hreg[C::HREGS];
    unsigned hok = 0;
    // tile-invariant per-piece offsets: global byte offset relative to the halo origin, LDS byte offset
    unsigned hrel[C::HOFF];
    int hlds[C::HOFF];
    if constexpr (C::HOIST) {
#pragma unroll
        for (int i = 0; i < C::HPT; ++i) {
            const int pix = i * C::PPP + hslot;
            const bool in_tile = hvalid && (i < C::HPT - 1 || pix < C::NPIX);
            const int iy = pix / C::IW, ix = pix % C::IW;
            hrel[i] = in_tile ? (unsigned)(((iy * a.Win + ix) * CIN + hc * EPB) * ES) : 0x80000000u;  // -> out of bounds
            hlds[i] = in_tile ? iy * C::ROWSTRIDE + ix * C::PSTRIDE + hc * 16 : -1;
        }
    }
    // one halo piece, general form: branch-free buffer load; padding / out-of-range pieces use an
    // out-of-bounds offset -> zeros
    auto piece_load = [&](int t, int i, uint4& v) __attribute__((always_inline)) -> bool {
        const int pix = i * C::PPP + hslot;
        const int hy0 = (t / a.tiles_x) * C::TH * C::SXY - 1, hx0 = (t % a.tiles_x) * C::TW * C::SXY - 1;
        const int gy = hy0 + pix / C::IW, gx = hx0 + pix % C::IW;
        const bool ok = hvalid && (i < C::HPT - 1 || pix < C::NPIX) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        v = buf_load16(in_rsrc, ok ? (unsigned)(((gy * a.Win + gx) * CIN + hc * EPB) * ES) : kOOB);
        return ok;
    };
    // all pieces of tile t: interior tiles (halo entirely inside the image) add one scalar base to the hoisted
    // offsets; border tiles take the general form.  Sets hok (bit i = piece i holds real data).
    auto halo_load_all = [&](int t) __attribute__((always_inline)) {
        const int hy0 = (t / a.tiles_x) * C::TH * C::SXY - 1, hx0 = (t % a.tiles_x) * C::TW * C::SXY - 1;
        const bool interior = hy0 >= 0 && hx0 >= 0 && hy0 + C::IH <= a.Hin && hx0 + C::IW <= a.Win;  // wave-uniform
        if (interior) {
            const unsigned base = (unsigned)((hy0 * a.Win + hx0) * CIN * ES);
#pragma unroll
            for (int i = 0; i < C::HPT; ++i) hreg[i] = buf_load16(in_rsrc, base + hrel[i]);
            hok = 0xFFFFFFFFu;
        } else {
            hok = 0;
#pragma unroll
            for (int i = 0; i < C::HPT; ++i)
                if (piece_load(t, i, hreg[i])) hok |= 1u << i;
        }
    };
    // GroupNorm affine (+SiLU) in registers, zero the padding AFTER the transform (select, no branch), one LDS
    // write.  XF is a compile-time tag (the caller branches once, wave-uniformly, on a.xf).
    auto piece_xf = [&](auto xf_tag, uint4 v, bool ok) __attribute__((always_inline)) -> uint4 {
        constexpr int XF = decltype(xf_tag)::value;
        if (XF != XF_NONE) {
            f32x2_t f[NP];
            Pairs<T>::unpack(v, f);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (XF == XF_SILU_AFFINE) f[j] = silu2(f[j]);
                f[j] = fma2(f[j], sc[j], sh[j]);
                if (XF == XF_AFFINE_SILU) f[j] = silu2(f[j]);
            }
            const uint4 tv = Pairs<T>::pack(f);
            v.x = ok ? tv.x : 0u; v.y = ok ? tv.y : 0u; v.z = ok ? tv.z : 0u; v.w = ok ? tv.w : 0u;
        }
        return v;
    };
    auto piece_store = [&](auto xf_tag, int i, uint4 v, bool ok) __attribute__((always_inline)) {
        const int pix = i * C::PPP + hslot;
        v = piece_xf(xf_tag, v, ok);
        if (i < C::HPT - 1 || pix < C::NPIX)
            *(uint4*)(halo + (pix / C::IW) * C::ROWSTRIDE + (pix % C::IW) * C::PSTRIDE + hc * 16) = v;
    };
    // issue the global loads of tile t's halo (no wait) -- prefetch configurations only
    auto halo_issue = [&](int t) __attribute__((always_inline)) {
        if constexpr (C::PREFETCH) {
            halo_load_all(t);
            // pin the issue point: without this LLVM sinks the loads below the MFMA block, next to their use
            // (no waitcnt is implied: nothing here reads the destination registers)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // write tile t's halo to LDS (from the prefetch registers, or loading it now)
    // pre: the tile's pieces are already in hreg (prefetch configurations always; the first tile of a hoisted one)
    auto halo_commit_xf = [&](auto xf_tag, int t, bool pre) __attribute__((always_inline)) {
        if (!hvalid) return;  // lanes beyond the channel pieces of a pixel (C/8 not a power of two)
        if constexpr (C::HOIST) {
            if (!C::PREFETCH && !pre) halo_load_all(t);
            if (hok == 0xFFFFFFFFu) {  // interior tile: every piece is real data
#pragma unroll
                for (int i = 0; i < C::HPT; ++i)
                    if (i < C::HPT - 1 || hlds[i] >= 0) *(uint4*)(halo + hlds[i]) = piece_xf(xf_tag, hreg[i], true);
            } else {
#pragma unroll
                for (int i = 0; i < C::HPT; ++i)
                    if (i < C::HPT - 1 || hlds[i] >= 0) *(uint4*)(halo + hlds[i]) = piece_xf(xf_tag, hreg[i], (hok >> i) & 1u);
            }
        } else {
#pragma unroll 1
            for (int i0 = 0; i0 < C::HPT; i0 += 4) {
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ok[u] = (i0 + u < C::HPT) && piece_load(t, i0 + u, hreg[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < C::HPT) piece_store(xf_tag, i0 + u, hreg[u], ok[u]);
            }
        }
    };
    auto halo_commit = [&](int t, bool pre = false) __attribute__((always_inline)) {
        if (a.xf == XF_AFFINE_SILU) halo_commit_xf(std::integral_constant<int, XF_AFFINE_SILU>(), t, pre);
        else if (a.xf == XF_SILU_AFFINE) halo_commit_xf(std::integral_constant<int, XF_SILU_AFFINE>(), t, pre);
        else if (a.xf == XF_AFFINE) halo_commit_xf(std::integral_constant<int, XF_AFFINE>(), t, pre);
        else halo_commit_xf(std::integral_constant<int, XF_NONE>(), t, pre);
    };

    // ---- per-lane MFMA operand offsets (tile independent) ----------------------------------------------
    int pixoff[C::MT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
        const int p = (wm * C::MT + m) * 32 + l31;
        pixoff[m] = (p / C::TW) * C::SXY * C::ROWSTRIDE + (p % C::TW) * C::SXY * C::PSTRIDE + h * 16;
    }
    const int woff = (wn * C::NT * 32 + l31) * C::WROW + h * 16;

    // ---- epilogue constants ---------------------------------------------------------------------------------
    const int oc = tid % C::OLPP, oslot = tid / C::OLPP;
    const bool ovalid = oc < C::OPP;
    f32x2_t st_s[NP], st_q[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
    const int Hout = (C::MODE == UP4) ? 2 * a.Hv : a.Hv;
    const size_t rowlen = (size_t)a.Wv * NOUT;  // elements per output row (UP4: Wv*2*Cprev = Wout*Cprev)
    const unsigned out_bytes = (unsigned)((size_t)Hout * rowlen * ES);
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc((T*)a.out + (size_t)bs * Hout * rowlen, out_bytes);
    const void* const side = C::BWD ? a.aux : a.skip;  // the second tensor of the epilogue: skip addend, or aux of the backward statistics
    const __amdgpu_buffer_rsrc_t skip_rsrc = make_rsrc(side ? (const T*)side + (size_t)bs * Hout * rowlen : (const T*)a.out, side ? out_bytes : 0u);
    // backward statistics, mode 2: the folded GroupNorm (scale, shift) of this thread's output channels
    f32x2_t bsc[C::BWD ? NP : 1], bsh[C::BWD ? NP : 1];
    if constexpr (C::BWD) {
#pragma unroll
        for (int j = 0; j < NP; ++j) { bsc[j] = 1.f; bsh[j] = 0.f; }
        if (a.bwd_mode == 2 && ovalid) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                bsc[j] = *(const f32x2_t*)(a.aux_scale + (size_t)bs * NOUT + cout0 + oc * EPB + 2 * j);
                bsh[j] = *(const f32x2_t*)(a.aux_shift + (size_t)bs * NOUT + cout0 + oc * EPB + 2 * j);
            }
        }
    }

    // ---- prologue: resident weights (if they fit in one chunk) and the first halo ------------------------
    int wstage = 0;  // ring stage holding the chunk about to be multiplied
    // Order of issue: (GroupNorm partials, above) -> weights -> the first tile's halo -> then the first wait.  Hoisted
    // configurations keep all halo pieces of a tile in registers, so the first tile's loads overlap the weights and the
    // partials instead of following them.
    // (All loads of the prologue are issued here, after every address computation above: a load issued earlier can stall
    // unrelated integer code through a register pair the allocator happens to share with its destination.  The scheduling
    // barrier and the empty asm statements (which "use" the hoisted offsets) keep that code from sinking below the loads.)
    if constexpr (C::HOIST) {
#pragma unroll
        for (int i = 0; i < C::HPT; ++i) asm volatile("" ::"v"(hrel[i]), "v"(hlds[i]) : "memory");
    }
#pragma unroll
    for (int j = 0; j < C::DMA_PER_WAVE; ++j) asm volatile("" ::"v"(wrel[j]) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    // per-cout epilogue addend (bias + this sample's timestep-embedding slice): loaded first and without branches (absent
    // operands read a valid dummy address and are dropped by a select), stored to LDS once everything else is in flight
    constexpr int AIT = (NB + C::NTHREADS - 1) / C::NTHREADS;
    float add_b[AIT], add_c[AIT];
    {
        const float* pb = a.bias ? a.bias + cout0 : (const float*)a.w;
        const float* pc = a.chan_add ? a.chan_add + (size_t)bs * a.chan_add_stride + cout0 : (const float*)a.w;
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = tid + k * C::NTHREADS;
            const int ic = i < NB ? i : NB - 1;
            add_b[k] = pb[ic];
            add_c[k] = pc[ic];
        }
    }
    // consumer-side GroupNorm finalisation (gn_fused.h): this sample's group partials are summed first -- their loads are the
    // oldest in the queue, the weight and halo loads below overlap the reduction
    const bool gn_fused = a.xf != XF_NONE && a.gn.stats != nullptr;  // uniform
    GnInLoads gn_ld;
    if (gn_fused) gn_in_issue(a.gn, bs, tid, C::NTHREADS, gn_ld);

    // one load sequence for both sources (a second, branch-separated one made the compiler wait for every load in flight at
    // the merge): folded (scale, shift) of this sample, or -- consumer-side finalisation -- (gamma, beta) to be folded with
    // the group statistics after the reduction below.  beta == null: gamma is read in its place and dropped.
    if (a.xf != XF_NONE && hvalid) {
        const float* psc = gn_fused ? a.gn.gamma + hc * EPB : a.in_scale + (size_t)bs * CIN + hc * EPB;
        const float* psh = gn_fused ? (a.gn.beta ? a.gn.beta : a.gn.gamma) + hc * EPB : a.in_shift + (size_t)bs * CIN + hc * EPB;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            sc[j] = *(const f32x2_t*)(psc + 2 * j);
            sh[j] = *(const f32x2_t*)(psh + 2 * j);
        }
    }
    dma_issue(0, 0);
    if constexpr (!C::RESIDENT_W) dma_issue(1 % C::NCHUNKS, 1);
    if constexpr (C::PREFETCH) {
        halo_issue(t_begin);
    } else if constexpr (C::HOIST) {
        halo_load_all(t_begin);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int i = tid + k * C::NTHREADS;
        if (i < NB) addv[i] = (a.bias ? add_b[k] : 0.f) + (a.chan_add ? add_c[k] : 0.f);
    }
    if (gn_fused) {
        gn_in_reduce(a.gn, bs, tid, C::NTHREADS, gn_ld, gnscr);
        __syncthreads();  // the waves' group sums are in gnscr
        if (hvalid) {
            float gam[EPB], bet[EPB], fs[EPB], fh[EPB];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                gam[2 * j] = sc[j].x; gam[2 * j + 1] = sc[j].y;
                bet[2 * j] = a.gn.beta ? sh[j].x : 0.f; bet[2 * j + 1] = a.gn.beta ? sh[j].y : 0.f;
            }
            gn_in_fold<EPB>(a.gn, gnscr, C::NWAVES, CIN, hc * EPB, gam, bet, fs, fh);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                sc[j].x = fs[2 * j]; sc[j].y = fs[2 * j + 1];
                sh[j].x = fh[2 * j]; sh[j].y = fh[2 * j + 1];
            }
        }
    }
    halo_commit(t_begin, C::HOIST);
    // resident weights: the chunk must have landed before the first MFMA (LDS-DMA is inline asm, invisible to the
    // compiler's waitcnt insertion; the streamed ring has its own counted waits)
    if constexpr (C::RESIDENT_W) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DDIMX_STAMP_DECL

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int y0 = (t / a.tiles_x) * C::TH, x0 = (t % a.tiles_x) * C::TW;
        if (t + 1 < t_end) halo_issue(t + 1);  // in flight during the MFMAs below
        DDIMX_STAMP_AT(0);

        f32x16_t acc[C::NT][C::MT];
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

        // one chunk: MFMAs over TPC taps x KC channels from LDS buffer `buf`
        auto chunk_mma = [&](int ch, int buf) __attribute__((always_inline)) {
            const char* wb = wbuf + buf * C::WSTAGE_BYTES + woff;
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
        };
        if (C::RESIDENT_W) {
            chunk_mma(0, 0);
        } else {
#pragma unroll 1
            for (int ch = 0; ch < C::NCHUNKS; ++ch) {
                const int s2 = wstage >= 1 ? wstage - 1 : 2;  // (wstage + 2) % 3: free since the last barrier
                dma_issue((ch + 2) % C::NCHUNKS, s2);
                chunk_mma(ch, wstage);
                // this wave's part of chunk g+1 has landed once all but its DMA_PER_WAVE youngest VM ops are done.
                // lgkmcnt(0): this wave's ds_reads of chunk g must have RETURNED before the barrier releases the others --
                // the next iteration's DMA overwrites a stage two barriers later at the earliest, but hipcc sinks the
                // chunk's last MFMAs (and the lgkmcnt waits in front of them) below a raw s_barrier, which left reads of
                // stage g in flight while another wave was already past the barrier; with a second kernel loading the
                // CU's LDS pipeline (two streams) such a read was occasionally served after a later DMA had landed.
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::DMA_PER_WAVE) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                wstage = wstage == 2 ? 0 : wstage + 1;
            }
        }
        DDIMX_STAMP_AT(1);
        if (C::RESIDENT_W) __syncthreads();  // barrier A: every wave is done reading this tile's halo
        DDIMX_STAMP_AT(2);

        // ---- epilogue 1: accumulators -> (+bias, +chan_add, act) -> LDS out tile [pixel][cout] ------------
        auto epi1 = [&](auto act_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = (wn * C::NT + n) * 32 + q * 8 + h * 4;  // local cout of this register quad
                    const float4 av = *(const float4*)(addv + cl);
                    const f32x2_t a01 = {av.x, av.y}, a23 = {av.z, av.w};
#pragma unroll
                    for (int m = 0; m < C::MT; ++m) {
                        const int p = (wm * C::MT + m) * 32 + l31;
                        f32x2_t v01 = {acc[n][m][q * 4 + 0], acc[n][m][q * 4 + 1]};
                        f32x2_t v23 = {acc[n][m][q * 4 + 2], acc[n][m][q * 4 + 3]};
                        v01 += a01;
                        v23 += a23;
                        if (ACT) { v01 = silu2(v01); v23 = silu2(v23); }
                        char* dst = otile + p * C::OSTRIDE + cl * ES;
                        if constexpr (ES == 4) {
                            *(float4*)dst = make_float4(v01.x, v01.y, v23.x, v23.y);
                        } else {
                            *(uint2*)dst = make_uint2(Piece<__bf16>::pk(v01.x, v01.y), Piece<__bf16>::pk(v23.x, v23.y));
                        }
                    }
                }
            }
        };
        if (a.act == 1) epi1(std::integral_constant<int, 1>()); else epi1(std::integral_constant<int, 0>());
        DDIMX_STAMP_AT(3);
        if (C::SEPARATE_OUT && t + 1 < t_end) halo_commit(t + 1);  // halo region is free: stage the next tile now
        DDIMX_STAMP_AT(4);
        __syncthreads();                                     // barrier B: out tile (and next halo) visible
        DDIMX_STAMP_AT(5);

        // ---- epilogue 2: coalesced stores of whole pixel rows, + skip, per-channel statistics --------------
        // FULL tiles (entirely inside the image) need no per-lane validity; ragged ones drop stores through the
        // buffer bounds check and mask their statistics.
        auto epi2 = [&](auto full_tag, auto skip_tag, auto ssilu_tag, auto bwd_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int BWDM = decltype(bwd_tag)::value;       // 0, or the backward-statistics mode (ConvArgs::bwd_mode)
            constexpr bool SKIP = decltype(skip_tag)::value || BWDM != 0;  // a second tensor is loaded
            constexpr bool SSILU = decltype(ssilu_tag)::value;  // statistics of SiLU(stored value)
            constexpr int STEP = C::NTHREADS / C::OLPP;          // pixels per pass
            constexpr int NPASS = (C::P + STEP - 1) / STEP;
            if (!ovalid) return;
            const unsigned rowb = (unsigned)(a.Wv * NOUT * ES);  // bytes per virtual output row
            const unsigned cbase = (unsigned)((cout0 + oc * EPB) * ES);
            // pass 1: offsets, and ALL skip loads of the tile in flight before the first store (vmcnt is in-order
            // and counts stores: a load issued behind a store would wait for that store to complete)
            unsigned offs[NPASS];
            uint4 skv[SKIP ? NPASS : 1];
            bool vld[NPASS];
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const int p = oslot + k * STEP;                  // TW is a power of two: shifts and masks
                const int vy = y0 + p / C::TW, vx = x0 + p % C::TW;
                vld[k] = (C::P % STEP == 0 || p < C::P) && (FULL || (vy < a.Hv && vx < a.Wv));
                const unsigned oy = (C::MODE == UP4) ? 2u * vy + cls : (unsigned)vy;
                offs[k] = vld[k] ? oy * rowb + (unsigned)vx * (NOUT * ES) + cbase : kOOB;
                if (SKIP) skv[k] = buf_load16(skip_rsrc, offs[k]);
            }
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const int p = oslot + k * STEP;
                if (C::P % STEP != 0 && p >= C::P) break;
                uint4 v = *(const uint4*)(otile + p * C::OSTRIDE + oc * 16);
                f32x2_t f[NP];
                Pairs<T>::unpack(v, f);
                if (SKIP && BWDM == 0) {
                    f32x2_t kk[NP];
                    Pairs<T>::unpack(skv[k], kk);
#pragma unroll
                    for (int j = 0; j < NP; ++j) f[j] += kk[j];
                    v = Pairs<T>::pack(f);
                    Pairs<T>::unpack(v, f);  // statistics of the values as stored
                }
                buf_store16(out_rsrc, offs[k], v);
                const float msk = vld[k] ? 1.f : 0.f;
                if constexpr (BWDM != 0) {  // GroupNorm-backward partial sums (P, Q) of the stored gradient against aux
                    f32x2_t au[NP];
                    Pairs<T>::unpack(skv[k], au);
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        f32x2_t g = f[j], vv;
                        if (BWDM == 1) {
                            vv = silu2(au[j]);
                        } else {
                            g = g * dsilu2(fma2(au[j], bsc[j], bsh[j]));
                            vv = au[j];
                        }
                        if (!FULL) g *= msk;
                        st_s[j] += g;
                        st_q[j] = fma2(g, vv, st_q[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        if (SSILU) f[j] = silu2(f[j]);
                        if (!FULL) f[j] *= msk;
                        st_s[j] += f[j];
                        st_q[j] = fma2(f[j], f[j], st_q[j]);
                    }
                }
            }
        };
        {
            const bool full = y0 + C::TH <= a.Hv && x0 + C::TW <= a.Wv;  // wave-uniform
            typedef std::integral_constant<int, 0> B0;
            if constexpr (C::BWD) {
                typedef std::integral_constant<int, 1> B1;
                typedef std::integral_constant<int, 2> B2;
                if (a.bwd_mode == 1) { if (full) epi2(std::true_type(), std::false_type(), std::false_type(), B1()); else epi2(std::false_type(), std::false_type(), std::false_type(), B1()); }
                else { if (full) epi2(std::true_type(), std::false_type(), std::false_type(), B2()); else epi2(std::false_type(), std::false_type(), std::false_type(), B2()); }
            } else if (a.act == 2) { if (full) epi2(std::true_type(), std::false_type(), std::true_type(), B0()); else epi2(std::false_type(), std::false_type(), std::true_type(), B0()); }
            else if (a.skip) { if (full) epi2(std::true_type(), std::true_type(), std::false_type(), B0()); else epi2(std::false_type(), std::true_type(), std::false_type(), B0()); }
            else { if (full) epi2(std::true_type(), std::false_type(), std::false_type(), B0()); else epi2(std::false_type(), std::false_type(), std::false_type(), B0()); }
        }
        DDIMX_STAMP_AT(6);
        if (!C::SEPARATE_OUT && t + 1 < t_end) {
            __syncthreads();  // barrier C: out tile fully read before the halo region is overwritten
            DDIMX_STAMP_AT(7);
            halo_commit(t + 1);
            DDIMX_STAMP_AT(8);
            __syncthreads();  // barrier D
            DDIMX_STAMP_AT(9);
        }
    }

    DDIMX_STAMP_AT(10);
    // The weight ring runs two chunks ahead, so the last tile leaves two chunks of LDS-DMA in flight that nothing consumes.
    // They must have landed before LDS is reused below (`red` overlays the ring): the compiler's barrier only waits for
    // lgkmcnt (the DMA is inline asm, invisible to its waitcnt pass), and a DMA landing late -- seen when another kernel
    // shares the CU -- would overwrite this workgroup's statistics.
    if constexpr (!C::RESIDENT_W) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- statistics: one partial per workgroup -----------------------------------------------------------------
    if (a.stats) {  // uniform branch
        float* const red = (float*)smem;
        // lanes with equal oc inside a wave: xor-reduce over the pixel-slot bits of the lane id
#pragma unroll
        for (int o = C::OLPP; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                st_s[j].x += __shfl_xor(st_s[j].x, o, 64);
                st_s[j].y += __shfl_xor(st_s[j].y, o, 64);
                st_q[j].x += __shfl_xor(st_q[j].x, o, 64);
                st_q[j].y += __shfl_xor(st_q[j].y, o, 64);
            }
        }
        __syncthreads();  // everyone is done with wbuf / halo / out tile
        if (lane < C::OLPP && ovalid) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                red[(wave * NB + oc * EPB + 2 * j) * 2 + 0] = st_s[j].x;
                red[(wave * NB + oc * EPB + 2 * j) * 2 + 1] = st_q[j].x;
                red[(wave * NB + oc * EPB + 2 * j + 1) * 2 + 0] = st_s[j].y;
                red[(wave * NB + oc * EPB + 2 * j + 1) * 2 + 1] = st_q[j].y;
            }
        }
        __syncthreads();
        const int nparts = a.wgs_per_sample * gridDim.z;
        const int part = wg * gridDim.z + cls;
        if (a.stats_groups_c) {  // uniform: group format -- wave 0 folds the waves' channel sums straight into the 8 bins
            if (wave == 0)
                gn_bins_store<C::NWAVES>(red, NB * 2, NB, cout0, a.stats_groups_c,
                              a.stats + (((size_t)bs * nparts + part) * gridDim.y + blockIdx.y) * kGnSlab, lane);
        } else {
            for (int i = tid; i < NB * 2; i += C::NTHREADS) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < C::NWAVES; ++w) t += red[w * NB * 2 + i];
                a.stats[(((size_t)bs * nparts + part) * NOUT + cout0) * 2 + i] = t;
            }
        }
    }
    DDIMX_STAMP_AT(11);
    DDIMX_STAMP_FLUSH();
}


template <class C>
hipError_t launch_conv_cfg(const ConvArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<C>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid(a.wgs_per_sample * a.B, C::NOUT / C::NB, C::MODE == UP4 ? 2 : 1);
    hipLaunchKernelGGL(conv_mfma_kernel<C>, grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// Host-side description of one configuration (tile geometry the caller needs for grids and slabs).
struct ConvGeom {
    int th, tw, nb, nout, classes, lds_bytes, nthreads;
};

// Implemented in conv_inst_*.hip: fills geometry / launches for (dtype, mode, cin, cout).
// ``cout`` is the real output channel count (UP4: Cprev); returns hipErrorInvalidValue if unsupported.
hipError_t conv_geometry(int dtype, int mode, int cin, int cout, int var, ConvGeom* g);
hipError_t conv_launch(int dtype, int mode, int cin, int cout, int var, ConvArgs& a, hipStream_t stream);
// variant choice for a problem size: the small-tile variant when the large one gives < 200 workgroups
int conv_pick_variant(int dtype, int mode, int cin, int cout, int B, int Hv, int Wv);

}  // namespace ddimx
