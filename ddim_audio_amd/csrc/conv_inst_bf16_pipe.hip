// Instantiations of conv3_pipe_kernel (conv_pipe.h): the Residual_Block convs of levels 0-1, software-pipelined inside each wave.
#include "conv_pipe.h"

namespace ddimx {

// C = 32: 8-row tiles, four waves, two workgroups per CU (69 KB of LDS, <= 256 registers): the second workgroup of a CU may belong
// to the other batch shard's launch (ddimx_unet_fwd_forked); C = 64: 144 registers of weights per wave, one four-wave workgroup per CU.
//      C   TH  WM  MINW
#define DDIMX_PIPE(X)  \
    X(32, 8, 4, 2)     \
    X(64, 8, 2, 1)

hipError_t pipe_geometry(int c, PipeGeom* g) {
#define DDIMX_G(C_, TH_, WM_, MW_)                                                                        \
    if (c == C_) {                                                                                        \
        typedef PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_> F;                                                  \
        g->th = F::TH; g->tw = F::TW; g->lds_bytes = F::LDS_BYTES; g->nthreads = F::NTHREADS;             \
        return hipSuccess;                                                                                \
    }
    DDIMX_PIPE(DDIMX_G)
    return hipErrorInvalidValue;
}
hipError_t pipe_launch(int c, int xf, const WregArgs& a, hipStream_t stream) {
#ifdef DDIMX_STAMP  // diagnostic build: stage-less variants (timing only), chosen by WregArgs::dbg
#define DDIMX_LD(C_, TH_, WM_, MW_)                                                                      \
    if (c == C_ && a.dbg == 1) return xf == XF_AFFINE ? launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_, 1>>(a, stream) : launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE_SILU, MW_, 1>>(a, stream); \
    if (c == C_ && a.dbg == 2) return xf == XF_AFFINE ? launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_, 2>>(a, stream) : launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE_SILU, MW_, 2>>(a, stream); \
    if (c == C_ && a.dbg == 4) return xf == XF_AFFINE ? launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_, 4>>(a, stream) : launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE_SILU, MW_, 4>>(a, stream); \
    if (c == C_ && a.dbg == 6) return xf == XF_AFFINE ? launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_, 6>>(a, stream) : launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE_SILU, MW_, 6>>(a, stream); \
    if (c == C_ && a.dbg == 7) return xf == XF_AFFINE ? launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_, 7>>(a, stream) : launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE_SILU, MW_, 7>>(a, stream);
    DDIMX_PIPE(DDIMX_LD)
#endif
#define DDIMX_L(C_, TH_, WM_, MW_)                                                                        \
    if (c == C_ && xf == XF_AFFINE) return launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_>>(a, stream); \
    if (c == C_ && xf == XF_AFFINE_SILU) return launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE_SILU, MW_>>(a, stream);
    DDIMX_PIPE(DDIMX_L)
    return hipErrorInvalidValue;
}

}  // namespace ddimx
