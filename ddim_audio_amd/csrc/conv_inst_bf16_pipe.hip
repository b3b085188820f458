// Instantiations of conv3_pipe_kernel (conv_pipe.h): the Residual_Block convs of levels 0-1, software-pipelined inside each wave.
#include "conv_pipe.h"

namespace ddimx {

//      C   TH  WM  MINW
#define DDIMX_PIPE(X)  \
    X(32, 16, 4, 1)    \
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
#define DDIMX_L(C_, TH_, WM_, MW_)                                                                        \
    if (c == C_ && xf == XF_AFFINE) return launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE, MW_>>(a, stream); \
    if (c == C_ && xf == XF_AFFINE_SILU) return launch_pipe_cfg<PipeCfg<C_, TH_, WM_, XF_AFFINE_SILU, MW_>>(a, stream);
    DDIMX_PIPE(DDIMX_L)
    return hipErrorInvalidValue;
}

}  // namespace ddimx
