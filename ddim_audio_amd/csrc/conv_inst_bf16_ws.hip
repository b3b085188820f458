// Instantiations of conv3_ws_kernel (conv_ws.h): Residual_Block convolutions with specialised MFMA / loader waves.
#include "conv_ws.h"

namespace ddimx {

//        C  TH  TW  WM  NL  D  TPW      (MFMA waves = WM x C/32; 8 waves per workgroup, one workgroup per CU)
#define DDIMX_WS(X)             \
    X(32, 16, 32, 4, 4, 6, 16)  \
    X(64, 8, 32, 2, 4, 6, 8)

hipError_t ws_geometry(int c, WsGeom* g) {
#define DDIMX_G(C_, TH_, TW_, WM_, NL_, D_, TPW_)                                                            \
    if (c == C_) {                                                                                          \
        typedef WsCfg<C_, TH_, TW_, WM_, NL_, D_, TPW_> F;                                                  \
        g->th = F::TH; g->tw = F::TW; g->lds_bytes = F::LDS_BYTES; g->nthreads = F::NTHREADS; g->tiles_per_wg = F::TPW; \
        return hipSuccess;                                                                                  \
    }
    DDIMX_WS(DDIMX_G)
    return hipErrorInvalidValue;
}
hipError_t ws_launch(int c, const WregArgs& a, hipStream_t stream) {
#define DDIMX_L(C_, TH_, TW_, WM_, NL_, D_, TPW_) \
    if (c == C_) return launch_ws_cfg<WsCfg<C_, TH_, TW_, WM_, NL_, D_, TPW_>>(a, stream);
    DDIMX_WS(DDIMX_L)
    return hipErrorInvalidValue;
}

}  // namespace ddimx
