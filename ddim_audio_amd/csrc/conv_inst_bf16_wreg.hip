// Instantiations of conv3_wreg_kernel (conv_wreg.h): convolutions with the weights streamed straight into registers.
#include "conv_wreg.h"

namespace ddimx {

//      CIN  COUT  MODE   TH  TW  WM WN  D  NS KS
#define DDIMX_WREG(X)                         \
    X(64, 64, CONV3, 8, 32, 2, 2, 6, 1, 1)       \
    X(96, 96, CONV3, 8, 32, 4, 3, 6, 1, 1)       \
    X(128, 128, CONV3, 4, 32, 2, 2, 8, 2, 1)     \
    X(192, 192, CONV3, 4, 16, 2, 2, 12, 3, 1)   \
    X(256, 256, CONV3, 4, 8, 1, 2, 12, 4, 2)     \
    X(32, 64, DOWN4, 8, 16, 2, 2, 8, 1, 1)       \
    X(64, 96, DOWN4, 4, 16, 2, 3, 8, 1, 1)       \
    X(96, 128, DOWN4, 4, 16, 2, 4, 8, 1, 1)      \
    X(128, 192, DOWN4, 4, 8, 1, 6, 8, 1, 1)     \
    X(192, 256, DOWN4, 4, 8, 1, 8, 8, 1, 1)      \
    X(256, 384, UP4, 4, 8, 1, 4, 8, 3, 1)        \
    X(192, 256, UP4, 4, 16, 2, 4, 8, 2, 1)       \
    X(128, 192, UP4, 4, 32, 1, 6, 8, 1, 1)       \
    X(96, 128, UP4, 4, 32, 1, 4, 6, 1, 1)        \
    X(64, 64, UP4, 8, 32, 2, 2, 6, 1, 1)

hipError_t wreg_geometry(int mode, int cin, int cout, WregGeom* g) {
#define DDIMX_G(CI, CO, MO, TH_, TW_, WM_, WN_, D_, NS_, KS_)                                                     \
    if (mode == MO && cin == CI && cout == CO) {                                                            \
        typedef WregCfg<CI, CO, MO, TH_, TW_, WM_, WN_, D_, NS_, KS_> F;                                         \
        g->th = F::TH; g->tw = F::TW; g->lds_bytes = F::LDS_BYTES; g->nthreads = F::NTHREADS; g->nsplit = F::NS; \
        return hipSuccess;                                                                                  \
    }
    DDIMX_WREG(DDIMX_G)
    return hipErrorInvalidValue;
}
hipError_t wreg_launch(int mode, int cin, int cout, const WregArgs& a, hipStream_t stream) {
#define DDIMX_L(CI, CO, MO, TH_, TW_, WM_, WN_, D_, NS_, KS_) \
    if (mode == MO && cin == CI && cout == CO) return launch_wreg_cfg<WregCfg<CI, CO, MO, TH_, TW_, WM_, WN_, D_, NS_, KS_>>(a, stream);
    DDIMX_WREG(DDIMX_L)
    return hipErrorInvalidValue;
}

}  // namespace ddimx
