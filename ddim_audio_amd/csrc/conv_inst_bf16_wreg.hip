// Instantiations of conv3_wreg_kernel (conv_wreg.h): 3x3 convolutions with the weights streamed straight into registers.
#include "conv_wreg.h"

namespace ddimx {

//                 C   TH  TW  WM WN  D  NS
#define DDIMX_WREG(X)                \
    X(64, 8, 32, 2, 2, 6, 1)         \
    X(96, 8, 32, 2, 3, 6, 1)         \
    X(128, 4, 32, 2, 2, 8, 2)        \
    X(192, 4, 16, 2, 2, 12, 3)       \
    X(256, 4, 8, 1, 2, 16, 4)

hipError_t wreg_geometry(int C, WregGeom* g) {
#define DDIMX_G(CC, TH_, TW_, WM_, WN_, D_, NS_)                                                                \
    if (C == CC) {                                                                                          \
        typedef WregCfg<CC, TH_, TW_, WM_, WN_, D_, NS_> F;                                                      \
        g->th = F::TH; g->tw = F::TW; g->lds_bytes = F::LDS_BYTES; g->nthreads = F::NTHREADS; g->nsplit = F::NS;                \
        return hipSuccess;                                                                                  \
    }
    DDIMX_WREG(DDIMX_G)
    return hipErrorInvalidValue;
}
hipError_t wreg_launch(int C, const WregArgs& a, hipStream_t stream) {
#define DDIMX_L(CC, TH_, TW_, WM_, WN_, D_, NS_) \
    if (C == CC) return launch_wreg_cfg<WregCfg<CC, TH_, TW_, WM_, WN_, D_, NS_>>(a, stream);
    DDIMX_WREG(DDIMX_L)
    return hipErrorInvalidValue;
}

}  // namespace ddimx
