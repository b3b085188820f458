// Instantiations of the weight-gradient kernel (wgrad_mfma.h) for T = float: one per (mode, a-channels, du-channels)
// pair of the reference network widths (configs/audio.yml:48: ch = [32, 64, 96, 128, 192, 256]).
#include "wgrad_mfma.h"

namespace ddimx {

#define DDIMX_WGRAD_TABLE(X) \
    X(CONV3, 32, 32) X(CONV3, 64, 64) X(CONV3, 96, 96) X(CONV3, 128, 128) X(CONV3, 192, 192) X(CONV3, 256, 256) \
    X(DOWN4, 32, 64) X(DOWN4, 64, 96) X(DOWN4, 96, 128) X(DOWN4, 128, 192) X(DOWN4, 192, 256)

hipError_t wgrad_geometry_f32(int mode, int ci, int co, WgradGeom* g) {
#define X(M, CI, CO)                                                   \
    if (mode == M && ci == CI && co == CO) {                           \
        typedef WgCfg<float, CI, CO, M> C;                                \
        g->th = C::TH; g->tw = C::TW; g->ntaps = C::NTAPS; g->grid_y = C::GRID_Y; \
        return hipSuccess;                                             \
    }
    DDIMX_WGRAD_TABLE(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t wgrad_launch_f32(int mode, int ci, int co, const WgradArgs& a, int nsplit, hipStream_t s) {
#define X(M, CI, CO) \
    if (mode == M && ci == CI && co == CO) return launch_wgrad_cfg<WgCfg<float, CI, CO, M>>(a, nsplit, s);
    DDIMX_WGRAD_TABLE(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace ddimx
