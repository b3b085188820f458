// Instantiations of conv_mfma_kernel: DDIMX_DOWNUP_F32 (see conv_configs.h).
#include "conv_mfma.h"
#include "conv_configs.h"

namespace ddimx {

#define DDIMX_GEOM(T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, VAR, OVL)                  \
    if (mode == MODE && cin == CIN && nout == NOUT && var == VAR) {                               \
        g->th = TH; g->tw = TW; g->nb = NB; g->nout = NOUT; g->classes = (MODE == UP4 ? 2 : 1); \
        g->lds_bytes = ConvCfg<T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, OVL>::LDS_BYTES;               \
        g->nthreads = ConvCfg<T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, OVL>::NTHREADS;                 \
        return hipSuccess;                                                          \
    }
#define DDIMX_LAUNCH(T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, VAR, OVL)                \
    if (mode == MODE && cin == CIN && nout == NOUT && var == VAR)                   \
        return launch_conv_cfg<ConvCfg<T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, OVL>>(a, stream);

hipError_t conv_geometry_f32_du(int mode, int cin, int nout, int var, ConvGeom* g) {
    DDIMX_DOWNUP_F32(DDIMX_GEOM)
    return hipErrorInvalidValue;
}
hipError_t conv_launch_f32_du(int mode, int cin, int nout, int var, ConvArgs& a, hipStream_t stream) {
    DDIMX_DOWNUP_F32(DDIMX_LAUNCH)
    return hipErrorInvalidValue;
}

}  // namespace ddimx
