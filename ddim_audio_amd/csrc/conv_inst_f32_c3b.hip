// Instantiations of conv_mfma_kernel with the GroupNorm-backward statistics epilogue (ConvCfg::BWD): the data-gradient convs of
// the training step, tile variants 0 and 1 of DDIMX_CONV3_F32 (see conv_configs.h).  Separate kernels, so that the
// inference / forward instantiations keep their register budgets.
#include "conv_mfma.h"
#include "conv_configs.h"

namespace ddimx {

template <typename T, int CIN, int NOUT, int NB, int MODE, int TH, int TW, int WM, int WN, int KC, int TPC, int VAR, int OVL>
static hipError_t launch_bwd(ConvArgs& a, hipStream_t stream) {
    if constexpr (VAR <= 1) return launch_conv_cfg<ConvCfg<T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, OVL, 1>>(a, stream);
    else return hipErrorInvalidValue;
}
#define DDIMX_LAUNCH_B(T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, VAR, OVL)                \
    if (mode == MODE && cin == CIN && nout == NOUT && var == VAR)                   \
        return launch_bwd<T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, VAR, OVL>(a, stream);

hipError_t conv_launch_f32_c3b(int mode, int cin, int nout, int var, ConvArgs& a, hipStream_t stream) {
    DDIMX_CONV3_F32(DDIMX_LAUNCH_B)
    return hipErrorInvalidValue;
}

}  // namespace ddimx
