// Instantiations of conv3_fold_kernel (conv_fold.h): 3x3 convolutions whose GroupNorm input is folded into the weights.
#include "conv_fold.h"

namespace ddimx {

//                C   TH  TW  WM WN
typedef FoldCfg<32, 8, 32, 4, 1> Fold32;

hipError_t fold_geometry(int C, FoldGeom* g) {
    if (C == 32) { g->th = Fold32::TH; g->tw = Fold32::TW; g->lds_bytes = Fold32::LDS_BYTES; g->nthreads = Fold32::NTHREADS; return hipSuccess; }
    return hipErrorInvalidValue;
}
hipError_t fold_launch(int C, const FoldArgs& a, hipStream_t stream) {
    if (C == 32) return launch_fold_cfg<Fold32>(a, stream);
    return hipErrorInvalidValue;
}

}  // namespace ddimx
