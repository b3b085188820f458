// Launch wrappers of the backward-pass kernels (train_kernels.hip).  Same rules as kernels.h: enqueue on the given
// stream, never allocate or synchronise.
#pragma once
#include "common.h"

namespace ddimx {

// ---- GroupNorm backward around the fused convolutions (three steps; see train_kernels.hip) -----------------
// mode 0: the norm is fed by SiLU(u) (GN1, GN2); mode 1: the norm is followed by SiLU and fed by x = u (GN0)
hipError_t gn_bwd_stats_launch(int dtype, int mode, const void* g, const void* u, const float* scale, const float* shift,
                               float* stats /*[B][nparts][C][2]*/, int B, int HW, int C, hipStream_t s);
// mr: saved (mean, rstd) [B][8][2]; coef out [B][3][C]; dgb out [B][2][C] = per-sample (dgamma, dbeta) terms
hipError_t gn_bwd_finalize_launch(const float* stats, int nparts, int C, double count, const float* gamma, const float* mr,
                                  float* coef, float* dgb, int B, hipStream_t s);
// mode 0: out = (ca*g + cb*SiLU(u) + cc)*SiLU'(u), sums [B][nparts][C] of out (nullable)
// mode 1: out = gy + ca*(g*SiLU'(scale*u+shift)) + cb*u + cc (+ extra)
hipError_t gn_bwd_apply_launch(int dtype, int mode, const void* g, const void* u, const void* gy, const void* extra,
                               const float* coef, const float* scale, const float* shift, void* out, float* sums, int B,
                               int HW, int C, hipStream_t s);
hipError_t colsum_launch(const float* src, int B, long long stride, int C, float* dst, hipStream_t s);
hipError_t partsum_launch(const float* src, int B, int nparts, int C, float* dst, long long dst_stride, hipStream_t s);
// data-gradient weights of a 3x3 conv: dst[tap'][ci][co] = w[co][ci][8 - tap'] in the activation dtype
hipError_t pack_conv_dgrad_launch(int dtype, const float* w, void* dst, int O, int I, hipStream_t s);

}  // namespace ddimx
