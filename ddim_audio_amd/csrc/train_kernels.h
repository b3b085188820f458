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
// mode 1: out = gy + ca*(g*SiLU'(scale*u+shift)) + cb*u + cc (+ extra); with nstats (and nu) also the slabs gn_bwd_stats_launch(mode 0)
//         would write for (g = out, u = nu): the first statistics pass of the block that takes `out` as its dy, bit for bit
hipError_t gn_bwd_apply_launch(int dtype, int mode, const void* g, const void* u, const void* gy, const void* extra,
                               const float* coef, const float* scale, const float* shift, void* out, float* sums, int B,
                               int HW, int C, hipStream_t s, const void* nu = nullptr, float* nstats = nullptr);
hipError_t colsum_launch(const float* src, int B, long long stride, int C, float* dst, hipStream_t s);
struct ColsumBatch {
    static constexpr int kMax = 96;
    const float* src[kMax]; float* dst[kMax]; long long stride[kMax]; int B[kMax]; int C[kMax]; int count;
};
hipError_t colsum_multi_launch(const ColsumBatch& q, hipStream_t s);
// dst[b][c] = sum_p src[((b*nparts + p)*C + c) * src_step]   (src_step = 2 reads the `sum` half of (sum, sumsq) slabs)
hipError_t partsum_launch(const float* src, int B, int nparts, int C, float* dst, long long dst_stride, hipStream_t s,
                          int src_step = 1);
// the same for many (src, dst) pairs in one launch (src_step 1): the backward defers the per-sample channel sums of its blocks
// (conv.1.bias and timestep-embedding terms: nothing on the data-gradient chain reads them) and flushes them with the batch sums
struct PartsumBatch {
    static constexpr int kMax = 32;
    const float* src[kMax]; float* dst[kMax]; long long dst_stride[kMax]; int nparts[kMax], C[kMax], B[kMax]; int count;
};
hipError_t partsum_multi_launch(const PartsumBatch& q, hipStream_t s);
// data-gradient weights of a 3x3 conv: dst[tap'][ci][co] = w[co][ci][8 - tap'] in the activation dtype
hipError_t pack_conv_dgrad_launch(int dtype, const float* w, void* dst, int O, int I, hipStream_t s);


// ---- FNet bottleneck, training ---------------------------------------------------------------------------
// seed_ctr (nullable, device memory): *seed_ctr is added to the seed when the kernel runs (graph-replayed training steps)
hipError_t dropout_apply_launch(const float* src, float* dst, long long n, float p, unsigned long long seed, unsigned stream,
                                hipStream_t s, const unsigned long long* seed_ctr = nullptr);
// y = LN(drop(x) + add[m % add_rows]); sum_out (nullable) keeps the pre-norm rows, stat [M][2] = (mean, rstd)
hipError_t ln_train_launch(int x_dtype, const void* x, const float* add, int add_rows, const float* gamma, const float* beta,
                           float eps, float* y, float* sum_out, float* stat, int M, int N, float p, unsigned long long seed,
                           unsigned stream, hipStream_t s, const unsigned long long* seed_ctr = nullptr);
int ln_bwd_nblocks(int M);
// partial: ln_bwd_nblocks(M) * 2 * N floats
hipError_t ln_bwd_launch(int x_dtype, const float* dy, const void* x, const float* add, int add_rows, const float* stat,
                         const float* gamma, float* dx, float* partial, float* dgamma, float* dbeta, int M, int N, hipStream_t s);
// mode 0: dst = gelu_new(src); mode 1: dst = src * gelu_new'(aux)
hipError_t gelu_launch(const float* src, const float* aux, float* dst, long long n, int mode, hipStream_t s);
hipError_t transpose_launch(const float* src, float* dst, int R, int C, int act_gelu, hipStream_t s);
hipError_t cast_f32_launch(int dtype, const void* src, float* dst, long long n, hipStream_t s);

// ---- timestep-embedding MLP backward -------------------------------------------------------------------------
hipError_t linear_bwd_w_launch(const float* dy, const float* x, const int64_t* idx, float* dW, float* db, int B, int N, int K,
                               int x_silu, hipStream_t s);
hipError_t linear_bwd_x_launch(const float* dy, const float* W, const float* xpre, float* dx, int B, int N, int K, hipStream_t s);

// ---- edge convolutions -----------------------------------------------------------------------------------------
hipError_t conv_out_bwd_data_launch(int dtype, const float* d_eps, const float* w /*packed [9][cout][C0]*/, void* ds, int B, int C0,
                                    int cout, int H, int W, hipStream_t s);
size_t edge_wgrad_partial_floats(int dtype, int B, int C, int NI, int H, int W);
// mode 0: input conv (G = d hidden[0], S = x); mode 1: output conv (G = g1 + g2 = x + hidden[0], S = d_eps)
hipError_t edge_wgrad_launch(int dtype, int mode, const void* g1, const void* g2, const float* S, float* partial, float* dW,
                             float* db, int B, int C, int NI, int H, int W, hipStream_t s);
// d_out[b] = 2 g[b] (out[b] - e[b]); g: upstream gradient of the per-sample losses [B]
hipError_t sqerr_bwd_launch(const float* e, const float* out, const float* g, float* d, int B, long long per, hipStream_t s, int with_mean = 0);

}  // namespace ddimx
