// Launch wrappers of the non-MFMA kernels (kernels.hip) and the MFMA GEMM (gemm.hip).
// All enqueue on the given stream, never allocate or synchronise (graph-capturable).
#pragma once
#include "common.h"
#include "gn_fused.h"

namespace ddimx {

// ---- U-Net edge convolutions (C_io = 2 side; HBM-bound; layout conversion NCHW fp32 <-> NHWC T) ----
// in-conv: reference models/diffusion.py:189-198.  x [B][2][H][W] fp32 -> out [B][H][W][C0] T, + stats partials
hipError_t conv_in_launch(int dtype, const float* x, const float* w /*[C0][cin][3][3]*/, const float* bias, void* out,
                          float* stats, int B, int cin, int C0, int H, int W, hipStream_t s, int groups = 0);
// groups = 1 (here and below): the statistics partials are written folded to the 8 groups, [B][nparts][8][2] (gn_fused.h)
int conv_in_nparts(int H, int W);
// out-conv: models/diffusion.py:199-208 preceded by x + hidden[0] (:284).  (a + b) NHWC T -> eps [B][cout][H][W] fp32
hipError_t conv_out_launch(int dtype, const void* a, const void* b, const float* w /*packed [9][cout][C0] fp32*/,
                           const float* bias, float* out, int B, int C0, int cout, int H, int W, hipStream_t s);

// ---- GroupNorm statistics -> folded per-(sample, channel) scale / shift ---------------------------
// stats [B][nparts][Cs][2]; channel vc of the slab is real channel vc % C.  count = elements per group.
hipError_t gn_finalize_launch(const float* stats, int nparts, int Cs, int C, double count, const float* gamma,
                              const float* beta /*nullable*/, float eps, float* scale, float* shift, int B,
                              hipStream_t s, float* mean_rstd_out /*[B][8][2], nullable*/ = nullptr);
// the same from group-format partials (gn.stats [B][gn.np][8][2])
hipError_t gn_finalize_groups_launch(const GnIn& gn, int C, float* scale, float* shift, int B, int nthreads, hipStream_t s);
int resid_threads(int dtype, int C);  // block size of resid_kernel / tensor_stats for C channels

// ---- residual pass: y = x + (h*scale + shift)  (block tail, models/diffusion.py:54-56), or y = x + h ----
// h_f32 = 1: h is fp32 (FNet output) and no affine is applied; h_f32 = 2: y = x + SiLU(h)*scale + shift (training
// forward, h = pre-activation).  stats nullable.  Elements per sample = HW*C.
// gn != null (gn->stats set): scale / shift are derived in-kernel from the group partials of h (consumer-side finalisation)
hipError_t resid_launch(int dtype, const void* x, const void* h, int h_f32, const float* scale, const float* shift,
                        void* y, float* stats, int B, int HW, int C, hipStream_t s, const GnIn* gn = nullptr, int groups = 0);
int resid_nparts(int dtype, int HW, int C);
int resid_iters(int dtype, int HW, int C);  // 16-byte pieces per thread of the element-wise passes (sample size only)
// per-channel (sum, sumsq) partials of an NHWC tensor, same partitioning as resid_nparts
hipError_t tensor_stats_launch(int dtype, const void* x, float* stats, int B, int HW, int C, hipStream_t s, int groups = 0);
hipError_t to_nhwc_launch(int dtype, const float* in, void* out, int B, int C, int HW, hipStream_t s);
hipError_t from_nhwc_launch(int dtype, const void* in, float* out, int B, int C, int HW, hipStream_t s);

// ---- small dense layers (timestep embedding MLP, models/diffusion.py:110-120) ------------------------
// y[b][n] = act(sum_k x[row(b)][k] * W[n][k] + bias[n]); row(b) = idx ? idx[b] : b
// in_silu: SiLU is applied to x while it is read (training keeps the pre-activations)
hipError_t linear_rows_launch(const float* x, const int64_t* idx, const float* W, const float* bias, float* y, int B,
                              int N, int K, int act_silu, hipStream_t s, int in_silu = 0);

hipError_t temb_gather_launch(const float* table /*[n_timesteps][E]*/, const int64_t* t, float* out, int B, int E, hipStream_t s);

// ---- LayerNorm over rows -----------------------------------------------------------------------------
// y = LN(x [+ add[(m % add_rows)]]) * gamma + beta;  x is T (dtype) or fp32 (dtype = DT_F32)
// chunk_rows > 0: y is written chunk-major for fnet_dense_kernel, rows of a sample = chunk_rows (<= 32)
hipError_t layernorm_launch(int x_dtype, const void* x, const float* add, int add_rows, const float* gamma,
                            const float* beta, float eps, float* y, int M, int N, hipStream_t s, int chunk_rows = 0);

// ---- "NT" GEMM (gemm.hip): C[z][M][N] (+)= A[z][M][K] * B[z][N][K]^T; fp32 or bf16 MFMA, optional split-K ----
struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;      // [N] or null
    const float* resid;     // same layout as C, added in the epilogue, or null
    float* partial;         // split-K workspace [batch*splitk][M][N] (required when splitk > 1)
    int M, N, K, lda, ldb, ldc;
    long long sA, sB, sC;   // batch strides (elements)
    int batch;
    int splitk;             // K split over blockIdx.z (1 = none)
    int accumulate;         // C += ...
    int act;                // 0 none, 1 gelu_new
    int bf16;               // 1: round operands to bf16 while staging, v_mfma_f32_32x32x16_bf16
};
hipError_t gemm_launch(GemmArgs g, hipStream_t s);
int gemm_pick_splitk(int M, int N, int K, int batch, int bf16);
// Z[b] = Re(FFT2(X[b])) + X[b] in one launch (gemm.hip); dft_hidden [2hid][hid] interleaved cos/sin rows, dft_seq [S][2S]
bool fnet_mix_supported(int S, int hid);
// zc / zstats (nullable, together): ALSO write Z chunk-major ([B][hid/4][32 rows][4], fnet_dense.hip) and, per row and
// workgroup column block (16 features), the pair (sum, centred sum of squares) of Z as [B][hid/32][32 rows][2 x 2] -- the
// LayerNorm statistics of the rows for fnet_dense_kernel; Z itself (row-major) may then be null
hipError_t fnet_mix_launch(const float* dft_hidden, const float* dft_seq, const float* X, float* Z, int B, int S, int hid,
                           hipStream_t s, float* zc = nullptr, float* zstats = nullptr);

// ---- dense layers of the FNet at S <= 32 without split-K workspace / LayerNorm launches (fnet_dense.hip) ----------------
// Layouts: "chunk-major" = [sample][k / 4][32 rows][4 fp32] (bf16: [k / 8][32][8]); statistics [sample][part / 2][32][2 x 2].
struct FnetDenseArgs {
    const void* W;         // FRAGMENT order (fnet_fold_launch); bf16 when the launch is bf16, else fp32
    const float* bias;     // [N]
    const void* X;         // tokens: row-major fp32 [B*S][K], or chunk-major (x_chunk) fp32 / bf16 (x_bf16)
    const float* xstats;   // non-null: the operand is (x - mean_row) * rstd_row, from xnp parts of xn elements each
    int xnp, xn;
    void* out;             // row-major fp32 [B*S][N], or chunk-major (out_chunk) fp32 / bf16 (out_bf16)
    int x_chunk, x_bf16, out_chunk, out_bf16;
    int act;               // 1: gelu_new
    const float* R;        // non-null (chunk-major fp32): + LayerNorm(R)[row][n] * rgamma[n] + rbeta[n], statistics rstats
    const float* rstats; const float* rgamma; const float* rbeta;
    int rnp, rn;
    float* ostats;         // nullable: row statistics of `out` as written, gridDim.x parts of 32 or 64 features
    float eps;
    int S, K, N;
};
// Fourier mixing over those layouts with the previous layer's output LayerNorm taken on the fly (fnet_dense.hip)
struct FnetMixArgs {
    const float* tab;      // hidden-DFT table of this layer, gamma folded in, fragment order (fnet_table_launch)
    const float* dft_seq;  // [S][2S] = [cos | -sin]
    const float* V;        // chunk-major fp32 input rows
    const float* vstats;   // their statistics (16 parts of hid / 16); null: the rows are used as they are
    const float* gamma; const float* beta; const float* bc;  // LayerNorm affine and C_H beta (with vstats)
    float* zc; float* zstats;  // chunk-major Z and its row statistics (hid / 16 parts of 16)
    float eps;
    int S, hid;
};
hipError_t fnet_mix2_launch(const FnetMixArgs& a, int B, hipStream_t s);
hipError_t fnet_table_launch(const float* gamma, const float* beta, float* tab, float* bc, int H, hipStream_t s);
bool fnet_dense_supported(int S, int K, int N);
hipError_t fnet_dense_launch(const FnetDenseArgs& a, int B, int bf16, hipStream_t s);
// Wf = W * diag(gamma) (gamma null: W) in MFMA fragment order, optionally rounded to bf16; bf = bias + W * beta (beta null: not written)
hipError_t fnet_fold_launch(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, int wf_bf16,
                            float* bf, int N, int K, hipStream_t s);
// out = LayerNorm(A*B^T + bias + resid) * gamma + beta (rows of N <= 2048), GEMM via the partial workspace
hipError_t gemm_ln_launch(GemmArgs g, const float* gamma, const float* beta, float eps, float* out, hipStream_t s);

// ---- sampler / training elementwise ------------------------------------------------------------------
// coef rows: (t, sqrt(1-at), sqrt(at), sqrt(at_next), c2, c1) fp32; step is a device counter
hipError_t step_begin_launch(const float* coef, const int* step, int64_t* t, int B, int stride, hipStream_t s);
hipError_t step_end_launch(int* step, hipStream_t s);
hipError_t ddim_update_launch(float* xt, const float* et, const float* noise, float* x0, const float* coef,
                              const int* step, long long n, hipStream_t s);
hipError_t ddpm_update_launch(const float* x, const float* e, const float* noise, float* x0, float* xn, const float* coef,
                              const int* step, long long n, hipStream_t s);
hipError_t qsample_launch(const float* x0, const float* e, const float* alphas, const int64_t* t, float* x, int B,
                          long long per, hipStream_t s);
hipError_t sqerr_launch(const float* e, const float* out, float* partial, float* loss_per, int B, long long per,
                        hipStream_t s);
int sqerr_nparts();
hipError_t ema_multi_launch(const long long* shadow_ptrs, const long long* param_ptrs, const long long* sizes,
                            const int* blk_tensor, const long long* blk_off, int nblocks, float mu, hipStream_t s);
int ema_block_elems();
// training-step tail (multi-tensor, pointer tables as for ema_multi)
hipError_t grad_norm_multi_launch(const long long* ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                                  int nblocks, float max_norm, float* partial, float* out, hipStream_t s);
struct AdamArgs {
    const long long* p; const long long* g; const long long* m; const long long* v; const long long* sizes;
    const int* blk_tensor; const long long* blk_off; const float* clip;
    float lr, b1, b2, eps, wd, bc1, bc2s; int decoupled;
    const float* dyn;  // nullable, device: {lr, bc1, bc2s} read when the kernel RUNS (graph-replayed steps) instead of the values above
};
hipError_t adam_multi_launch(const AdamArgs& a, int nblocks, hipStream_t s);
hipError_t scale_multi_launch(const long long* ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                              int nblocks, const float* coef, hipStream_t s);

// ---- weight packing ------------------------------------------------------------------------------------
hipError_t pack_copy_launch(const float* src, float* dst, long long n, hipStream_t s);
struct PackCopyBatch {
    static constexpr int kMax = 96;
    const float* src[kMax]; float* dst[kMax]; long long n[kMax]; int count;
};
hipError_t pack_copy_multi_launch(const PackCopyBatch& b, hipStream_t s);
struct PackConvBatch {
    static constexpr int kMax = 64;
    const float* src[kMax]; void* dst[kMax]; int O[kMax], I[kMax], KK[kMax]; unsigned char mode[kMax], f32[kMax]; int count;
    // queues one packing; launches the batch when it is full
    hipError_t push(const float* w, void* d, int o, int i, int kk, int md, int dtype, hipStream_t s);
};
hipError_t pack_conv_multi_launch(const PackConvBatch& b, hipStream_t s);
inline hipError_t PackConvBatch::push(const float* w, void* d, int o, int i, int kk, int md, int dtype, hipStream_t s) {
    src[count] = w; dst[count] = d; O[count] = o; I[count] = i; KK[count] = kk; mode[count] = (unsigned char)md;
    f32[count] = dtype == DT_F32;
    if (++count == kMax) { hipError_t e = pack_conv_multi_launch(*this, s); count = 0; return e; }
    return hipSuccess;
}
hipError_t pack_conv_launch(int dtype, const float* w /*[O][I][KH][KW]*/, void* dst /*[KH*KW][O][I]*/, int O, int I,
                            int KH, int KW, hipStream_t s);
hipError_t pack_convT_launch(int dtype, const float* w /*[I][O][4][4]*/, void* dst /*[2][6][2*O][I]*/, int I, int O,
                             hipStream_t s);
// dst[r][f*C + c] = src[r][c*Fr + f]   (token-order permutation of the FNet boundary, rows r)
hipError_t pack_perm_cols_launch(const float* src, float* dst, int rows, int C, int Fr, hipStream_t s);
// dst[(f*C + c)][k] = src[(c*Fr + f)][k]
hipError_t pack_perm_rows_launch(const float* src, float* dst, int C, int Fr, int K, hipStream_t s);

}  // namespace ddimx
