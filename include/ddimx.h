/* libddimx -- C ABI of the MI355X-native DDIM denoising hot path (gfx950).
 *
 * The reference (klae01/ddim-audio) is pure Python and has no FFI of its own; its boundary for this
 * path is the Python API  Model(config) / model(x, t) / generalized_steps(...)  (SURVEY.md section 8b).
 * This header is the native interface introduced *underneath* that API: each entry point names the
 * reference code whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; ddimx_last_error() gives the message;
 *     nothing aborts or exits (the reference's main.py logs exceptions, main.py:212-223).
 *   - all pointers except `ddimx_config*`, host pointer arrays and host scalars are DEVICE pointers
 *     owned by the caller (PyTorch); the library borrows them for the duration of one call, never
 *     allocates device memory, never synchronises: every call only enqueues work on `stream`
 *     (a hipStream_t passed as void*), so call sequences can be captured into a hipGraph.
 *   - activations inside the network are NHWC ([B][T'][F'][C], channels innermost) in the
 *     activation dtype (DDIMX_F32 or DDIMX_BF16); the network boundary is the reference's
 *     NCHW fp32 [B][C][T][F] (models/diffusion.py:238-240).
 */
#ifndef DDIMX_H
#define DDIMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDIMX_ABI_VERSION 2
#define DDIMX_MAX_LEVELS 8
#define DDIMX_F32 0
#define DDIMX_BF16 1

/* config.model.* and config.diffusion.num_diffusion_timesteps as read by Model.__init__
 * (models/diffusion.py:170-235; configs/audio.yml:25-61). */
typedef struct {
    int in_channels;                 /* model.channels (2) */
    int f_size;                      /* model.f_size (256) */
    int n_levels;                    /* len(model.ch) */
    int ch[DDIMX_MAX_LEVELS];        /* model.ch */
    int res[DDIMX_MAX_LEVELS];       /* model.res */
    int krn[DDIMX_MAX_LEVELS];       /* model.krn (3 only) */
    int n_timesteps;                 /* diffusion.num_diffusion_timesteps */
    int fnet_hidden;                 /* transformers.kwargs.hidden_size */
    int fnet_layers;                 /* transformers.kwargs.num_hidden_layers */
    int fnet_inter;                  /* transformers.kwargs.intermediate_size */
    float fnet_ln_eps;               /* transformers.kwargs.layer_norm_eps */
    int act_dtype;                   /* model.dtype: DDIMX_F32 (parity mode) or DDIMX_BF16 */
    int fnet_dtype;                  /* transformers.dtype (models/diffusion.py:242-246): operand type of the FNet's dense-weight
                                        GEMMs.  DDIMX_F32 = the reference's mixed mode (convs in act_dtype, transformer fp32);
                                        DDIMX_BF16 = operands rounded to bf16 (needs act_dtype == DDIMX_BF16).  The DFT factors,
                                        LayerNorms and accumulators are fp32 either way. */
} ddimx_config;

/* Host-built constant tables for a given T (device pointers, fp32):
 *   posenc     [S][width]    Add_Encoding table (models/diffusion.py:81-92,131-140) in NHWC token order
 *   dft_hidden [2*hid][hid]  interleaved rows: 2k = cos(2 pi k n / hid), 2k+1 = sin(2 pi k n / hid)
 *   dft_seq    [S][2*S]      row k = [cos(2 pi k n / S), n < S | -sin(2 pi k n / S), n < S]
 * with S = T / 2^(n_levels-1), width = ch[-1] * f_size / 2^(n_levels-1). */
typedef struct {
    const float* posenc;
    const float* dft_hidden;
    const float* dft_seq;
    const float* temb_table; /* optional (may be null): [n_timesteps][E] = BetaEmbedding(t) for every t (models/diffusion.py:
                                110-120), built once per weight set with ddimx_temb_fwd over t = 0..n_timesteps-1; when given,
                                ddimx_unet_fwd copies row t[b] instead of running the three-layer MLP (eval mode only: the
                                rows are bit-identical to the MLP's, which computes every row independently) */
} ddimx_tables;

typedef struct ddimx_ctx* ddimx_handle;

int ddimx_abi_version(void);
const char* ddimx_last_error(void);

/* Model.__init__ (models/diffusion.py:170-235): validates the configuration, builds the packing and
 * launch plan.  No device memory is allocated. */
int ddimx_create(const ddimx_config* cfg, ddimx_handle* out);
int ddimx_destroy(ddimx_handle h);
/* Training under hipGraph replay: *counter (device memory, or null to switch off) is added to the dropout seed of every
 * launch of the training forward / backward when it RUNS; an eager step passes a fresh seed by value instead. */
int ddimx_set_dropout_counter(ddimx_handle h, const unsigned long long* counter);

/* Number of state_dict entries (388 parameters + temb.te = 389 for configs/audio.yml), in the
 * reference's registration order; name/shape of entry i for cross-checking the host mirror. */
int ddimx_num_params(ddimx_handle h);
int ddimx_param_info(ddimx_handle h, int i, const char** name, long long* numel);

/* Bytes of the packed-weight buffer / of the workspace for a [B,2,T,F] forward. */
long long ddimx_packed_bytes(ddimx_handle h);
long long ddimx_workspace_bytes(ddimx_handle h, int B, int T);

/* Convert the fp32 parameters (device pointers, state_dict order) to the internal layouts
 * (implicit-GEMM conv weights in the activation dtype, sub-pixel ConvTranspose weights, FNet boundary
 * in NHWC token order).  Must be re-run whenever parameters change (optimizer step, load_state_dict,
 * EMAHelper.ema: models/ema.py:25-30). */
int ddimx_pack_weights(ddimx_handle h, const void* const* params, int n_params, void* packed, void* stream);
/* Inference-only second copies, derived from what ddimx_pack_weights wrote into `packed`; call it after ddimx_pack_weights when the
 * weights are packed for eval mode (a training step re-packs after every optimizer step and reads none of them):
 * - the conv weights of Residual_Block / Downsample / Upsample in MFMA fragment order (csrc/conv_wreg.h, conv_pipe.h;
 *   models/diffusion.py:28-40,59-78) -- without them the convolutions run through conv_mfma_kernel's LDS weight path;
 * - the FNet weights for the launch-lean bottleneck path (csrc/fnet_dense.hip: fragment order, the LayerNorms' gamma / beta folded
 *   into the next matrix, one hidden-DFT table per layer; models/diffusion.py:131-167, modeling_fnet.py:138-279) -- without them
 *   the forward takes the GEMM path: 77 launches instead of 39.
 * Same results within rounding either way; a later ddimx_pack_weights into the same buffer invalidates both. */
int ddimx_pack_fnet_inference(ddimx_handle h, void* packed, void* stream);

/* Model.forward (models/diffusion.py:237-294), eval mode: x [B][C][T][F] fp32, t [B] int64 -> eps. */
int ddimx_unet_fwd(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace,
                   long long workspace_bytes, const float* x, const int64_t* t, float* eps, int B, int T,
                   void* stream);

/* The same forward with part of the network run as two batch shards on two streams (samples [0, B/2) on `stream`, [B/2, B) on
 * `aux_stream`, forked and joined with caller-owned hipEvent_t -- one per fork and one per join, at most 2 * n_levels + 4 of them,
 * none recorded twice in a call; everything is joined into `stream` on return, and the calls are capturable into one hipGraph).  fork_mask bit l (0 <= l < n_levels): the ops whose output lives on level l run
 * sharded; bit 16: the FNet bottleneck.  Every op of the path is per sample (GroupNorm / LayerNorm / FFT are per sample:
 * models/diffusion.py:42-56,148-167), so the result is bit-identical for every mask; aux_stream = NULL or fork_mask = 0 is
 * ddimx_unet_fwd. */
int ddimx_unet_fwd_forked(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace, long long workspace_bytes,
                          const float* x, const int64_t* t, float* eps, int B, int T, void* stream, void* aux_stream,
                          void* const* events, int n_events, unsigned fork_mask);

/* ---- training: forward that keeps a tape + whole-network backward ------------------------------------
 * The reference trains through autograd: functions/losses.py:12-18 builds the graph of Model.forward
 * (models/diffusion.py:237-294, training mode: dropout hidden_dropout_prob after the FNet projection and after each FNet
 * FFN) and runners/diffusion.py:150 `loss.backward()` walks it.  Here the forward stores, per Residual_Block, the two
 * pre-activation conv outputs and the GroupNorm constants (plus the FNet rows) in `tape`, and ddimx_unet_bwd turns
 * d_eps into all 388 parameter gradients.
 *   packed_bwd : extra weight packings of the backward (data-gradient conv layouts, transposed FNet matrices), built by
 *                ddimx_pack_weights_bwd from the same parameter tensors + the forward's packed buffer;
 *   tape       : ddimx_train_tape_bytes(B, T) bytes, written by the forward, read by the backward;
 *   workspace  : ddimx_train_workspace_bytes(B, T) bytes of scratch (shared by both calls);
 *   grads      : fp32 [ddimx_grad_floats()], parameter i (plan order, as ddimx_param_info) at ddimx_grad_offset(i) in the
 *                parameter's own shape; gradients are WRITTEN (not accumulated); the temb.te buffer's slot is untouched;
 *   dropout    : masks are a pure function of (seed, layer, element), so the backward regenerates them: pass the same
 *                (dropout_p, seed) to both calls.  dropout_p = 0 gives the deterministic function the parity tests use.
 * The gradient w.r.t. the input x is not produced (the training step never needs it). */
long long ddimx_packed_bwd_bytes(ddimx_handle h);
int ddimx_pack_weights_bwd(ddimx_handle h, const void* const* params, int n_params, const void* packed, void* packed_bwd,
                           void* stream);
long long ddimx_train_tape_bytes(ddimx_handle h, int B, int T);
long long ddimx_train_workspace_bytes(ddimx_handle h, int B, int T);
long long ddimx_grad_floats(ddimx_handle h);
long long ddimx_grad_offset(ddimx_handle h, int i);
int ddimx_unet_fwd_train(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace,
                         long long workspace_bytes, void* tape, long long tape_bytes, const float* x, const int64_t* t,
                         float* eps, int B, int T, float dropout_p, unsigned long long seed, void* stream);
int ddimx_unet_bwd(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                   long long workspace_bytes, const void* tape, long long tape_bytes, const float* x, const int64_t* t,
                   const float* d_eps, float* grads, int B, int T, float dropout_p, unsigned long long seed, void* stream);
/* Data-parallel training (the reference has no counterpart: runners/diffusion.py:216 is a commented DataParallel):
 * the same backward, which additionally records three caller-owned hipEvent_t on `stream` as soon as a bucket of the flat
 * gradient buffer is final -- bucket 0 = up_modules.* (ready after the up path), 1 = transformer.* (after the bottleneck),
 * 2 = temb.* + down_modules.* (at the end) -- so that the caller can all-reduce each bucket on a second stream while the rest of
 * the backward still runs.  ddimx_grad_buckets fills ranges[6] = {begin0, end0, begin1, end1, begin2, end2} (float offsets into
 * the flat gradient buffer, each bucket one contiguous run).  n_events must be 0 (plain ddimx_unet_bwd) or 3. */
int ddimx_grad_buckets(ddimx_handle h, long long* ranges);
int ddimx_unet_bwd_staged(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                          long long workspace_bytes, const void* tape, long long tape_bytes, const float* x, const int64_t* t,
                          const float* d_eps, float* grads, int B, int T, float dropout_p, unsigned long long seed,
                          void* const* bucket_events, int n_events, void* stream);
/* The same backward with its weight gradients on a second stream (runners/diffusion.py:150 `loss.backward()`; autograd has no
 * counterpart -- it runs one stream).  A conv's weight gradient feeds nothing but its parameter's slot, so it leaves the
 * data-gradient chain: each is issued on `side_stream` behind an event of `side_events` (caller-owned hipEvent_t, at least
 * ddimx_bwd_side_events(h) of them, none re-recorded within one call so that the call can be captured into a hipGraph) while the
 * chain goes on; the branch is joined into `stream` before the call returns.  Results are bit-identical to ddimx_unet_bwd_staged
 * (same kernels, partitions and order of additions).  With n_events = 3, bucket 0's event is recorded on `side_stream` (behind the
 * up path's last weight gradient and the chain's batch sums), buckets 1 and 2 on `stream`.  side_stream null: one stream. */
int ddimx_bwd_side_events(ddimx_handle h);
int ddimx_unet_bwd_forked(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                          long long workspace_bytes, const void* tape, long long tape_bytes, const float* x, const int64_t* t,
                          const float* d_eps, float* grads, int B, int T, float dropout_p, unsigned long long seed,
                          void* const* bucket_events, int n_events, void* stream, void* side_stream, void* const* side_events,
                          int n_side_events);
/* backward of the per-sample squared-error loss (functions/losses.py:18): d_out[b] = 2 g[b] (out[b] - e[b]) */
int ddimx_sqerr_loss_bwd(const float* e, const float* out, const float* g_per_sample, float* d_out, int B,
                         long long per_sample, void* stream);
/* the same against the gradient of ddimx_sqerr_loss's whole [B + 1] vector (per-sample losses, then their batch mean --
 * functions/losses.py:16-18 keepdim / mean): d_out[b] = 2 (g[b] + g[B] / B) (out[b] - e[b]) */
int ddimx_sqerr_loss_bwd_mean(const float* e, const float* out, const float* g, float* d_out, int B, long long per_sample, void* stream);

/* ---- per-op entry points (same kernels as ddimx_unet_fwd; used by the parity tests) ------------- */
/* layout helpers: NCHW fp32 <-> NHWC activation dtype */
int ddimx_to_nhwc(int dtype, const float* nchw, void* nhwc, int B, int C, int H, int W, void* stream);
int ddimx_from_nhwc(int dtype, const void* nhwc, float* nchw, int B, int C, int H, int W, void* stream);
/* weight packing of single layers: Conv2d [O][I][KH][KW] -> [KH*KW][O][I]; ConvTranspose2d(k4,s2,p1)
 * [I][O][4][4] -> [2][6][2*O][I] (sub-pixel form, see csrc/kernels.hip) */
int ddimx_pack_conv(int dtype, const float* w, void* dst, int O, int I, int KH, int KW, void* stream);
int ddimx_pack_convT(int dtype, const float* w, void* dst, int I, int O, void* stream);
long long ddimx_op_workspace_bytes(int dtype, int B, int C, int H, int W);

/* Residual_Block.forward (models/diffusion.py:42-56) on NHWC x -> y (may alias x).  gn*_ are fp32
 * [C]; w0/w1 packed by ddimx_pack_conv; temb [B][temb_stride] fp32 (already offset to this block). */
int ddimx_resblock_fwd(int dtype, int C, const void* x, void* y, const float* temb, int temb_stride,
                       const float* gn0_w, const float* gn0_b, const void* w0, const float* gn1_w,
                       const float* gn1_b, const void* w1, const float* bias1, const float* gn2_w, void* workspace,
                       int B, int H, int W, void* stream);
/* ---- training (autograd of the same reference code: runners/diffusion.py:150 `loss.backward()`) ----
 * Residual_Block forward that keeps what its backward needs: u1 = conv.0(...) + temb and u2 = conv.1(...) + bias
 * (PRE-activation, NHWC [B][H][W][C] in the activation dtype) and tape_small (ddimx_rb_tape_floats(B, C) floats:
 * folded GroupNorm scale/shift and (mean, rstd) of the three norms).  Same result as ddimx_resblock_fwd. */
long long ddimx_rb_tape_floats(int B, int C);
int ddimx_resblock_fwd_train(int dtype, int C, const void* x, void* y, const float* temb, int temb_stride,
                             const float* gn0_w, const float* gn0_b, const void* w0, const float* gn1_w,
                             const float* gn1_b, const void* w1, const float* bias1, const float* gn2_w, void* u1,
                             void* u2, float* tape_small, void* workspace, int B, int H, int W, void* stream);
/* data-gradient packing of a 3x3 Conv2d.weight [O][I][3][3]: dst[tap][I][O] = w[.., 8 - tap] (activation dtype) */
int ddimx_pack_conv_dgrad(int dtype, const float* w, void* dst, int O, int I, void* stream);
/* Backward of Residual_Block (models/diffusion.py:42-56): dy -> dx and the gradients of its 8 parameters (fp32, the
 * parameters' own shapes, WRITTEN not accumulated) and of its timestep-embedding chunk d_temb [B][d_temb_stride]
 * (nullable).  workspace: ddimx_resblock_bwd_workspace_bytes(). */
long long ddimx_resblock_bwd_workspace_bytes(int dtype, int B, int C, int H, int W);
int ddimx_resblock_bwd(int dtype, int C, const void* x, const void* u1, const void* u2, const float* tape_small,
                       const void* dy, void* dx, const float* gn0_w, const float* gn1_w, const float* gn2_w,
                       const void* w0_dgrad, const void* w1_dgrad, float* d_gn0_w, float* d_gn0_b, float* d_w0,
                       float* d_gn1_w, float* d_gn1_b, float* d_w1, float* d_bias1, float* d_gn2_w, float* d_temb,
                       int d_temb_stride, void* workspace, int B, int H, int W, void* stream);
/* One fused 3x3 convolution of a Residual_Block (models/diffusion.py:28-40,46-53): the GroupNorm
 * affine (xf = 1) or affine + SiLU (xf = 2) folded as per-(sample, channel) scale/shift [B][C] is applied
 * to the input while it is staged; epilogue adds bias [C] and chan_add [B][chan_add_stride] (either may
 * be null), applies SiLU when act = 1, and writes per-channel (sum, sumsq) partials to stats (nullable;
 * ddimx_conv3x3_stats_floats() floats).  x, y: NHWC [B][H][W][C]. */
int ddimx_conv3x3_fwd(int dtype, int C, const void* x, const void* w, const float* bias, const float* chan_add,
                      int chan_add_stride, const float* in_scale, const float* in_shift, int xf, int act, void* y,
                      float* stats, int B, int H, int W, void* stream);
long long ddimx_conv3x3_stats_floats(int dtype, int C, int B, int H, int W);
/* The same convolution through the register-streamed-weights kernel (csrc/conv_wreg.h; bf16, C in {64, 96, 128, 192, 256}, whole
 * tiles): w_frag = the weights in MFMA fragment order, ddimx_pack_conv_frag(w [O][I][3][3] fp32 -> 9*O*I bf16).  This is what
 * ddimx_unet_fwd launches for the Residual_Block convs (models/diffusion.py:46-53) of those widths; fails if the shape is not
 * eligible (ddimx_conv3x3_fwd is the general entry). */
int ddimx_pack_conv_frag(const float* w, void* dst, int O, int I, void* stream);
/* Diagnostic builds (-DDDIMX_STAMP) only: ddimx_conv3x3_wreg_fwd writes its per-wave phase stamps here (null: off). */
int ddimx_debug_set_stamps(unsigned long long* stamps);
/* The same for any tap count (KK = KH * KW; Downsample: 16), and Downsample.forward (models/diffusion.py:70-78) through the
 * register-streamed kernel: x [B][H][W][Cin] bf16 -> y [B][H/2][W/2][Cout], w_frag = ddimx_pack_conv_frag_k(w [Cout][Cin][4][4], 16);
 * stats: per-channel (sum, sumsq) partials of y (sized for one partial per 32 output pixels), nullable. */
int ddimx_pack_conv_frag_k(const float* w, void* dst, int O, int I, int KK, void* stream);
int ddimx_downsample_wreg_fwd(int Cin, int Cout, const void* x, const void* w_frag, const float* bias, void* y, float* stats, int B,
                              int H, int W, void* stream);
/* Upsample.forward + the skip addition (models/diffusion.py:59-67,284) through the register-streamed kernel: w_frag = both
 * row-parity classes of the sub-pixel form (ddimx_pack_convT) re-ordered by ddimx_pack_frag_from_taps(class a of the convT
 * packing, 6 taps, NOUT = 2 * Cout, Cin) one after the other; bias2, skip, y, stats as ddimx_upsample_add_fwd. */
int ddimx_pack_frag_from_taps(const void* taps, void* dst, int ntaps, int NOUT, int CIN, void* stream);
int ddimx_upsample_add_wreg_fwd(int Cin, int Cout, const void* x, const void* w_frag, const float* bias2, const void* skip, void* y,
                                float* stats, int B, int H, int W, void* stream);
/* Both convs of Residual_Block (models/diffusion.py:46-53: conv(SiLU(GN(x))) + temb -> SiLU, conv(GN(h)) + bias -> SiLU) through
 * the software-pipelined kernel (csrc/conv_pipe.h: weights resident in registers, the next tile's halo transform and the previous
 * block's epilogue issued in the MFMA gaps of the same wave; bf16, C = 32 / 64, H and W whole numbers of its tiles) -- what
 * ddimx_unet_fwd launches for those widths.  xf = 1 (affine input) or 2 (affine + SiLU); the output activation is SiLU.
 * group_stats: [B][workgroups per sample][32] floats = 8 groups x (sum, sum of squares) of the fp32 values before the bf16
 * rounding + zero padding (gn_fused.h), ddimx_conv3x3_pipe_stats_floats floats (-1: shape not eligible); nullable. */
int ddimx_conv3x3_pipe_fwd(int C, const void* x, const void* w_frag, const float* bias, const float* chan_add, int chan_add_stride,
                           const float* in_scale, const float* in_shift, int xf, void* y, float* group_stats, int B, int H, int W,
                           void* stream);
long long ddimx_conv3x3_pipe_stats_floats(int C, int B, int H, int W);
int ddimx_conv3x3_wreg_fwd(int C, const void* x, const void* w, const void* w_frag, const float* bias, const float* chan_add,
                           int chan_add_stride, const float* in_scale, const float* in_shift, int xf, int act, void* y,
                           float* stats, int B, int H, int W, void* stream);
/* Diagnostic only: as ddimx_conv3x3_fwd (xf = 2, act = 1); in a -DDDIMX_STAMP build of the library the kernel
 * also writes per-wave per-phase cycle sums to stamps[waves][12] (tools/conv_stamps.py). */
int ddimx_debug_conv3x3_stamps(int dtype, int C, const void* x, const void* w, const float* chan_add,
                                const float* in_scale, const float* in_shift, void* y, float* stats,
                                unsigned long long* stamps, int B, int H, int W, void* stream);
/* y = x + (h * scale + shift): tail of Residual_Block (models/diffusion.py:54-56), NHWC, stats nullable */
int ddimx_resid_gn_fwd(int dtype, int C, const void* x, const void* h, const float* scale, const float* shift, void* y,
                       float* stats, int B, int H, int W, void* stream);
/* Downsample.forward (models/diffusion.py:70-78): [B][H][W][Cin] -> [B][H/2][W/2][Cout] */
int ddimx_downsample_fwd(int dtype, int Cin, int Cout, const void* x, const void* w, const float* bias, void* y,
                         int B, int H, int W, void* stream);
/* Upsample.forward + the skip add that follows it (models/diffusion.py:59-67,284):
 * [B][H][W][Cin] -> [B][2H][2W][Cout] + skip.  bias2 is the bias repeated twice ([2*Cout]). */
int ddimx_upsample_add_fwd(int dtype, int Cin, int Cout, const void* x, const void* w, const float* bias2,
                           const void* skip, void* y, int B, int H, int W, void* stream);
/* BetaEmbedding.forward (models/diffusion.py:110-120): t [B] -> out [B][E]; h1/h2 [B][512] scratch */
int ddimx_temb_fwd(const float* te, const int64_t* t, const float* w0, const float* b0, const float* w1,
                   const float* b1, const float* w2, const float* b2, float* h1, float* h2, float* out, int B,
                   int pos_ch, int emb_ch, int E, void* stream);

/* Input convolution, `down_modules[0]` = Conv2d(C_io -> ch[0], k3, p1) (models/diffusion.py:189-198,255-256):
 * x [B][Cin][H][W] fp32 (the reference's NCHW boundary) -> y [B][H][W][C0] NHWC in `dtype`; also writes the GroupNorm
 * statistics partials of y into `stats` (ddimx_conv_in_stats_floats() floats).  w [C0][Cin][3][3] fp32 as stored. */
long long ddimx_conv_in_stats_floats(int B, int C0, int H, int W);
int ddimx_conv_in_fwd(int dtype, const float* x, const float* w, const float* bias, void* y, float* stats, int B, int Cin,
                      int C0, int H, int W, void* stream);
/* Output convolution, `up_modules[-1]` = Conv2d(ch[0] -> C_io, k3, p1) applied to `x + hidden[0]`
 * (models/diffusion.py:199-208,283-292): a, b NHWC [B][H][W][C0] in `dtype` (summed on the fly) -> eps [B][Cout][H][W] fp32.
 * w_packed: [9][Cout][C0] fp32 = ddimx_pack_conv(DDIMX_F32, w, ..). */
int ddimx_conv_out_fwd(int dtype, const void* a, const void* b, const float* w_packed, const float* bias, float* eps, int B,
                       int C0, int Cout, int H, int W, void* stream);
/* Transformer_Module.forward (models/diffusion.py:148-167: TransformerEmbedding :131-145, FNetEncoder x num_hidden_layers
 * transformers modeling_fnet.py:138-279, compute_out), eval mode.  x: the bottleneck activation NHWC
 * [B][S][Fr][C_last] in act_dtype viewed as tokens [B*S][width]; out: fp32 [B*S][width], both in the library's token
 * order f*C_last + c (the reference's is c*Fr + f: models/diffusion.py:273-278).  workspace: ddimx_workspace_bytes(B, T). */
int ddimx_fnet_fwd(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace, long long workspace_bytes,
                   const void* x, float* out, int B, int T, void* stream);

/* Transformer_Module alone in TRAINING mode (dropout after the projection and after every FNet FFN, tape kept) and its
 * backward -- the `_bwd` twin of ddimx_fnet_fwd (models/diffusion.py:148-167; autograd through TransformerEmbedding :131-145,
 * FNetEncoder modeling_fnet.py:138-279 and compute_out).  workspace / tape: the whole-network layouts for the same (B, T),
 * ddimx_train_workspace_bytes / ddimx_train_tape_bytes; the launches are the ones ddimx_unet_fwd_train / ddimx_unet_bwd issue for
 * the bottleneck.  x: tokens as for ddimx_fnet_fwd; out / d_out / d_x: fp32 [B*S][width] in the library's token order.
 * ddimx_fnet_bwd WRITES every transformer.* gradient at its ddimx_grad_offset() of `grads` (ddimx_grad_floats() floats; the other
 * entries are left untouched) and the gradient w.r.t. the tokens into d_x. */
int ddimx_fnet_fwd_train(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace, long long workspace_bytes,
                         void* tape, long long tape_bytes, const void* x, float* out, int B, int T, float dropout_p,
                         unsigned long long seed, void* stream);
int ddimx_fnet_bwd(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                   long long workspace_bytes, const void* tape, long long tape_bytes, const void* x, const float* d_out, float* d_x,
                   float* grads, int B, int T, float dropout_p, unsigned long long seed, void* stream);

/* ---- backward twins of the per-op forwards (autograd of the reference modules; gradients are WRITTEN, fp32, in the
 * parameter's own layout).  The whole-network ddimx_unet_bwd issues exactly these launches. -------------------------------
 * Downsample (models/diffusion.py:70-78): x [B][H][W][Cin] the forward input, dy [B][H/2][W/2][Cout]; w_dgrad = the SAME weight
 * tensor packed with ddimx_pack_convT(dtype, w, .., I = Cout, O = Cin) (the data gradient of a stride-2 conv is the sub-pixel
 * transposed conv); dx = dx_add (nullable, same shape) + d(input).  workspace: ddimx_downup_bwd_workspace_bytes(dtype, Cout, Cin,
 * B, H/2, W/2) bytes. */
long long ddimx_downup_bwd_workspace_bytes(int dtype, int Csmall, int Cbig, int B, int Hsmall, int Wsmall);
int ddimx_downsample_bwd(int dtype, int Cin, int Cout, const void* x, const void* dy, const void* w_dgrad, const void* dx_add, void* dx,
                         float* d_w, float* d_b, void* workspace, int B, int H, int W, void* stream);
/* Upsample + skip add (models/diffusion.py:59-67,284): x [B][H][W][Cin] the forward input, dy [B][2H][2W][Cout] (also the
 * gradient of the skip tensor, unchanged); w_dgrad = the ConvTranspose2d weight packed with ddimx_pack_conv(dtype, w, .., O = Cin,
 * I = Cout, 4, 4).  d_w in the ConvTranspose2d layout [Cin][Cout][4][4].  workspace: (dtype, Cin, Cout, B, H, W). */
int ddimx_upsample_add_bwd(int dtype, int Cin, int Cout, const void* x, const void* dy, const void* w_dgrad, void* dx, float* d_w,
                           float* d_b, void* workspace, int B, int H, int W, void* stream);
/* Edge convolutions (models/diffusion.py:189-208).  conv_in: dy NHWC gradient of its output, x the NCHW fp32 network input.
 * conv_out: d_sum = gradient w.r.t. (a + b) NHWC (it is the gradient of both summands); weight / bias gradients from a + b.
 * partial: ddimx_edge_bwd_workspace_floats() floats. */
long long ddimx_edge_bwd_workspace_floats(int dtype, int B, int C0, int Cio, int H, int W);
int ddimx_conv_in_bwd(int dtype, const void* dy, const float* x, float* partial, float* d_w, float* d_b, int B, int Cin, int C0, int H,
                      int W, void* stream);
int ddimx_conv_out_bwd(int dtype, const float* d_eps, const void* a, const void* b, const float* w_packed, void* d_sum, float* partial,
                       float* d_w, float* d_b, int B, int C0, int Cout, int H, int W, void* stream);
/* BetaEmbedding (models/diffusion.py:110-120), training: the forward keeps the two pre-activations ([B][emb_ch] each), the backward
 * returns every weight / bias gradient (d_h2, d_h1: [B][emb_ch] scratch). */
int ddimx_temb_fwd_train(const float* te, const int64_t* t, const float* w0, const float* b0, const float* w1, const float* b1,
                         const float* w2, const float* b2, float* h1_pre, float* h2_pre, float* out, int B, int pos_ch, int emb_ch, int E,
                         void* stream);
int ddimx_temb_bwd(const float* d_out, const float* te, const int64_t* t, const float* w1, const float* w2, const float* h1_pre,
                   const float* h2_pre, float* d_h2, float* d_h1, float* d_w0, float* d_b0, float* d_w1, float* d_b1, float* d_w2,
                   float* d_b2, int B, int pos_ch, int emb_ch, int E, void* stream);

/* ---- sampler (functions/denoising.py:10-52) ------------------------------------------------------- */
/* coef [n_iter][6] fp32 rows (t, sqrt(1-at), sqrt(at), sqrt(at_next), c2, c1); step: device int counter.
 * step_begin fills t[B] with the current timestep; ddim_update performs lines :27 and :41-43 in one pass,
 * writing the x0 prediction to x0 and x_{t-1} in place; step_end advances the counter. */
/* FNet Fourier mixing with its residual, transformers modeling_fnet.py:138-166 (`fftn(x, dim=(1, 2)).real`) + :182:
 * z[b] = Re(FFT2(x[b])) + x[b] over fp32 [B][S][hid]; dft_hidden / dft_seq as in ddimx_tables.  fused = 1: one launch
 * (when ddimx_fnet_mix_supported(S, hid)); fused = 0: two exact-fp32 GEMMs through ut [B][2*hid][S] and the split-K
 * scratch partial [8*B*2*hid*S floats].  The operator is symmetric, so it is also its own backward. */
int ddimx_fnet_mix_supported(int S, int hid);
int ddimx_fnet_mix(const float* dft_hidden, const float* dft_seq, const float* x, float* z, float* ut, float* partial, int B,
                   int S, int hid, int fused, void* stream);
int ddimx_step_begin(const float* coef, const int* step, int64_t* t, int B, void* stream);
/* as ddimx_step_begin for coefficient tables with another row stride (ddpm_steps: 7) */
int ddimx_step_begin_ex(const float* coef, int row_stride, const int* step, int64_t* t, int B, void* stream);
int ddimx_ddim_update(float* xt, const float* et, const float* noise, float* x0, const float* coef, const int* step,
                      long long n, void* stream);
/* ddpm_steps update (functions/denoising.py:72-90): coef [n_iter][7] fp32 rows (t, (1/at).sqrt(), (1/at-1).sqrt(),
 * atm1.sqrt()*beta_t, (1-beta_t).sqrt()*(1-atm1), 1-at, mask*exp(0.5*log(beta_t))); x0 = clamp(x0 pred), xn = sample */
int ddimx_ddpm_update(const float* x, const float* et, const float* noise, float* x0, float* xn, const float* coef,
                      const int* step, long long n, void* stream);
int ddimx_step_end(int* step, void* stream);

/* ---- training-step pieces (functions/losses.py:4-18, models/ema.py:16-23) ---------------------------- */
int ddimx_qsample(const float* x0, const float* e, const float* alphas, const int64_t* t, float* x, int B,
                  long long per_sample, void* stream);
/* loss[0..B-1] = per-sample sum of squared error, loss[B] = batch mean; partial: [B*64] scratch */
int ddimx_sqerr_loss(const float* e, const float* out, float* partial, float* loss, int B, long long per_sample,
                     void* stream);
int ddimx_ema_block_elems(void);
/* shadow = (1-mu)*param + mu*shadow for a list of tensors in one launch (pointer tables on device) */
int ddimx_ema_update_multi(const long long* shadow_ptrs, const long long* param_ptrs, const long long* sizes,
                           const int* blk_tensor, const long long* blk_off, int nblocks, float mu, void* stream);

/* ---- optimizer tail of train_step (runners/diffusion.py:155-173; functions/__init__.py:5-23) ------------------
 * Multi-tensor launches over device pointer tables (block tables as for ddimx_ema_update_multi).
 * grad_norm: out[0] = global L2 norm of all gradients, out[1] = min(1, max_norm/(norm+1e-6)) -- the coefficient of
 * torch.nn.utils.clip_grad_norm_ -- kept on the device (no host sync); partial: [nblocks] scratch.
 * adam: g *= clip[1] (if clip != null), then torch.optim.Adam (decoupled = 0) / AdamW (decoupled = 1), amsgrad off;
 * decoupled = 2: AdaBelief as published (Zhuang et al. 2020; decoupled decay, no rectification) -- the reference's default
 * optimizer (functions/__init__.py:24-42) comes from an un-vendored submodule, so this mode has no reference pin. */
int ddimx_grad_norm_multi(const long long* grad_ptrs, const long long* sizes, const int* blk_tensor,
                          const long long* blk_off, int nblocks, float max_norm, float* partial, float* out, void* stream);
/* every tensor *= coef[0] (device scalar): the in-place scaling of clip_grad_norm_ */
int ddimx_scale_multi(const long long* ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                      int nblocks, const float* coef, void* stream);
int ddimx_adam_multi(const long long* param_ptrs, const long long* grad_ptrs, const long long* m_ptrs,
                     const long long* v_ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                     int nblocks, const float* clip, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int step, int decoupled, void* stream);
/* The same with the per-step scalars in device memory, dyn = {lr, 1 - beta1^step, sqrt(1 - beta2^step)} (fp32), read when
 * the kernel RUNS: a training step captured once into a hipGraph is replayed with new values written to dyn between replays
 * (LambdaLR, functions/__init__.py:53-60, and the bias corrections change every step). */
int ddimx_adam_multi_dyn(const long long* param_ptrs, const long long* grad_ptrs, const long long* m_ptrs,
                         const long long* v_ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                         int nblocks, const float* clip, const float* dyn, float beta1, float beta2, float eps,
                         float weight_decay, int decoupled, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDIMX_H */
