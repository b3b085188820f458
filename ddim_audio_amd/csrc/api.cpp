// C ABI of libddimx (include/ddimx.h): parameter plan, weight packing, workspace carving and the
// network walk of Model.forward (reference models/diffusion.py:237-294) over the HIP kernels.
#include "../../include/ddimx.h"

#include <stdarg.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

#include "conv_mfma.h"
#include "kernels.h"
#include "train_kernels.h"
#include "wgrad_mfma.h"
#include "conv_pipe.h"

namespace ddimx {
// Tuning hooks (A/B runs of tools/*.py only): the DDIMX_* environment variables are read ONCE per process, at the first
// library call that needs one, never per launch.
struct Knobs {
    int fnet_dense, conv_pipe, pipe_tpw, bwd_stats_fused, gn_dbg, conv_wreg, conv_wps, conv_var, wgrad_split, wgrad_side, wgrad_hold;
    Knobs() {
        auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        conv_var = geti("DDIMX_CONV_VAR", -1);     // tools/conv_tune.py: force a candidate tile variant of conv_mfma_kernel
        conv_wps = geti("DDIMX_CONV_WPS", 0);      // tools/conv_tune.py: workgroups per sample
        wgrad_split = geti("DDIMX_WGRAD_SPLIT", 0);  // tools/wgrad_one.py
        wgrad_side = geti("DDIMX_WGRAD_SIDE", 1);    // tools/wgside_trace.sh A/B: 0 = the weight gradients stay on the backward's stream, 2 = forked early
        // ... and the up path's weight gradients of levels < wgrad_hold wait (in `du` buffers of their own) for the bottleneck's
        // backward, whose launch-bound FNet kernels leave the chip idle (WgSide::held)
        wgrad_hold = geti("DDIMX_WGRAD_HOLD", 2);
        // A/B, GroupNorm-backward statistics -- bit 0: of GN1 / GN0 in the data-gradient convs' epilogue, bit 1: of GN2 in the previous
        // block's last apply pass; 0 = every statistics pass on its own
        bwd_stats_fused = geti("DDIMX_BWD_STATS_FUSED", 3);
        gn_dbg = geti("DDIMX_GN_DBG", 0);          // tools/gn_dbg.sh: 1 = resid, 2 = convs take their GroupNorm input from a finalize launch
        fnet_dense = geti("DDIMX_FNET_DENSE", 1);  // tools/fnet_ab.sh: 0 = the GEMM path for the FNet at S <= 32
        conv_wreg = geti("DDIMX_CONV_WREG", 1);    // tools/step_ab.sh: 0 = the convs of C >= 64 keep the LDS weight ring (conv_mfma_kernel)
        // conv3_pipe_kernel in the walk, bit 0: C = 32, bit 1: C = 64.  Default: level 0 only.  With the two batch shards in flight the
        // C = 64 form (one four-wave workgroup per CU: 144 registers of weights per wave) runs its B = 4 launches on half the chip,
        // 51-63 us against conv3_wreg_kernel's 43 (profiles/r04/pipe_v2_forked_step_kernels.txt); alone on the chip it is level
        // (58 / 65 vs 57 / 67 us at B = 8) and in the single-stream step it wins (DDIMX_FORK_MASK=0: +5.5 % with both levels on).
        conv_pipe = geti("DDIMX_CONV_PIPE", 1);
        pipe_tpw = geti("DDIMX_PIPE_TPW", 0);      // tools/pipe_time.py: tiles per workgroup of conv3_pipe_kernel
    }
};
static const Knobs& knobs() {
    static const Knobs k;
    return k;
}
hipError_t conv_geometry_bf16_c3(int, int, int, int, ConvGeom*);
hipError_t conv_geometry_bf16_du(int, int, int, int, ConvGeom*);
hipError_t conv_geometry_f32_c3(int, int, int, int, ConvGeom*);
hipError_t conv_geometry_f32_du(int, int, int, int, ConvGeom*);
hipError_t conv_launch_bf16_c3(int, int, int, int, ConvArgs&, hipStream_t);
hipError_t conv_launch_bf16_du(int, int, int, int, ConvArgs&, hipStream_t);
hipError_t conv_launch_f32_c3(int, int, int, int, ConvArgs&, hipStream_t);
hipError_t conv_launch_f32_du(int, int, int, int, ConvArgs&, hipStream_t);
hipError_t conv_launch_bf16_c3b(int, int, int, int, ConvArgs&, hipStream_t);  // + GroupNorm-backward statistics epilogue
hipError_t conv_launch_f32_c3b(int, int, int, int, ConvArgs&, hipStream_t);

hipError_t conv_geometry(int dtype, int mode, int cin, int cout, int var, ConvGeom* g) {
    const int nout = mode == UP4 ? 2 * cout : cout;
    if (dtype == DT_BF16)
        return mode == CONV3 ? conv_geometry_bf16_c3(mode, cin, nout, var, g) : conv_geometry_bf16_du(mode, cin, nout, var, g);
    return mode == CONV3 ? conv_geometry_f32_c3(mode, cin, nout, var, g) : conv_geometry_f32_du(mode, cin, nout, var, g);
}
hipError_t conv_launch(int dtype, int mode, int cin, int cout, int var, ConvArgs& a, hipStream_t s) {
    const int nout = mode == UP4 ? 2 * cout : cout;
    if (a.bwd_mode) {
        if (mode != CONV3 || !a.aux || !a.stats) return hipErrorInvalidValue;
        return dtype == DT_BF16 ? conv_launch_bf16_c3b(mode, cin, nout, var, a, s) : conv_launch_f32_c3b(mode, cin, nout, var, a, s);
    }
    if (dtype == DT_BF16)
        return mode == CONV3 ? conv_launch_bf16_c3(mode, cin, nout, var, a, s) : conv_launch_bf16_du(mode, cin, nout, var, a, s);
    return mode == CONV3 ? conv_launch_f32_c3(mode, cin, nout, var, a, s) : conv_launch_f32_du(mode, cin, nout, var, a, s);
}
int conv_pick_variant(int dtype, int mode, int cin, int cout, int B, int Hv, int Wv) {
    ConvGeom g0, g1;
    if (const int v = knobs().conv_var; v >= 0) {  // tuning hook: force a candidate variant where one exists
        if (conv_geometry(dtype, mode, cin, cout, v, &g0) == hipSuccess) return v;
    }
    if (conv_geometry(dtype, mode, cin, cout, 0, &g0) != hipSuccess) return 0;
    if (conv_geometry(dtype, mode, cin, cout, 1, &g1) != hipSuccess) return 0;
    // The choice depends on the SAMPLE's size only (never on B): a sample then runs through the same kernels, with the same
    // statistics partition, alone or in any batch / on any number of GPUs -> bit-identical results.  The threshold is the
    // one measured at the headline batch of 8 (the large tile wins from ~256 workgroups up, i.e. >= 25 tiles per sample).
    // (The training step passes its real batch: gradients depend on the whole batch anyway, and the large tile is faster
    // once B x tiles fills the GPU.  B <= 0 selects the batch-independent rule.)
    const long long per_sample = (long long)((Wv + g0.tw - 1) / g0.tw) * ((Hv + g0.th - 1) / g0.th) * g0.classes * (g0.nout / g0.nb);
    return per_sample * (B > 0 ? B : 8) < 200 ? 1 : 0;
}
}  // namespace ddimx

using namespace ddimx;

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)               \
    do {                        \
        int r_ = (expr);        \
        if (r_) return r_;      \
    } while (0)

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t esz(int dtype) { return dtype == DT_BF16 ? 2 : 4; }

// ------------------------------------------------------------------------------------------ plan
enum PackKind { PK_COPY, PK_CONV, PK_CONVT, PK_BIAS2, PK_PERM_COLS, PK_PERM_ROWS, PK_CONV_F32 };

struct ParamSpec {
    std::string name;
    long long numel;
    int kind;
    int d0, d1, d2, d3;  // shape (unused dims = 1)
    size_t off, bytes;   // in the packed buffer
};

struct RBW {
    int g0, b0, g1, b1, g2, w0, w1, bias1;  // indices into specs
};

struct ddimx_ctx {
    ddimx_config cfg;
    int dtype;
    int fnet_bf16;  // operands of the FNet's dense-weight GEMMs rounded to bf16 (transformers.dtype), else exact fp32 MFMA
    int L;
    int E;      // total timestep-embedding width
    int width;  // FNet token width
    int Fr;     // frequency bins at the bottleneck
    std::vector<ParamSpec> specs;
    std::vector<size_t> frag_off;  // per spec: offset of a second, fragment-order copy of a 3x3 conv weight (conv_wreg.h), 0 = none
    size_t packed_bytes;
    // indices
    int te, tw[3], tb[3];
    int in_w, in_b, out_w, out_b;
    std::vector<std::vector<RBW>> down_rb, up_rb;  // [level][r]
    std::vector<int> down_w, down_b, up_w, up_b;   // per level (level 0 unused)
    int ln0_w, ln0_b, proj_w, proj_b, cout_w, cout_b;
    struct FL { int ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b; };
    std::vector<FL> fl;
    // second copies for fnet_dense_kernel (fnet_dense.hip; offsets into the packed buffer, 0 = none): the first FFN matrix with
    // the preceding LayerNorm's gamma folded in (+ the bias with its beta), compute_out likewise with the last layer's output
    // LayerNorm; all of them in MFMA fragment order and -- bf16 FNet -- pre-rounded to bf16
    struct FX { size_t w1f, b1f, w2c, tab, bc; };  // tab / bc: the layer's hidden-DFT table with the PREVIOUS layer's output LayerNorm folded in
    std::vector<FX> fx;
    size_t fx_proj = 0, fx_coutf = 0, fx_coutb = 0;
    bool fx_on = false;
    const void* frag_packed = nullptr;  // the packed buffer whose fragment-order conv copies (frag_off) are current: written by the
                                        // eval-only pack (ddimx_pack_fnet_inference), stale after every ddimx_pack_weights
    const void* fx_packed = nullptr;  // the packed buffer whose fnet_dense copies are current (ddimx_pack_fnet_inference), else null
    std::vector<int> emb_off_down, emb_off_up;  // temb chunk offsets per block, execution order
    const unsigned long long* dropout_ctr = nullptr;  // device counter added to every dropout seed (ddimx_set_dropout_counter)
};

static int add_spec(ddimx_ctx* c, const std::string& name, int kind, int d0, int d1 = 1, int d2 = 1, int d3 = 1) {
    ParamSpec s;
    s.name = name; s.kind = kind; s.d0 = d0; s.d1 = d1; s.d2 = d2; s.d3 = d3;
    s.numel = (long long)d0 * d1 * d2 * d3;
    size_t bytes;
    const size_t es = esz(c->dtype);
    switch (kind) {
        case PK_CONV: bytes = (size_t)s.numel * es; break;
        case PK_CONVT: bytes = (size_t)2 * 6 * 2 * d1 * d0 * es; break;  // [I][O][4][4] -> [2][6][2*O][I]
        case PK_BIAS2: bytes = (size_t)2 * d0 * 4; break;
        default: bytes = (size_t)s.numel * 4;
    }
    s.off = c->packed_bytes;
    s.bytes = bytes;
    c->packed_bytes += al256(bytes);
    c->specs.push_back(s);
    return (int)c->specs.size() - 1;
}

static RBW add_rb(ddimx_ctx* c, const std::string& p, int C, int k) {
    RBW r;
    r.g0 = add_spec(c, p + "norm.0.weight", PK_COPY, C);
    r.b0 = add_spec(c, p + "norm.0.bias", PK_COPY, C);
    r.g1 = add_spec(c, p + "norm.1.weight", PK_COPY, C);
    r.b1 = add_spec(c, p + "norm.1.bias", PK_COPY, C);
    r.g2 = add_spec(c, p + "norm.2.weight", PK_COPY, C);
    r.w0 = add_spec(c, p + "conv.0.weight", PK_CONV, C, C, k, k);
    r.w1 = add_spec(c, p + "conv.1.weight", PK_CONV, C, C, k, k);
    r.bias1 = add_spec(c, p + "conv.1.bias", PK_COPY, C);
    WregGeom wg;
    PipeGeom pg;
    if (c->dtype == DT_BF16 && k == 3 && (wreg_geometry(CONV3, C, C, &wg) == hipSuccess || pipe_geometry(C, &pg) == hipSuccess)) {  // second copy in MFMA fragment order
        c->frag_off.resize(c->specs.size(), 0);
        for (int i : {r.w0, r.w1}) {
            c->frag_off[i] = c->packed_bytes;
            c->packed_bytes += al256((size_t)9 * C * C * 2);
        }
    }
    return r;
}

static int build_plan(ddimx_ctx* c) {
    const ddimx_config& f = c->cfg;
    const int L = f.n_levels;
    c->L = L;
    c->dtype = f.act_dtype;
    if (L < 1 || L > DDIMX_MAX_LEVELS) return fail("n_levels %d out of range", L);
    if (f.act_dtype != DDIMX_F32 && f.act_dtype != DDIMX_BF16) return fail("act_dtype %d not supported", f.act_dtype);
    if (f.fnet_dtype != DDIMX_F32 && f.fnet_dtype != DDIMX_BF16) return fail("fnet_dtype %d not supported", f.fnet_dtype);
    if (f.fnet_dtype == DDIMX_BF16 && f.act_dtype != DDIMX_BF16) return fail("fnet_dtype bf16 needs act_dtype bf16");
    c->fnet_bf16 = f.fnet_dtype == DDIMX_BF16;
    for (int l = 0; l < L; ++l) {
        if (f.krn[l] != 3) return fail("kernel size %d at level %d: only 3 is implemented", f.krn[l], l);
        if (f.ch[l] % 32) return fail("channel width %d at level %d must be a multiple of 32", f.ch[l], l);
        ConvGeom g;
        if (conv_geometry(c->dtype, CONV3, f.ch[l], f.ch[l], 0, &g) != hipSuccess)
            return fail("no 3x3 conv kernel instantiated for %d channels", f.ch[l]);
        if (l > 0) {
            if (conv_geometry(c->dtype, DOWN4, f.ch[l - 1], f.ch[l], 0, &g) != hipSuccess)
                return fail("no downsample kernel instantiated for %d->%d", f.ch[l - 1], f.ch[l]);
            if (conv_geometry(c->dtype, UP4, f.ch[l], f.ch[l - 1], 0, &g) != hipSuccess)
                return fail("no upsample kernel instantiated for %d->%d", f.ch[l], f.ch[l - 1]);
        }
    }
    if (f.f_size % (1 << (L - 1))) return fail("f_size %d not divisible by 2^(levels-1)", f.f_size);
    c->Fr = f.f_size >> (L - 1);
    c->width = f.ch[L - 1] * c->Fr;
    if (c->width > 2048) return fail("FNet token width %d > 2048 not supported by the LayerNorm kernel", c->width);
    if (f.fnet_hidden > 2048) return fail("fnet hidden %d > 2048", f.fnet_hidden);
    c->packed_bytes = 0;
    c->E = 0;
    for (int l = 0; l < L; ++l) c->E += 2 * f.res[l] * f.ch[l];

    c->te = add_spec(c, "temb.te", PK_COPY, f.n_timesteps, 128);
    const int tdim[3][2] = {{512, 128}, {512, 512}, {c->E, 512}};
    for (int i = 0; i < 3; ++i) {
        c->tw[i] = add_spec(c, "temb.weight." + std::to_string(i) + ".weight", PK_COPY, tdim[i][0], tdim[i][1]);
        c->tb[i] = add_spec(c, "temb.weight." + std::to_string(i) + ".bias", PK_COPY, tdim[i][0]);
    }
    c->in_w = add_spec(c, "down_modules.0.weight", PK_COPY, f.ch[0], f.in_channels, 3, 3);
    c->in_b = add_spec(c, "down_modules.0.bias", PK_COPY, f.ch[0]);
    c->down_rb.resize(L); c->up_rb.resize(L);
    c->down_w.assign(L, -1); c->down_b.assign(L, -1); c->up_w.assign(L, -1); c->up_b.assign(L, -1);
    for (int l = 0; l < L; ++l) {
        const std::string base = "down_modules." + std::to_string(l + 1) + ".";
        int j = 0;
        if (l > 0) {
            c->down_w[l] = add_spec(c, base + "0.conv.weight", PK_CONV, f.ch[l], f.ch[l - 1], 4, 4);
            {
                WregGeom wg;
                if (c->dtype == DT_BF16 && wreg_geometry(DOWN4, f.ch[l - 1], f.ch[l], &wg) == hipSuccess) {
                    c->frag_off.resize(c->specs.size(), 0);
                    c->frag_off[c->down_w[l]] = c->packed_bytes;
                    c->packed_bytes += al256((size_t)16 * f.ch[l] * f.ch[l - 1] * 2);
                }
            }
            c->down_b[l] = add_spec(c, base + "0.conv.bias", PK_COPY, f.ch[l]);
            j = 1;
        }
        for (int r = 0; r < f.res[l]; ++r) c->down_rb[l].push_back(add_rb(c, base + std::to_string(j + r) + ".", f.ch[l], 3));
    }
    for (int k = 0; k < L; ++k) {
        const int l = L - 1 - k;
        const std::string base = "up_modules." + std::to_string(k) + ".";
        for (int r = 0; r < f.res[l]; ++r) c->up_rb[l].push_back(add_rb(c, base + std::to_string(r) + ".", f.ch[l], 3));
        if (l > 0) {
            c->up_w[l] = add_spec(c, base + std::to_string(f.res[l]) + ".conv.weight", PK_CONVT, f.ch[l], f.ch[l - 1], 4, 4);
            {
                WregGeom wg;
                if (c->dtype == DT_BF16 && wreg_geometry(UP4, f.ch[l], 2 * f.ch[l - 1], &wg) == hipSuccess) {
                    c->frag_off.resize(c->specs.size(), 0);
                    c->frag_off[c->up_w[l]] = c->packed_bytes;
                    c->packed_bytes += al256((size_t)2 * 6 * 2 * f.ch[l - 1] * f.ch[l] * 2);
                }
            }
            c->up_b[l] = add_spec(c, base + std::to_string(f.res[l]) + ".conv.bias", PK_BIAS2, f.ch[l - 1]);
        }
    }
    c->out_w = add_spec(c, "up_modules." + std::to_string(L) + ".weight", PK_CONV_F32, f.in_channels, f.ch[0], 3, 3);
    c->out_b = add_spec(c, "up_modules." + std::to_string(L) + ".bias", PK_COPY, f.in_channels);
    const int hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width;
    c->ln0_w = add_spec(c, "transformer.embedding.LayerNorm.weight", PK_PERM_COLS, 1, width);
    c->ln0_b = add_spec(c, "transformer.embedding.LayerNorm.bias", PK_PERM_COLS, 1, width);
    c->proj_w = add_spec(c, "transformer.embedding.projection.weight", PK_PERM_COLS, hid, width);
    c->proj_b = add_spec(c, "transformer.embedding.projection.bias", PK_COPY, hid);
    for (int i = 0; i < f.fnet_layers; ++i) {
        const std::string p = "transformer.encoder.layer." + std::to_string(i) + ".";
        ddimx_ctx::FL fl;
        fl.ln1_w = add_spec(c, p + "fourier.output.LayerNorm.weight", PK_COPY, hid);
        fl.ln1_b = add_spec(c, p + "fourier.output.LayerNorm.bias", PK_COPY, hid);
        fl.w1 = add_spec(c, p + "intermediate.dense.weight", PK_COPY, inter, hid);
        fl.b1 = add_spec(c, p + "intermediate.dense.bias", PK_COPY, inter);
        fl.w2 = add_spec(c, p + "output.dense.weight", PK_COPY, hid, inter);
        fl.b2 = add_spec(c, p + "output.dense.bias", PK_COPY, hid);
        fl.ln2_w = add_spec(c, p + "output.LayerNorm.weight", PK_COPY, hid);
        fl.ln2_b = add_spec(c, p + "output.LayerNorm.bias", PK_COPY, hid);
        c->fl.push_back(fl);
    }
    c->cout_w = add_spec(c, "transformer.compute_out.weight", PK_PERM_ROWS, width, hid);
    c->cout_b = add_spec(c, "transformer.compute_out.bias", PK_PERM_COLS, 1, width);
    // exactly the shapes fnet_dense_launch instantiates: K = hid = 512 -> N = inter / width ("wide"), and K = inter / width -> N = hid
    // only as the "deep" form (K >= 4 N: inter, width >= 2048) -- smaller widths take the GEMM path instead of failing in the launcher
    c->fx_on = f.fnet_layers > 0 && hid == 512 && inter % 512 == 0 && width % 512 == 0 && inter >= 2048 && width >= 2048;
    if (c->fx_on) {
        const size_t wes = c->fnet_bf16 ? 2 : 4;
        auto take = [&](size_t bytes) { const size_t o = c->packed_bytes; c->packed_bytes += al256(bytes); return o; };
        for (int i = 0; i < f.fnet_layers; ++i) {
            ddimx_ctx::FX x;
            x.w1f = take((size_t)inter * hid * wes);
            x.b1f = take((size_t)inter * 4);
            x.w2c = take((size_t)hid * inter * wes);
            x.tab = take((size_t)2 * hid * hid * 4);
            x.bc = take((size_t)hid * 4);
            c->fx.push_back(x);
        }
        c->fx_proj = take((size_t)hid * width * wes);
        c->fx_coutf = take((size_t)width * hid * wes);
        c->fx_coutb = take((size_t)width * 4);
    }
    // timestep-embedding chunk offsets in execution order (models/diffusion.py:178-184,249-250)
    int off = 0;
    for (int l = 0; l < L; ++l)
        for (int r = 0; r < f.res[l]; ++r) { c->emb_off_down.push_back(off); off += f.ch[l]; }
    for (int l = L - 1; l >= 0; --l)
        for (int r = 0; r < f.res[l]; ++r) { c->emb_off_up.push_back(off); off += f.ch[l]; }
    return 0;
}

// Split-K of the FNet GEMMs is chosen from the PER-SAMPLE problem (rows of one sample, never the batch): a sample's rows are
// then summed in the same order alone, inside any batch and on any number of GPUs (bit-identical results).
constexpr int kMaxSplitK = 8;
static inline int sample_splitk(int rows_per_sample, int N, int K, int bf16) {
    int s = gemm_pick_splitk(rows_per_sample, N, K, 1, bf16);
    return s;
}

// ------------------------------------------------------------------------------------------ workspace
struct Carver {
    char* base;
    size_t off;
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += al256(bytes);
        return p;
    }
};

struct Ws {
    float *temb_h1, *temb_h2, *temb;
    void* A;                  // in-conv output (hidden[0])
    std::vector<void*> xd, xu;
    void *h1, *h2;
    float *stats, *stats2, *scale, *shift;  // stats / stats2: the inference walk alternates (a kernel reads one, writes the other)
    float *ln0, *X, *Ut, *Z, *Y, *Hb, *O, *gpart;
    float *pz, *pv, *zc, *hc, *vc;  // fnet_dense.hip: row statistics of Z and of the last FFN output; chunk-major Z, FFN hidden, last FFN output
    size_t total;
    size_t stats_per_sample, gpart_per_sample;  // floats: the statistics / split-K scratch one sample can need (max over ops)
    size_t h_per_sample;                        // bytes of h1 / h2 one sample can need (its largest level)
    int cmax;
};

static size_t conv_stats_floats(int dtype, int mode, int cin, int cout, int B, int Hv, int Wv) {
    size_t mx = 0;
    for (int var = 0; var < 8; ++var) {
        ConvGeom g;
        if (conv_geometry(dtype, mode, cin, cout, var, &g) != hipSuccess) continue;
        const size_t n = (size_t)B * cdiv(Wv, g.tw) * cdiv(Hv, g.th) * g.classes * g.nout * 2;
        if (n > mx) mx = n;
    }
    if (mode == CONV3 && cin == cout && dtype == DT_BF16) {  // the specialised kernels partition a sample into their own tiles
        WregGeom wg;
        if (wreg_geometry(CONV3, cin, cout, &wg) == hipSuccess) {
            const size_t n = (size_t)B * cdiv(Wv, wg.tw) * cdiv(Hv, wg.th) * cout * 2;
            if (n > mx) mx = n;
        }
        PipeGeom pg;  // (one 32-float slab per workgroup of >= 1 tile)
        if (pipe_geometry(cin, &pg) == hipSuccess) {
            const size_t n = (size_t)B * cdiv(Wv, pg.tw) * cdiv(Hv, pg.th) * kGnSlab;
            if (n > mx) mx = n;
        }
    }
    if (mode != CONV3 && dtype == DT_BF16) {
        WregGeom wg;
        const int nout = mode == UP4 ? 2 * cout : cout, ncls = mode == UP4 ? 2 : 1;
        if (wreg_geometry(mode, cin, nout, &wg) == hipSuccess) {
            const size_t n = (size_t)B * cdiv(Wv, wg.tw) * cdiv(Hv, wg.th) * ncls * nout * 2;
            if (n > mx) mx = n;
        }
    }
    return mx;
}

static void carve(const ddimx_ctx* c, char* base, int B, int T, Ws* w) {
    const ddimx_config& f = c->cfg;
    const int L = c->L;
    const size_t es = esz(c->dtype);
    Carver cv{base, 0};
    w->temb_h1 = (float*)cv.take((size_t)B * 512 * 4);
    w->temb_h2 = (float*)cv.take((size_t)B * 512 * 4);
    w->temb = (float*)cv.take((size_t)B * c->E * 4);
    const size_t lvl0 = (size_t)B * T * f.f_size * f.ch[0] * es;
    w->A = cv.take(lvl0);
    w->xd.resize(L); w->xu.resize(L);
    // per-channel slabs (training) and one 128-byte group slab per partial (inference, gn_fused.h): size for the larger
    size_t stats_f = (size_t)B * conv_in_nparts(T, f.f_size) * (f.ch[0] * 2 > kGnSlab ? f.ch[0] * 2 : kGnSlab);
    size_t hmax = 0;
    int cmax = 0;
    for (int l = 0; l < L; ++l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        const size_t bytes = (size_t)B * H * W * C * es;
        w->xd[l] = cv.take(bytes);
        w->xu[l] = cv.take(bytes);
        if (bytes > hmax) hmax = bytes;
        if (C > cmax) cmax = C;
        size_t s = conv_stats_floats(c->dtype, CONV3, C, C, B, H, W);
        if (s > stats_f) stats_f = s;
        s = (size_t)B * resid_nparts(c->dtype, H * W, C) * (C * 2 > kGnSlab ? C * 2 : kGnSlab);
        if (s > stats_f) stats_f = s;
        if (l > 0) {
            s = conv_stats_floats(c->dtype, DOWN4, f.ch[l - 1], C, B, H, W);
            if (s > stats_f) stats_f = s;
            s = conv_stats_floats(c->dtype, UP4, C, f.ch[l - 1], B, H, W);
            if (s > stats_f) stats_f = s;
        }
    }
    w->h1 = cv.take(hmax);
    w->h2 = cv.take(hmax);
    w->stats_per_sample = stats_f / B;  // every term above is B x (per-sample slab)
    w->h_per_sample = hmax / B;
    w->cmax = cmax;
    w->stats = (float*)cv.take(stats_f * 4);
    w->stats2 = (float*)cv.take(stats_f * 4);  // (a conv's group slabs, 128 B per >= 32-channel workgroup, never exceed its per-channel ones)
    w->scale = (float*)cv.take((size_t)B * cmax * 4);
    w->shift = (float*)cv.take((size_t)B * cmax * 4);
    const int S = T >> (L - 1);
    const size_t M = (size_t)B * S;
    const int hid = f.fnet_hidden, inter = f.fnet_inter;
    w->ln0 = (float*)cv.take((size_t)B * (S > 32 ? S : 32) * c->width * 4);  // (chunk-major for the dense FNet path: 32 rows per sample)
    w->X = (float*)cv.take(M * hid * 4);
    w->Ut = (float*)cv.take((size_t)B * 2 * hid * S * 4);
    w->Z = (float*)cv.take(M * hid * 4);
    w->Y = (float*)cv.take(M * hid * 4);
    w->Hb = (float*)cv.take(M * inter * 4);
    w->O = (float*)cv.take(M * c->width * 4);
    w->pz = (float*)cv.take((size_t)B * (hid / 16) * 64 * 4);  // (blocks of 32 rows per sample whatever S)
    w->pv = (float*)cv.take((size_t)B * (hid / 32) * 64 * 4);
    w->zc = (float*)cv.take((size_t)B * 32 * hid * 4);
    w->hc = (float*)cv.take((size_t)B * 32 * inter * 4);
    w->vc = (float*)cv.take((size_t)B * 32 * hid * 4);
    {   // split-K partial tiles of the skinny FNet GEMMs
        const int bf = c->fnet_bf16;
        const int shp[6][5] = {{(int)M, hid, c->width, 1, bf}, {2 * hid, S, hid, B, 0}, {S, hid, S, B, 0},
                               {(int)M, inter, hid, 1, bf}, {(int)M, hid, inter, 1, bf}, {(int)M, c->width, hid, 1, bf}};
        size_t mx = 0;
        for (auto& q : shp) {
            const size_t n = (size_t)kMaxSplitK * q[3] * q[0] * q[1];  // the split depends on the per-sample shape only; size for the cap
            if (n > mx) mx = n;
        }
        w->gpart_per_sample = mx / B;  // every shape above has B in its row or batch count
        w->gpart = (float*)cv.take(mx * 4);
    }
    w->total = cv.off;
}

// ------------------------------------------------------------------------------------------ building blocks
struct ConvCall {
    int dtype, mode, cin, cout;
    const void* in; const void* w; const float* bias; const float* chan_add; int chan_add_stride;
    const float* in_scale; const float* in_shift; int xf; int act;
    const void* skip; void* out; float* stats;
    int B, Hin, Win;
    unsigned long long* stamps = nullptr;
    bool batch_plan = false;  // training: choose the tile variant from the real batch (inference: sample size only)
    const void* aux = nullptr; const float* aux_scale = nullptr; const float* aux_shift = nullptr; int bwd_mode = 0;  // ConvArgs, same names
    GnIn gn = {};             // gn.stats set: the input's GroupNorm is finished inside the kernel (in_scale / in_shift unused)
    bool groups = false;      // statistics partials in group format (gn_fused.h)
    const void* wf = nullptr; // the same weights in MFMA fragment order (conv_wreg.h), if the caller has them
    int kernel_pref = 0;      // 0: the walk's choice; 1: never the software-pipelined kernel (conv_pipe.h); 2: only it (per-op exports)
};

// set for the duration of the whole-network training calls (see ConvCall::batch_plan)
static thread_local bool g_batch_plan = false;
struct BatchPlanScope {
    BatchPlanScope() { g_batch_plan = true; }
    ~BatchPlanScope() { g_batch_plan = false; }
};
// Plan of one conv launch: tile configuration and the persistent-workgroup split.
struct ConvPlan { ConvGeom g; int var, Hv, Wv, tiles_x, tiles_y, tiles_per_wg, wgs_per_sample; bool wreg, pipe; };
// The software-pipelined kernel (conv_pipe.h) takes the Residual_Block convs of the inference walk at the widths it is instantiated
// for (C = 32, 64): bf16, GroupNorm-affine (+ SiLU) input, SiLU output, group-format statistics, fragment-order weights, whole
// tiles.  The choice depends on the sample's size only (never on the batch).
static bool pipe_eligible(const ConvCall& q, PipeGeom* pg) {
    if (q.kernel_pref == 1 || !((knobs().conv_pipe >> (q.cin == 32 ? 0 : 1)) & 1 || q.kernel_pref == 2)) return false;  // bit 0: C = 32, bit 1: C = 64
    if (!q.wf || q.dtype != DT_BF16 || q.mode != CONV3 || q.cin != q.cout || q.act != 1 || q.aux || q.bwd_mode || q.skip || q.batch_plan || g_batch_plan)
        return false;
    if (q.xf != XF_AFFINE && q.xf != XF_AFFINE_SILU) return false;
    if (q.stats && !q.groups) return false;
    if (pipe_geometry(q.cin, pg) != hipSuccess) return false;
    return q.Hin % pg->th == 0 && q.Win % pg->tw == 0;
}
// The register-streamed-weights kernel (conv_wreg.h) takes the 3x3 convs of the inference walk from C = 64 up when the caller has
// the fragment-order weights and the image is a whole number of its tiles (sample size only, never the batch).
static bool wreg_eligible(const ConvCall& q, WregGeom* wg) {
    if (!knobs().conv_wreg || !q.wf || q.dtype != DT_BF16 || q.act > 1 || q.aux || q.bwd_mode || q.batch_plan || g_batch_plan) return false;
    if (q.skip && q.mode != UP4) return false;
    if (q.xf != XF_NONE && q.xf != XF_AFFINE && q.xf != XF_AFFINE_SILU) return false;
    if (wreg_geometry(q.mode, q.cin, q.mode == UP4 ? 2 * q.cout : q.cout, wg) != hipSuccess) return false;
    const int sxy = q.mode == DOWN4 ? 2 : 1;
    return q.Hin % (wg->th * sxy) == 0 && q.Win % (wg->tw * sxy) == 0;
}
static int conv_plan(const ConvCall& q, ConvPlan* p) {
    ConvGeom& g = p->g;
    p->wreg = false;
    p->pipe = false;
    PipeGeom pgm;
    if (pipe_eligible(q, &pgm)) {
        p->pipe = true;
        p->Hv = q.Hin; p->Wv = q.Win; p->var = 0;
        g.th = pgm.th; g.tw = pgm.tw; g.nb = g.nout = q.cout; g.classes = 1; g.lds_bytes = pgm.lds_bytes; g.nthreads = pgm.nthreads;
        p->tiles_x = q.Win / pgm.tw;
        p->tiles_y = q.Hin / pgm.th;
        const int tiles_s = p->tiles_x * p->tiles_y;
        // persistent workgroups of 8 tiles: C = 32 (8 x 32 tiles, two workgroups per CU): 128 workgroups per T = 1024 sample, a shard of
        // four samples = one round of 512; C = 64 (one workgroup per CU): 32 per sample.  Long samples keep the tile count per workgroup
        int tpw = 8;
        if (const int v = knobs().pipe_tpw; v > 0) tpw = q.cin == 32 ? (v & 0xff) : ((v >> 8) ? (v >> 8) : tpw);  // tuning: L0 | L1 << 8
        if (tpw > tiles_s) tpw = tiles_s;
        p->tiles_per_wg = tpw;
        p->wgs_per_sample = cdiv(tiles_s, tpw);
        return 0;
    }
    if (q.kernel_pref == 2) return fail("conv %d->%d %dx%d xf=%d act=%d: not eligible for the software-pipelined kernel", q.cin, q.cout, q.Hin, q.Win, q.xf, q.act);
    WregGeom wgm;
    if (wreg_eligible(q, &wgm)) {
        p->wreg = true;
        const int sxy = q.mode == DOWN4 ? 2 : 1;
        p->Hv = q.Hin / sxy; p->Wv = q.Win / sxy; p->var = 0;
        g.th = wgm.th; g.tw = wgm.tw; g.nout = q.mode == UP4 ? 2 * q.cout : q.cout; g.nb = g.nout / wgm.nsplit; g.classes = q.mode == UP4 ? 2 : 1;
        g.lds_bytes = wgm.lds_bytes; g.nthreads = wgm.nthreads;
        p->tiles_x = p->Wv / wgm.tw;
        p->tiles_y = p->Hv / wgm.th;
        const int tiles_s = p->tiles_x * p->tiles_y;
        int wps = tiles_s < 128 ? tiles_s : 128;
        if (tiles_s / 4 > wps) wps = tiles_s / 4;
        if (const int v = knobs().conv_wps; v > 0) wps = v < tiles_s ? v : tiles_s;
        // level 2 (C = 96, twelve-wave workgroups): two tiles per workgroup -- the 6.7 us prologue (GroupNorm partials, weight
        // warm-up, first halo) is paid once per 2 x 5 us of tile work instead of once per 5: +1.5-2 % sample-fwd/s at B = 8 with
        // the two shards in flight (same-box A/B, round 3; four tiles: -4 %; the same at C = 64 / 128: -1 / -2.5 %)
        if (q.mode == CONV3 && q.cin == 96 && tiles_s >= 4 && wps > tiles_s / 2) wps = tiles_s / 2;
        // Down / Upsample: at least two tiles per workgroup from 64 tiles per sample up (+0.5-1 %, round 3)
        if (q.mode != CONV3 && tiles_s >= 64 && cdiv(tiles_s, wps) < 2) wps = tiles_s / 2;
        p->tiles_per_wg = cdiv(tiles_s, wps);
        p->wgs_per_sample = cdiv(tiles_s, p->tiles_per_wg);
        return 0;
    }
    if (q.mode == DOWN4 && ((q.Hin | q.Win) & 1)) return fail("downsample needs even H, W (got %d x %d)", q.Hin, q.Win);
    p->Hv = q.mode == DOWN4 ? q.Hin / 2 : q.Hin;
    p->Wv = q.mode == DOWN4 ? q.Win / 2 : q.Win;
    p->var = conv_pick_variant(q.dtype, q.mode, q.cin, q.cout, (q.batch_plan || g_batch_plan) ? q.B : 0, p->Hv, p->Wv);
    if (conv_geometry(q.dtype, q.mode, q.cin, q.cout, p->var, &g) != hipSuccess)
        return fail("conv %d->%d mode %d dtype %d: no kernel", q.cin, q.cout, q.mode, q.dtype);
    p->tiles_x = cdiv(p->Wv, g.tw);
    p->tiles_y = cdiv(p->Hv, g.th);
    // persistent workgroups: each walks tiles_per_wg consecutive tiles of ONE sample.  The split depends only
    // on the sample's size, never on the batch, so a sample's statistics partials (and hence its result, bit
    // for bit) are the same alone, inside any batch, or on any number of GPUs.
    const int tiles_s = p->tiles_x * p->tiles_y;
    int wps = tiles_s < 128 ? tiles_s : 128;
    if (tiles_s / 4 > wps) wps = tiles_s / 4;  // long spectrograms (T >= 2048): at most 4 tiles per workgroup, so that a
                                               // single sample still fills the 256 CUs
    if (q.cin >= 64 && tiles_s >= 512 && wps < 256) wps = 256;  // streamed-weight levels of long samples: 2 tiles per workgroup
    // Down / Upsample with exactly one tile per workgroup (levels 1-2 at T = 1024): two tiles per workgroup halve the
    // per-workgroup costs (82 KB of weights, statistics tail): 114 -> 106 / 97 -> 88 / 61 -> 58 us at B = 8
    // (profiles/r02/downup_wps.txt); a single short sample pays about 12 us per launch for the emptier grid.
    if (q.mode != CONV3 && tiles_s == 128) wps = 64;
    if (const int v = knobs().conv_wps; v > 0) wps = v < tiles_s ? v : tiles_s;
    p->tiles_per_wg = cdiv(tiles_s, wps);
    p->wgs_per_sample = cdiv(tiles_s, p->tiles_per_wg);
    return 0;
}
constexpr int kNumCUs = 256;  // MI355X
// consumer-side GroupNorm finalisation pays 1-1.5 us per round of workgroups (prologue + the producer's tail), a finalize launch
// 5 us + a kernel boundary that the other batch shard partly fills: B = 8 stays launch-free, B >= 32 mostly does not
constexpr int kGnFuseConvRounds = 3, kGnFuseResidRounds = 2;
// how many times over the launch fills the chip (workgroup "rounds" per CU slot)
static int conv_rounds(const ConvPlan& p, int B) {
    const long long wgs = (long long)p.wgs_per_sample * B * (p.g.nout / p.g.nb) * p.g.classes;
    int per_cu = (160 * 1024) / p.g.lds_bytes;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2048 / p.g.nthreads) per_cu = 2048 / p.g.nthreads;
    return (int)((wgs + (long long)kNumCUs * per_cu - 1) / ((long long)kNumCUs * per_cu));
}
// launches one fused conv; returns the stats slab geometry (nparts, Cs) it produced
static int run_conv(const ConvCall& q, hipStream_t s, int* nparts, int* Cs) {
    ConvPlan pl;
    CHK(conv_plan(q, &pl));
    const ConvGeom& g = pl.g;
    if (pl.wreg || pl.pipe) {
        WregArgs f;
        memset(&f, 0, sizeof(f));
        f.in = q.in; f.wf = q.wf; f.skip = q.skip; f.bias = q.bias; f.chan_add = q.chan_add; f.chan_add_stride = q.chan_add_stride;
        f.in_scale = q.in_scale; f.in_shift = q.in_shift; f.gn = q.gn; f.out = q.out; f.stats = q.stats;
        f.stats_groups_c = q.groups ? q.cout : 0; f.xf = q.xf; f.act = q.act; f.stamps = q.stamps;
#ifdef DDIMX_STAMP
        { static const int dbg = getenv("DDIMX_PIPE_DBG") ? atoi(getenv("DDIMX_PIPE_DBG")) : 0; f.dbg = dbg; }
#endif
        if (q.gn.stats && q.gn.np > kGnFuseMaxParts) return fail("conv: %d statistics partials per sample cannot be finished in-kernel", q.gn.np);
        if (q.xf != XF_NONE && !q.gn.stats && (!q.in_scale || !q.in_shift)) return fail("conv: affine input without scale / shift");
        f.B = q.B; f.H = q.Hin; f.W = q.Win;
        f.tiles_x = pl.tiles_x; f.tiles_y = pl.tiles_y; f.tiles_per_wg = pl.tiles_per_wg; f.wgs_per_sample = pl.wgs_per_sample;
        if (nparts) *nparts = f.wgs_per_sample * g.classes * (q.groups ? g.nout / g.nb : 1);
        if (Cs) *Cs = g.nout;
        if (pl.pipe) HIPCHK(pipe_launch(q.cin, q.xf, f, s));
        else HIPCHK(wreg_launch(q.mode, q.cin, g.nout, f, s));
        return 0;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = q.in; a.w = q.w; a.bias = q.bias; a.chan_add = q.chan_add; a.chan_add_stride = q.chan_add_stride;
    a.in_scale = q.in_scale; a.in_shift = q.in_shift; a.xf = q.xf; a.act = q.act;
    a.skip = q.skip; a.out = q.out; a.stats = q.stats;
    a.gn = q.gn;
    a.aux = q.aux; a.aux_scale = q.aux_scale; a.aux_shift = q.aux_shift; a.bwd_mode = q.bwd_mode;
    a.stats_groups_c = q.groups ? q.cout : 0;
    if (q.gn.stats && q.gn.np > kGnFuseMaxParts) return fail("conv: %d statistics partials per sample cannot be finished in-kernel", q.gn.np);
    if (q.groups && q.cout % kGroups) return fail("conv: group-format statistics need cout %% 8 == 0");
    a.B = q.B; a.Hin = q.Hin; a.Win = q.Win;
    a.stamps = q.stamps;
    a.Hv = pl.Hv; a.Wv = pl.Wv;
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y;
    a.tiles_per_wg = pl.tiles_per_wg; a.wgs_per_sample = pl.wgs_per_sample;
    const int var = pl.var;
    if (nparts) *nparts = a.wgs_per_sample * g.classes * (q.groups ? g.nout / g.nb : 1);
    if (Cs) *Cs = g.nout;
    HIPCHK(conv_launch(q.dtype, q.mode, q.cin, q.cout, var, a, s));
    return 0;
}

struct RBPtrs {
    const float *g0, *b0, *g1, *b1, *g2, *bias1;
    const void *w0, *w1;
    const void *w0f = nullptr, *w1f = nullptr;  // fragment-order copies (conv_wreg.h) or null
};

// What the training forward of one Residual_Block keeps for its backward: the two pre-activation tensors and the
// GroupNorm constants.  small: [6][B][C] folded (scale, shift) of GN0, GN1, GN2 then [3][B][8][2] (mean, rstd).
struct RBTape {
    void *u1, *u2;
    float* small;
    float* sc(int i, int B, int C) const { return small + (size_t)(2 * i) * B * C; }
    float* sh(int i, int B, int C) const { return small + (size_t)(2 * i + 1) * B * C; }
    float* mr(int i, int B, int C) const { return small + (size_t)6 * B * C + (size_t)i * B * kGroups * 2; }
};
static inline size_t rb_tape_small_floats(int B, int C) { return (size_t)6 * B * C + (size_t)3 * B * kGroups * 2; }

// Residual_Block (models/diffusion.py:42-56).  The stats of x must already be in `stats`
// ([B][x_nparts][x_Cs][2]).  On return, if want_stats, `stats` holds those of y (*y_nparts, Cs = C).
// tape != null: training forward -- the convs store their PRE-activation outputs (u1 = conv0 + temb, u2 = conv1 + bias)
// in the tape and the consumers apply SiLU while loading; the result is the same function of the same inputs.
// stats2 != null (inference walk, no tape): launch-free GroupNorm (gn_fused.h).  All partials are in group format
// ([B][np][8][2], x_Cs unused), every consumer finishes its input's normalisation itself, and the two buffers alternate
// because a kernel now reads its input's partials while it writes its output's: x in `stats` -> conv0 -> `stats2` ->
// conv1 -> `stats` -> resid -> `stats2` = those of y (the caller swaps).  Samples with more than kGnFuseMaxParts partials
// (long spectrograms, shallow levels) take one gn_finalize_groups launch instead, per GroupNorm.
static int run_resblock(int dtype, int C, const void* x, void* y, const float* temb, int temb_stride, const RBPtrs& p,
                        void* h1, void* h2, float* stats, float* scale, float* shift, int x_nparts, int x_Cs,
                        bool want_stats, int* y_nparts, int B, int H, int W, hipStream_t s, const RBTape* tape = nullptr,
                        float* stats2 = nullptr) {
    const double cnt = (double)H * W * (C / kGroups);
    const float eps = 1e-6f;
    int np = 0, cs = 0;
    if (stats2) {
        if (tape) return fail("run_resblock: the launch-free GroupNorm path keeps no tape");
        // GroupNorm input of one consumer: finished inside the consumer, or -- scale / shift from one launch -- when the sample
        // has too many partials, or when the consumer launch fills the chip so many times over that its per-workgroup prologue
        // (about a microsecond per round) costs more than the launch it saves (large batches).  Either way the numbers are the
        // same bit for bit: gn_finalize_groups runs the consumer's own reduction with the consumer's block size.
        auto gn_of = [&](const float* st, int n, const float* gamma, const float* beta, GnIn* g, bool* fused, int which, int rounds,
                         int max_rounds, int nthreads) -> int {
            *g = GnIn{st, gamma, beta, 1.0 / cnt, eps, n};
            *fused = n <= kGnFuseMaxParts && rounds <= max_rounds && !(knobs().gn_dbg & which);
            if (!*fused) HIPCHK(gn_finalize_groups_launch(*g, C, scale, shift, B, nthreads, s));
            return 0;
        };
        GnIn g; bool fu;
        // (each conv is planned exactly as run_conv will plan it -- kernel family, block size -- BEFORE its GroupNorm input is
        // decided: the finalize launch must reduce with the block size of the kernel that would otherwise do it in its prologue)
        ConvCall k1 = {dtype, CONV3, C, C, x, p.w0, nullptr, temb, temb_stride, scale, shift, XF_AFFINE_SILU, 1, nullptr, h1, stats2, B, H, W};
        k1.groups = true;
        k1.wf = p.w0f;
        ConvPlan pl;
        CHK(conv_plan(k1, &pl));
        CHK(gn_of(stats, x_nparts, p.g0, p.b0, &g, &fu, 2, conv_rounds(pl, B), kGnFuseConvRounds, pl.g.nthreads));
        if (fu) k1.gn = g;
        CHK(run_conv(k1, s, &np, &cs));
        ConvCall k2 = {dtype, CONV3, C, C, h1, p.w1, p.bias1, nullptr, 0, scale, shift, XF_AFFINE, 1, nullptr, h2, stats, B, H, W};
        k2.groups = true;
        k2.wf = p.w1f;
        CHK(conv_plan(k2, &pl));
        CHK(gn_of(stats2, np, p.g1, p.b1, &g, &fu, 2, conv_rounds(pl, B), kGnFuseConvRounds, pl.g.nthreads));
        if (fu) k2.gn = g;
        CHK(run_conv(k2, s, &np, &cs));
        const int rparts = resid_nparts(dtype, H * W, C);
        const int rrounds = (int)(((long long)rparts * B + kNumCUs * 8 - 1) / (kNumCUs * 8));
        CHK(gn_of(stats, np, p.g2, nullptr, &g, &fu, 1, rrounds, kGnFuseResidRounds, resid_threads(dtype, C)));
        HIPCHK(resid_launch(dtype, x, h2, 0, scale, shift, y, want_stats ? stats2 : nullptr, B, H * W, C, s, fu ? &g : nullptr, 1));
        if (y_nparts) *y_nparts = rparts;
        return 0;
    }
    float *sc0 = scale, *sh0 = shift, *sc1 = scale, *sh1 = shift, *sc2 = scale, *sh2 = shift;
    float *mr0 = nullptr, *mr1 = nullptr, *mr2 = nullptr;
    if (tape) {
        sc0 = tape->sc(0, B, C); sh0 = tape->sh(0, B, C); sc1 = tape->sc(1, B, C); sh1 = tape->sh(1, B, C);
        sc2 = tape->sc(2, B, C); sh2 = tape->sh(2, B, C);
        mr0 = tape->mr(0, B, C); mr1 = tape->mr(1, B, C); mr2 = tape->mr(2, B, C);
        h1 = tape->u1; h2 = tape->u2;
    }
    const int act = tape ? 2 : 1;
    HIPCHK(gn_finalize_launch(stats, x_nparts, x_Cs, C, cnt, p.g0, p.b0, eps, sc0, sh0, B, s, mr0));
    ConvCall k1 = {dtype, CONV3, C, C, x, p.w0, nullptr, temb, temb_stride, sc0, sh0, XF_AFFINE_SILU, act,
                   nullptr, h1, stats, B, H, W};
    CHK(run_conv(k1, s, &np, &cs));
    HIPCHK(gn_finalize_launch(stats, np, cs, C, cnt, p.g1, p.b1, eps, sc1, sh1, B, s, mr1));
    ConvCall k2 = {dtype, CONV3, C, C, h1, p.w1, p.bias1, nullptr, 0, sc1, sh1, tape ? XF_SILU_AFFINE : XF_AFFINE, act,
                   nullptr, h2, stats, B, H, W};
    CHK(run_conv(k2, s, &np, &cs));
    HIPCHK(gn_finalize_launch(stats, np, cs, C, cnt, p.g2, nullptr, eps, sc2, sh2, B, s, mr2));
    HIPCHK(resid_launch(dtype, x, h2, tape ? 2 : 0, sc2, sh2, y, want_stats ? stats : nullptr, B, H * W, C, s));
    if (y_nparts) *y_nparts = resid_nparts(dtype, H * W, C);
    return 0;
}

// ---- weight gradient of one convolution: MFMA partial slabs + fixed-order reduction into dst[co][ci][taps] ----
static void wgrad_plan(const WgradGeom& g, int B, int Hd, int Wd, int* tiles_x, int* tiles_y, int* nsplit, int* per) {
    *tiles_x = cdiv(Wd, g.tw);
    *tiles_y = cdiv(Hd, g.th);
    const int total = B * *tiles_x * *tiles_y;
    // workgroups per launch: 2 per CU alone on the chip; 1.5 per CU where the launch shares the chip with the data-gradient chain (the
    // weight-gradient branch, WgSide: 49.1-49.4 vs 49.5-49.9 ms per step, profiles/r04/wgside/wgrad_split_ab.txt)
    int want = (knobs().wgrad_side != 0 ? 384 : 512) / g.grid_y;
    if (const int v = knobs().wgrad_split; v > 0) want = v / g.grid_y;  // tuning hook
    if (want < 1) want = 1;
    if (want > total) want = total;
    *per = cdiv(total, want);
    *nsplit = cdiv(total, *per);
}
static size_t wgrad_partial_floats(int dtype, int mode, int ci, int co, int B, int Hd, int Wd) {
    WgradGeom g;
    if (wgrad_geometry(dtype, mode, ci, co, &g) != hipSuccess) return 0;
    int tx, ty, ns, per;
    wgrad_plan(g, B, Hd, Wd, &tx, &ty, &ns, &per);
    return (size_t)ns * g.ntaps * co * ci;
}
// a: [B][Ha][Wa][ci] (halo operand, transformed by xf), du: [B][Hd][Wd][co]; dst fp32 [co][ci][taps]
static int run_wgrad(int dtype, int mode, int ci, int co, const void* a_t, const void* du, const float* a_scale,
                     const float* a_shift, int xf, float* partial, float* dst, int B, int Hd, int Wd, hipStream_t s) {
    WgradGeom g;
    if (wgrad_geometry(dtype, mode, ci, co, &g) != hipSuccess)
        return fail("weight gradient %d x %d mode %d dtype %d: no kernel", ci, co, mode, dtype);
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.a = a_t; a.du = du; a.a_scale = a_scale; a.a_shift = a_shift; a.xf = xf; a.partial = partial;
    a.B = B; a.Hd = Hd; a.Wd = Wd;
    a.Ha = mode == DOWN4 ? 2 * Hd : Hd;
    a.Wa = mode == DOWN4 ? 2 * Wd : Wd;
    int ns;
    wgrad_plan(g, B, Hd, Wd, &a.tiles_x, &a.tiles_y, &ns, &a.tiles_per_wg);
    a.total_tiles = B * a.tiles_x * a.tiles_y;
    HIPCHK(wgrad_launch(dtype, mode, ci, co, a, ns, s));
    HIPCHK(wgrad_reduce_launch(partial, ns, g.ntaps, co, ci, dst, s));
    return 0;
}

// The weight-gradient branch of the backward (ddimx_unet_bwd_forked).  A conv's weight gradient needs its output gradient `du`
// and the saved forward tensor and feeds nothing but the parameter's gradient slot, so it leaves the data-gradient chain: it is
// issued on a second stream behind an event and the chain goes on.  WHEN it is issued decides what it shares the chip with, and
// that decides whether anything is gained (profiles/r04/wgside/): next to the data-gradient convs (forked as soon as `du`
// exists) both kernels want the same VALU + matrix cycles and each simply takes longer (wgrad 128 -> 204 us, conv 137 -> 196 us
// on average: zero sum); next to the GroupNorm-backward passes (HBM only) the two overlap for real.  So both weight gradients of a
// block are forked behind its LAST data-gradient conv and run beside the block's final apply pass and the next block's statistics
// and first apply pass.  The branch owns its slab buffer and four `du` buffers: a block writes du2 / du1 into the pair of its
// parity, and before the block after next overwrites that pair it waits for the event behind the pair's last reader.  Every fork,
// release and join records an event of its own (nothing is re-recorded inside one capture).
struct WgSide {
    hipStream_t st = nullptr;  // null: one stream, nothing below is used
    void* const* ev = nullptr;
    int n = 0, used = 0;
    int early = 0;             // DDIMX_WGRAD_SIDE=2 (A/B): fork each weight gradient as soon as its `du` exists
    bool early_block = false;  // ... for the block about to run only (the walk's last block: nothing follows that its branch could run beside)
    float* partial = nullptr;
    void* du[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t du_free[2] = {nullptr, nullptr};
    int blk = 0;               // Residual_Blocks seen so far (parity -> buffer pair)
    // Weight gradients that wait for the bottleneck: the up path's blocks of the largest levels write their du2 / du1 into buffers
    // of their own (`hold`, non-null for such a block) and queue their launches here; the queue is issued on the branch when the
    // chain enters the FNet backward -- ~3.5 ms of launch-bound kernels at 32 samples, under which these run almost for free.
    void* const* hold = nullptr;
    std::vector<std::function<int()>> held;
    bool on() const { return st != nullptr; }
    int flush_held(hipStream_t s) {
        if (held.empty()) return 0;
        CHK(fork(s));
        for (auto& f : held) CHK(f());
        held.clear();
        return 0;
    }
    int next(hipEvent_t* e) {
        if (used >= n) return fail("ddimx_unet_bwd_forked: %d events are not enough (ddimx_bwd_side_events)", n);
        *e = (hipEvent_t)ev[used++];
        return 0;
    }
    // the branch may read what `s` has produced so far
    int fork(hipStream_t s) {
        hipEvent_t e;
        CHK(next(&e));
        HIPCHK(hipEventRecord(e, s));
        HIPCHK(hipStreamWaitEvent(st, e, 0));
        return 0;
    }
    // everything the branch has been given so far reads buffer pair p no more
    int release(int p) {
        hipEvent_t e;
        CHK(next(&e));
        HIPCHK(hipEventRecord(e, st));
        du_free[p] = e;
        return 0;
    }
    // `s` is about to overwrite buffer pair p
    int claim(int p, hipStream_t s) {
        if (du_free[p]) HIPCHK(hipStreamWaitEvent(s, du_free[p], 0));
        du_free[p] = nullptr;
        return 0;
    }
    int join(hipStream_t s) {
        hipEvent_t e;
        CHK(next(&e));
        HIPCHK(hipEventRecord(e, st));
        HIPCHK(hipStreamWaitEvent(s, e, 0));
        du_free[0] = du_free[1] = nullptr;
        return 0;
    }
};

// Gradient destinations of one Residual_Block (fp32, the parameters' own layouts); dtemb: [B][stride] slice.
struct RBGrads {
    float *g0, *b0, *g1, *b1, *g2, *w0, *w1, *bias1;
    float* dtemb; int dtemb_stride;
};
// Scratch of the block backward (carved by the caller)
struct RBBwdWs {
    void *du, *dg;          // activation-sized
    float *stats, *coef, *dgb, *sums, *partial;
    // whole-network backward: the batch sums of the per-sample parameter-gradient terms are deferred and flushed many at
    // a time (colsum_multi); each block then gets its own 4 slots of [B][2][C] floats in `slots` (null: sum immediately)
    ColsumBatch* defer = nullptr;
    float* slots = nullptr;
    // ... and so are the per-sample channel sums of du2 / du1 (conv.1.bias, the timestep-embedding chunk): each block then writes
    // them into two slabs of its own (`sums2`: [2][sums floats]) and queues the reductions (partsum_multi)
    PartsumBatch* pdefer = nullptr;
    float* sums2 = nullptr;
    size_t sums_f = 0;
};
static int push_colsum(const RBBwdWs& w, const float* src, int B, long long stride, int C, float* dst, hipStream_t s) {
    if (!w.defer) { HIPCHK(colsum_launch(src, B, stride, C, dst, s)); return 0; }
    ColsumBatch& q = *w.defer;
    q.src[q.count] = src; q.dst[q.count] = dst; q.stride[q.count] = stride; q.B[q.count] = B; q.C[q.count] = C;
    ++q.count;  // the caller flushes before the slots are reused (capacity is checked there)
    return 0;
}
static size_t rb_bwd_stats_floats(int dtype, int B, int HW, int C) { return (size_t)B * resid_nparts(dtype, HW, C) * C * 2; }

// Backward of Residual_Block (autograd of models/diffusion.py:42-56).  dy -> dx (+ extra if given); parameter
// gradients are WRITTEN (not accumulated).  wd0 / wd1: data-gradient packings of conv.0 / conv.1.
static int run_resblock_bwd(int dtype, int C, const void* x, const RBTape& tp, const void* dy, const void* extra, void* dx,
                            const float* gam0, const float* gam1, const float* gam2, const void* wd0, const void* wd1,
                            const RBGrads& gr, const RBBwdWs& w, int B, int H, int W, hipStream_t s, WgSide* sd = nullptr,
                            bool stats_ready = false, const void* next_u2 = nullptr) {
    const int HW = H * W;
    const double cnt = (double)HW * (C / kGroups);
    const int np = resid_nparts(dtype, HW, C);
    const bool side = sd && sd->on();
    void* const* hold = side ? sd->hold : nullptr;         // this block's weight gradients wait for the bottleneck
    const int par = side && !hold ? (sd->blk++ & 1) : 0;
    void* const du2 = hold ? hold[0] : (side ? sd->du[2 * par] : w.du);      // gradient of conv.1's output / of conv.0's output
    void* const du1 = hold ? hold[1] : (side ? sd->du[2 * par + 1] : w.du);
    hipStream_t const sw = side ? sd->st : s;              // the weight gradients' stream
    float* const wpart = side ? sd->partial : w.partial;
    const bool early = side && (sd->early || sd->early_block) && !hold;
    const void* const u1 = tp.u1;
    const float *const sc1 = tp.sc(1, B, C), *const sh1 = tp.sh(1, B, C), *const sc0 = tp.sc(0, B, C), *const sh0 = tp.sh(0, B, C);
    float *const gw1 = gr.w1, *const gw0 = gr.w0;
    auto wgrad1 = [=]() { return run_wgrad(dtype, CONV3, C, C, u1, du2, sc1, sh1, XF_SILU_AFFINE, wpart, gw1, B, H, W, sw); };
    auto wgrad0 = [=]() { return run_wgrad(dtype, CONV3, C, C, x, du1, sc0, sh0, XF_AFFINE_SILU, wpart, gw0, B, H, W, sw); };
    float* const slot0 = w.slots ? w.slots : w.dgb;  // [B][2][C] each; with deferral every use keeps its own slot
    const size_t slot_f = (size_t)B * 2 * C;
    float* const dgb2 = slot0;
    float* const sumb = w.slots ? slot0 + slot_f : w.dgb;
    float* const dgb1 = w.slots ? slot0 + 2 * slot_f : w.dgb;
    float* const dgb0 = w.slots ? slot0 + 3 * slot_f : w.dgb;
    // ---- GN2 (fed by SiLU(u2), weight only) and the SiLU in front of it: du2
    // (stats_ready: the kernel that produced dy -- the previous block's last apply pass -- has left these slabs in w.stats)
    if (!stats_ready) HIPCHK(gn_bwd_stats_launch(dtype, 0, dy, tp.u2, nullptr, nullptr, w.stats, B, HW, C, s));
    HIPCHK(gn_bwd_finalize_launch(w.stats, np, C, cnt, gam2, tp.mr(2, B, C), w.coef, dgb2, B, s));
    CHK(push_colsum(w, dgb2, B, 2 * C, C, gr.g2, s));
    if (side && !hold) CHK(sd->claim(par, s));
    float* const sums_a = w.pdefer ? w.sums2 : w.sums;
    float* const sums_b = w.pdefer ? w.sums2 + w.sums_f : w.sums;
    auto psum = [&](const float* src, float* dst, long long stride) -> int {
        if (!w.pdefer) { HIPCHK(partsum_launch(src, B, np, C, dst, stride, s)); return 0; }
        PartsumBatch& q = *w.pdefer;
        if (q.count >= PartsumBatch::kMax) return fail("partsum queue overflow");
        q.src[q.count] = src; q.dst[q.count] = dst; q.dst_stride[q.count] = stride; q.nparts[q.count] = np; q.C[q.count] = C; q.B[q.count] = B;
        ++q.count;
        return 0;
    };
    HIPCHK(gn_bwd_apply_launch(dtype, 0, dy, tp.u2, nullptr, nullptr, w.coef, nullptr, nullptr, du2, sums_a, B, HW, C, s));
    CHK(psum(sums_a, sumb, C));                                     // per-sample channel sums of du2
    CHK(push_colsum(w, sumb, B, C, C, gr.bias1, s));                // conv.1.bias
    // ---- conv.1: weight gradient against GN1(SiLU(u1)), data gradient -> dg
    if (early) CHK(sd->fork(s));
    if (!side || early) CHK(wgrad1());
    // The data-gradient convs take the GroupNorm-backward partial sums of their own output in their epilogue (ConvCfg::BWD:
    // one more read of u1 / x there instead of a pass over dg and u1 / x); the slab count is then the conv's, not resid's.
    auto fused_stats = [&](ConvCall& d, const void* aux, const float* asc, const float* ash, int mode, int* nparts) -> int {
        ConvPlan pl;
        CHK(conv_plan(d, &pl));
        *nparts = pl.wgs_per_sample * pl.g.classes;
        if (!(knobs().bwd_stats_fused & 1) || *nparts > np) { *nparts = 0; return 0; }  // (slabs are sized for resid's partition)
        d.aux = aux; d.aux_scale = asc; d.aux_shift = ash; d.bwd_mode = mode; d.stats = w.stats;
        return 0;
    };
    ConvCall d1 = {dtype, CONV3, C, C, du2, wd1, nullptr, nullptr, 0, nullptr, nullptr, XF_NONE, 0, nullptr, w.dg, nullptr, B, H, W};
    int np1 = 0;
    CHK(fused_stats(d1, tp.u1, nullptr, nullptr, 1, &np1));
    CHK(run_conv(d1, s, nullptr, nullptr));
    // ---- GN1 (fed by SiLU(u1)) and the SiLU in front of it: du1
    if (!np1) { HIPCHK(gn_bwd_stats_launch(dtype, 0, w.dg, tp.u1, nullptr, nullptr, w.stats, B, HW, C, s)); np1 = np; }
    HIPCHK(gn_bwd_finalize_launch(w.stats, np1, C, cnt, gam1, tp.mr(1, B, C), w.coef, dgb1, B, s));
    CHK(push_colsum(w, dgb1, B, 2 * C, C, gr.g1, s));
    CHK(push_colsum(w, dgb1 + C, B, 2 * C, C, gr.b1, s));
    HIPCHK(gn_bwd_apply_launch(dtype, 0, w.dg, tp.u1, nullptr, nullptr, w.coef, nullptr, nullptr, du1, sums_b, B, HW, C, s));
    if (gr.dtemb) CHK(psum(sums_b, gr.dtemb, gr.dtemb_stride));     // timestep-embedding chunk
    // ---- conv.0: weight gradient against SiLU(GN0(x)), data gradient -> dg
    if (early) CHK(sd->fork(s));
    if (!side || early) CHK(wgrad0());
    ConvCall d0 = {dtype, CONV3, C, C, du1, wd0, nullptr, nullptr, 0, nullptr, nullptr, XF_NONE, 0, nullptr, w.dg, nullptr, B, H, W};
    int np0 = 0;
    CHK(fused_stats(d0, x, tp.sc(0, B, C), tp.sh(0, B, C), 2, &np0));
    CHK(run_conv(d0, s, nullptr, nullptr));
    if (hold) {
        sd->held.push_back(wgrad1);
        sd->held.push_back(wgrad0);
    } else if (side && !early) {  // both weight gradients behind the last data-gradient conv: beside the HBM-bound passes that follow
        CHK(sd->fork(s));
        CHK(wgrad1());
        CHK(wgrad0());
    }
    if (side && !hold) CHK(sd->release(par));
    // ---- SiLU behind GN0, GN0 itself, and the identity path
    if (!np0) { HIPCHK(gn_bwd_stats_launch(dtype, 1, w.dg, x, tp.sc(0, B, C), tp.sh(0, B, C), w.stats, B, HW, C, s)); np0 = np; }
    HIPCHK(gn_bwd_finalize_launch(w.stats, np0, C, cnt, gam0, tp.mr(0, B, C), w.coef, dgb0, B, s));
    CHK(push_colsum(w, dgb0, B, 2 * C, C, gr.g0, s));
    CHK(push_colsum(w, dgb0 + C, B, 2 * C, C, gr.b0, s));
    // next_u2: dx is the dy of a block of the same shape whose saved u2 this is -- its first statistics pass rides this kernel
    HIPCHK(gn_bwd_apply_launch(dtype, 1, w.dg, x, dy, extra, w.coef, tp.sc(0, B, C), tp.sh(0, B, C), dx, nullptr, B, HW, C, s,
                               next_u2, next_u2 ? w.stats : nullptr));
    return 0;
}

static inline const float* pf(const ddimx_ctx* c, const void* packed, int i) {
    return (const float*)((const char*)packed + c->specs[i].off);
}
static inline const void* pv(const ddimx_ctx* c, const void* packed, int i) {
    return (const void*)((const char*)packed + c->specs[i].off);
}
static RBPtrs rb_ptrs(const ddimx_ctx* c, const void* packed, const RBW& r) {
    RBPtrs p;
    p.g0 = pf(c, packed, r.g0); p.b0 = pf(c, packed, r.b0); p.g1 = pf(c, packed, r.g1); p.b1 = pf(c, packed, r.b1);
    p.g2 = pf(c, packed, r.g2); p.bias1 = pf(c, packed, r.bias1);
    p.w0 = pv(c, packed, r.w0); p.w1 = pv(c, packed, r.w1);
    if (c->frag_packed == packed && (size_t)r.w1 < c->frag_off.size() && c->frag_off[r.w0]) {
        p.w0f = (const char*)packed + c->frag_off[r.w0];
        p.w1f = (const char*)packed + c->frag_off[r.w1];
    }
    return p;
}

static int run_temb(const float* te, const int64_t* t, const float* w0, const float* b0, const float* w1,
                    const float* b1, const float* w2, const float* b2, float* h1, float* h2, float* out, int B,
                    int pos_ch, int emb_ch, int E, hipStream_t s) {
    HIPCHK(linear_rows_launch(te, t, w0, b0, h1, B, emb_ch, pos_ch, 1, s));
    HIPCHK(linear_rows_launch(h1, nullptr, w1, b1, h2, B, emb_ch, emb_ch, 1, s));
    HIPCHK(linear_rows_launch(h2, nullptr, w2, b2, out, B, E, emb_ch, 0, s));
    return 0;
}

// Transformer_Module (models/diffusion.py:131-167 + transformers modeling_fnet.py:138-279), eval mode.
// x: NHWC bottleneck activation viewed as tokens [B*S][width]; writes O [B*S][width] fp32.
// Dense-weight GEMMs use bf16 MFMA in bf16 mode; the DFT factors always run on the exact fp32 MFMA.
static int fnet_gemm(const Ws& w, hipStream_t s, const float* A, const float* Bm, float* C, int M, int N, int K, int lda,
                     int ldb, int ldc, const float* bias, const float* resid, int act, int accumulate, int bf16,
                     int batch = 1, long long sA = 0, long long sB = 0, long long sC = 0, int srows = 0) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = Bm; g.C = C; g.bias = bias; g.resid = resid; g.partial = w.gpart;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = sA; g.sB = sB; g.sC = sC; g.batch = batch; g.accumulate = accumulate; g.act = act; g.bf16 = bf16;
    g.splitk = sample_splitk(srows > 0 ? srows : M, N, K, bf16);  // srows: rows of ONE sample (M itself for batched GEMMs)
    HIPCHK(gemm_launch(g, s));
    return 0;
}

static int run_fnet(const ddimx_ctx* c, const void* packed, const ddimx_tables* tb, const Ws& w, const void* x, int B,
                    int S, hipStream_t s) {
    const ddimx_config& f = c->cfg;
    const int hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width, M = B * S;
    const float eps = f.fnet_ln_eps;
    const int bf = c->fnet_bf16;
    const bool dense = c->fx_on && c->fx_packed == packed && knobs().fnet_dense != 0 && fnet_mix_supported(S, hid) &&
                       fnet_dense_supported(S, hid, inter) && fnet_dense_supported(S, inter, hid) && fnet_dense_supported(S, width, hid) &&
                       fnet_dense_supported(S, hid, width);
    HIPCHK(layernorm_launch(c->dtype, x, tb->posenc, S, pf(c, packed, c->ln0_w), pf(c, packed, c->ln0_b), eps, w.ln0, M,
                            width, s, dense ? S : 0));
    if (dense) {
        // Three launches per layer instead of six (fnet_dense.hip): the Fourier mixing normalises its input rows on the fly (the
        // previous layer's output LayerNorm: statistics from that layer's last kernel, gamma folded into a per-layer DFT table)
        // and emits the row statistics of its output; the first FFN matrix normalises its operand from them (gamma / beta folded
        // into the packed weights) and applies bias + gelu_new; the second adds bias and the recomputed LayerNorm(Z) residual
        // and emits the statistics of ITS output; compute_out absorbs the last output LayerNorm the same way.
        const char* pk = (const char*)packed;
        const int npz = hid / 16, npv = hid / 32;
        FnetDenseArgs d;
        memset(&d, 0, sizeof(d));
        d.eps = eps; d.S = S;
        d.W = pk + c->fx_proj; d.bias = pf(c, packed, c->proj_b); d.X = w.ln0; d.x_chunk = 1; d.out = w.vc; d.out_chunk = 1; d.K = width; d.N = hid;
        HIPCHK(fnet_dense_launch(d, B, bf, s));
        for (int i = 0; i < f.fnet_layers; ++i) {
            const ddimx_ctx::FL& L = c->fl[i];
            const ddimx_ctx::FX& X = c->fx[i];
            FnetMixArgs m;
            memset(&m, 0, sizeof(m));
            m.tab = (const float*)(pk + X.tab); m.dft_seq = tb->dft_seq; m.V = w.vc; m.zc = w.zc; m.zstats = w.pz; m.eps = eps; m.S = S; m.hid = hid;
            if (i > 0) {
                m.vstats = w.pv; m.gamma = pf(c, packed, c->fl[i - 1].ln2_w); m.beta = pf(c, packed, c->fl[i - 1].ln2_b);
                m.bc = (const float*)(pk + X.bc);
            }
            HIPCHK(fnet_mix2_launch(m, B, s));
            memset(&d, 0, sizeof(d));
            d.eps = eps; d.S = S;
            d.W = pk + X.w1f; d.bias = (const float*)(pk + X.b1f); d.X = w.zc; d.x_chunk = 1; d.xstats = w.pz; d.xnp = npz; d.xn = 16;
            d.out = w.hc; d.out_chunk = 1; d.out_bf16 = bf; d.act = 1; d.K = hid; d.N = inter;
            HIPCHK(fnet_dense_launch(d, B, bf, s));
            memset(&d, 0, sizeof(d));
            d.eps = eps; d.S = S;
            d.W = pk + X.w2c; d.bias = pf(c, packed, L.b2); d.X = w.hc; d.x_chunk = 1; d.x_bf16 = bf; d.K = inter; d.N = hid;
            d.out = w.vc; d.out_chunk = 1; d.ostats = w.pv;
            d.R = w.zc; d.rstats = w.pz; d.rgamma = pf(c, packed, L.ln1_w); d.rbeta = pf(c, packed, L.ln1_b); d.rnp = npz; d.rn = 16;
            HIPCHK(fnet_dense_launch(d, B, bf, s));
        }
        memset(&d, 0, sizeof(d));
        d.eps = eps; d.S = S;
        d.W = pk + c->fx_coutf; d.bias = (const float*)(pk + c->fx_coutb); d.X = w.vc; d.x_chunk = 1; d.xstats = w.pv; d.xnp = npv; d.xn = 32;
        d.out = w.O; d.K = hid; d.N = width;
        HIPCHK(fnet_dense_launch(d, B, bf, s));
        return 0;
    }
    CHK(fnet_gemm(w, s, w.ln0, pf(c, packed, c->proj_w), w.X, M, hid, width, width, width, hid, pf(c, packed, c->proj_b),
                  nullptr, 0, 0, bf, 1, 0, 0, 0, S));
    float* cur = w.X;
    float* other = w.Y;
    for (int i = 0; i < f.fnet_layers; ++i) {
        const ddimx_ctx::FL& L = c->fl[i];
        // Ut[b] = D_H * X[b]^T -> [2*hid][S]; D_H rows interleaved (2k: cos_k, 2k+1: sin_k), so that row pair k of
        // Ut[b] is one contiguous K-vector [cos-part(S) | sin-part(S)] for the sequence transform
        if (fnet_mix_supported(S, hid)) {
            HIPCHK(fnet_mix_launch(tb->dft_hidden, tb->dft_seq, cur, w.Z, B, S, hid, s));
        } else {
        CHK(fnet_gemm(w, s, tb->dft_hidden, cur, w.Ut, 2 * hid, S, hid, hid, hid, S, nullptr, nullptr, 0, 0, 0, B, 0,
                      (long long)S * hid, (long long)2 * hid * S));
        // Z[b] = [C_S | -S_S] * Ut[b]^T + X[b]   (Re(FFT2) + residual) in one GEMM with K = 2S
        CHK(fnet_gemm(w, s, tb->dft_seq, w.Ut, w.Z, S, hid, 2 * S, 2 * S, 2 * S, hid, nullptr, cur, 0, 0, 0, B, 0,
                      (long long)2 * hid * S, (long long)S * hid));
        }
        HIPCHK(layernorm_launch(DT_F32, w.Z, nullptr, 1, pf(c, packed, L.ln1_w), pf(c, packed, L.ln1_b), eps, other, M, hid, s));
        // FFN; the second GEMM's split-K reduce also applies bias, residual and output.LayerNorm
        CHK(fnet_gemm(w, s, other, pf(c, packed, L.w1), w.Hb, M, inter, hid, hid, hid, inter, pf(c, packed, L.b1), nullptr, 1, 0, bf, 1, 0, 0, 0, S));
        {
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.A = w.Hb; g.B = pf(c, packed, L.w2); g.C = w.Z; g.bias = pf(c, packed, L.b2); g.resid = other; g.partial = w.gpart;
            g.M = M; g.N = hid; g.K = inter; g.lda = inter; g.ldb = inter; g.ldc = hid; g.batch = 1; g.bf16 = bf;
            g.splitk = sample_splitk(S, hid, inter, bf);
            HIPCHK(gemm_ln_launch(g, pf(c, packed, L.ln2_w), pf(c, packed, L.ln2_b), eps, cur, s));
        }
    }
    CHK(fnet_gemm(w, s, cur, pf(c, packed, c->cout_w), w.O, M, width, hid, hid, hid, width, pf(c, packed, c->cout_b), nullptr,
                  0, 0, bf, 1, 0, 0, 0, S));
    return 0;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

int ddimx_abi_version(void) { return DDIMX_ABI_VERSION; }
const char* ddimx_last_error(void) { return g_err; }

int ddimx_create(const ddimx_config* cfg, ddimx_handle* out) {
    if (!cfg || !out) return fail("ddimx_create: null argument");
    ddimx_ctx* c = new ddimx_ctx();
    c->cfg = *cfg;
    if (build_plan(c)) { delete c; return 1; }
    *out = c;
    return 0;
}
int ddimx_destroy(ddimx_handle h) { delete h; return 0; }
int ddimx_set_dropout_counter(ddimx_handle h, const unsigned long long* counter) {
    if (!h) return fail("ddimx_set_dropout_counter: null handle");
    h->dropout_ctr = counter;
    return 0;
}
int ddimx_num_params(ddimx_handle h) { return h ? (int)h->specs.size() : 0; }
int ddimx_param_info(ddimx_handle h, int i, const char** name, long long* numel) {
    if (!h || i < 0 || i >= (int)h->specs.size()) return fail("ddimx_param_info: index %d out of range", i);
    if (name) *name = h->specs[i].name.c_str();
    if (numel) *numel = h->specs[i].numel;
    return 0;
}
long long ddimx_packed_bytes(ddimx_handle h) { return h ? (long long)h->packed_bytes : 0; }
long long ddimx_workspace_bytes(ddimx_handle h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    Ws w;
    carve(h, nullptr, B, T, &w);
    return (long long)w.total;
}

int ddimx_pack_weights(ddimx_handle h, const void* const* params, int n_params, void* packed, void* stream) {
    if (!h || !params || !packed) return fail("ddimx_pack_weights: null argument");
    if (n_params != (int)h->specs.size()) return fail("ddimx_pack_weights: got %d tensors, plan has %zu", n_params, h->specs.size());
    hipStream_t s = (hipStream_t)stream;
    if (h->fx_packed == packed) h->fx_packed = nullptr;  // the folded FNet copies of this buffer are stale from here on
    if (h->frag_packed == packed) h->frag_packed = nullptr;  // ... and so are the fragment-order conv copies
    const int C5 = h->cfg.ch[h->L - 1], Fr = h->Fr;
    PackCopyBatch batch;  // plain copies (norm weights, biases, dense matrices) go out in batches of kMax per launch
    batch.count = 0;
    auto push_copy = [&](const float* src, float* dst, long long n) -> hipError_t {
        batch.src[batch.count] = src; batch.dst[batch.count] = dst; batch.n[batch.count] = n;
        if (++batch.count == PackCopyBatch::kMax) { hipError_t e = pack_copy_multi_launch(batch, s); batch.count = 0; return e; }
        return hipSuccess;
    };
    PackConvBatch convs;  // ... and the conv weights likewise (64 per launch)
    convs.count = 0;
    for (int i = 0; i < n_params; ++i) {
        const ParamSpec& p = h->specs[i];
        const float* src = (const float*)params[i];
        void* dst = (char*)packed + p.off;
        if (!src) return fail("ddimx_pack_weights: parameter %d (%s) is null", i, p.name.c_str());
        switch (p.kind) {
            case PK_COPY: HIPCHK(push_copy(src, (float*)dst, p.numel)); break;
            case PK_CONV: HIPCHK(convs.push(src, dst, p.d0, p.d1, p.d2 * p.d3, 0, h->dtype, s)); break;
            case PK_CONV_F32: HIPCHK(convs.push(src, dst, p.d0, p.d1, p.d2 * p.d3, 0, DT_F32, s)); break;
            case PK_CONVT:
                HIPCHK(pack_convT_launch(h->dtype, src, dst, p.d0, p.d1, s));
                break;
            case PK_BIAS2:
                HIPCHK(push_copy(src, (float*)dst, p.d0));
                HIPCHK(push_copy(src, (float*)dst + p.d0, p.d0));
                break;
            case PK_PERM_COLS: HIPCHK(pack_perm_cols_launch(src, (float*)dst, p.d0, C5, Fr, s)); break;
            case PK_PERM_ROWS: HIPCHK(pack_perm_rows_launch(src, (float*)dst, C5, Fr, p.d1, s)); break;
            default: return fail("bad pack kind");
        }
    }
    HIPCHK(pack_copy_multi_launch(batch, s));
    HIPCHK(pack_conv_multi_launch(convs, s));
    return 0;
}

// The inference-only second copies of the FNet weights (fnet_dense.hip: fragment order, LayerNorm gamma / beta folded in, per-layer
// DFT tables), from the fp32 copies ddimx_pack_weights wrote into `packed` (stream order).  Separate because a training step
// re-packs after every optimizer step and never reads them (38 launches); the host calls it when it packs for eval mode.
// Without it the inference walk must not take the dense FNet path: `ready` is kept in the handle.
int ddimx_pack_fnet_inference(ddimx_handle h, void* packed, void* stream) {
    if (!h || !packed) return fail("ddimx_pack_fnet_inference: null argument");
    hipStream_t s = (hipStream_t)stream;
    h->fx_packed = nullptr;
    h->frag_packed = nullptr;
    // fragment-order copies of the conv weights (conv_wreg.h / conv_pipe.h: inference walk only -- a training step re-packs after every
    // optimizer step and never reads them: 79 launches per step), re-ordered from the tap-layout copies just packed
    for (size_t i = 0; i < h->frag_off.size() && i < h->specs.size(); ++i) {
        if (!h->frag_off[i]) continue;
        const ParamSpec& p = h->specs[i];
        const char* taps = (const char*)packed + p.off;
        if (p.kind == PK_CONV) {
            HIPCHK(pack_frag_from_taps_launch(taps, (char*)packed + h->frag_off[i], p.d2 * p.d3, p.d0, p.d1, s));
        } else if (p.kind == PK_CONVT) {  // both row-parity classes of the sub-pixel form
            const size_t cls_bytes = (size_t)6 * 2 * p.d1 * p.d0 * 2;
            for (int a = 0; a < 2; ++a)
                HIPCHK(pack_frag_from_taps_launch(taps + a * cls_bytes, (char*)packed + h->frag_off[i] + a * cls_bytes, 6, 2 * p.d1, p.d0, s));
        }
    }
    h->frag_packed = packed;
    if (h->fx_on) {
        const ddimx_ctx* c = h;
        const ddimx_config& f = c->cfg;
        const int hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width, bf = c->fnet_bf16;
        char* pk = (char*)packed;
        for (int i = 0; i < f.fnet_layers; ++i) {
            const ddimx_ctx::FL& L = c->fl[i];
            const ddimx_ctx::FX& x = c->fx[i];
            HIPCHK(fnet_fold_launch(pf(c, packed, L.w1), pf(c, packed, L.ln1_w), pf(c, packed, L.ln1_b), pf(c, packed, L.b1),
                                    pk + x.w1f, bf, (float*)(pk + x.b1f), inter, hid, s));
            HIPCHK(fnet_fold_launch(pf(c, packed, L.w2), nullptr, nullptr, nullptr, pk + x.w2c, bf, nullptr, hid, inter, s));
            if (i == 0)
                HIPCHK(fnet_table_launch(nullptr, nullptr, (float*)(pk + x.tab), nullptr, hid, s));
            else
                HIPCHK(fnet_table_launch(pf(c, packed, c->fl[i - 1].ln2_w), pf(c, packed, c->fl[i - 1].ln2_b), (float*)(pk + x.tab),
                                         (float*)(pk + x.bc), hid, s));
        }
        HIPCHK(fnet_fold_launch(pf(c, packed, c->proj_w), nullptr, nullptr, nullptr, pk + c->fx_proj, bf, nullptr, hid, width, s));
        const ddimx_ctx::FL& LL = c->fl[f.fnet_layers - 1];
        HIPCHK(fnet_fold_launch(pf(c, packed, c->cout_w), pf(c, packed, LL.ln2_w), pf(c, packed, LL.ln2_b), pf(c, packed, c->cout_b),
                                pk + c->fx_coutf, bf, (float*)(pk + c->fx_coutb), width, hid, s));
        h->fx_packed = packed;
    }
    return 0;
}

int ddimx_unet_fwd(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace,
                   long long workspace_bytes, const float* x, const int64_t* t, float* eps, int B, int T, void* stream) {
    return ddimx_unet_fwd_forked(h, packed, tables, workspace, workspace_bytes, x, t, eps, B, T, stream, nullptr, nullptr, 0, 0);
}

// Model.forward with part of the network run as TWO batch shards on two streams.  Every op is per sample and its launch plan
// depends on the sample's size only, so an op over samples [0, B) equals the same op over [0, B/2) and [B/2, B): results are
// bit-identical whatever the mask.  fork_mask bit l: the ops whose OUTPUT lives on level l (its Residual_Blocks, the Downsample
// into it, the Upsample into it, the edge convs for level 0) run as two shards, shard 0 on `stream`, shard 1 on `aux_stream`;
// bit 16: the FNet bottleneck.  Consecutive sharded ops stay forked (the shards drift apart freely); the streams are joined
// in front of the next unsharded op.  Where it pays is measured, not assumed: the full-chip, HBM-bound levels
// and the FNet gain from a second stream covering launch gaps and GroupNorm finalisation; the latency-bound deep levels lose.
int ddimx_unet_fwd_forked(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace, long long workspace_bytes,
                          const float* x, const int64_t* t, float* eps, int B, int T, void* stream, void* aux_stream,
                          void* const* events, int n_events, unsigned fork_mask) {
    if (!h || !packed || !tables || !workspace || !x || !t || !eps) return fail("ddimx_unet_fwd: null argument");
    const ddimx_ctx* c = h;
    const ddimx_config& f = c->cfg;
    const int L = c->L;
    if (B < 1) return fail("batch %d", B);
    if (T < (1 << (L - 1)) || T % (1 << (L - 1))) return fail("T=%d must be a positive multiple of %d", T, 1 << (L - 1));
    Ws w;
    carve(c, (char*)workspace, B, T, &w);
    if ((long long)w.total > workspace_bytes) return fail("workspace too small: need %zu bytes, got %lld", w.total, workspace_bytes);
    hipStream_t s = (hipStream_t)stream, sa = (hipStream_t)aux_stream;
    if (!sa || !events || n_events < 2 || B < 2) fork_mask = 0;
    fork_mask &= 0xFFFFFu;
    int ev_used = 0;  // every fork and every join records an event of its own: nothing is re-recorded inside one capture
    const int dt = c->dtype;
    const size_t es = esz(dt);
    const int bh = B / 2;  // shard 0 = samples [0, bh), shard 1 = [bh, B)
    bool forked = false;
    // fork / join in front of an op, as its shardedness requires
    auto sync_for = [&](bool sharded) -> int {
        if (sharded == forked) return 0;
        if (ev_used >= n_events) return fail("ddimx_unet_fwd_forked: %d events are not enough for this fork mask", n_events);
        hipEvent_t ev = (hipEvent_t)events[ev_used++];
        if (sharded) {
            HIPCHK(hipEventRecord(ev, s));
            HIPCHK(hipStreamWaitEvent(sa, ev, 0));
        } else {
            HIPCHK(hipEventRecord(ev, sa));
            HIPCHK(hipStreamWaitEvent(s, ev, 0));
        }
        forked = sharded;
        return 0;
    };
    auto lvl_on = [&](int l) { return (fork_mask >> l) & 1u; };
    auto act_bytes = [&](int l) { return (size_t)(T >> l) * (f.f_size >> l) * f.ch[l] * es; };  // one sample's activation on level l
    struct Lane { int b0, n; hipStream_t st; };
    // runs `op(lane)` once over the whole batch or once per shard
    auto for_lanes = [&](bool sharded, auto&& op) -> int {
        CHK(sync_for(sharded));
        if (!sharded) return op(Lane{0, B, s});
        CHK(op(Lane{bh, B - bh, sa}));
        CHK(op(Lane{0, bh, s}));
        return 0;
    };
    auto at = [&](const void* p, size_t per_sample, int b0) { return (void*)((char*)const_cast<void*>(p) + per_sample * b0); };
    // scratch shared by all levels (h1 / h2, statistics, scale / shift): a shard's share starts at b0 x (the most one sample
    // can need on ANY level) -- the two shards drift apart and may be on different levels at the same time
    // statistics partials (group format, gn_fused.h): buffer `cur` holds those of the tensor the next GroupNorm reads
    int cur = 0;
    auto stats_of = [&](const Lane& ln, int which) { return (which ? w.stats2 : w.stats) + w.stats_per_sample * ln.b0; };
    auto scale_of = [&](const Lane& ln) { return w.scale + (size_t)w.cmax * ln.b0; };
    auto shift_of = [&](const Lane& ln) { return w.shift + (size_t)w.cmax * ln.b0; };

    if (tables->temb_table) {
        HIPCHK(temb_gather_launch(tables->temb_table, t, w.temb, B, c->E, s));
    } else {
        CHK(run_temb(pf(c, packed, c->te), t, pf(c, packed, c->tw[0]), pf(c, packed, c->tb[0]), pf(c, packed, c->tw[1]),
                     pf(c, packed, c->tb[1]), pf(c, packed, c->tw[2]), pf(c, packed, c->tb[2]), w.temb_h1, w.temb_h2, w.temb,
                     B, 128, 512, c->E, s));
    }
    auto resblock = [&](int l, const void* in, void* out, const float* temb_chunk, const RBW& rbw, int np, int cs, bool want_stats,
                        int* ynp) -> int {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        return for_lanes(lvl_on(l), [&](const Lane& ln) -> int {
            return run_resblock(dt, C, at(in, act_bytes(l), ln.b0), at(out, act_bytes(l), ln.b0), temb_chunk + (size_t)c->E * ln.b0, c->E,
                                rb_ptrs(c, packed, rbw), at(w.h1, w.h_per_sample, ln.b0), at(w.h2, w.h_per_sample, ln.b0), stats_of(ln, cur),
                                scale_of(ln), shift_of(ln), np, cs, want_stats, ynp, ln.n, H, W, ln.st, nullptr, stats_of(ln, cur ^ 1));
        });
    };

    cur = 0;
    // ---- down path (models/diffusion.py:252-264) ----
    const size_t in_per = (size_t)f.in_channels * T * f.f_size;  // fp32 NCHW elements per sample at the network boundary
    CHK(for_lanes(lvl_on(0), [&](const Lane& ln) -> int {
        HIPCHK(conv_in_launch(dt, x + in_per * ln.b0, pf(c, packed, c->in_w), pf(c, packed, c->in_b), at(w.A, act_bytes(0), ln.b0),
                              stats_of(ln, 0), ln.n, f.in_channels, f.ch[0], T, f.f_size, ln.st, 1));
        return 0;
    }));
    int np = conv_in_nparts(T, f.f_size), cs = f.ch[0];
    const void* xcur = w.A;
    int bi = 0;
    for (int l = 0; l < L; ++l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        if (l > 0) {
            CHK(for_lanes(lvl_on(l), [&](const Lane& ln) -> int {
                ConvCall d = {dt, DOWN4, f.ch[l - 1], C, at(xcur, act_bytes(l - 1), ln.b0), pv(c, packed, c->down_w[l]),
                              pf(c, packed, c->down_b[l]), nullptr, 0, nullptr, nullptr, XF_NONE, 0, nullptr,
                              at(w.xd[l], act_bytes(l), ln.b0), stats_of(ln, 0), ln.n, H * 2, W * 2};
                d.groups = true;
                if (c->frag_packed == packed && (size_t)c->down_w[l] < c->frag_off.size() && c->frag_off[c->down_w[l]])
                    d.wf = (const char*)packed + c->frag_off[c->down_w[l]];
                return run_conv(d, ln.st, &np, &cs);
            }));
            cur = 0;
            xcur = w.xd[l];
        }
        for (int r = 0; r < f.res[l]; ++r, ++bi) {
            const bool last = (r == f.res[l] - 1);
            int ynp = 0;
            CHK(resblock(l, xcur, w.xd[l], w.temb + c->emb_off_down[bi], c->down_rb[l][r], np, cs, !last, &ynp));
            xcur = w.xd[l];
            cur ^= 1;
            np = ynp; cs = C;
        }
        if (f.res[l] == 0 && l == 0) return fail("level 0 needs at least one residual block");
    }
    // ---- bottleneck (models/diffusion.py:267-279) + first skip add (:284) ----
    const int S = T >> (L - 1), CL = f.ch[L - 1];
    CHK(for_lanes((fork_mask >> 16) & 1u, [&](const Lane& ln) -> int {
        Ws v = w;  // this shard's rows of every token matrix, its share of the split-K scratch
        const size_t rows = (size_t)ln.b0 * S;
        v.ln0 = w.ln0 + (size_t)ln.b0 * (S > 32 ? S : 32) * c->width; v.X = w.X + rows * f.fnet_hidden; v.Z = w.Z + rows * f.fnet_hidden;
        v.Y = w.Y + rows * f.fnet_hidden; v.Hb = w.Hb + rows * f.fnet_inter; v.O = w.O + rows * c->width;
        v.Ut = w.Ut + (size_t)ln.b0 * 2 * f.fnet_hidden * S;
        v.pz = w.pz + (size_t)ln.b0 * (f.fnet_hidden / 16) * 64; v.pv = w.pv + (size_t)ln.b0 * (f.fnet_hidden / 32) * 64;
        v.zc = w.zc + (size_t)ln.b0 * 32 * f.fnet_hidden; v.hc = w.hc + (size_t)ln.b0 * 32 * f.fnet_inter;
        v.vc = w.vc + (size_t)ln.b0 * 32 * f.fnet_hidden;
        v.gpart = w.gpart + w.gpart_per_sample * ln.b0;
        return run_fnet(c, packed, tables, v, at(w.xd[L - 1], act_bytes(L - 1), ln.b0), ln.n, S, ln.st);
    }));
    CHK(for_lanes(lvl_on(L - 1), [&](const Lane& ln) -> int {
        HIPCHK(resid_launch(dt, at(w.xd[L - 1], act_bytes(L - 1), ln.b0), w.O + (size_t)ln.b0 * S * c->width, 1, nullptr, nullptr,
                            at(w.xu[L - 1], act_bytes(L - 1), ln.b0), stats_of(ln, 0), ln.n, S * c->Fr, CL, ln.st, nullptr, 1));
        return 0;
    }));
    cur = 0;
    np = resid_nparts(dt, S * c->Fr, CL); cs = CL;
    // ---- up path (models/diffusion.py:281-292) ----
    bi = 0;
    for (int l = L - 1; l >= 0; --l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        for (int r = 0; r < f.res[l]; ++r, ++bi) {
            const bool last = (r == f.res[l] - 1);
            int ynp = 0;
            CHK(resblock(l, w.xu[l], w.xu[l], w.temb + c->emb_off_up[bi], c->up_rb[l][r], np, cs, !last, &ynp));
            cur ^= 1;
            np = ynp; cs = C;
        }
        if (l > 0) {
            CHK(for_lanes(lvl_on(l - 1), [&](const Lane& ln) -> int {
                ConvCall u = {dt, UP4, C, f.ch[l - 1], at(w.xu[l], act_bytes(l), ln.b0), pv(c, packed, c->up_w[l]), pf(c, packed, c->up_b[l]),
                              nullptr, 0, nullptr, nullptr, XF_NONE, 0, at(w.xd[l - 1], act_bytes(l - 1), ln.b0),
                              at(w.xu[l - 1], act_bytes(l - 1), ln.b0), stats_of(ln, 0), ln.n, H, W};
                u.groups = true;
                if (c->frag_packed == packed && (size_t)c->up_w[l] < c->frag_off.size() && c->frag_off[c->up_w[l]])
                    u.wf = (const char*)packed + c->frag_off[c->up_w[l]];
                return run_conv(u, ln.st, &np, &cs);
            }));
            cur = 0;
        }
    }
    CHK(for_lanes(lvl_on(0), [&](const Lane& ln) -> int {
        HIPCHK(conv_out_launch(dt, at(w.xu[0], act_bytes(0), ln.b0), at(w.A, act_bytes(0), ln.b0), pf(c, packed, c->out_w),
                               pf(c, packed, c->out_b), eps + in_per * ln.b0, ln.n, f.ch[0], f.in_channels, T, f.f_size, ln.st));
        return 0;
    }));
    CHK(sync_for(false));  // leave with everything joined into `stream`
    return 0;
}

// ======================================================================================================
// Training: forward that keeps a tape, and the whole-network backward.
// (reference: functions/losses.py:12-18 builds the graph, runners/diffusion.py:150 `loss.backward()` walks it)
// ======================================================================================================
}  // extern "C"

// Extra weight packings the backward needs: data-gradient layouts of the convs and transposed FNet matrices.
struct BwdPack {
    std::vector<std::vector<size_t>> dn_wd0, dn_wd1, up_wd0, up_wd1;  // [level][r] offsets
    std::vector<size_t> down_dg, up_dg;                              // per level (level 0 unused)
    size_t projT, coutT;
    std::vector<size_t> w1T, w2T;
    size_t total;
};
static void plan_bwd_pack(const ddimx_ctx* c, BwdPack* b) {
    const ddimx_config& f = c->cfg;
    const int L = c->L;
    const size_t es = esz(c->dtype);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
    b->dn_wd0.assign(L, {}); b->dn_wd1.assign(L, {}); b->up_wd0.assign(L, {}); b->up_wd1.assign(L, {});
    b->down_dg.assign(L, 0); b->up_dg.assign(L, 0);
    for (int l = 0; l < L; ++l) {
        const size_t cc = (size_t)9 * f.ch[l] * f.ch[l] * es;
        for (int r = 0; r < f.res[l]; ++r) {
            b->dn_wd0[l].push_back(take(cc)); b->dn_wd1[l].push_back(take(cc));
            b->up_wd0[l].push_back(take(cc)); b->up_wd1[l].push_back(take(cc));
        }
        if (l > 0) {
            b->down_dg[l] = take((size_t)2 * 6 * 2 * f.ch[l - 1] * f.ch[l] * es);  // Conv2d weight read as ConvTranspose2d [I=C][O=Cprev]
            b->up_dg[l] = take((size_t)16 * f.ch[l] * f.ch[l - 1] * es);           // ConvTranspose2d weight read as Conv2d [O=C][I=Cprev]
        }
    }
    const size_t hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width;
    b->projT = take(width * hid * 4);
    b->coutT = take(hid * width * 4);
    for (int i = 0; i < f.fnet_layers; ++i) { b->w1T.push_back(take(hid * inter * 4)); b->w2T.push_back(take(inter * hid * 4)); }
    b->total = off;
}

// What the training forward keeps (carved from the caller's `tape` buffer; depends on B and T).
struct TrainTape {
    float *temb_h1p, *temb_h2p, *temb;
    void* A;
    std::vector<void*> dn_in, up_in;
    std::vector<std::vector<RBTape>> dn_rb, up_rb;
    std::vector<std::vector<void*>> dn_y, up_y;
    float *ln0, *ln0_stat, *X0;
    struct FLT { float *Z, *zstat, *Y1, *pre, *s, *sstat, *Xout; };
    std::vector<FLT> fl;
    size_t total;
};
static void carve_tape(const ddimx_ctx* c, char* base, int B, int T, TrainTape* t) {
    const ddimx_config& f = c->cfg;
    const int L = c->L;
    const size_t es = esz(c->dtype);
    Carver cv{base, 0};
    t->temb_h1p = (float*)cv.take((size_t)B * 512 * 4);
    t->temb_h2p = (float*)cv.take((size_t)B * 512 * 4);
    t->temb = (float*)cv.take((size_t)B * c->E * 4);
    t->A = cv.take((size_t)B * T * f.f_size * f.ch[0] * es);
    t->dn_in.assign(L, nullptr); t->up_in.assign(L, nullptr);
    t->dn_rb.assign(L, {}); t->up_rb.assign(L, {}); t->dn_y.assign(L, {}); t->up_y.assign(L, {});
    for (int l = 0; l < L; ++l) {
        const size_t act = (size_t)B * (T >> l) * (f.f_size >> l) * f.ch[l] * es;
        t->dn_in[l] = l == 0 ? t->A : cv.take(act);
        t->up_in[l] = cv.take(act);
        for (int pass = 0; pass < 2; ++pass)
            for (int r = 0; r < f.res[l]; ++r) {
                RBTape rb;
                rb.u1 = cv.take(act); rb.u2 = cv.take(act);
                rb.small = (float*)cv.take(rb_tape_small_floats(B, f.ch[l]) * 4);
                (pass ? t->up_rb : t->dn_rb)[l].push_back(rb);
                (pass ? t->up_y : t->dn_y)[l].push_back(cv.take(act));
            }
    }
    const int S = T >> (L - 1);
    const size_t M = (size_t)B * S, hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width;
    t->ln0 = (float*)cv.take(M * width * 4);
    t->ln0_stat = (float*)cv.take(M * 2 * 4);
    t->X0 = (float*)cv.take(M * hid * 4);
    t->fl.clear();
    for (int i = 0; i < f.fnet_layers; ++i) {
        TrainTape::FLT q;
        q.Z = (float*)cv.take(M * hid * 4); q.zstat = (float*)cv.take(M * 2 * 4);
        q.Y1 = (float*)cv.take(M * hid * 4); q.pre = (float*)cv.take(M * inter * 4);
        q.s = (float*)cv.take(M * hid * 4); q.sstat = (float*)cv.take(M * 2 * 4);
        q.Xout = (float*)cv.take(M * hid * 4);
        t->fl.push_back(q);
    }
    t->total = cv.off;
}

// Residual blocks whose deferred batch sums fit one ColsumBatch (7 entries each)
constexpr int kDeferBlocks = ColsumBatch::kMax / 7;
// Scratch shared by the training forward and the backward.
struct TrainWs {
    float *stats, *scale, *shift;
    float *Ut, *Hb, *O, *gpart;
    std::vector<void*> Ga, Gb, GS;
    void *gA, *du, *dg, *du_b[3];          // du_b / partial_b: the weight-gradient branch's further `du` buffers and its own slabs (WgSide)
    float *coef, *dgb, *sums, *partial, *partial_b, *slots, *sums_ring;
    size_t sums_f;
    std::vector<std::vector<void*>> hold;  // [level][2 r + {du2, du1}]: the up path's held weight gradients (WgSide::held)
    float *dtemb, *dh2, *dh1;
    float *dO, *dXa, *dXb, *dZ, *dH, *T1, *T2, *lnpart, *dTok, *pgrad;
    size_t total;
};
static void carve_train_ws(const ddimx_ctx* c, char* base, int B, int T, TrainWs* w) {
    const ddimx_config& f = c->cfg;
    const int L = c->L, dt = c->dtype;
    const size_t es = esz(dt);
    Carver cv{base, 0};
    size_t stats_f = (size_t)B * conv_in_nparts(T, f.f_size) * f.ch[0] * 2, hmax = 0, part_f = 0, sums_f = 0;
    int cmax = 0;
    w->Ga.assign(L, nullptr); w->Gb.assign(L, nullptr); w->GS.assign(L, nullptr);
    for (int l = 0; l < L; ++l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        const size_t act = (size_t)B * H * W * C * es;
        w->Ga[l] = cv.take(act); w->Gb[l] = cv.take(act); w->GS[l] = cv.take(act);
        if (act > hmax) hmax = act;
        if (C > cmax) cmax = C;
        size_t q = conv_stats_floats(dt, CONV3, C, C, B, H, W);
        if (q > stats_f) stats_f = q;
        q = (size_t)B * resid_nparts(dt, H * W, C) * C * 2;
        if (q > stats_f) stats_f = q;
        if (q / 2 > sums_f) sums_f = q / 2;
        q = wgrad_partial_floats(dt, CONV3, C, C, B, H, W);
        if (q > part_f) part_f = q;
        if (l > 0) {
            q = conv_stats_floats(dt, DOWN4, f.ch[l - 1], C, B, H, W);
            if (q > stats_f) stats_f = q;
            q = conv_stats_floats(dt, UP4, C, f.ch[l - 1], B, H, W);
            if (q > stats_f) stats_f = q;
            q = wgrad_partial_floats(dt, DOWN4, f.ch[l - 1], C, B, H, W);
            if (q > part_f) part_f = q;
        }
    }
    {
        const size_t q = edge_wgrad_partial_floats(dt, B, f.ch[0], f.in_channels, T, f.f_size);
        if (q > part_f) part_f = q;
    }
    w->gA = cv.take((size_t)B * T * f.f_size * f.ch[0] * es);
    w->du = cv.take(hmax);
    w->dg = cv.take(hmax);
    for (int i = 0; i < 3; ++i) w->du_b[i] = cv.take(hmax);
    w->hold.assign(L, {});
    for (int l = 0; l < L - 1 && l < knobs().wgrad_hold; ++l)
        for (int i = 0; i < 2 * f.res[l]; ++i) w->hold[l].push_back(cv.take((size_t)B * (T >> l) * (f.f_size >> l) * f.ch[l] * es));
    w->stats = (float*)cv.take(stats_f * 4);
    w->scale = (float*)cv.take((size_t)B * cmax * 4);
    w->shift = (float*)cv.take((size_t)B * cmax * 4);
    w->coef = (float*)cv.take((size_t)B * 3 * cmax * 4);
    w->dgb = (float*)cv.take((size_t)B * 2 * cmax * 4);
    w->slots = (float*)cv.take((size_t)kDeferBlocks * 4 * B * 2 * cmax * 4);
    w->sums = (float*)cv.take(sums_f * 4);
    w->sums_f = sums_f;
    w->sums_ring = (float*)cv.take((size_t)kDeferBlocks * 2 * sums_f * 4);
    w->partial = (float*)cv.take(part_f * 4);
    w->partial_b = (float*)cv.take(part_f * 4);
    w->dtemb = (float*)cv.take((size_t)B * c->E * 4);
    w->dh2 = (float*)cv.take((size_t)B * 512 * 4);
    w->dh1 = (float*)cv.take((size_t)B * 512 * 4);
    const int S = T >> (L - 1);
    const size_t M = (size_t)B * S, hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width;
    const size_t big = inter > width ? inter : width;
    w->Ut = (float*)cv.take((size_t)B * 2 * hid * S * 4);
    w->Hb = (float*)cv.take(M * inter * 4);
    w->O = (float*)cv.take(M * width * 4);
    w->dO = (float*)cv.take(M * width * 4);
    w->dXa = (float*)cv.take(M * hid * 4);
    w->dXb = (float*)cv.take(M * hid * 4);
    w->dZ = (float*)cv.take(M * hid * 4);
    w->dH = (float*)cv.take(M * inter * 4);
    w->T1 = (float*)cv.take(M * big * 4);
    w->T2 = (float*)cv.take(M * big * 4);
    w->lnpart = (float*)cv.take((size_t)ln_bwd_nblocks((int)M) * 2 * big * 4);
    w->dTok = (float*)cv.take(M * width * 4);
    w->pgrad = (float*)cv.take(hid * width * 4);
    {   // split-K partial tiles: forward shapes and the backward GEMMs (weight gradients contract over M)
        const int bf = c->fnet_bf16, Mi = (int)M, h = (int)hid, in = (int)inter, wd = (int)width;
        const int shp[][5] = {{Mi, h, wd, 1, bf}, {2 * h, S, h, B, 0}, {S, h, 2 * S, B, 0}, {Mi, in, h, 1, bf}, {Mi, h, in, 1, bf},
                              {Mi, wd, h, 1, bf}, {wd, h, Mi, 1, bf}, {h, in, Mi, 1, bf}, {in, h, Mi, 1, bf}, {h, wd, Mi, 1, bf}};
        size_t mx = 0;
        for (auto& q : shp) {
            const size_t n = (size_t)kMaxSplitK * q[3] * q[0] * q[1];  // the split depends on the per-sample shape only; size for the cap
            if (n > mx) mx = n;
        }
        w->gpart = (float*)cv.take(mx * 4);
    }
    w->total = cv.off;
}

// GEMM helper over the training scratch (same call shape as fnet_gemm)
static int tgemm(const TrainWs& w, hipStream_t s, const float* A, const float* Bm, float* C, int M, int N, int K, const float* bias,
                 const float* resid, int bf16, int batch = 1, long long sA = 0, long long sB = 0, long long sC = 0, int lda = -1,
                 int ldb = -1, int srows = 0) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = Bm; g.C = C; g.bias = bias; g.resid = resid; g.partial = w.gpart;
    g.M = M; g.N = N; g.K = K; g.lda = lda < 0 ? K : lda; g.ldb = ldb < 0 ? K : ldb; g.ldc = N;
    g.sA = sA; g.sB = sB; g.sC = sC; g.batch = batch; g.bf16 = bf16;
    g.splitk = sample_splitk(srows > 0 ? srows : M, N, K, bf16);
    HIPCHK(gemm_launch(g, s));
    return 0;
}
// Re(FFT2(X)) + X over [B][S][hid] token matrices (linear and symmetric: also its own backward)
static int fourier_mix(const ddimx_ctx* c, const ddimx_tables* tb, const TrainWs& w, const float* X, float* Z, int B, int S,
                       hipStream_t s) {
    const int hid = c->cfg.fnet_hidden;
    if (fnet_mix_supported(S, hid)) {
        HIPCHK(fnet_mix_launch(tb->dft_hidden, tb->dft_seq, X, Z, B, S, hid, s));
        return 0;
    }
    CHK(tgemm(w, s, tb->dft_hidden, X, w.Ut, 2 * hid, S, hid, nullptr, nullptr, 0, B, 0, (long long)S * hid, (long long)2 * hid * S));
    CHK(tgemm(w, s, tb->dft_seq, w.Ut, Z, S, hid, 2 * S, nullptr, X, 0, B, 0, (long long)2 * hid * S, (long long)S * hid));
    return 0;
}

// Transformer_Module in training mode (models/diffusion.py:148-167 with the FNet layers of modeling_fnet.py:138-279): tokens
// `x` (NHWC bottleneck activation = [B*S][width] rows) -> w.O [B*S][width] fp32, keeping the tape rows the backward needs.
static int fnet_fwd_train_part(const ddimx_ctx* c, const void* packed, const ddimx_tables* tables, const TrainWs& w, const TrainTape& tp,
                               const void* x, int B, int S, float dropout_p, unsigned long long seed, hipStream_t s) {
    const ddimx_config& f = c->cfg;
    const int dt = c->dtype;
    const int hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width, M = B * S;
    const float eps_ln = f.fnet_ln_eps;
    const int bf = c->fnet_bf16;
    HIPCHK(ln_train_launch(dt, x, tables->posenc, S, pf(c, packed, c->ln0_w), pf(c, packed, c->ln0_b), eps_ln, tp.ln0, nullptr,
                           tp.ln0_stat, M, width, 0.f, seed, 0, s, c->dropout_ctr));
    CHK(tgemm(w, s, tp.ln0, pf(c, packed, c->proj_w), tp.X0, M, hid, width, pf(c, packed, c->proj_b), nullptr, bf, 1, 0, 0, 0, -1, -1, S));
    if (dropout_p > 0.f) HIPCHK(dropout_apply_launch(tp.X0, tp.X0, (long long)M * hid, dropout_p, seed, 0, s, c->dropout_ctr));
    const float* xc = tp.X0;
    for (int i = 0; i < f.fnet_layers; ++i) {
        const ddimx_ctx::FL& Lw = c->fl[i];
        const TrainTape::FLT& q = tp.fl[i];
        CHK(fourier_mix(c, tables, w, xc, q.Z, B, S, s));
        HIPCHK(ln_train_launch(DT_F32, q.Z, nullptr, 1, pf(c, packed, Lw.ln1_w), pf(c, packed, Lw.ln1_b), eps_ln, q.Y1, nullptr,
                               q.zstat, M, hid, 0.f, seed, 0, s, c->dropout_ctr));
        CHK(tgemm(w, s, q.Y1, pf(c, packed, Lw.w1), q.pre, M, inter, hid, pf(c, packed, Lw.b1), nullptr, bf, 1, 0, 0, 0, -1, -1, S));
        HIPCHK(gelu_launch(q.pre, nullptr, w.Hb, (long long)M * inter, 0, s));
        CHK(tgemm(w, s, w.Hb, pf(c, packed, Lw.w2), w.dXa, M, hid, inter, pf(c, packed, Lw.b2), nullptr, bf, 1, 0, 0, 0, -1, -1, S));
        HIPCHK(ln_train_launch(DT_F32, w.dXa, q.Y1, M, pf(c, packed, Lw.ln2_w), pf(c, packed, Lw.ln2_b), eps_ln, q.Xout, q.s,
                               q.sstat, M, hid, dropout_p, seed, (unsigned)(i + 1), s, c->dropout_ctr));
        xc = q.Xout;
    }
    CHK(tgemm(w, s, xc, pf(c, packed, c->cout_w), w.O, M, width, hid, pf(c, packed, c->cout_b), nullptr, bf, 1, 0, 0, 0, -1, -1, S));
    return 0;
}

// Backward of the Transformer_Module: w.dO [B*S][width] fp32 (gradient of its output) -> every transformer.* parameter gradient
// (written at its plan offset of `grads`) and w.dTok [B*S][width] fp32 (gradient of its input tokens `x`).
static int fnet_bwd_part(const ddimx_ctx* c, const void* packed, const char* pb, const BwdPack& bp, const ddimx_tables* tables,
                         const TrainWs& w, const TrainTape& tp, const void* Dlast, float* grads, const std::vector<long long>& goff,
                         int B, int S, float dropout_p, unsigned long long seed, hipStream_t s) {
    const ddimx_config& f = c->cfg;
    const int dt = c->dtype;
    const int hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width, M = B * S;
    const int bf = c->fnet_bf16;
    const int C5 = f.ch[c->L - 1], Fr = c->Fr;
    auto G = [&](int i) { return grads + goff[i]; };
    {   // compute_out: O = Xlast Wc^T + bc   (parameters live in the token-order permutation; gradients are un-permuted)
        const float* Xlast = f.fnet_layers ? tp.fl[f.fnet_layers - 1].Xout : tp.X0;
        HIPCHK(colsum_launch(w.dO, M, width, width, w.pgrad, s));
        HIPCHK(pack_perm_cols_launch(w.pgrad, G(c->cout_b), 1, Fr, C5, s));
        HIPCHK(transpose_launch(w.dO, w.T1, M, width, 0, s));
        HIPCHK(transpose_launch(Xlast, w.T2, M, hid, 0, s));
        CHK(tgemm(w, s, w.T1, w.T2, w.pgrad, width, hid, M, nullptr, nullptr, bf));
        HIPCHK(pack_perm_rows_launch(w.pgrad, G(c->cout_w), Fr, C5, hid, s));
        CHK(tgemm(w, s, w.dO, (const float*)(pb + bp.coutT), w.dXa, M, hid, width, nullptr, nullptr, bf));
    }
    for (int i = f.fnet_layers - 1; i >= 0; --i) {
        const ddimx_ctx::FL& Lw = c->fl[i];
        const TrainTape::FLT& q = tp.fl[i];
        // output.LayerNorm(s), s = Y1 + dropout(FFN)
        HIPCHK(ln_bwd_launch(DT_F32, w.dXa, q.s, nullptr, 1, q.sstat, pf(c, packed, Lw.ln2_w), w.dXb, w.lnpart, G(Lw.ln2_w), G(Lw.ln2_b),
                             M, hid, s));
        const float* dO2 = w.dXb;
        if (dropout_p > 0.f) {
            HIPCHK(dropout_apply_launch(w.dXb, w.dZ, (long long)M * hid, dropout_p, seed, (unsigned)(i + 1), s, c->dropout_ctr));
            dO2 = w.dZ;
        }
        HIPCHK(colsum_launch(dO2, M, hid, hid, G(Lw.b2), s));
        HIPCHK(transpose_launch(dO2, w.T1, M, hid, 0, s));
        HIPCHK(transpose_launch(q.pre, w.T2, M, inter, 1, s));                                   // gelu(pre)^T
        CHK(tgemm(w, s, w.T1, w.T2, G(Lw.w2), hid, inter, M, nullptr, nullptr, bf));             // dW2 [hid][inter]
        CHK(tgemm(w, s, dO2, (const float*)(pb + bp.w2T[i]), w.dH, M, inter, hid, nullptr, nullptr, bf));
        HIPCHK(gelu_launch(w.dH, q.pre, w.dH, (long long)M * inter, 1, s));                      // d(pre)
        HIPCHK(colsum_launch(w.dH, M, inter, inter, G(Lw.b1), s));
        HIPCHK(transpose_launch(w.dH, w.T1, M, inter, 0, s));
        HIPCHK(transpose_launch(q.Y1, w.T2, M, hid, 0, s));
        CHK(tgemm(w, s, w.T1, w.T2, G(Lw.w1), inter, hid, M, nullptr, nullptr, bf));             // dW1 [inter][hid]
        CHK(tgemm(w, s, w.dH, (const float*)(pb + bp.w1T[i]), w.dXa, M, hid, inter, nullptr, w.dXb, bf));  // dY1 = ds + dpre W1
        // fourier.output.LayerNorm(Z), Z = X + Re(FFT2(X))
        HIPCHK(ln_bwd_launch(DT_F32, w.dXa, q.Z, nullptr, 1, q.zstat, pf(c, packed, Lw.ln1_w), w.dXb, w.lnpart, G(Lw.ln1_w), G(Lw.ln1_b),
                             M, hid, s));
        CHK(fourier_mix(c, tables, w, w.dXb, w.dXa, B, S, s));
    }
    {   // embedding: X0 = dropout(LN0(tok + posenc) Wp^T + bp)
        if (dropout_p > 0.f) HIPCHK(dropout_apply_launch(w.dXa, w.dXa, (long long)M * hid, dropout_p, seed, 0, s, c->dropout_ctr));
        HIPCHK(colsum_launch(w.dXa, M, hid, hid, G(c->proj_b), s));
        HIPCHK(transpose_launch(w.dXa, w.T1, M, hid, 0, s));
        HIPCHK(transpose_launch(tp.ln0, w.T2, M, width, 0, s));
        CHK(tgemm(w, s, w.T1, w.T2, w.pgrad, hid, width, M, nullptr, nullptr, bf));
        HIPCHK(pack_perm_cols_launch(w.pgrad, G(c->proj_w), hid, Fr, C5, s));
        CHK(tgemm(w, s, w.dXa, (const float*)(pb + bp.projT), w.dO, M, width, hid, nullptr, nullptr, bf));
        HIPCHK(ln_bwd_launch(dt, w.dO, Dlast, tables->posenc, S, tp.ln0_stat, pf(c, packed, c->ln0_w), w.dTok, w.lnpart, w.pgrad,
                             w.pgrad + width, M, width, s));
        HIPCHK(pack_perm_cols_launch(w.pgrad, G(c->ln0_w), 1, Fr, C5, s));
        HIPCHK(pack_perm_cols_launch(w.pgrad + width, G(c->ln0_b), 1, Fr, C5, s));
    }
    return 0;
}

static inline int down_bi(const ddimx_config& f, int l, int r) { int b = 0; for (int i = 0; i < l; ++i) b += f.res[i]; return b + r; }
static inline int up_bi(const ddimx_config& f, int L, int l, int r) { int b = 0; for (int i = L - 1; i > l; --i) b += f.res[i]; return b + r; }

// per-channel sums of an NHWC tensor over (batch, pixels) -> dst[C]   (bias gradients of Downsample / Upsample)
static int channel_sums(int dt, const void* x, const TrainWs& w, float* dst, int B, int HW, int C, hipStream_t s) {
    HIPCHK(tensor_stats_launch(dt, x, w.stats, B, HW, C, s));
    HIPCHK(partsum_launch(w.stats, B, resid_nparts(dt, HW, C), C, w.dgb, C, s, 2));
    HIPCHK(colsum_launch(w.dgb, B, C, C, dst, s));
    return 0;
}

extern "C" {

long long ddimx_packed_bwd_bytes(ddimx_handle h) {
    if (!h) return 0;
    BwdPack b;
    plan_bwd_pack(h, &b);
    return (long long)b.total;
}
long long ddimx_train_tape_bytes(ddimx_handle h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    TrainTape t;
    carve_tape(h, nullptr, B, T, &t);
    return (long long)t.total;
}
long long ddimx_train_workspace_bytes(ddimx_handle h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    TrainWs w;
    carve_train_ws(h, nullptr, B, T, &w);
    return (long long)w.total;
}
long long ddimx_grad_floats(ddimx_handle h) {
    long long n = 0;
    if (h) for (auto& p : h->specs) n += (p.numel + 63) & ~63ll;
    return n;
}
long long ddimx_grad_offset(ddimx_handle h, int i) {
    if (!h || i < 0 || i >= (int)h->specs.size()) return -1;
    long long n = 0;
    for (int k = 0; k < i; ++k) n += (h->specs[k].numel + 63) & ~63ll;
    return n;
}

int ddimx_pack_weights_bwd(ddimx_handle h, const void* const* params, int n_params, const void* packed, void* packed_bwd,
                           void* stream) {
    if (!h || !params || !packed || !packed_bwd) return fail("ddimx_pack_weights_bwd: null argument");
    if (n_params != (int)h->specs.size()) return fail("ddimx_pack_weights_bwd: got %d tensors, plan has %zu", n_params, h->specs.size());
    const ddimx_ctx* c = h;
    const ddimx_config& f = c->cfg;
    hipStream_t s = (hipStream_t)stream;
    BwdPack b;
    plan_bwd_pack(c, &b);
    char* base = (char*)packed_bwd;
    const int dt = c->dtype;
    PackConvBatch convs;  // all conv packings of the backward in two launches (pack_conv_multi_kernel)
    convs.count = 0;
    for (int l = 0; l < c->L; ++l) {
        const int C = f.ch[l];
        for (int r = 0; r < f.res[l]; ++r) {
            HIPCHK(convs.push((const float*)params[c->down_rb[l][r].w0], base + b.dn_wd0[l][r], C, C, 9, 1, dt, s));
            HIPCHK(convs.push((const float*)params[c->down_rb[l][r].w1], base + b.dn_wd1[l][r], C, C, 9, 1, dt, s));
            HIPCHK(convs.push((const float*)params[c->up_rb[l][r].w0], base + b.up_wd0[l][r], C, C, 9, 1, dt, s));
            HIPCHK(convs.push((const float*)params[c->up_rb[l][r].w1], base + b.up_wd1[l][r], C, C, 9, 1, dt, s));
        }
        if (l > 0) {
            // d(input) of Conv2d(W [C][Cprev][4][4], s2 p1) = ConvTranspose2d with the same tensor read as [I = C][O = Cprev]
            HIPCHK(pack_convT_launch(dt, (const float*)params[c->down_w[l]], base + b.down_dg[l], C, f.ch[l - 1], s));
            // d(input) of ConvTranspose2d(W [C][Cprev][4][4]) = Conv2d with the same tensor read as [O = C][I = Cprev]
            HIPCHK(convs.push((const float*)params[c->up_w[l]], base + b.up_dg[l], C, f.ch[l - 1], 16, 0, dt, s));
        }
    }
    HIPCHK(pack_conv_multi_launch(convs, s));
    const int hid = f.fnet_hidden, inter = f.fnet_inter, width = c->width;
    HIPCHK(transpose_launch(pf(c, packed, c->proj_w), (float*)(base + b.projT), hid, width, 0, s));   // [hid][width] -> [width][hid]
    HIPCHK(transpose_launch(pf(c, packed, c->cout_w), (float*)(base + b.coutT), width, hid, 0, s));   // [width][hid] -> [hid][width]
    for (int i = 0; i < f.fnet_layers; ++i) {
        HIPCHK(transpose_launch(pf(c, packed, c->fl[i].w1), (float*)(base + b.w1T[i]), inter, hid, 0, s));
        HIPCHK(transpose_launch(pf(c, packed, c->fl[i].w2), (float*)(base + b.w2T[i]), hid, inter, 0, s));
    }
    return 0;
}

int ddimx_unet_fwd_train(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace,
                         long long workspace_bytes, void* tape, long long tape_bytes, const float* x, const int64_t* t, float* eps,
                         int B, int T, float dropout_p, unsigned long long seed, void* stream) {
    if (!h || !packed || !tables || !workspace || !tape || !x || !t || !eps) return fail("ddimx_unet_fwd_train: null argument");
    const ddimx_ctx* c = h;
    const ddimx_config& f = c->cfg;
    const int L = c->L, dt = c->dtype;
    if (B < 1) return fail("batch %d", B);
    if (T < (1 << (L - 1)) || T % (1 << (L - 1))) return fail("T=%d must be a positive multiple of %d", T, 1 << (L - 1));
    if (dropout_p < 0.f || dropout_p >= 1.f) return fail("dropout probability %g out of [0, 1)", (double)dropout_p);
    for (int l = 0; l < L; ++l) if (f.res[l] < 1) return fail("training needs at least one residual block per level");
    BatchPlanScope plan_scope;
    TrainWs w;
    carve_train_ws(c, (char*)workspace, B, T, &w);
    if ((long long)w.total > workspace_bytes) return fail("training workspace too small: need %zu bytes, got %lld", w.total, workspace_bytes);
    TrainTape tp;
    carve_tape(c, (char*)tape, B, T, &tp);
    if ((long long)tp.total > tape_bytes) return fail("tape too small: need %zu bytes, got %lld", tp.total, tape_bytes);
    hipStream_t s = (hipStream_t)stream;

    // timestep embedding, keeping the pre-activations (models/diffusion.py:110-120)
    HIPCHK(linear_rows_launch(pf(c, packed, c->te), t, pf(c, packed, c->tw[0]), pf(c, packed, c->tb[0]), tp.temb_h1p, B, 512, 128, 0, s));
    HIPCHK(linear_rows_launch(tp.temb_h1p, nullptr, pf(c, packed, c->tw[1]), pf(c, packed, c->tb[1]), tp.temb_h2p, B, 512, 512, 0, s, 1));
    HIPCHK(linear_rows_launch(tp.temb_h2p, nullptr, pf(c, packed, c->tw[2]), pf(c, packed, c->tb[2]), tp.temb, B, c->E, 512, 0, s, 1));

    HIPCHK(conv_in_launch(dt, x, pf(c, packed, c->in_w), pf(c, packed, c->in_b), tp.A, w.stats, B, f.in_channels, f.ch[0], T, f.f_size, s));
    int np = conv_in_nparts(T, f.f_size), cs = f.ch[0];
    const void* cur = tp.A;
    for (int l = 0; l < L; ++l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        if (l > 0) {
            ConvCall d = {dt, DOWN4, f.ch[l - 1], C, cur, pv(c, packed, c->down_w[l]), pf(c, packed, c->down_b[l]), nullptr, 0,
                          nullptr, nullptr, XF_NONE, 0, nullptr, tp.dn_in[l], w.stats, B, H * 2, W * 2};
            CHK(run_conv(d, s, &np, &cs));
            cur = tp.dn_in[l];
        }
        for (int r = 0; r < f.res[l]; ++r) {
            int ynp = 0;
            CHK(run_resblock(dt, C, cur, tp.dn_y[l][r], tp.temb + c->emb_off_down[down_bi(f, l, r)], c->E,
                             rb_ptrs(c, packed, c->down_rb[l][r]), nullptr, nullptr, w.stats, w.scale, w.shift, np, cs,
                             r != f.res[l] - 1, &ynp, B, H, W, s, &tp.dn_rb[l][r]));
            cur = tp.dn_y[l][r];
            np = ynp; cs = C;
        }
    }
    // bottleneck (models/diffusion.py:267-279), training mode: dropout after the projection and after each FFN
    const int S = T >> (L - 1), CL = f.ch[L - 1];
    CHK(fnet_fwd_train_part(c, packed, tables, w, tp, cur, B, S, dropout_p, seed, s));
    HIPCHK(resid_launch(dt, cur, w.O, 1, nullptr, nullptr, tp.up_in[L - 1], w.stats, B, S * c->Fr, CL, s));
    np = resid_nparts(dt, S * c->Fr, CL); cs = CL;
    for (int l = L - 1; l >= 0; --l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        cur = tp.up_in[l];
        for (int r = 0; r < f.res[l]; ++r) {
            int ynp = 0;
            CHK(run_resblock(dt, C, cur, tp.up_y[l][r], tp.temb + c->emb_off_up[up_bi(f, L, l, r)], c->E,
                             rb_ptrs(c, packed, c->up_rb[l][r]), nullptr, nullptr, w.stats, w.scale, w.shift, np, cs,
                             r != f.res[l] - 1, &ynp, B, H, W, s, &tp.up_rb[l][r]));
            cur = tp.up_y[l][r];
            np = ynp; cs = C;
        }
        if (l > 0) {
            ConvCall u = {dt, UP4, C, f.ch[l - 1], cur, pv(c, packed, c->up_w[l]), pf(c, packed, c->up_b[l]), nullptr, 0,
                          nullptr, nullptr, XF_NONE, 0, tp.dn_y[l - 1].back(), tp.up_in[l - 1], w.stats, B, H, W};
            CHK(run_conv(u, s, &np, &cs));
        }
    }
    HIPCHK(conv_out_launch(dt, cur, tp.A, pf(c, packed, c->out_w), pf(c, packed, c->out_b), eps, B, f.ch[0], f.in_channels, T, f.f_size, s));
    return 0;
}

// Backward of the whole network: d_eps [B][cio][T][F] fp32 -> every parameter gradient, WRITTEN into `grads`
// (fp32, ddimx_grad_floats() floats; parameter i at ddimx_grad_offset(i) in its own shape; the temb.te buffer's slot is
// left untouched).  x, t: the forward's inputs.  The gradient w.r.t. x is not produced (nothing upstream needs it).
int ddimx_unet_bwd(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                   long long workspace_bytes, const void* tape, long long tape_bytes, const float* x, const int64_t* t,
                   const float* d_eps, float* grads, int B, int T, float dropout_p, unsigned long long seed, void* stream) {
    return ddimx_unet_bwd_staged(h, packed, packed_bwd, tables, workspace, workspace_bytes, tape, tape_bytes, x, t, d_eps, grads, B, T,
                                 dropout_p, seed, nullptr, 0, stream);
}

// Gradient buckets in the order the backward completes them (reverse execution order of the forward): the flat gradient buffer
// is in plan order  temb | down_modules | up_modules | transformer, so the three buckets are contiguous ranges of it.
int ddimx_grad_buckets(ddimx_handle h, long long* ranges) {
    if (!h || !ranges) return fail("ddimx_grad_buckets: null argument");
    long long lo[3] = {-1, -1, -1}, hi[3] = {0, 0, 0}, n = 0;
    for (auto& p : h->specs) {
        const long long e = n + ((p.numel + 63) & ~63ll);
        const int b = p.name.rfind("up_modules.", 0) == 0 ? 0 : (p.name.rfind("transformer.", 0) == 0 ? 1 : 2);
        if (lo[b] < 0) lo[b] = n;
        hi[b] = e;
        n = e;
    }
    for (int b = 0; b < 3; ++b) { ranges[2 * b] = lo[b] < 0 ? 0 : lo[b]; ranges[2 * b + 1] = lo[b] < 0 ? 0 : hi[b]; }
    // a bucket must be one contiguous run of the plan: temb + down_modules come first, then up_modules, then transformer
    if (!(ranges[4] == 0 && ranges[5] == ranges[0] && ranges[1] == ranges[2] && ranges[3] == n)) return fail("ddimx_grad_buckets: plan order changed");
    return 0;
}

int ddimx_unet_bwd_staged(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                          long long workspace_bytes, const void* tape, long long tape_bytes, const float* x, const int64_t* t,
                          const float* d_eps, float* grads, int B, int T, float dropout_p, unsigned long long seed,
                          void* const* bucket_events, int n_events, void* stream) {
    return ddimx_unet_bwd_forked(h, packed, packed_bwd, tables, workspace, workspace_bytes, tape, tape_bytes, x, t, d_eps, grads, B, T,
                                 dropout_p, seed, bucket_events, n_events, stream, nullptr, nullptr, 0);
}

// Events ddimx_unet_bwd_forked needs for its weight-gradient branch: per Residual_Block two forks and two buffer releases, one fork
// per Downsample / Upsample weight and for the output conv's, the fork in front of bucket 0's event, the final join.
int ddimx_bwd_side_events(ddimx_handle h) {
    if (!h) return 0;
    int blocks = 0;
    for (int l = 0; l < h->L; ++l) blocks += 2 * h->cfg.res[l];
    return 4 * blocks + 2 * (h->L - 1) + 3;
}

// The backward with its weight gradients on `side_stream` (WgSide above; results are bit-identical to the one-stream call: same
// kernels, same partitions, same order of additions).  The branch is joined into `stream` before the call returns; bucket 0's event
// is recorded on the side stream (behind the up path's last weight gradient AND the chain's batch sums), the other two on `stream`.
// side_stream null (or DDIMX_WGRAD_SIDE=0): one stream.
int ddimx_unet_bwd_forked(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                          long long workspace_bytes, const void* tape, long long tape_bytes, const float* x, const int64_t* t,
                          const float* d_eps, float* grads, int B, int T, float dropout_p, unsigned long long seed,
                          void* const* bucket_events, int n_events, void* stream, void* side_stream, void* const* side_events,
                          int n_side_events) {
    if (!h || !packed || !packed_bwd || !tables || !workspace || !tape || !x || !t || !d_eps || !grads)
        return fail("ddimx_unet_bwd: null argument");
    if (n_events != 0 && (n_events != 3 || !bucket_events)) return fail("ddimx_unet_bwd_staged: pass 0 or 3 bucket events");
    if (side_stream && (!side_events || n_side_events < ddimx_bwd_side_events(h)))
        return fail("ddimx_unet_bwd_forked: %d side events, the plan needs %d", n_side_events, ddimx_bwd_side_events(h));
    const ddimx_ctx* c = h;
    const ddimx_config& f = c->cfg;
    const int L = c->L, dt = c->dtype;
    if (B < 1 || T < (1 << (L - 1)) || T % (1 << (L - 1))) return fail("ddimx_unet_bwd: bad shape B=%d T=%d", B, T);
    for (int l = 0; l < L; ++l) if (f.res[l] < 1) return fail("training needs at least one residual block per level");
    if (dropout_p < 0.f || dropout_p >= 1.f) return fail("dropout probability %g out of [0, 1)", (double)dropout_p);
    BatchPlanScope plan_scope;
    TrainWs w;
    carve_train_ws(c, (char*)workspace, B, T, &w);
    if ((long long)w.total > workspace_bytes) return fail("training workspace too small: need %zu bytes, got %lld", w.total, workspace_bytes);
    TrainTape tp;
    carve_tape(c, (char*)const_cast<void*>(tape), B, T, &tp);
    if ((long long)tp.total > tape_bytes) return fail("tape too small: need %zu bytes, got %lld", tp.total, tape_bytes);
    BwdPack bp;
    plan_bwd_pack(c, &bp);
    const char* pb = (const char*)packed_bwd;
    hipStream_t s = (hipStream_t)stream;
    std::vector<long long> goff(c->specs.size());
    {
        long long n = 0;
        for (size_t i = 0; i < c->specs.size(); ++i) { goff[i] = n; n += (c->specs[i].numel + 63) & ~63ll; }
    }
    auto G = [&](int i) { return grads + goff[i]; };
    RBBwdWs rw = {w.du, w.dg, w.stats, w.coef, w.dgb, w.sums, w.partial};
    WgSide sd;
    if (side_stream && side_stream != stream && knobs().wgrad_side != 0) {
        sd.st = (hipStream_t)side_stream; sd.ev = side_events; sd.n = n_side_events;
        sd.early = knobs().wgrad_side == 2;
        sd.partial = w.partial_b; sd.du[0] = w.du; sd.du[1] = w.du_b[0]; sd.du[2] = w.du_b[1]; sd.du[3] = w.du_b[2];
    }
    hipStream_t const sw = sd.on() ? sd.st : s;            // the weight gradients' stream ...
    float* const wpart = sd.on() ? sd.partial : w.partial;  // ... and slab buffer
    ColsumBatch defer;
    defer.count = 0;
    rw.defer = &defer;
    PartsumBatch pdefer;
    pdefer.count = 0;
    rw.pdefer = &pdefer;
    rw.sums_f = w.sums_f;
    auto flush_sums = [&]() -> int {  // the queued per-sample sums first: some of the batch sums read them
        HIPCHK(partsum_multi_launch(pdefer, s));
        pdefer.count = 0;
        HIPCHK(colsum_multi_launch(defer, s));
        defer.count = 0;
        return 0;
    };
    int deferred_blocks = 0;
    int cmax = 0;
    for (int l = 0; l < L; ++l) if (f.ch[l] > cmax) cmax = f.ch[l];
    auto next_slots = [&]() -> int {  // hands the next block its slots; flushes the batch when the arena is full
        if (deferred_blocks == kDeferBlocks) {
            CHK(flush_sums());
            deferred_blocks = 0;
        }
        rw.slots = w.slots + (size_t)deferred_blocks * 4 * B * 2 * cmax;
        rw.sums2 = w.sums_ring + (size_t)deferred_blocks * 2 * w.sums_f;
        ++deferred_blocks;
        return 0;
    };
    auto rb_grads = [&](const RBW& r, float* dtemb) {
        RBGrads g = {G(r.g0), G(r.b0), G(r.g1), G(r.b1), G(r.g2), G(r.w0), G(r.w1), G(r.bias1), dtemb, c->E};
        return g;
    };

    // ---- output conv (models/diffusion.py:283-292): gradient of `x + hidden[0]`, weight / bias gradients
    HIPCHK(conv_out_bwd_data_launch(dt, d_eps, pf(c, packed, c->out_w), w.gA, B, f.ch[0], f.in_channels, T, f.f_size, s));
    if (sd.on()) CHK(sd.fork(s));
    HIPCHK(edge_wgrad_launch(dt, 1, tp.up_y[0].back(), tp.A, d_eps, wpart, G(c->out_w), G(c->out_b), B, f.ch[0], f.in_channels, T,
                             f.f_size, sw));
    // ---- up path, last level first executed = level 0 ... L-1
    const void* gy = w.gA;
    const bool chain_stats = (knobs().bwd_stats_fused & 2) != 0;
    bool have_stats = false;  // w.stats holds the first statistics pass of the block about to run
    for (int l = 0; l < L; ++l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        for (int r = f.res[l] - 1; r >= 0; --r) {
            const void* xin = r ? tp.up_y[l][r - 1] : tp.up_in[l];
            void* dx = r == 0 ? w.GS[l] : (gy == w.Gb[l] ? w.Ga[l] : w.Gb[l]);
            const RBW& rbw = c->up_rb[l][r];
            CHK(next_slots());
            sd.hold = sd.on() && !sd.early && !w.hold[l].empty() ? &w.hold[l][2 * r] : nullptr;
            // (block r - 1 of the level takes dx as its dy: its first statistics pass rides this block's last kernel)
            const void* nu2 = chain_stats && r > 0 ? tp.up_rb[l][r - 1].u2 : nullptr;
            CHK(run_resblock_bwd(dt, C, xin, tp.up_rb[l][r], gy, nullptr, dx, pf(c, packed, rbw.g0), pf(c, packed, rbw.g1),
                                 pf(c, packed, rbw.g2), pb + bp.up_wd0[l][r], pb + bp.up_wd1[l][r],
                                 rb_grads(rbw, w.dtemb + c->emb_off_up[up_bi(f, L, l, r)]), rw, B, H, W, s, &sd, have_stats, nu2));
            have_stats = nu2 != nullptr;
            gy = dx;
        }
        sd.hold = nullptr;
        // now GS[l] = d(up_in[l]) = gradient of the skip D_l as well
        if (l < L - 1) {
            const int Cn = f.ch[l + 1];
            // up_in[l] = ConvTranspose2d(up_y[l+1].back()) + D_l
            if (sd.on()) CHK(sd.fork(s));  // (GS[l] is not written again in this call)
            CHK(run_wgrad(dt, DOWN4, C, Cn, w.GS[l], tp.up_y[l + 1].back(), nullptr, nullptr, XF_NONE, wpart, G(c->up_w[l + 1]), B,
                          H / 2, W / 2, sw));
            CHK(channel_sums(dt, w.GS[l], w, G(c->up_b[l + 1]), B, H * W, C, s));
            ConvCall d = {dt, DOWN4, C, Cn, w.GS[l], pb + bp.up_dg[l + 1], nullptr, nullptr, 0, nullptr, nullptr, XF_NONE, 0, nullptr,
                          w.Ga[l + 1], nullptr, B, H, W};
            CHK(run_conv(d, s, nullptr, nullptr));
            gy = w.Ga[l + 1];
        }
    }
    if (n_events) {  // bucket 0: every up_modules.* gradient is final once the deferred batch sums are flushed
        CHK(flush_sums());
        deferred_blocks = 0;
    }
    if (sd.on()) CHK(sd.flush_held(s));  // the held weight gradients of the up path: now, under the bottleneck's launch-bound kernels
    if (n_events) {
        if (sd.on()) CHK(sd.fork(s));  // the chain does not wait for the branch here: the event goes behind both on the branch's stream
        HIPCHK(hipEventRecord((hipEvent_t)bucket_events[0], sw));
    }
    // ---- bottleneck: up_in[L-1] = D_{L-1} + O
    const int S = T >> (L - 1), CL = f.ch[L - 1];
    const int width = c->width, M = B * S, Fr = c->Fr;
    const void* Dlast = tp.dn_y[L - 1].back();
    HIPCHK(cast_f32_launch(dt, w.GS[L - 1], w.dO, (long long)M * width, s));
    CHK(fnet_bwd_part(c, packed, pb, bp, tables, w, tp, Dlast, grads, goff, B, S, dropout_p, seed, s));
    if (n_events) HIPCHK(hipEventRecord((hipEvent_t)bucket_events[1], s));  // bucket 1: transformer.* gradients are final
    // d(D_{L-1}) = skip gradient + gradient through the bottleneck
    HIPCHK(resid_launch(dt, w.GS[L - 1], w.dTok, 1, nullptr, nullptr, w.Ga[L - 1], nullptr, B, S * Fr, CL, s));
    gy = w.Ga[L - 1];
    // ---- down path
    for (int l = L - 1; l >= 0; --l) {
        const int H = T >> l, W = f.f_size >> l, C = f.ch[l];
        for (int r = f.res[l] - 1; r >= 0; --r) {
            const void* xin = r ? tp.dn_y[l][r - 1] : tp.dn_in[l];
            void* dx = gy == w.Gb[l] ? w.Ga[l] : w.Gb[l];
            const RBW& rbw = c->down_rb[l][r];
            CHK(next_slots());
            const void* nu2 = chain_stats && r > 0 ? tp.dn_rb[l][r - 1].u2 : nullptr;
            // the walk's last block forks its weight gradients as soon as their `du` exists: nothing follows that they could run beside,
            // so they start under the block's own data-gradient convs (48.05 vs 48.20 ms per step, profiles/r04/wgside/last_block_early_ab.txt)
            sd.early_block = l == 0 && r == 0;
            CHK(run_resblock_bwd(dt, C, xin, tp.dn_rb[l][r], gy, (l == 0 && r == 0) ? w.gA : nullptr, dx, pf(c, packed, rbw.g0),
                                 pf(c, packed, rbw.g1), pf(c, packed, rbw.g2), pb + bp.dn_wd0[l][r], pb + bp.dn_wd1[l][r],
                                 rb_grads(rbw, w.dtemb + c->emb_off_down[down_bi(f, l, r)]), rw, B, H, W, s, &sd, have_stats, nu2));
            have_stats = nu2 != nullptr;
            gy = dx;
        }
        if (l > 0) {
            const int Cp = f.ch[l - 1];
            // dn_in[l] = Conv2d(D_{l-1}, k4 s2 p1)
            if (sd.on()) CHK(sd.fork(s));  // (level l's gradient buffers are not written again in this call)
            CHK(run_wgrad(dt, DOWN4, Cp, C, tp.dn_y[l - 1].back(), gy, nullptr, nullptr, XF_NONE, wpart, G(c->down_w[l]), B, H, W, sw));
            CHK(channel_sums(dt, gy, w, G(c->down_b[l]), B, H * W, C, s));
            ConvCall u = {dt, UP4, C, Cp, gy, pb + bp.down_dg[l], nullptr, nullptr, 0, nullptr, nullptr, XF_NONE, 0, w.GS[l - 1],
                          w.Ga[l - 1], nullptr, B, H, W};
            CHK(run_conv(u, s, nullptr, nullptr));
            gy = w.Ga[l - 1];
        }
    }
    CHK(flush_sums());
    // ---- input conv (models/diffusion.py:255-256): gy = d(hidden[0]) including the skip into the output conv
    HIPCHK(edge_wgrad_launch(dt, 0, gy, nullptr, x, w.partial, G(c->in_w), G(c->in_b), B, f.ch[0], f.in_channels, T, f.f_size, s));
    // ---- timestep-embedding MLP
    HIPCHK(linear_bwd_w_launch(w.dtemb, tp.temb_h2p, nullptr, G(c->tw[2]), G(c->tb[2]), B, c->E, 512, 1, s));
    HIPCHK(linear_bwd_x_launch(w.dtemb, pf(c, packed, c->tw[2]), tp.temb_h2p, w.dh2, B, c->E, 512, s));
    HIPCHK(linear_bwd_w_launch(w.dh2, tp.temb_h1p, nullptr, G(c->tw[1]), G(c->tb[1]), B, 512, 512, 1, s));
    HIPCHK(linear_bwd_x_launch(w.dh2, pf(c, packed, c->tw[1]), tp.temb_h1p, w.dh1, B, 512, 512, s));
    HIPCHK(linear_bwd_w_launch(w.dh1, pf(c, packed, c->te), t, G(c->tw[0]), G(c->tb[0]), B, 512, 128, 0, s));
    if (sd.on()) CHK(sd.join(s));
    if (n_events) HIPCHK(hipEventRecord((hipEvent_t)bucket_events[2], s));  // bucket 2: temb.* and down_modules.*
    return 0;
}

int ddimx_sqerr_loss_bwd(const float* e, const float* out, const float* g_per_sample, float* d_out, int B, long long per_sample,
                         void* stream) {
    HIPCHK(sqerr_bwd_launch(e, out, g_per_sample, d_out, B, per_sample, (hipStream_t)stream));
    return 0;
}
int ddimx_sqerr_loss_bwd_mean(const float* e, const float* out, const float* g, float* d_out, int B, long long per_sample, void* stream) {
    HIPCHK(sqerr_bwd_launch(e, out, g, d_out, B, per_sample, (hipStream_t)stream, 1));
    return 0;
}

// ---- per-op entry points ---------------------------------------------------------------------------
int ddimx_to_nhwc(int dtype, const float* nchw, void* nhwc, int B, int C, int H, int W, void* stream) {
    HIPCHK(to_nhwc_launch(dtype, nchw, nhwc, B, C, H * W, (hipStream_t)stream));
    return 0;
}
int ddimx_from_nhwc(int dtype, const void* nhwc, float* nchw, int B, int C, int H, int W, void* stream) {
    HIPCHK(from_nhwc_launch(dtype, nhwc, nchw, B, C, H * W, (hipStream_t)stream));
    return 0;
}
int ddimx_pack_conv(int dtype, const float* w, void* dst, int O, int I, int KH, int KW, void* stream) {
    HIPCHK(pack_conv_launch(dtype, w, dst, O, I, KH, KW, (hipStream_t)stream));
    return 0;
}
int ddimx_pack_convT(int dtype, const float* w, void* dst, int I, int O, void* stream) {
    HIPCHK(pack_convT_launch(dtype, w, dst, I, O, (hipStream_t)stream));
    return 0;
}

struct OpWs { void *h1, *h2; float *stats, *stats2, *scale, *shift; size_t total; };
static void carve_op(char* base, int dtype, int B, int C, int H, int W, OpWs* o) {
    Carver cv{base, 0};
    const size_t act = (size_t)B * H * W * C * esz(dtype);
    o->h1 = cv.take(act);
    o->h2 = cv.take(act);
    size_t sf = conv_stats_floats(dtype, CONV3, C, C, B, H, W);
    const size_t s2 = (size_t)B * resid_nparts(dtype, H * W, C) * (C * 2 > kGnSlab ? C * 2 : kGnSlab);
    if (s2 > sf) sf = s2;
    o->stats = (float*)cv.take(sf * 4);
    o->stats2 = (float*)cv.take(sf * 4);
    o->scale = (float*)cv.take((size_t)B * C * 4);
    o->shift = (float*)cv.take((size_t)B * C * 4);
    o->total = cv.off;
}
long long ddimx_op_workspace_bytes(int dtype, int B, int C, int H, int W) {
    OpWs o;
    carve_op(nullptr, dtype, B, C, H, W, &o);
    return (long long)o.total;
}

int ddimx_resblock_fwd(int dtype, int C, const void* x, void* y, const float* temb, int temb_stride, const float* gn0_w,
                       const float* gn0_b, const void* w0, const float* gn1_w, const float* gn1_b, const void* w1,
                       const float* bias1, const float* gn2_w, void* workspace, int B, int H, int W, void* stream) {
    if (!x || !y || !workspace) return fail("ddimx_resblock_fwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    OpWs o;
    carve_op((char*)workspace, dtype, B, C, H, W, &o);
    HIPCHK(tensor_stats_launch(dtype, x, o.stats, B, H * W, C, s, 1));
    RBPtrs p = {gn0_w, gn0_b, gn1_w, gn1_b, gn2_w, bias1, w0, w1};
    return run_resblock(dtype, C, x, y, temb, temb_stride, p, o.h1, o.h2, o.stats, o.scale, o.shift,
                        resid_nparts(dtype, H * W, C), C, false, nullptr, B, H, W, s, nullptr, o.stats2);
}
// ---- training: Residual_Block forward that keeps its tape, and its backward --------------------------
long long ddimx_rb_tape_floats(int B, int C) { return (long long)rb_tape_small_floats(B, C); }
int ddimx_pack_conv_dgrad(int dtype, const float* w, void* dst, int O, int I, void* stream) {
    HIPCHK(pack_conv_dgrad_launch(dtype, w, dst, O, I, (hipStream_t)stream));
    return 0;
}
int ddimx_resblock_fwd_train(int dtype, int C, const void* x, void* y, const float* temb, int temb_stride, const float* gn0_w,
                             const float* gn0_b, const void* w0, const float* gn1_w, const float* gn1_b, const void* w1,
                             const float* bias1, const float* gn2_w, void* u1, void* u2, float* tape_small, void* workspace,
                             int B, int H, int W, void* stream) {
    if (!x || !y || !workspace || !u1 || !u2 || !tape_small) return fail("ddimx_resblock_fwd_train: null argument");
    hipStream_t s = (hipStream_t)stream;
    OpWs o;
    carve_op((char*)workspace, dtype, B, C, H, W, &o);
    HIPCHK(tensor_stats_launch(dtype, x, o.stats, B, H * W, C, s));
    RBPtrs p = {gn0_w, gn0_b, gn1_w, gn1_b, gn2_w, bias1, w0, w1};
    RBTape tp = {u1, u2, tape_small};
    return run_resblock(dtype, C, x, y, temb, temb_stride, p, nullptr, nullptr, o.stats, o.scale, o.shift,
                        resid_nparts(dtype, H * W, C), C, false, nullptr, B, H, W, s, &tp);
}
static void carve_rb_bwd(char* base, int dtype, int B, int C, int H, int W, RBBwdWs* w, size_t* total) {
    Carver cv{base, 0};
    const size_t act = (size_t)B * H * W * C * esz(dtype);
    w->du = cv.take(act);
    w->dg = cv.take(act);
    w->stats = (float*)cv.take(rb_bwd_stats_floats(dtype, B, H * W, C) * 4);
    w->coef = (float*)cv.take((size_t)B * 3 * C * 4);
    w->dgb = (float*)cv.take((size_t)B * 2 * C * 4);
    w->sums = (float*)cv.take((size_t)B * resid_nparts(dtype, H * W, C) * C * 4);
    w->partial = (float*)cv.take(wgrad_partial_floats(dtype, CONV3, C, C, B, H, W) * 4);
    *total = cv.off;
}
long long ddimx_resblock_bwd_workspace_bytes(int dtype, int B, int C, int H, int W) {
    RBBwdWs w;
    size_t total;
    carve_rb_bwd(nullptr, dtype, B, C, H, W, &w, &total);
    return (long long)total;
}
int ddimx_resblock_bwd(int dtype, int C, const void* x, const void* u1, const void* u2, const float* tape_small,
                       const void* dy, void* dx, const float* gn0_w, const float* gn1_w, const float* gn2_w,
                       const void* w0_dgrad, const void* w1_dgrad, float* d_gn0_w, float* d_gn0_b, float* d_w0,
                       float* d_gn1_w, float* d_gn1_b, float* d_w1, float* d_bias1, float* d_gn2_w, float* d_temb,
                       int d_temb_stride, void* workspace, int B, int H, int W, void* stream) {
    if (!x || !u1 || !u2 || !tape_small || !dy || !dx || !workspace) return fail("ddimx_resblock_bwd: null argument");
    RBBwdWs w;
    size_t total;
    carve_rb_bwd((char*)workspace, dtype, B, C, H, W, &w, &total);
    RBTape tp = {const_cast<void*>(u1), const_cast<void*>(u2), const_cast<float*>(tape_small)};
    RBGrads gr = {d_gn0_w, d_gn0_b, d_gn1_w, d_gn1_b, d_gn2_w, d_w0, d_w1, d_bias1, d_temb, d_temb_stride};
    return run_resblock_bwd(dtype, C, x, tp, dy, nullptr, dx, gn0_w, gn1_w, gn2_w, w0_dgrad, w1_dgrad, gr, w, B, H, W,
                            (hipStream_t)stream);
}
int ddimx_conv3x3_fwd(int dtype, int C, const void* x, const void* w, const float* bias, const float* chan_add,
                      int chan_add_stride, const float* in_scale, const float* in_shift, int xf, int act, void* y,
                      float* stats, int B, int H, int W, void* stream) {
    ConvCall k = {dtype, CONV3, C, C, x, w, bias, chan_add, chan_add_stride, in_scale, in_shift, xf, act, nullptr, y, stats, B, H, W};
    return run_conv(k, (hipStream_t)stream, nullptr, nullptr);
}
static unsigned long long* g_debug_stamps = nullptr;
int ddimx_debug_set_stamps(unsigned long long* stamps) { g_debug_stamps = stamps; return 0; }  // diagnostic builds: next conv launches stamp here
int ddimx_pack_conv_frag(const float* w, void* dst, int O, int I, void* stream) {
    if (!w || !dst) return fail("ddimx_pack_conv_frag: null argument");
    HIPCHK(pack_conv_frag_launch(w, dst, O, I, 9, (hipStream_t)stream));
    return 0;
}
int ddimx_pack_conv_frag_k(const float* w, void* dst, int O, int I, int KK, void* stream) {
    if (!w || !dst) return fail("ddimx_pack_conv_frag_k: null argument");
    HIPCHK(pack_conv_frag_launch(w, dst, O, I, KK, (hipStream_t)stream));
    return 0;
}
int ddimx_downsample_wreg_fwd(int Cin, int Cout, const void* x, const void* w_frag, const float* bias, void* y, float* stats, int B,
                              int H, int W, void* stream) {
    ConvCall d = {DT_BF16, DOWN4, Cin, Cout, x, w_frag, bias, nullptr, 0, nullptr, nullptr, XF_NONE, 0, nullptr, y, stats, B, H, W};
    d.wf = w_frag;
    ConvPlan pl;
    CHK(conv_plan(d, &pl));
    if (!pl.wreg) return fail("ddimx_downsample_wreg_fwd: %d->%d %dx%d is not eligible for the register-streamed kernel", Cin, Cout, H, W);
    return run_conv(d, (hipStream_t)stream, nullptr, nullptr);
}
int ddimx_pack_frag_from_taps(const void* taps, void* dst, int ntaps, int NOUT, int CIN, void* stream) {
    if (!taps || !dst) return fail("ddimx_pack_frag_from_taps: null argument");
    HIPCHK(pack_frag_from_taps_launch(taps, dst, ntaps, NOUT, CIN, (hipStream_t)stream));
    return 0;
}
int ddimx_upsample_add_wreg_fwd(int Cin, int Cout, const void* x, const void* w_frag, const float* bias2, const void* skip, void* y,
                                float* stats, int B, int H, int W, void* stream) {
    ConvCall u = {DT_BF16, UP4, Cin, Cout, x, w_frag, bias2, nullptr, 0, nullptr, nullptr, XF_NONE, 0, skip, y, stats, B, H, W};
    u.wf = w_frag;
    ConvPlan pl;
    CHK(conv_plan(u, &pl));
    if (!pl.wreg) return fail("ddimx_upsample_add_wreg_fwd: %d->%d %dx%d is not eligible for the register-streamed kernel", Cin, Cout, H, W);
    return run_conv(u, (hipStream_t)stream, nullptr, nullptr);
}
int ddimx_conv3x3_pipe_fwd(int C, const void* x, const void* w_frag, const float* bias, const float* chan_add, int chan_add_stride,
                           const float* in_scale, const float* in_shift, int xf, void* y, float* group_stats, int B, int H, int W,
                           void* stream) {
    ConvCall k = {DT_BF16, CONV3, C, C, x, nullptr, bias, chan_add, chan_add_stride, in_scale, in_shift, xf, 1, nullptr, y, group_stats, B, H, W};
    k.wf = w_frag;
    k.groups = true;
    k.kernel_pref = 2;
    k.stamps = g_debug_stamps;
    return run_conv(k, (hipStream_t)stream, nullptr, nullptr);
}
long long ddimx_conv3x3_pipe_stats_floats(int C, int B, int H, int W) {
    ConvCall k = {DT_BF16, CONV3, C, C, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, XF_AFFINE, 1, nullptr, nullptr, nullptr, B, H, W};
    k.wf = &k;  // (any non-null value: only the plan is asked for)
    k.groups = true;
    k.kernel_pref = 2;
    ConvPlan pl;
    if (conv_plan(k, &pl)) return -1;
    return (long long)B * pl.wgs_per_sample * kGnSlab;
}
int ddimx_conv3x3_wreg_fwd(int C, const void* x, const void* w, const void* w_frag, const float* bias, const float* chan_add,
                           int chan_add_stride, const float* in_scale, const float* in_shift, int xf, int act, void* y, float* stats,
                           int B, int H, int W, void* stream) {
    ConvCall k = {DT_BF16, CONV3, C, C, x, w, bias, chan_add, chan_add_stride, in_scale, in_shift, xf, act, nullptr, y, stats, B, H, W};
    k.wf = w_frag;
    k.kernel_pref = 1;
    k.stamps = g_debug_stamps;
    ConvPlan pl;
    CHK(conv_plan(k, &pl));
    if (!pl.wreg) return fail("ddimx_conv3x3_wreg_fwd: C=%d %dx%d xf=%d is not eligible for the register-streamed kernel", C, H, W, xf);
    return run_conv(k, (hipStream_t)stream, nullptr, nullptr);
}
int ddimx_debug_conv3x3_stamps(int dtype, int C, const void* x, const void* w, const float* chan_add, const float* in_scale,
                                const float* in_shift, void* y, float* stats, unsigned long long* stamps, int B, int H, int W,
                                void* stream) {
    // DDIMX_STAMP_XF=1 (diagnostic runs): stamp the block's second conv (affine input) instead of its first
    static const int xf = getenv("DDIMX_STAMP_XF") ? atoi(getenv("DDIMX_STAMP_XF")) : XF_AFFINE_SILU;
    ConvCall k = {dtype, CONV3, C, C, x, w, nullptr, chan_add, C, in_scale, in_shift, xf, 1, nullptr, y, stats, B, H, W};
    k.stamps = stamps;
    return run_conv(k, (hipStream_t)stream, nullptr, nullptr);
}
long long ddimx_conv3x3_stats_floats(int dtype, int C, int B, int H, int W) {
    return (long long)conv_stats_floats(dtype, CONV3, C, C, B, H, W);
}
int ddimx_resid_gn_fwd(int dtype, int C, const void* x, const void* h, const float* scale, const float* shift, void* y,
                       float* stats, int B, int H, int W, void* stream) {
    HIPCHK(resid_launch(dtype, x, h, 0, scale, shift, y, stats, B, H * W, C, (hipStream_t)stream));
    return 0;
}
int ddimx_downsample_fwd(int dtype, int Cin, int Cout, const void* x, const void* w, const float* bias, void* y, int B,
                         int H, int W, void* stream) {
    ConvCall d = {dtype, DOWN4, Cin, Cout, x, w, bias, nullptr, 0, nullptr, nullptr, XF_NONE, 0, nullptr, y, nullptr, B, H, W};
    return run_conv(d, (hipStream_t)stream, nullptr, nullptr);
}
int ddimx_upsample_add_fwd(int dtype, int Cin, int Cout, const void* x, const void* w, const float* bias2,
                           const void* skip, void* y, int B, int H, int W, void* stream) {
    ConvCall u = {dtype, UP4, Cin, Cout, x, w, bias2, nullptr, 0, nullptr, nullptr, XF_NONE, 0, skip, y, nullptr, B, H, W};
    return run_conv(u, (hipStream_t)stream, nullptr, nullptr);
}
// ---- edge convolutions and the FNet bottleneck as single ops (the whole-network call runs exactly these) ----------
long long ddimx_conv_in_stats_floats(int B, int C0, int H, int W) { return (long long)B * conv_in_nparts(H, W) * C0 * 2; }
int ddimx_conv_in_fwd(int dtype, const float* x, const float* w, const float* bias, void* y, float* stats, int B, int Cin, int C0,
                      int H, int W, void* stream) {
    if (!x || !w || !bias || !y || !stats) return fail("ddimx_conv_in_fwd: null argument");
    HIPCHK(conv_in_launch(dtype, x, w, bias, y, stats, B, Cin, C0, H, W, (hipStream_t)stream));
    return 0;
}
int ddimx_conv_out_fwd(int dtype, const void* a, const void* b, const float* w_packed, const float* bias, float* eps, int B, int C0,
                       int Cout, int H, int W, void* stream) {
    if (!a || !b || !w_packed || !bias || !eps) return fail("ddimx_conv_out_fwd: null argument");
    HIPCHK(conv_out_launch(dtype, a, b, w_packed, bias, eps, B, C0, Cout, H, W, (hipStream_t)stream));
    return 0;
}
int ddimx_fnet_fwd(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace, long long workspace_bytes,
                   const void* x, float* out, int B, int T, void* stream) {
    if (!h || !packed || !tables || !workspace || !x || !out) return fail("ddimx_fnet_fwd: null argument");
    const ddimx_ctx* c = h;
    const int L = c->L;
    if (B < 1 || T < (1 << (L - 1)) || T % (1 << (L - 1))) return fail("ddimx_fnet_fwd: bad shape B=%d T=%d", B, T);
    Ws w;
    carve(c, (char*)workspace, B, T, &w);
    if ((long long)w.total > workspace_bytes) return fail("workspace too small: need %zu bytes, got %lld", w.total, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int S = T >> (L - 1);
    CHK(run_fnet(c, packed, tables, w, x, B, S, s));
    HIPCHK(hipMemcpyAsync(out, w.O, (size_t)B * S * c->width * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ---- backward twins of the per-op forwards (the whole-network backward runs exactly these launches) ---------------------
// Transformer_Module alone, training mode + its backward (the `_bwd` twin of ddimx_fnet_fwd).  Both use the whole-network
// scratch / tape layouts (ddimx_train_workspace_bytes, ddimx_train_tape_bytes for the same B, T) and run exactly the launches
// ddimx_unet_fwd_train / ddimx_unet_bwd issue for the bottleneck.
int ddimx_fnet_fwd_train(ddimx_handle h, const void* packed, const ddimx_tables* tables, void* workspace, long long workspace_bytes,
                         void* tape, long long tape_bytes, const void* x, float* out, int B, int T, float dropout_p,
                         unsigned long long seed, void* stream) {
    if (!h || !packed || !tables || !workspace || !tape || !x || !out) return fail("ddimx_fnet_fwd_train: null argument");
    const ddimx_ctx* c = h;
    const int L = c->L;
    if (B < 1 || T < (1 << (L - 1)) || T % (1 << (L - 1))) return fail("ddimx_fnet_fwd_train: bad shape B=%d T=%d", B, T);
    if (dropout_p < 0.f || dropout_p >= 1.f) return fail("dropout probability %g out of [0, 1)", (double)dropout_p);
    TrainWs w;
    carve_train_ws(c, (char*)workspace, B, T, &w);
    if ((long long)w.total > workspace_bytes) return fail("training workspace too small: need %zu bytes, got %lld", w.total, workspace_bytes);
    TrainTape tp;
    carve_tape(c, (char*)tape, B, T, &tp);
    if ((long long)tp.total > tape_bytes) return fail("tape too small: need %zu bytes, got %lld", tp.total, tape_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int S = T >> (L - 1);
    CHK(fnet_fwd_train_part(c, packed, tables, w, tp, x, B, S, dropout_p, seed, s));
    HIPCHK(hipMemcpyAsync(out, w.O, (size_t)B * S * c->width * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}
int ddimx_fnet_bwd(ddimx_handle h, const void* packed, const void* packed_bwd, const ddimx_tables* tables, void* workspace,
                   long long workspace_bytes, const void* tape, long long tape_bytes, const void* x, const float* d_out, float* d_x,
                   float* grads, int B, int T, float dropout_p, unsigned long long seed, void* stream) {
    if (!h || !packed || !packed_bwd || !tables || !workspace || !tape || !x || !d_out || !d_x || !grads)
        return fail("ddimx_fnet_bwd: null argument");
    const ddimx_ctx* c = h;
    const int L = c->L;
    if (B < 1 || T < (1 << (L - 1)) || T % (1 << (L - 1))) return fail("ddimx_fnet_bwd: bad shape B=%d T=%d", B, T);
    if (dropout_p < 0.f || dropout_p >= 1.f) return fail("dropout probability %g out of [0, 1)", (double)dropout_p);
    TrainWs w;
    carve_train_ws(c, (char*)workspace, B, T, &w);
    if ((long long)w.total > workspace_bytes) return fail("training workspace too small: need %zu bytes, got %lld", w.total, workspace_bytes);
    TrainTape tp;
    carve_tape(c, (char*)const_cast<void*>(tape), B, T, &tp);
    if ((long long)tp.total > tape_bytes) return fail("tape too small: need %zu bytes, got %lld", tp.total, tape_bytes);
    BwdPack bp;
    plan_bwd_pack(c, &bp);
    std::vector<long long> goff(c->specs.size());
    {
        long long n = 0;
        for (size_t i = 0; i < c->specs.size(); ++i) { goff[i] = n; n += (c->specs[i].numel + 63) & ~63ll; }
    }
    hipStream_t s = (hipStream_t)stream;
    const int S = T >> (L - 1);
    const size_t bytes = (size_t)B * S * c->width * 4;
    HIPCHK(hipMemcpyAsync(w.dO, d_out, bytes, hipMemcpyDeviceToDevice, s));
    CHK(fnet_bwd_part(c, packed, (const char*)packed_bwd, bp, tables, w, tp, x, grads, goff, B, S, dropout_p, seed, s));
    HIPCHK(hipMemcpyAsync(d_x, w.dTok, bytes, hipMemcpyDeviceToDevice, s));
    return 0;
}

struct DuBwdWs { float *partial, *stats, *dgb; size_t total; };
static void carve_du_bwd(char* base, int dtype, int Cs, int Cb, int B, int Hs, int Ws, DuBwdWs* o) {
    // Cs/Hs/Ws: the SMALL (low-resolution) side, Cb the big side's channels; the bias gradient sums run over whichever side
    // carries the bias (Downsample: small side, Upsample: big side), so size for the larger of the two
    Carver cv{base, 0};
    o->partial = (float*)cv.take(wgrad_partial_floats(dtype, DOWN4, Cb, Cs, B, Hs, Ws) * 4);
    const size_t s_small = (size_t)B * resid_nparts(dtype, Hs * Ws, Cs) * Cs * 2;
    const size_t s_big = (size_t)B * resid_nparts(dtype, 4 * Hs * Ws, Cb) * Cb * 2;
    o->stats = (float*)cv.take((s_small > s_big ? s_small : s_big) * 4);
    o->dgb = (float*)cv.take((size_t)B * (Cs > Cb ? Cs : Cb) * 4);
    o->total = cv.off;
}
static int op_channel_sums(int dt, const void* x, const DuBwdWs& w, float* dst, int B, int HW, int C, hipStream_t s) {
    HIPCHK(tensor_stats_launch(dt, x, w.stats, B, HW, C, s));
    HIPCHK(partsum_launch(w.stats, B, resid_nparts(dt, HW, C), C, w.dgb, C, s, 2));
    HIPCHK(colsum_launch(w.dgb, B, C, C, dst, s));
    return 0;
}
long long ddimx_downup_bwd_workspace_bytes(int dtype, int Csmall, int Cbig, int B, int Hsmall, int Wsmall) {
    DuBwdWs o;
    carve_du_bwd(nullptr, dtype, Csmall, Cbig, B, Hsmall, Wsmall, &o);
    return (long long)o.total;
}
int ddimx_downsample_bwd(int dtype, int Cin, int Cout, const void* x, const void* dy, const void* w_dgrad, const void* dx_add, void* dx,
                         float* d_w, float* d_b, void* workspace, int B, int H, int W, void* stream) {
    if (!x || !dy || !w_dgrad || !dx || !d_w || !d_b || !workspace) return fail("ddimx_downsample_bwd: null argument");
    if ((H | W) & 1) return fail("ddimx_downsample_bwd: H, W must be even (got %d x %d)", H, W);
    hipStream_t s = (hipStream_t)stream;
    DuBwdWs o;
    carve_du_bwd((char*)workspace, dtype, Cout, Cin, B, H / 2, W / 2, &o);
    CHK(run_wgrad(dtype, DOWN4, Cin, Cout, x, dy, nullptr, nullptr, XF_NONE, o.partial, d_w, B, H / 2, W / 2, s));
    CHK(op_channel_sums(dtype, dy, o, d_b, B, (H / 2) * (W / 2), Cout, s));
    ConvCall u = {dtype, UP4, Cout, Cin, dy, w_dgrad, nullptr, nullptr, 0, nullptr, nullptr, XF_NONE, 0, dx_add, dx, nullptr, B, H / 2, W / 2};
    u.batch_plan = true;
    return run_conv(u, s, nullptr, nullptr);
}
int ddimx_upsample_add_bwd(int dtype, int Cin, int Cout, const void* x, const void* dy, const void* w_dgrad, void* dx, float* d_w,
                           float* d_b, void* workspace, int B, int H, int W, void* stream) {
    if (!x || !dy || !w_dgrad || !dx || !d_w || !d_b || !workspace) return fail("ddimx_upsample_add_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    DuBwdWs o;
    carve_du_bwd((char*)workspace, dtype, Cin, Cout, B, H, W, &o);
    // ConvTranspose2d weight [Cin][Cout][4][4]: its gradient is the stride-2 weight gradient with the roles of input and output
    // swapped (the big tensor dy plays the halo operand, the small tensor x the output gradient)
    CHK(run_wgrad(dtype, DOWN4, Cout, Cin, dy, x, nullptr, nullptr, XF_NONE, o.partial, d_w, B, H, W, s));
    CHK(op_channel_sums(dtype, dy, o, d_b, B, 4 * H * W, Cout, s));
    ConvCall d = {dtype, DOWN4, Cout, Cin, dy, w_dgrad, nullptr, nullptr, 0, nullptr, nullptr, XF_NONE, 0, nullptr, dx, nullptr, B, 2 * H, 2 * W};
    d.batch_plan = true;
    return run_conv(d, s, nullptr, nullptr);
}
long long ddimx_edge_bwd_workspace_floats(int dtype, int B, int C0, int Cio, int H, int W) {
    return (long long)edge_wgrad_partial_floats(dtype, B, C0, Cio, H, W);
}
int ddimx_conv_in_bwd(int dtype, const void* dy, const float* x, float* partial, float* d_w, float* d_b, int B, int Cin, int C0, int H,
                      int W, void* stream) {
    if (!dy || !x || !partial || !d_w || !d_b) return fail("ddimx_conv_in_bwd: null argument");
    HIPCHK(edge_wgrad_launch(dtype, 0, dy, nullptr, x, partial, d_w, d_b, B, C0, Cin, H, W, (hipStream_t)stream));
    return 0;
}
int ddimx_conv_out_bwd(int dtype, const float* d_eps, const void* a, const void* b, const float* w_packed, void* d_sum, float* partial,
                       float* d_w, float* d_b, int B, int C0, int Cout, int H, int W, void* stream) {
    if (!d_eps || !a || !b || !w_packed || !d_sum || !partial || !d_w || !d_b) return fail("ddimx_conv_out_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(conv_out_bwd_data_launch(dtype, d_eps, w_packed, d_sum, B, C0, Cout, H, W, s));
    HIPCHK(edge_wgrad_launch(dtype, 1, a, b, d_eps, partial, d_w, d_b, B, C0, Cout, H, W, s));
    return 0;
}
int ddimx_temb_fwd_train(const float* te, const int64_t* t, const float* w0, const float* b0, const float* w1, const float* b1,
                         const float* w2, const float* b2, float* h1_pre, float* h2_pre, float* out, int B, int pos_ch, int emb_ch, int E,
                         void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(linear_rows_launch(te, t, w0, b0, h1_pre, B, emb_ch, pos_ch, 0, s));
    HIPCHK(linear_rows_launch(h1_pre, nullptr, w1, b1, h2_pre, B, emb_ch, emb_ch, 0, s, 1));
    HIPCHK(linear_rows_launch(h2_pre, nullptr, w2, b2, out, B, E, emb_ch, 0, s, 1));
    return 0;
}
int ddimx_temb_bwd(const float* d_out, const float* te, const int64_t* t, const float* w1, const float* w2, const float* h1_pre,
                   const float* h2_pre, float* d_h2, float* d_h1, float* d_w0, float* d_b0, float* d_w1, float* d_b1, float* d_w2,
                   float* d_b2, int B, int pos_ch, int emb_ch, int E, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(linear_bwd_w_launch(d_out, h2_pre, nullptr, d_w2, d_b2, B, E, emb_ch, 1, s));
    HIPCHK(linear_bwd_x_launch(d_out, w2, h2_pre, d_h2, B, E, emb_ch, s));
    HIPCHK(linear_bwd_w_launch(d_h2, h1_pre, nullptr, d_w1, d_b1, B, emb_ch, emb_ch, 1, s));
    HIPCHK(linear_bwd_x_launch(d_h2, w1, h1_pre, d_h1, B, emb_ch, emb_ch, s));
    HIPCHK(linear_bwd_w_launch(d_h1, te, t, d_w0, d_b0, B, emb_ch, pos_ch, 0, s));
    return 0;
}

int ddimx_temb_fwd(const float* te, const int64_t* t, const float* w0, const float* b0, const float* w1, const float* b1,
                   const float* w2, const float* b2, float* h1, float* h2, float* out, int B, int pos_ch, int emb_ch, int E,
                   void* stream) {
    return run_temb(te, t, w0, b0, w1, b1, w2, b2, h1, h2, out, B, pos_ch, emb_ch, E, (hipStream_t)stream);
}

// Z[b] = Re(FFT2(X[b])) + X[b] over [B][S][hid] fp32 token matrices (the FNet mixing + residual; also its own backward).
// fused = 1: the single-launch kernel (needs ddimx_fnet_mix_supported); 0: two GEMMs through `ut` ([B][2*hid][S]) and
// `partial` (split-K scratch, 8*B*2*hid*S floats).
int ddimx_fnet_mix_supported(int S, int hid) { return fnet_mix_supported(S, hid) ? 1 : 0; }
int ddimx_fnet_mix(const float* dft_hidden, const float* dft_seq, const float* x, float* z, float* ut, float* partial, int B, int S,
                   int hid, int fused, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (fused) {
        if (!fnet_mix_supported(S, hid)) return fail("ddimx_fnet_mix: S=%d hid=%d not supported by the fused kernel", S, hid);
        HIPCHK(fnet_mix_launch(dft_hidden, dft_seq, x, z, B, S, hid, s));
        return 0;
    }
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dft_hidden; g.B = x; g.C = ut; g.partial = partial; g.M = 2 * hid; g.N = S; g.K = hid; g.lda = hid; g.ldb = hid; g.ldc = S;
    g.sB = (long long)S * hid; g.sC = (long long)2 * hid * S; g.batch = B; g.splitk = sample_splitk(2 * hid, S, hid, 0);
    HIPCHK(gemm_launch(g, s));
    memset(&g, 0, sizeof(g));
    g.A = dft_seq; g.B = ut; g.C = z; g.resid = x; g.partial = partial; g.M = S; g.N = hid; g.K = 2 * S; g.lda = 2 * S; g.ldb = 2 * S;
    g.ldc = hid; g.sB = (long long)2 * hid * S; g.sC = (long long)S * hid; g.batch = B; g.splitk = sample_splitk(S, hid, 2 * S, 0);
    HIPCHK(gemm_launch(g, s));
    return 0;
}

int ddimx_step_begin(const float* coef, const int* step, int64_t* t, int B, void* stream) {
    HIPCHK(step_begin_launch(coef, step, t, B, 6, (hipStream_t)stream));
    return 0;
}
int ddimx_step_begin_ex(const float* coef, int row_stride, const int* step, int64_t* t, int B, void* stream) {
    HIPCHK(step_begin_launch(coef, step, t, B, row_stride, (hipStream_t)stream));
    return 0;
}
int ddimx_ddim_update(float* xt, const float* et, const float* noise, float* x0, const float* coef, const int* step,
                      long long n, void* stream) {
    HIPCHK(ddim_update_launch(xt, et, noise, x0, coef, step, n, (hipStream_t)stream));
    return 0;
}
int ddimx_ddpm_update(const float* x, const float* et, const float* noise, float* x0, float* xn, const float* coef,
                      const int* step, long long n, void* stream) {
    HIPCHK(ddpm_update_launch(x, et, noise, x0, xn, coef, step, n, (hipStream_t)stream));
    return 0;
}
int ddimx_step_end(int* step, void* stream) {
    HIPCHK(step_end_launch(step, (hipStream_t)stream));
    return 0;
}
int ddimx_qsample(const float* x0, const float* e, const float* alphas, const int64_t* t, float* x, int B,
                  long long per_sample, void* stream) {
    HIPCHK(qsample_launch(x0, e, alphas, t, x, B, per_sample, (hipStream_t)stream));
    return 0;
}
int ddimx_sqerr_loss(const float* e, const float* out, float* partial, float* loss, int B, long long per_sample,
                     void* stream) {
    HIPCHK(sqerr_launch(e, out, partial, loss, B, per_sample, (hipStream_t)stream));
    return 0;
}
int ddimx_ema_block_elems(void) { return ema_block_elems(); }
int ddimx_ema_update_multi(const long long* shadow_ptrs, const long long* param_ptrs, const long long* sizes,
                           const int* blk_tensor, const long long* blk_off, int nblocks, float mu, void* stream) {
    HIPCHK(ema_multi_launch(shadow_ptrs, param_ptrs, sizes, blk_tensor, blk_off, nblocks, mu, (hipStream_t)stream));
    return 0;
}

int ddimx_grad_norm_multi(const long long* grad_ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                          int nblocks, float max_norm, float* partial, float* out, void* stream) {
    HIPCHK(grad_norm_multi_launch(grad_ptrs, sizes, blk_tensor, blk_off, nblocks, max_norm, partial, out, (hipStream_t)stream));
    return 0;
}
int ddimx_scale_multi(const long long* ptrs, const long long* sizes, const int* blk_tensor, const long long* blk_off, int nblocks,
                      const float* coef, void* stream) {
    HIPCHK(scale_multi_launch(ptrs, sizes, blk_tensor, blk_off, nblocks, coef, (hipStream_t)stream));
    return 0;
}
int ddimx_adam_multi(const long long* param_ptrs, const long long* grad_ptrs, const long long* m_ptrs, const long long* v_ptrs,
                     const long long* sizes, const int* blk_tensor, const long long* blk_off, int nblocks, const float* clip,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int step, int decoupled, void* stream) {
    if (step < 1) return fail("ddimx_adam_multi: step must be >= 1");
    AdamArgs a;
    a.p = param_ptrs; a.g = grad_ptrs; a.m = m_ptrs; a.v = v_ptrs; a.sizes = sizes; a.blk_tensor = blk_tensor; a.blk_off = blk_off;
    a.clip = clip; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.decoupled = decoupled;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.dyn = nullptr;
    HIPCHK(adam_multi_launch(a, nblocks, (hipStream_t)stream));
    return 0;
}

int ddimx_adam_multi_dyn(const long long* param_ptrs, const long long* grad_ptrs, const long long* m_ptrs, const long long* v_ptrs,
                         const long long* sizes, const int* blk_tensor, const long long* blk_off, int nblocks, const float* clip,
                         const float* dyn, float beta1, float beta2, float eps, float weight_decay, int decoupled, void* stream) {
    if (!dyn) return fail("ddimx_adam_multi_dyn: null dyn");
    AdamArgs a;
    a.p = param_ptrs; a.g = grad_ptrs; a.m = m_ptrs; a.v = v_ptrs; a.sizes = sizes; a.blk_tensor = blk_tensor; a.blk_off = blk_off;
    a.clip = clip; a.lr = 0.f; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.decoupled = decoupled;
    a.bc1 = 1.f; a.bc2s = 1.f;
    a.dyn = dyn;
    HIPCHK(adam_multi_launch(a, nblocks, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
