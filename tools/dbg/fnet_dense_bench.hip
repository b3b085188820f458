// Stand-alone timing harness for fnet_dense_kernel variants (no Python, no library): back-to-back launches over rotating weight
// buffers (cold L2, like inside the step) timed with HIP events, plus s_memtime phase stamps of every workgroup's first wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ddim_audio_amd/csrc -I include -o /tmp/fdb tools/dbg/fnet_dense_bench.hip && /tmp/fdb
#define DDIMX_FD_STAMP
#include "../../ddim_audio_amd/csrc/fnet_dense.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <string>

using namespace ddimx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bufs {
    static constexpr int NW = 48;
    char* W[NW];  // 4 MB each (enough for fp32 [2048][512])
    float *bias, *X, *R, *stats, *gamma, *out, *ostats;
    unsigned long long* stamps;
};

template <typename F>
static void run(const char* name, Bufs& bf, int B, dim3 grid, dim3 block, F launch) {
    const int NREP = 240;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* nul = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(fd_stamps), &nul, sizeof(nul)));
    for (int i = 0; i < 10; ++i) launch(i % Bufs::NW);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < NREP; ++i) launch(i % Bufs::NW);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // stamps of one cold launch
    const int nwg = grid.x * grid.y;
    CK(hipMemset(bf.stamps, 0, (size_t)nwg * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(fd_stamps), &bf.stamps, sizeof(bf.stamps)));
    launch(17);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nwg * 8);
    CK(hipMemcpy(st.data(), bf.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t4 = 0;
    double ph[4] = {0, 0, 0, 0};
    for (int w = 0; w < nwg; ++w) {
        t0 = std::min(t0, st[w * 8]);
        t4 = std::max(t4, st[w * 8 + 4]);
        for (int k = 0; k < 4; ++k) ph[k] += (double)(st[w * 8 + k + 1] - st[w * 8 + k]) / nwg;
    }
    unsigned long long first_end = ~0ull, last_start = 0;
    for (int w = 0; w < nwg; ++w) { first_end = std::min(first_end, st[w * 8 + 4]); last_start = std::max(last_start, st[w * 8]); }
    // s_memtime ticks at 100 MHz on gfx950: 10 ns
    printf("%-44s B=%d grid %4d x %4d thr: %6.2f us/launch | stamps (x10 ns): span %5llu  last start +%4llu first end +%4llu | issue %5.0f  loads+mfma %5.0f  drain+barrier %5.0f  epilogue %5.0f\n",
           name, B, nwg, block.x, ms * 1000.f / NREP, t4 - t0, last_start - t0, first_end - t0, ph[0], ph[1], ph[2], ph[3]);
    fflush(stdout);
}

int main() {
    Bufs bf;
    for (int i = 0; i < Bufs::NW; ++i) { CK(hipMalloc(&bf.W[i], 4 << 20)); CK(hipMemset(bf.W[i], 0x11, 4 << 20)); }
    const int Bmax = 8, S = 32;
    CK(hipMalloc(&bf.bias, 2048 * 4)); CK(hipMemset(bf.bias, 0, 2048 * 4));
    CK(hipMalloc(&bf.gamma, 2048 * 4)); CK(hipMemset(bf.gamma, 0, 2048 * 4));
    CK(hipMalloc(&bf.X, (size_t)Bmax * S * 2048 * 4)); CK(hipMemset(bf.X, 0x22, (size_t)Bmax * S * 2048 * 4));
    CK(hipMalloc(&bf.R, (size_t)Bmax * S * 2048 * 4)); CK(hipMemset(bf.R, 0, (size_t)Bmax * S * 2048 * 4));
    CK(hipMalloc(&bf.out, (size_t)Bmax * S * 2048 * 4));
    CK(hipMalloc(&bf.stats, (size_t)Bmax * S * 64 * 4));
    {   // plausible statistics (sum 0, m2 16 per part)
        std::vector<float> h((size_t)Bmax * S * 64);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (i & 1) ? 16.f : 0.f;
        CK(hipMemcpy(bf.stats, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&bf.ostats, (size_t)Bmax * S * 64 * 4));
    CK(hipMalloc(&bf.stamps, (size_t)4096 * 8 * 8));

    for (int B : {8, 4}) {
        FnetDenseArgs w;  // wide: K = 512, N = 2048, normalised fp32 tokens, gelu, bf16 out (the first FFN matrix)
        memset(&w, 0, sizeof(w));
        w.bias = bf.bias; w.X = bf.X; w.x_chunk = 1; w.xstats = bf.stats; w.xnp = 32; w.xn = 16; w.out = bf.out; w.out_chunk = 1; w.out_bf16 = 1; w.act = 1;
        w.eps = 1e-6f; w.S = S; w.K = 512; w.N = 2048;
        FnetDenseArgs d;  // deep: K = 2048, N = 512, bf16 tokens, LayerNorm residual, statistics out (the second FFN matrix)
        memset(&d, 0, sizeof(d));
        d.bias = bf.bias; d.X = bf.X; d.x_chunk = 1; d.x_bf16 = 1; d.out = bf.out; d.R = bf.R; d.rstats = bf.stats; d.rgamma = bf.gamma; d.rbeta = bf.gamma;
        d.rnp = 32; d.rn = 16; d.ostats = bf.ostats; d.eps = 1e-6f; d.S = S; d.K = 2048; d.N = 512;
#define WIDE(WF, KS, XF, GP)                                                                                                    \
        run("wide bf16 WF" #WF " KS" #KS " XF" #XF " GP" #GP, bf, B, dim3(2048 / (32 * WF), B), dim3(64 * WF * KS), [&](int i) { \
            FnetDenseArgs a = w; a.W = bf.W[i]; if (!XF) a.xstats = nullptr;                                                     \
            hipLaunchKernelGGL((fnet_dense_kernel<1, false, true, WF, KS, XF, 1, GP, false>), dim3(2048 / (32 * WF), B), dim3(64 * WF * KS), 0, 0, a); })
#define DEEP(WF, KS, GP, RES)                                                                                                   \
        run("deep bf16 WF" #WF " KS" #KS " GP" #GP " R" #RES, bf, B, dim3(512 / (32 * WF), B), dim3(64 * WF * KS), [&](int i) {  \
            FnetDenseArgs a = d; a.W = bf.W[i]; if (!RES) { a.R = nullptr; a.ostats = nullptr; }                                 \
            if (a.R && (a.rnp % (8 * WF) || a.rnp / (8 * WF) > 4)) return;                                                       \
            hipLaunchKernelGGL((fnet_dense_kernel<1, true, false, WF, KS, 0, 1, GP, RES != 0>), dim3(512 / (32 * WF), B), dim3(64 * WF * KS), 0, 0, a); })
        WIDE(2, 4, 1, 8);
        WIDE(2, 4, 0, 8);
        WIDE(1, 4, 1, 8);
        WIDE(1, 8, 1, 4);
        WIDE(2, 2, 1, 16);
        WIDE(1, 2, 1, 16);
        DEEP(1, 8, 16, 1);
        DEEP(1, 8, 16, 0);
        DEEP(1, 8, 8, 1);
        DEEP(1, 4, 16, 1);
        DEEP(1, 4, 32, 1);
        DEEP(2, 4, 16, 1);
    }
    return 0;
}
