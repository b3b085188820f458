// How much vector work hides behind a wave's own MFMAs on gfx950?  One workgroup per CU, W waves per SIMD; each wave runs a loop of
// {one v_mfma_f32_32x32x16_bf16 (dependent chain on one accumulator), N filler instructions of one kind} and stamps s_memtime.
// Output: cycles per iteration for N = 0..NMAX, per filler kind, per W.     hipcc --offload-arch=gfx950 -O3 -o bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int KIND, int N, bool MFMA>
__global__ void __launch_bounds__(512) k(unsigned long long* out, float* sink, int iters) {
    f32x16_t acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x % 7 + j)); b[j] = (__bf16)(0.02f * (threadIdx.x % 5 + j)); }
    float f[16];
    f32x2_t p[8];
    for (int i = 0; i < 16; ++i) f[i] = 1.0f + 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 8; ++i) { p[i].x = f[2 * i]; p[i].y = f[2 * i + 1]; }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    const float c1 = 1.0001f, c2 = 0.5f;
    const f32x2_t q1 = {1.0001f, 1.0001f}, q2 = {0.5f, 0.5f};
    // the whole stream as volatile asm statements: the compiler keeps their order (plain builtins were hoisted out from between the MFMAs)
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int i = (u * N + n) % 16;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i % 8]) : "v"(q1), "v"(q2));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
                if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
                if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c1));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + f[r];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int N, bool MFMA>
double run(int waves_per_simd, unsigned long long* d_out, float* d_sink) {
    const int iters = 200, nb = 256, nt = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<KIND, N, MFMA>), dim3(nb), dim3(nt), 0, 0, d_out, d_sink, iters);
    hipLaunchKernelGGL((k<KIND, N, MFMA>), dim3(nb), dim3(nt), 0, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 8];
    hipMemcpy(h, d_out, sizeof(unsigned long long) * nb * (nt / 64), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb * (nt / 64); ++i) s += (double)h[i];
    return s / (nb * (nt / 64)) / (iters * 8.0);
}

template <int KIND, bool MFMA>
void sweep(const char* name, unsigned long long* d_out, float* d_sink) {
    for (int w = 1; w <= 2; ++w) {
        printf("%-12s %s W=%d cycles per {MFMA + N fillers}:", name, MFMA ? "mfma+" : "alone", w);
        printf(" N=0 %.1f", run<KIND, 0, MFMA>(w, d_out, d_sink));
        printf(" 2 %.1f", run<KIND, 2, MFMA>(w, d_out, d_sink));
        printf(" 4 %.1f", run<KIND, 4, MFMA>(w, d_out, d_sink));
        printf(" 6 %.1f", run<KIND, 6, MFMA>(w, d_out, d_sink));
        printf(" 8 %.1f", run<KIND, 8, MFMA>(w, d_out, d_sink));
        printf(" 12 %.1f", run<KIND, 12, MFMA>(w, d_out, d_sink));
        printf(" 16 %.1f", run<KIND, 16, MFMA>(w, d_out, d_sink));
        printf(" 24 %.1f\n", run<KIND, 24, MFMA>(w, d_out, d_sink));
    }
}

int main() {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 8);
    hipMalloc(&d_sink, sizeof(float) * 256 * 512);
    sweep<0, true>("v_fma_f32", d_out, d_sink);   sweep<0, false>("v_fma_f32", d_out, d_sink);
    sweep<1, true>("v_pk_fma_f32", d_out, d_sink); sweep<1, false>("v_pk_fma_f32", d_out, d_sink);
    sweep<2, true>("v_exp_f32", d_out, d_sink);   sweep<2, false>("v_exp_f32", d_out, d_sink);
    sweep<3, true>("v_rcp_f32", d_out, d_sink);
    sweep<4, true>("v_cvt_pk_bf16", d_out, d_sink); sweep<4, false>("v_cvt_pk_bf16", d_out, d_sink);
    return 0;
}
