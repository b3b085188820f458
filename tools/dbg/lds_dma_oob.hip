// Micro-test: does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer resource) write ZEROS to LDS for lanes whose
// offset is out of range, and does the range check see voffset only or voffset + soffset?   hipcc --offload-arch=gfx950 -o t t.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((vector_size(16))) unsigned int u32x4_t;
__global__ void k(const uint4* src, unsigned nbytes, unsigned soff, uint4* out) {
    __shared__ __attribute__((aligned(1024))) uint4 buf[128];
    buf[threadIdx.x] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    buf[threadIdx.x + 64] = make_uint4(0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu);
    __syncthreads();
    const uint64_t p = (uint64_t)src;
    u32x4_t rs = {(unsigned)p, (unsigned)(p >> 32), nbytes, 0x00020000u};
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)buf);
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    unsigned voff = threadIdx.x * 32;   // 16-byte pieces at stride 32: lanes >= nbytes/32 are out of range
    if (threadIdx.x == 63) voff = 0x80000000u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(so) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = buf[threadIdx.x];
    out[threadIdx.x + 64] = buf[threadIdx.x + 64];
}
int main() {
    const int N = 1 << 16;
    std::vector<uint32_t> h(N);
    for (int i = 0; i < N; ++i) h[i] = 0x10000000u + i;
    uint4 *src, *out;
    hipMalloc(&src, N * 4); hipMalloc(&out, 128 * 16);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    for (int cas = 0; cas < 3; ++cas) {
        const unsigned nbytes = 1024, soff = cas == 1 ? 512 : (cas == 2 ? 1024 : 0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, nbytes, soff, out);
        std::vector<uint32_t> r(128 * 4);
        hipMemcpy(r.data(), out, 128 * 16, hipMemcpyDeviceToHost);
        printf("case %d: num_records %u soffset %u\n", cas, nbytes, soff);
        for (int l : {0, 1, 15, 16, 31, 32, 33, 47, 48, 62, 63}) printf("  lane %2d voff %5u -> %08x %08x %08x %08x\n", l, l == 63 ? 0x80000000u : l * 32, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
        printf("  second KiB untouched: %08x\n", r[64 * 4]);
    }
    return 0;
}
