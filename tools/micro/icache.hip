// What a cold instruction stream costs: straight-line scalar code of a given size, run three times inside one launch (first pass
// cold, the others warm) by one wave; launched several times (does the instruction cache survive a kernel boundary?), with another
// kernel in between or not.   hipcc --offload-arch=gfx950 -O3 tools/micro/icache.hip -o gpurun_out/icache && gpurun_out/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REPT(N) asm volatile(".rept " #N "\n s_add_u32 %0, %0, 1\n .endr" : "+s"(x))
template <int KB> __device__ __forceinline__ void body(unsigned &x);
template <> __device__ __forceinline__ void body<4>(unsigned &x) { REPT(1024); }
template <> __device__ __forceinline__ void body<16>(unsigned &x) { REPT(4096); }
template <> __device__ __forceinline__ void body<32>(unsigned &x) { REPT(8192); }
template <> __device__ __forceinline__ void body<64>(unsigned &x) { REPT(16384); }
template <int KB> __global__ void k_code(unsigned long long *out, int slot)
{
    unsigned x = 0;
    unsigned long long t[4];
    for (int rep = 0; rep < 3; rep++) {
        t[rep] = __builtin_amdgcn_s_memrealtime();
        body<KB>(x);
    }
    t[3] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { for (int r = 0; r < 3; r++) out[slot * 4 + r] = t[r + 1] - t[r]; out[slot * 4 + 3] = x; }
}
__global__ void k_other(unsigned long long *out) { if (threadIdx.x == 1000) out[0] = 1; }
template <int KB> void run(unsigned long long *d, bool between, int blocks)
{
    std::vector<unsigned long long> h(64);
    hipMemset(d, 0, 64 * 8);
    for (int l = 0; l < 6; l++) {
        hipLaunchKernelGGL(k_code<KB>, dim3(blocks), dim3(64), 0, 0, d, l);
        if (between) hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, 0, d + 60);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
    printf("%2d KB of code, %3d workgroup(s), %s: per launch (pass 1 / 2 / 3, us):", KB, blocks, between ? "another kernel in between" : "back to back");
    for (int l = 0; l < 6; l++) printf("  %.2f/%.2f/%.2f", h[l * 4] * 0.01, h[l * 4 + 1] * 0.01, h[l * 4 + 2] * 0.01);
    printf("\n");
}
int main()
{
    unsigned long long *d;
    hipMalloc(&d, 64 * 8);
    for (int b : {1, 65}) for (int bt = 0; bt < 2; bt++) { run<4>(d, bt, b); run<16>(d, bt, b); run<32>(d, bt, b); run<64>(d, bt, b); }
    return 0;
}
