// on the GPU box: what does one wavefront alone on its SIMD pay for a DEPENDENT VALU instruction?  (hipcc --offload-arch=gfx950 -O3 -o /tmp/dep
// tools/lab/depissue.hip && /tmp/dep)  One chain of dependent v_min3 / v_add_dpp against two interleaved independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ void one_chain(uint32_t *out, int n)
{
    uint32_t a = threadIdx.x, g = 3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i)
        asm volatile(REP16("v_add_u32 %0, %0, %1\n\tv_min3_u32 %0, %0, %1, %0\n\t") : "+v"(a) : "v"(g));
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[64] = (uint32_t)(t1 - t0);
}
__global__ void two_chains(uint32_t *out, int n)
{
    uint32_t a = threadIdx.x, b = threadIdx.x + 7, g = 3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i)
        asm volatile(REP16("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\tv_min3_u32 %0, %0, %2, %0\n\tv_min3_u32 %1, %1, %2, %1\n\t") : "+v"(a), "+v"(b) : "v"(g));
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + b;
    if (threadIdx.x == 0) out[64] = (uint32_t)(t1 - t0);
}
__global__ void dpp_chain(uint32_t *out, int n)
{
    uint32_t a = threadIdx.x, g = 3, t;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i)
        asm volatile(REP16("v_add_u32 %1, %0, %2\n\tv_min_u32 %1, %1, %2\n\tv_add_u32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_min_u32 %0, %0, %1\n\t")
                     : "+v"(a), "=&v"(t) : "v"(g));
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[64] = (uint32_t)(t1 - t0);
}
int main()
{
    uint32_t *d, h[65];
    hipMalloc(&d, 65 * 4);
    const int n = 2000;
    for (int k = 0; k < 3; ++k) {
        for (int rep = 0; rep < 2; ++rep) {
            if (k == 0) hipLaunchKernelGGL(one_chain, dim3(1), dim3(64), 0, 0, d, n);
            if (k == 1) hipLaunchKernelGGL(two_chains, dim3(1), dim3(64), 0, 0, d, n);
            if (k == 2) hipLaunchKernelGGL(dpp_chain, dim3(1), dim3(64), 0, 0, d, n);
            hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
        }
        const double instr = (k == 1 ? 64.0 : k == 2 ? 64.0 : 32.0) * n;
        printf("%s: %u ticks of s_memtime (100 MHz) for %.0f instructions = %.2f ns per instruction\n", k == 0 ? "one dependent chain" : k == 1 ? "two interleaved chains" : "chain with a DPP add", h[64], instr,
               h[64] * 10.0 / instr);
    }
    return 0;
}
