// Microbenchmark of the fill kernels' tile traffic pattern: every wavefront loads a 64x64 float window (row stride =
// raster width) and stores its 62x62 interior.  mode 0: 64 row loads of 4 B per lane; mode 1: same + stores;
// mode 2: 16 loads of 16 B per lane (4 rows x 16 lanes each) staged through LDS into the row-per-register layout;
// mode 3: mode 2 + stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256,2) void k(const float* __restrict__ src, float* __restrict__ dst, long stride, int ntr, int ntc, int mode) {
  __shared__ float lds[4][64*65];
  int lane = threadIdx.x & 63; int wave = threadIdx.x >> 6;
  long g = (long)blockIdx.x*4 + wave, nw = (long)gridDim.x*4; long nt = (long)ntr*ntc;
  for (long t = g; t < nt; t += nw) {
    long ti = t / ntc, tj = t % ntc; long base = ti*62*stride + tj*62;
    float v[64];
    if (mode < 2) {
      #pragma unroll
      for (int r = 0; r < 64; ++r) v[r] = src[base + r*stride + lane];
    } else {
      float* s = lds[wave];
      const int sub = lane >> 4, c4 = (lane & 15) * 4;
      #pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int r = q*4 + sub;
        const float* p = src + base + r*stride + c4;
        float x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];   // 8-byte aligned only: compiler emits dwordx2 pairs or x4
        s[r*65 + c4] = x0; s[r*65 + c4 + 1] = x1; s[r*65 + c4 + 2] = x2; s[r*65 + c4 + 3] = x3;
      }
      __builtin_amdgcn_wave_barrier();
      #pragma unroll
      for (int r = 0; r < 64; ++r) v[r] = s[r*65 + lane];
    }
    float sum = 0;
    #pragma unroll
    for (int r = 0; r < 64; ++r) sum += v[r];
    if (mode & 1) {
      #pragma unroll
      for (int r = 1; r < 63; ++r) if (lane>=1 && lane<=62) dst[base + r*stride + lane] = v[r] + sum;
    } else if (sum == 12345.f) dst[t] = sum;
  }
}
int main(int argc, char** argv) {
  long n = atol(argv[1]); long W = n, H = n;
  float *a, *b; hipMalloc(&a, H*W*4 + 4096); hipMalloc(&b, H*W*4 + 4096); hipMemset(a, 0, H*W*4); hipMemset(b,0,H*W*4);
  int ntr = (H-2)/62, ntc = (W-2)/62;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) for (int grid : {512}) {
    k<<<grid,256>>>(a,b,W,ntr,ntc,mode); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<grid,256>>>(a,b,W,ntr,ntc,mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double bytes = (double)ntr*ntc*64*64*4*((mode&1)?2:1);
    printf("mode=%d grid=%d tiles=%d: %.3f ms  %.1f GB/s  per-visit(us at 2048 slots)=%.1f\n", mode, grid, ntr*ntc, ms, bytes/ms/1e6, ms*1e3/((double)ntr*ntc/2048));
  }
  return 0;
}
