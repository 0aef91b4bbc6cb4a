#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// each wave loads 64 rows x 64 floats at row stride `stride` floats from tile origin; grid-stride over tiles
__global__ __launch_bounds__(256,2) void k(const float* __restrict__ src, float* __restrict__ dst, long stride, int ntr, int ntc, int mode) {
  int lane = threadIdx.x & 63; int wave = threadIdx.x >> 6;
  long g = (long)blockIdx.x*4 + wave, nw = (long)gridDim.x*4; long nt = (long)ntr*ntc;
  for (long t = g; t < nt; t += nw) {
    long ti = t / ntc, tj = t % ntc; long base = ti*62*stride + tj*62;
    float v[64];
    #pragma unroll
    for (int r = 0; r < 64; ++r) v[r] = src[base + r*stride + lane];
    float s = 0;
    #pragma unroll
    for (int r = 0; r < 64; ++r) s += v[r];
    if (mode == 1) {
      #pragma unroll
      for (int r = 1; r < 63; ++r) if (lane>=1 && lane<=62) dst[base + r*stride + lane] = v[r] + s;
    } else if (s == 12345.f) dst[t] = s;
  }
}
int main(int argc, char** argv) {
  long n = atol(argv[1]); long W = n, H = n; if (argc > 2) { W = atol(argv[2]); H = n*n/W; }
  float *a, *b; hipMalloc(&a, H*W*4 + 1024); hipMalloc(&b, H*W*4 + 1024); hipMemset(a, 0, H*W*4); hipMemset(b,0,H*W*4);
  int ntr = (H-2)/62, ntc = (W-2)/62;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) for (int grid : {512, 2048, 8192}) {
    k<<<grid,256>>>(a,b,W,ntr,ntc,mode); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<grid,256>>>(a,b,W,ntr,ntc,mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double bytes = (double)ntr*ntc*64*64*4*(mode?2:1);
    printf("H=%ld W=%ld mode=%d grid=%d tiles=%d: %.3f ms  %.1f GB/s  per-visit(us at 2048 slots)=%.1f\n", H, W, mode, grid, ntr*ntc, ms, bytes/ms/1e6, ms*1e3/((double)ntr*ntc/ (grid*4.0 < 2048? grid*4.0:2048)));
  }
  return 0;
}
