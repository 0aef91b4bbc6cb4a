// Microbenchmark 2 of the fill tile-load pattern: bandwidth vs waves/SIMD, tile step (62 = real, 64 = line aligned) and
// rows per load instruction.  Usage: ldtest2 N
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int STEP, int PERM = 0, int STORE = 0>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, long stride, int ntr, int ntc) {
  int lane = threadIdx.x & 63; int wave = threadIdx.x >> 6;
  long g = (long)blockIdx.x*4 + wave, nw = (long)gridDim.x*4; long nt = (long)ntr*ntc;
  for (long t0 = g; t0 < nt; t0 += nw) {
    long t = PERM ? (t0 * 40503L) % nt : t0;   // 40503 coprime with nt for the sizes used: a permutation of the tiles
    long ti = t / ntc, tj = t % ntc; long base = ti*STEP*stride + tj*STEP;
    float v[64];
    #pragma unroll
    for (int r = 0; r < 64; ++r) v[r] = src[base + r*stride + lane];
    float sum = 0;
    #pragma unroll
    for (int r = 0; r < 64; ++r) sum += v[r];
    if (STORE) {
      #pragma unroll
      for (int r = 1; r < 63; ++r) if (lane >= 1 && lane <= 62) dst[base + r*stride + lane] = v[r] + sum;
    } else if (sum == 12345.f) dst[t] = sum;
  }
}
// 32 rows x 128 columns per wave, 8 B per lane
__global__ __launch_bounds__(256) void k2(const float* __restrict__ src, float* __restrict__ dst, long stride, int ntr, int ntc) {
  int lane = threadIdx.x & 63; int wave = threadIdx.x >> 6;
  long g = (long)blockIdx.x*4 + wave, nw = (long)gridDim.x*4; long nt = (long)ntr*ntc;
  for (long t = g; t < nt; t += nw) {
    long ti = t / ntc, tj = t % ntc; long base = ti*30*stride + tj*126;
    float2 v[32];
    #pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = *(const float2*)(src + base + r*stride + 2*lane);
    float sum = 0;
    #pragma unroll
    for (int r = 0; r < 32; ++r) sum += v[r].x + v[r].y;
    if (sum == 12345.f) dst[t] = sum;
  }
}
// 16 rows x 256 columns per wave, 16 B per lane
__global__ __launch_bounds__(256) void k4(const float* __restrict__ src, float* __restrict__ dst, long stride, int ntr, int ntc) {
  int lane = threadIdx.x & 63; int wave = threadIdx.x >> 6;
  long g = (long)blockIdx.x*4 + wave, nw = (long)gridDim.x*4; long nt = (long)ntr*ntc;
  for (long t = g; t < nt; t += nw) {
    long ti = t / ntc, tj = t % ntc; long base = ti*14*stride + tj*252;
    float4 v[16];
    #pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = *(const float4*)(src + base + r*stride + 4*lane);
    float sum = 0;
    #pragma unroll
    for (int r = 0; r < 16; ++r) sum += v[r].x + v[r].y + v[r].z + v[r].w;
    if (sum == 12345.f) dst[t] = sum;
  }
}
int main(int argc, char** argv) {
  long n = atol(argv[1]); long W = n, H = n;
  float *a, *b; (void)hipMalloc(&a, H*W*4 + 65536); (void)hipMalloc(&b, H*W*4 + 4096); (void)hipMemset(a, 0, H*W*4);
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int variant = 0; variant < 7; ++variant) for (int grid : {256, 512, 1024, 2048, 4096}) {
    int ntr, ntc; 
    auto launch = [&]() {
      if (variant == 0) { ntr = (H-2)/62; ntc = (W-2)/62; k<62><<<grid,256>>>(a,b,W,ntr,ntc); }
      if (variant == 1) { ntr = H/64; ntc = W/64; k<64><<<grid,256>>>(a,b,W,ntr,ntc); }
      if (variant == 2) { ntr = (H-2)/30; ntc = (W-2)/126; k2<<<grid,256>>>(a,b,W,ntr,ntc); }
      if (variant == 4) { ntr = (H-2)/62; ntc = (W-2)/62; k<62,1,0><<<grid,256>>>(a,b,W,ntr,ntc); }
      if (variant == 5) { ntr = (H-2)/62; ntc = (W-2)/62; k<62,0,1><<<grid,256>>>(a,b,W,ntr,ntc); }
      if (variant == 6) { ntr = (H-2)/62; ntc = (W-2)/62; k<62,1,1><<<grid,256>>>(a,b,W,ntr,ntc); }
      if (variant == 3) { ntr = (H-2)/14; ntc = (W-4)/252; k4<<<grid,256>>>(a,b,W,ntr,ntc); }
    };
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1);
    double bytes = (double)ntr*ntc*64*64*4*(variant >= 5 ? 2 : 1);
    printf("variant=%d (%s) waves/SIMD=%.1f: %.3f ms  %.1f GB/s\n", variant, variant==0?"64x64 step62":variant==1?"64x64 aligned":variant==2?"32x128 8B":variant==3?"16x256 16B":variant==4?"64x64 permuted":variant==5?"64x64 +store":"64x64 permuted+store", grid/256.0, ms, bytes/ms/1e6);
  }
  return 0;
}
