// Microbenchmark: cost of cross-lane shifts on gfx950 (cycles per wave-instruction, one wave per SIMD and 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(float* out, int iters) {
  float v = threadIdx.x * 0.5f + 1.0f, acc = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float s;
      if (MODE == 0) s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));       // wave_shr:1
      else if (MODE == 1) s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));  // row_shr:1
      else if (MODE == 2) s = __shfl_up(v, 1);                                                                              // ds_bpermute
      else if (MODE == 3) s = v * 1.0001f;                                                                                   // plain VALU
      else if (MODE == 4) s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));  // wave_shl:1
      else s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xf, 0xf, true));                   // row_bcast15
      acc = fminf(acc + 1.0f, s);
      v = acc + s;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = acc; ((long long*)out)[1] = t1 - t0; }
  if (acc == 123.f) out[2] = v;
}
int main() {
  float* d; hipMalloc(&d, 64);
  const char* names[] = {"wave_shr:1", "row_shr:1", "ds_bpermute", "plain VALU", "wave_shl:1", "row_bcast15"};
  for (int waves = 1; waves <= 2; ++waves) for (int mode = 0; mode < 6; ++mode) {
    int iters = 4096; dim3 grid(256), block(256 * waves);
    void (*f)(float*, int) = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4> : k<5>;
    f<<<grid, block>>>(d, iters); hipDeviceSynchronize();
    f<<<grid, block>>>(d, iters); hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%d wave(s)/SIMD %-12s: %.1f cycles per (shift + 2 dependent VALU) group\n", waves, names[mode], (double)h[1] / (iters * 16.0));
  }
  return 0;
}
