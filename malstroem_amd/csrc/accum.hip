// accum.hip -- D8 flow accumulation as an in-degree driven topological walk with atomics (gfx950).
//
// Reference: flow.accumulated_flow (_flow.pyx:256-273, python flow.py:344-364) with
// trace_accumulated_flow (_flow.pyx:225-247): accum[c] = 1 + sum(accum[n]) over in-raster neighbours n whose
// flow direction points at c (flowdir[n] == (dir(c->n)+4)%8, codes > 7 never flow, _flow.pyx:212-222).
// Cells on a flow cycle, and everything downstream of one, stay 0.  Values are integers < 2**53, so any
// summation order is bit-exact in float64.
//
// Device schedule (Kahn): one 64-bit state word per cell lives in the output buffer itself:
//     bit 63      = source flag (in-degree 0 at start)
//     bits 56..59 = number of upstream neighbours that have not delivered yet
//     bits 0..55  = running sum (starts at 1 = the cell itself)
// A walker delivers its total to the downstream cell with ONE returning 64-bit atomic add (value - 2**56);
// the walker whose add brings the pending count to zero owns the now complete total and carries on.
// A NODIR cell (code > 7) receives but does not forward (the reference leaves that step undefined).
#include "common.hpp"

namespace mh {
namespace {

constexpr uint64_t SRC = 1ull << 63;
constexpr int DEG_SHIFT = 56;
constexpr uint64_t SUM_MASK = (1ull << DEG_SHIFT) - 1;

__device__ __forceinline__ bool flows_into(unsigned code, int k_from_me)  // neighbour in direction k has `code`
{
    return code <= 7u && code == (unsigned)((k_from_me + 4) & 7);
}

__global__ __launch_bounds__(256) void accum_init_kernel(const uint8_t *__restrict__ fd, uint64_t *__restrict__ st,
                                                        int64_t H, int64_t W)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int64_t r = i / W, c = i - r * W;
    unsigned deg = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t nr = r + dir_dr(k), nc = c + dir_dc(k);
        if (nr >= 0 && nr < H && nc >= 0 && nc < W) deg += flows_into(fd[nr * W + nc], k) ? 1u : 0u;
    }
    st[r * W + c] = ((uint64_t)deg << DEG_SHIFT) | 1ull | (deg == 0 ? SRC : 0ull);
}

__global__ __launch_bounds__(256) void accum_walk_kernel(const uint8_t *__restrict__ fd, uint64_t *st, int64_t H,
                                                        int64_t W)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    int64_t r = i / W, c = i - r * W;
    if (!(st[i] & SRC)) return;  // the flag is only ever written by accum_init_kernel
    uint64_t total = 1;
    for (;;) {
        const unsigned code = fd[r * W + c];
        if (code > 7u) break;
        r += dir_dr((int)code);
        c += dir_dc((int)code);
        if (r < 0 || r >= H || c < 0 || c >= W) break;
        const uint64_t delta = total - (1ull << DEG_SHIFT);
        const uint64_t old = atomicAdd(reinterpret_cast<unsigned long long *>(&st[r * W + c]), (unsigned long long)delta);
        const uint64_t now = old + delta;
        if ((now >> DEG_SHIFT) & 0xf) break;  // somebody else still has to arrive; the last arriver continues
        total = now & SUM_MASK;
    }
}

__global__ __launch_bounds__(256) void accum_finish_kernel(uint64_t *st, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = st[i];
    const double v = ((s >> DEG_SHIFT) & 0xf) ? 0.0 : (double)(s & SUM_MASK);  // unresolved (cycle) cells stay 0
    reinterpret_cast<double *>(st)[i] = v;
}

}  // namespace

int accum_dev(const uint8_t *d_fd, double *d_out, int64_t H, int64_t W, hipStream_t s)
{
    uint64_t *st = reinterpret_cast<uint64_t *>(d_out);
    const dim3 grid((unsigned)cdiv(H * W, 256));
    hipLaunchKernelGGL(accum_init_kernel, grid, dim3(256), 0, s, d_fd, st, H, W);
    hipLaunchKernelGGL(accum_walk_kernel, grid, dim3(256), 0, s, d_fd, st, H, W);
    hipLaunchKernelGGL(accum_finish_kernel, dim3((unsigned)cdiv(H * W, 256)), dim3(256), 0, s, st, H * W);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

}  // namespace mh
