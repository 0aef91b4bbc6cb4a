// check.hip -- run-time proof of a plain fill (fill.fill_terrain, reference fill.py:112-171): is the surface a fixed point of
//     W = max(dtm, min(W, 8 neighbours))   on interior cells,   W = dtm on the raster border?
//
// Why that is a proof and not a plausibility test.  The reference's result F* is the GREATEST fixed point with those border
// values, so every fixed point F satisfies F <= F*.  Both fill engines here only ever hold upper bounds of F* (the iterative
// schedule lowers cells from +inf through the monotone operator; the flood's seed levels start at +inf and every relaxation
// L[s] = min(L[s], max(L[t], w)) of its worklist keeps L >= the true minimax level): F >= F*.  A surface that passes this check
// is therefore F* itself; a lost wake-up or a dropped relaxation anywhere leaves a cell that can still drop, and the check sees
// it (it evaluates the reference's own update at EVERY cell).  Called on the flood's result at the end of every plain fill
// (pflood.hip: PfRun::finish); a failure hands the surface -- still an upper bound -- to the iterative schedule (FillRun::attach +
// certify), which queues exactly the tiles that can still move.
//
// HBM-bound, 8 B per cell (dem + filled, each read once): one wavefront streams a 256-column strip downwards with a rolling
// three-row window of horizontal minima in registers, 16-byte loads, neighbour columns through DPP wave shifts (as d8.hip).
// NaN follows the fill kernels: a NaN cell never moves and never wins a minimum (reference: `a <= b ? a : b`, _fill.pyx:22).
#include "common.hpp"

namespace mh {
namespace {

constexpr int CPL = 4, SW = 64 * CPL;
constexpr int DPP_WF_SL1 = 0x130, DPP_WF_SR1 = 0x138;
__device__ __forceinline__ float lane_left(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WF_SR1, 0xf, 0xf, true)); }
__device__ __forceinline__ float lane_right(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WF_SL1, 0xf, 0xf, true)); }
__device__ __forceinline__ float nan_to_inf(float v) { return v != v ? __builtin_inff() : v; }

struct RowF {
    float f[CPL];    // own columns, NaN replaced by +inf
    float h[CPL];    // min over columns c - 1, c, c + 1 of the same row
};

// row `rr` (clamped into the raster) of the strip that starts at column c0; edge_off: the neighbour column of lane 0 / 63
__device__ __forceinline__ void load_rowf(const float *__restrict__ F, int64_t rr, int64_t H, int64_t W, int64_t c0, int lane, int edge_off, RowF &o,
                                          float (&raw)[CPL])
{
    rr = rr < 0 ? 0 : (rr >= H ? H - 1 : rr);
    const float *row = F + rr * W + c0;
    const float4 a = *reinterpret_cast<const float4 *>(row + (unsigned)lane * CPL);
    raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w;
    float e = 0.0f;
    if (lane == 0 || lane == 63) e = row[edge_off];
#pragma unroll
    for (int k = 0; k < CPL; ++k) o.f[k] = nan_to_inf(raw[k]);
    e = nan_to_inf(e);
    const float l0 = lane_left(o.f[CPL - 1]), r0 = lane_right(o.f[0]);
    const float l = lane == 0 ? e : l0, r = lane == 63 ? e : r0;
    o.h[0] = fminf(fminf(l, o.f[0]), o.f[1]);
    o.h[1] = fminf(fminf(o.f[0], o.f[1]), o.f[2]);
    o.h[2] = fminf(fminf(o.f[1], o.f[2]), o.f[3]);
    o.h[3] = fminf(fminf(o.f[2], o.f[3]), r);
}

// fixed_top / fixed_bot: local row 0 / H - 1 is a band's halo row (a neighbour's cells: used as neighbours, not checked)
template <int RPW>
__global__ __launch_bounds__(256) void fill_check_kernel(const float *__restrict__ dem, const float *__restrict__ F, int64_t H, int64_t W,
                                                        int fixed_top, int fixed_bot, unsigned nbx, unsigned nstrips, unsigned int *flag)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned by = blockIdx.x / nbx, bx = blockIdx.x - by * nbx;
    const unsigned strip = bx * 4 + wave;
    if (strip >= nstrips) return;
    const int64_t c0 = (int64_t)strip * SW;
    const int64_t r_begin = (int64_t)by * RPW, r_end = r_begin + RPW < H ? r_begin + RPW : H;
    int edge_off = lane == 0 ? -1 : SW;
    edge_off = c0 + edge_off < 0 ? 0 : (c0 + edge_off > W - 1 ? SW - 1 : edge_off);
    RowF up, mid, dn;
    float raw[CPL], rawm[CPL], rawd[CPL];
    load_rowf(F, r_begin - 1, H, W, c0, lane, edge_off, up, raw);
    load_rowf(F, r_begin, H, W, c0, lane, edge_off, mid, rawm);
    bool bad = false;
    for (int64_t r = r_begin; r < r_end; ++r) {
        load_rowf(F, r + 1, H, W, c0, lane, edge_off, dn, rawd);
        const float4 dv = *reinterpret_cast<const float4 *>(dem + r * W + c0 + (unsigned)lane * CPL);
        const float d[CPL] = {dv.x, dv.y, dv.z, dv.w};
        const bool row_border = r == 0 || r == H - 1;
        const bool row_skip = (r == 0 && fixed_top) || (r == H - 1 && fixed_bot);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int64_t c = c0 + lane * CPL + j;
            const float own = rawm[j];
            if (row_border || c == 0 || c == W - 1) {
                // border cells hold the DEM (same bits; NaN stays NaN)
                bad |= !row_skip && __float_as_uint(own) != __float_as_uint(d[j]);
            } else {
                // own > max(dem, min8)  <=>  own > dem and min9 < own  (min9 includes the cell itself); NaN cells never move
                const float m9 = fminf(fminf(up.h[j], mid.h[j]), dn.h[j]);
                bad |= (own > d[j]) & (m9 < own);
                bad |= own < d[j];            // below the terrain: never a fill
            }
        }
        up = mid;
        mid = dn;
#pragma unroll
        for (int k = 0; k < CPL; ++k) rawm[k] = rawd[k];
    }
    if (__any(bad) && lane == 0) *flag = 1u;
}

// everything the strips do not cover (W % 4 != 0, the ragged last strip, tiny rasters): one thread per cell
__global__ __launch_bounds__(256) void fill_check_generic_kernel(const float *__restrict__ dem, const float *__restrict__ F, int64_t H, int64_t W,
                                                                int64_t c_first, int fixed_top, int fixed_bot, unsigned int *flag)
{
    const int64_t c = c_first + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (c >= W) return;
    const float own = F[r * W + c], dv = dem[r * W + c];
    bool bad;
    if (r == 0 || r == H - 1 || c == 0 || c == W - 1) {
        const bool skip = (r == 0 && fixed_top) || (r == H - 1 && fixed_bot);
        bad = !skip && __float_as_uint(own) != __float_as_uint(dv);
    } else {
        float m9 = __builtin_inff();
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) m9 = fminf(m9, nan_to_inf(F[(r + dr) * W + c + dc]));
        bad = ((own > dv) & (m9 < own)) | (own < dv);
    }
    if (bad) *flag = 1u;
}

}  // namespace

// *d_flag (device word, zeroed by the caller) becomes 1 when `filled` is not the fixed point described above
int fill_check_f32_dev(const float *d_dem, const float *d_filled, int64_t H, int64_t W, int fixed_top, int fixed_bot, hipStream_t s,
                       unsigned int *d_flag)
{
    constexpr int RPW = 32;
    const int64_t nfull = (W % 4 == 0) ? W / SW : 0;
    if (nfull > 0) {
        const unsigned nbx = (unsigned)cdiv(nfull, 4), nby = (unsigned)cdiv(H, RPW);
        hipLaunchKernelGGL((fill_check_kernel<RPW>), dim3(nbx * nby), dim3(256), 0, s, d_dem, d_filled, H, W, fixed_top, fixed_bot, nbx,
                           (unsigned)nfull, d_flag);
        MH_HIP(hipGetLastError());
    }
    const int64_t c_rest = nfull * SW;
    if (c_rest < W) {
        hipLaunchKernelGGL(fill_check_generic_kernel, dim3((unsigned)cdiv(W - c_rest, 256), (unsigned)H), dim3(256), 0, s, d_dem, d_filled, H, W,
                           c_rest, fixed_top, fixed_bot, d_flag);
        MH_HIP(hipGetLastError());
    }
    return MHIP_OK;
}

}  // namespace mh
