// d8.hip -- D8 flow direction stencil, float64 surface -> uint8 AGNPS codes (gfx950).
//
// Reference: flow.terrain_flowdirection (flow.py:142-167) -> _flow.terrain_flow (_flow.pyx:98-176)
// + set_edges_flow_outward (flow.py:118-139).
//   interior: i = 8 (NODIR), dzmax = 0.0; for k in U,UR,R,DR,D,DL,L,UL: dz = z - nbr_k, diagonals MULTIPLIED by
//   INV_SQRT2 = 1/2**0.5 (_flow.pyx:93-94,140); `if dz > dzmax` (strict: the first maximum wins).
//   border cells: NODIR, or the fixed outward codes when edges_flow_outward.
//
// HBM-bound: 8 B read + 1 B written per cell.  One wavefront streams a 256-column strip downwards keeping a
// rolling 3-row window in registers (each lane owns 4 adjacent columns: two 16-byte loads per row, the
// left/right neighbour columns come from the adjacent lanes via DPP wave shifts), so every surface row is
// fetched once per strip and the 4 codes of a lane leave as one 32-bit store.
#include "common.hpp"

namespace mh {
namespace {

constexpr int CPL = 4;             // columns per lane
constexpr int STRIP = 64 * CPL;    // columns per wavefront
constexpr int DPP_WF_SL1 = 0x130, DPP_WF_SR1 = 0x138;

__device__ __forceinline__ double lane_from_left(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, DPP_WF_SR1, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, DPP_WF_SR1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_right(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, DPP_WF_SL1, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, DPP_WF_SL1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

struct Row6 {
    double v[CPL + 2];  // [0] = left neighbour column, [1..4] = own columns, [5] = right neighbour column
};

// Loads raster row `rr` (clamped into the raster; clamped values are never used for an interior cell).
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void load_row(const double *__restrict__ z, int64_t rr, int64_t H, int64_t W, int64_t c,
                                         int lane, bool fast, Row6 &o)
{
    rr = rr < 0 ? 0 : (rr >= H ? H - 1 : rr);
    const double *row = z + rr * W;
    if (fast) {
        v2d a, b;
        if (NT) {   // streamed once: do not let the surface push the flow directions out of L2
            a = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(row + c));
            b = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(row + c + 2));
        } else {
            a = *reinterpret_cast<const v2d *>(row + c);
            b = *reinterpret_cast<const v2d *>(row + c + 2);
        }
        o.v[1] = a.x; o.v[2] = a.y; o.v[3] = b.x; o.v[4] = b.y;
        double l = lane_from_left(o.v[4]), r = lane_from_right(o.v[1]);
        if (lane == 0) l = row[c > 0 ? c - 1 : 0];
        if (lane == 63) r = row[c + CPL < W ? c + CPL : W - 1];
        o.v[0] = l;
        o.v[5] = r;
    } else {
#pragma unroll
        for (int k = 0; k < CPL + 2; ++k) {
            int64_t cc = c - 1 + k;
            cc = cc < 0 ? 0 : (cc >= W ? W - 1 : cc);
            o.v[k] = row[cc];
        }
    }
}

__device__ __forceinline__ unsigned d8_code(double z, double u, double ur, double r, double dr, double d, double dl,
                                            double l, double ul)
{
    const double INV_SQRT2 = 0.7071067811865475;  // 1 / 2**0.5, _flow.pyx:93-94
    // `if dz > dzmax: dzmax = dz; i = k` in the reference's order.  The running maximum is a v_max_f64 (same value as the
    // conditional move: it only changes when dz > dzmax; NaN drops are ignored by both), the index a 32-bit select.
    unsigned i = 8;
    double dzmax = 0.0, dz;
#define MH_D8_STEP(expr, k)          \
    dz = (expr);                     \
    i = dz > dzmax ? (k) : i;        \
    dzmax = fmax(dzmax, dz);
    MH_D8_STEP(__dsub_rn(z, u), 0u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, ur), INV_SQRT2), 1u)
    MH_D8_STEP(__dsub_rn(z, r), 2u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, dr), INV_SQRT2), 3u)
    MH_D8_STEP(__dsub_rn(z, d), 4u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, dl), INV_SQRT2), 5u)
    MH_D8_STEP(__dsub_rn(z, l), 6u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, ul), INV_SQRT2), 7u)
#undef MH_D8_STEP
    return i;
}

// flow.py:130-139: rows first, then columns overwrite, corners last (same order => same result on 1-wide rasters)
__device__ __forceinline__ unsigned edge_code(int64_t r, int64_t c, int64_t maxr, int64_t maxc)
{
    unsigned code = 8;
    if (r == 0) code = 0;
    if (r == maxr) code = 4;
    if (c == 0) code = 6;
    if (c == maxc) code = 2;
    if (r == 0 && c == 0) code = 7;
    if (r == 0 && c == maxc) code = 1;
    if (r == maxr && c == 0) code = 5;
    if (r == maxr && c == maxc) code = 3;
    return code;
}

// row_off / Hg: row band of a larger raster (local row r is global row r + row_off of Hg rows); only global border
// rows get the border codes, the band's halo rows are computed from clamped data and overwritten by the host.
// ROWS_PER_WAVE: rows a wavefront walks down (2 extra halo rows are re-read); PF: rows loaded ahead of the one being computed
template <int ROWS_PER_WAVE, bool NT, int PF>
__global__ __launch_bounds__(256) void d8_kernel(const double *__restrict__ z, uint8_t *__restrict__ out, int64_t H,
                                                int64_t W, int edges_outward, int64_t row_off, int64_t Hg, unsigned int *nodir)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t strip = (int64_t)blockIdx.x * 4 + wave;
    const int64_t c0 = strip * STRIP;
    if (c0 >= W) return;
    const int64_t c = c0 + (int64_t)lane * CPL;
    const int64_t r_begin = (int64_t)blockIdx.y * ROWS_PER_WAVE;
    const int64_t r_end = r_begin + ROWS_PER_WAVE < H ? r_begin + ROWS_PER_WAVE : H;
    const int64_t maxr = Hg - 1, maxc = W - 1;
    // wave-uniform: whole strip inside the raster and rows 16-byte aligned (clamped neighbour columns at the
    // raster edge are only ever consumed by border cells, whose code does not depend on the surface)
    const bool fast = (c0 + STRIP <= W) && ((W & 1) == 0);
    const bool store32 = (c0 + STRIP <= W) && ((W & 3) == 0);

    unsigned nnodir = 0;     // != 0: an interior cell without a downslope neighbour (none on a no-flats surface: the watersheds' fast path)
    Row6 up, mid, dn, nx;
    load_row<NT>(z, r_begin - 1, H, W, c, lane, fast, up);
    load_row<NT>(z, r_begin, H, W, c, lane, fast, mid);
    if (PF == 2) load_row<NT>(z, r_begin + 1, H, W, c, lane, fast, dn);
    for (int64_t r = r_begin; r < r_end; ++r) {
        if (PF == 2) load_row<NT>(z, r + 2, H, W, c, lane, fast, nx);   // in flight while row r is computed from (up, mid, dn)
        else load_row<NT>(z, r + 1, H, W, c, lane, fast, dn);
        unsigned packed = 0;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int64_t cc = c + j;
            unsigned code;
            const int64_t gr = r + row_off;
            const bool border = (gr == 0) || (gr == maxr) || (cc == 0) || (cc >= maxc);
            if (border)
                code = edges_outward ? edge_code(gr, cc, maxr, maxc) : 8u;
            else
                code = d8_code(mid.v[j + 1], up.v[j + 1], up.v[j + 2], mid.v[j + 2], dn.v[j + 2], dn.v[j + 1], dn.v[j],
                               mid.v[j], up.v[j]);
            packed |= code << (8 * j);
        }
        // with edges flowing outward every border cell has a direction: a NODIR byte is an interior cell without a downslope
        // neighbour ("does the word hold a zero byte" on packed ^ 0x08080808: four cells for four operations)
        const unsigned v8 = packed ^ 0x08080808u;
        nnodir |= (v8 - 0x01010101u) & ~v8 & 0x80808080u;
        if (store32) {
            *reinterpret_cast<uint32_t *>(out + r * W + c) = packed;
        } else {
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (c + j < W) out[r * W + c + j] = (uint8_t)(packed >> (8 * j));
        }
        up = mid;
        mid = dn;
        if (PF == 2) dn = nx;
    }
    if (nodir && edges_outward && __any(nnodir != 0u) && lane == 0) atomicAdd(nodir, 1u);   // (only "none" / "some" matters)
}

}  // namespace

int d8_dev(const double *d_z, uint8_t *d_out, int64_t H, int64_t W, int edges_outward, hipStream_t s, int64_t row_off,
           int64_t Hg, unsigned int *d_interior_nodir)
{
    if (Hg <= 0) Hg = H;
    // measured at 16384^2 (stage time, HIP events): 64 rows per wave 0.538 ms, 128 rows 0.524 (fewer halo rows re-read);
    // non-temporal loads 0.56-0.58, a second row in flight 0.54-0.62: the kernel is not waiting for memory latency
    constexpr int RPW = 128;
    hipLaunchKernelGGL((d8_kernel<RPW, false, 1>), dim3((unsigned)cdiv(cdiv(W, STRIP), 4), (unsigned)cdiv(H, RPW)), dim3(256), 0, s, d_z, d_out, H, W,
                       edges_outward, row_off, Hg, d_interior_nodir);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

}  // namespace mh
