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

// (d8_code / edge_code: common.hpp -- the no-flats fill's finishing pass computes the codes from the same registers)

// row_off / Hg: row band of a larger raster (local row r is global row r + row_off of Hg rows); only global border
// rows get the border codes, the band's halo rows are computed from clamped data and overwritten by the host.
// ROWS_PER_WAVE: rows a wavefront walks down (2 extra halo rows are re-read); PF: rows loaded ahead of the one being computed
template <int ROWS_PER_WAVE, bool NT, int PF>
__global__ __launch_bounds__(256) void d8_kernel(const double *__restrict__ z, uint8_t *__restrict__ out, int64_t H,
                                                int64_t W, int edges_outward, int64_t row_off, int64_t Hg, unsigned int *nodir, int64_t strip0)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t strip = strip0 + (int64_t)blockIdx.x * 4 + wave;
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
    // (only "none" / "some" matters: a plain store of the same value from every wave that saw one, no atomic on one address)
    if (nodir && edges_outward && __any(nnodir != 0u) && lane == 0) *nodir = 1u;
}


// ---- streaming kernel for whole strips of rasters with W % 4 == 0 (everything else: d8_kernel above) ---------------------------
// The drop from cell a towards its neighbour b is the exact negation of the drop from b towards a (IEEE subtraction is
// antisymmetric under round-to-nearest, and so is the product with INV_SQRT2), so a row only computes the drops of its cells
// towards R, DR, D and DL (and one extra R / DR / DL drop for the strip's neighbour columns); the drops towards U, UR, UL come
// from the row above through negated-source modifiers and the drop towards L from the cell to the left: 6 * CPL + 5 float64
// operations per lane and row instead of 12 * CPL.  The compare order U, UR, R, DR, D, DL, L, UL and the strict `>` are the
// reference's (_flow.pyx:128-170).  NB raw rows (own columns + one edge column on lanes 0 / 63) are in flight while a row is
// computed; the neighbour columns of the other lanes come from DPP wave shifts when a row is taken into use.  The row address
// stays in scalar registers (the strip's first column is wave-uniform), a lane adds a 32-bit offset.
// Template: CPL columns per lane, NB rows in flight, RPW rows per wave, WPS waves per SIMD the registers are budgeted for,
// WIDE the four-row transposed store, DIAG != 0 measurement builds of tools/lab/d8lab.hip (1: no stores, 2: no loads after the
// first rows) that never run in the library.
template <int CPL> struct RowX { double v[CPL + 2]; };             // [0] left neighbour column, [1..CPL] own, [CPL+1] right
template <int CPL> struct RowRaw { double v[CPL]; double e; };     // as loaded: own columns; e = the neighbour column of lane 0 / 63

template <int CPL>
__device__ __forceinline__ void issue_row(const double *__restrict__ z, int64_t rr, int64_t H, int64_t W, int64_t c0, int lane, int edge_off,
                                          RowRaw<CPL> &o)
{
    rr = rr >= H ? H - 1 : rr;
    const double *row = z + rr * W + c0;
    const unsigned lc = (unsigned)lane * CPL;
#pragma unroll
    for (int k = 0; k < CPL; k += 2) {
        const v2d a = *reinterpret_cast<const v2d *>(row + (lc + k));
        o.v[k] = a.x;
        o.v[k + 1] = a.y;
    }
    o.e = 0.0;
    if (lane == 0 || lane == 63) o.e = row[edge_off];
}

template <int CPL> __device__ __forceinline__ void take_row(const RowRaw<CPL> &a, int lane, RowX<CPL> &o)
{
#pragma unroll
    for (int k = 0; k < CPL; ++k) o.v[k + 1] = a.v[k];
    const double l = lane_from_left(a.v[CPL - 1]), r = lane_from_right(a.v[0]);
    o.v[0] = lane == 0 ? a.e : l;
    o.v[CPL + 1] = lane == 63 ? a.e : r;
}

// The reference's selection loop (_flow.pyx:128-170: i = 8, dzmax = 0.0; in the order U, UR, R, DR, D, DL, L, UL:
// `if dz > dzmax: dzmax = dz; i = k`) as 24 instructions: compare, select of the index, running maximum.  Written out because the
// compiler puts a canonicalising v_max in front of every fmax whose operand went round the row loop and copies the negated
// drops into registers of their own; here the drops towards U, UR, L, UL enter as the values whose NEGATION they are and are
// negated by source modifiers.  v_max_f64 returns the other operand when one is NaN, and the compare is false: a NaN drop is
// never taken, like in the reference.  (-x > 0 is written x < 0: the same for NaN and both zeros.)
__device__ __forceinline__ unsigned d8_pick(double nu, double nur, double r, double dr, double d, double dl, double nl, double nul)
{
    unsigned i;
    double m;
    asm("v_cmp_lt_f64 vcc, %[nu], 0\n\t"
        "v_max_f64 %[m], -%[nu], 0\n\t"
        "v_cndmask_b32_e64 %[i], 8, 0, vcc\n\t"
        "v_cmp_gt_f64 vcc, -%[nur], %[m]\n\t"
        "v_max_f64 %[m], -%[nur], %[m]\n\t"
        "v_cndmask_b32_e64 %[i], %[i], 1, vcc\n\t"
        "v_cmp_gt_f64 vcc, %[r], %[m]\n\t"
        "v_max_f64 %[m], %[r], %[m]\n\t"
        "v_cndmask_b32_e64 %[i], %[i], 2, vcc\n\t"
        "v_cmp_gt_f64 vcc, %[dr], %[m]\n\t"
        "v_max_f64 %[m], %[dr], %[m]\n\t"
        "v_cndmask_b32_e64 %[i], %[i], 3, vcc\n\t"
        "v_cmp_gt_f64 vcc, %[d], %[m]\n\t"
        "v_max_f64 %[m], %[d], %[m]\n\t"
        "v_cndmask_b32_e64 %[i], %[i], 4, vcc\n\t"
        "v_cmp_gt_f64 vcc, %[dl], %[m]\n\t"
        "v_max_f64 %[m], %[dl], %[m]\n\t"
        "v_cndmask_b32_e64 %[i], %[i], 5, vcc\n\t"
        "v_cmp_gt_f64 vcc, -%[nl], %[m]\n\t"
        "v_max_f64 %[m], -%[nl], %[m]\n\t"
        "v_cndmask_b32_e64 %[i], %[i], 6, vcc\n\t"
        "v_cmp_gt_f64 vcc, -%[nul], %[m]\n\t"
        "v_cndmask_b32_e64 %[i], %[i], 7, vcc"
        : [i] "=&v"(i), [m] "=&v"(m)
        : [nu] "v"(nu), [nur] "v"(nur), [r] "v"(r), [dr] "v"(dr), [d] "v"(d), [dl] "v"(dl), [nl] "v"(nl), [nul] "v"(nul)
        : "vcc");
    return i;
}

template <int CPL> struct Prev { double v[CPL], ul[CPL], ur[CPL]; };   // drops of a row towards D / DR / DL, indexed by the cell of the NEXT row they point at

// codes of row `zm` (interior columns; the caller overrides border cells) from the drops `p` of the row above and the row `zd` below;
// `n`: this row's drops for the row below
template <int CPL>
__device__ __forceinline__ unsigned d8_row(const RowX<CPL> &zm, const RowX<CPL> &zd, const Prev<CPL> &p, Prev<CPL> &n)
{
    const double K = 0.7071067811865475;   // 1 / 2**0.5, _flow.pyx:93-94
    double h[CPL + 1], ddr[CPL + 1], ddl[CPL + 1];
    // h[k] / ddr[k]: drop of column (c - 1 + k) towards R / DR;  ddl[k]: drop of column (c + k) towards DL;  k = 0 .. CPL
#pragma unroll
    for (int k = 0; k <= CPL; ++k) {
        h[k] = __dsub_rn(zm.v[k], zm.v[k + 1]);
        ddr[k] = __dmul_rn(__dsub_rn(zm.v[k], zd.v[k + 1]), K);
        ddl[k] = __dmul_rn(__dsub_rn(zm.v[k + 1], zd.v[k]), K);
    }
    unsigned packed = 0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const double dd = __dsub_rn(zm.v[j + 1], zd.v[j + 1]);
        const unsigned i = d8_pick(p.v[j], p.ur[j], h[j + 1], ddr[j + 1], dd, ddl[j], h[j], p.ul[j]);
        packed |= i << (8 * j);
        n.v[j] = dd;
        n.ul[j] = ddr[j];
        n.ur[j] = ddl[j + 1];
    }
    return packed;
}

template <int CPL, int NB, int RPW, int WPS, bool WIDE = true, int DIAG = 0>
__global__ __launch_bounds__(256, WPS) void d8s_kernel(const double *__restrict__ z, uint8_t *__restrict__ out, int64_t H, int64_t W,
                                                       int edges_outward, int64_t row_off, int64_t Hg, unsigned int *nodir, unsigned nbx,
                                                       unsigned nblocks, unsigned nstrips)
{
    static_assert(NB == 1 || NB == 2, "rows in flight");
    constexpr int SW = 64 * CPL;
    constexpr unsigned ONES = CPL == 4 ? 0x01010101u : 0x00000101u;
    const double K = 0.7071067811865475;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous range of (row block, strip group) pairs so
    // that the neighbour columns a strip reads beside itself are lines its own L2 already holds
    unsigned id = blockIdx.x;
    if ((nblocks & 7u) == 0) id = (id & 7u) * (nblocks >> 3) + (id >> 3);
    unsigned by = id / nbx;
    const unsigned bx = id - by * nbx;
    const unsigned strip = bx * 4 + wave;
    if (strip >= nstrips) return;
    const int64_t c0 = (int64_t)strip * SW;
    const int64_t r_begin = (int64_t)by * RPW;
    const int64_t r_end = r_begin + RPW < H ? r_begin + RPW : H;
    const int64_t maxr = Hg - 1, maxc = W - 1;
    // the neighbour column of lane 0 / lane 63 as an offset from c0, clamped at the raster edge (border cells ignore the value)
    int edge_off = lane == 0 ? -1 : SW;
    edge_off = c0 + edge_off < 0 ? 0 : (c0 + edge_off > maxc ? SW - 1 : edge_off);
    // cells of the first / last raster column carry a fixed code (flow.py:118-139); rows 0 / maxr are handled by a uniform branch
    unsigned ovr = 0, keep = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int64_t cc = c0 + lane * CPL + j;
        if (cc == 0 || cc == maxc) {
            keep &= ~(0xffu << (8 * j));
            ovr |= (edges_outward ? (cc == 0 ? 6u : 2u) : 8u) << (8 * j);
        }
    }

    RowX<CPL> za, zb;
    Prev<CPL> pa, pb;
    RowRaw<CPL> buf[NB];
    {
        RowRaw<CPL> r0, r1;
        RowX<CPL> zu;
        issue_row<CPL>(z, r_begin > 0 ? r_begin - 1 : 0, H, W, c0, lane, edge_off, r0);
        issue_row<CPL>(z, r_begin, H, W, c0, lane, edge_off, r1);
#pragma unroll
        for (int m = 0; m < NB; ++m) issue_row<CPL>(z, r_begin + 1 + m, H, W, c0, lane, edge_off, buf[m]);
        take_row<CPL>(r0, lane, zu);
        take_row<CPL>(r1, lane, za);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            pa.v[j] = __dsub_rn(zu.v[j + 1], za.v[j + 1]);
            pa.ul[j] = __dmul_rn(__dsub_rn(zu.v[j], za.v[j + 1]), K);
            pa.ur[j] = __dmul_rn(__dsub_rn(zu.v[j + 2], za.v[j + 1]), K);
        }
    }
    unsigned nnodir = 0;
    uint8_t *orow = out + r_begin * W + c0;
    auto finish = [&](unsigned packed, int64_t r) -> unsigned {
        packed = (packed & keep) | ovr;
        const int64_t gr = r + row_off;
        if (gr == 0 || gr == maxr) {     // (wave-uniform)
            packed = 0;
#pragma unroll
            for (int j = 0; j < CPL; ++j) packed |= (edges_outward ? edge_code(gr, c0 + lane * CPL + j, maxr, maxc) : 8u) << (8 * j);
        }
        // a NODIR byte = an interior cell without a downslope neighbour (with edges flowing outward every border cell has a code)
        const unsigned v8 = packed ^ (ONES * 8u);
        nnodir |= (v8 - ONES) & ~v8 & (ONES * 0x80u);
        return packed;
    };
    auto store_row = [&](unsigned packed) {
        if (DIAG != 1 || packed == 0xdeadbeefu) {
            if (CPL == 4) *reinterpret_cast<uint32_t *>(orow + (unsigned)lane * CPL) = packed;
            else *reinterpret_cast<uint16_t *>(orow + (unsigned)lane * CPL) = (uint16_t)packed;
        }
        orow += W;
    };
    // Four rows per turn.  The state alternates between (za, pa) and (zb, pb), so nothing is copied from one row to the next; the
    // codes of the four rows (one 32-bit word per lane and row) are transposed inside every quad of lanes (two butterfly stages
    // of DPP quad permutes), after which lane 4q + k holds the 16 bytes of row k above the quad's 16 columns and ONE 16-byte
    // store per lane writes all four rows (stores are paid per instruction: one instead of four).
    int64_t r = r_begin;
    if (CPL == 4 && WIDE) {
        const bool b0 = lane & 1, b1 = lane & 2;
        const unsigned soff = (unsigned)(lane & 3) * (unsigned)W + (unsigned)(lane >> 2) * 16u;
        for (; r + 4 <= r_end; r += 4) {
            unsigned w0, w1, w2, w3;
            take_row<CPL>(buf[0], lane, zb);
            if (DIAG != 2) issue_row<CPL>(z, r + 1 + NB, H, W, c0, lane, edge_off, buf[0]);
            w0 = finish(d8_row<CPL>(za, zb, pa, pb), r);
            take_row<CPL>(buf[NB - 1], lane, za);
            if (DIAG != 2) issue_row<CPL>(z, r + 2 + NB, H, W, c0, lane, edge_off, buf[NB - 1]);
            w1 = finish(d8_row<CPL>(zb, za, pb, pa), r + 1);
            take_row<CPL>(buf[0], lane, zb);
            if (DIAG != 2) issue_row<CPL>(z, r + 3 + NB, H, W, c0, lane, edge_off, buf[0]);
            w2 = finish(d8_row<CPL>(za, zb, pa, pb), r + 2);
            take_row<CPL>(buf[NB - 1], lane, za);
            if (DIAG != 2) issue_row<CPL>(z, r + 4 + NB, H, W, c0, lane, edge_off, buf[NB - 1]);
            w3 = finish(d8_row<CPL>(zb, za, pb, pa), r + 3);
            // 4 x 4 transpose in the quad: exchange with lane ^ 1 (quad_perm [1,0,3,2] = 0xB1), then with lane ^ 2 ([2,3,0,1] = 0x4E)
            // (the permutes are evaluated by every lane before anything is selected: a permute inside a conditional would read
            // lanes that are switched off)
            const unsigned s0 = (unsigned)__builtin_amdgcn_mov_dpp((int)w0, 0xB1, 0xf, 0xf, true);
            const unsigned s1 = (unsigned)__builtin_amdgcn_mov_dpp((int)w1, 0xB1, 0xf, 0xf, true);
            const unsigned s2 = (unsigned)__builtin_amdgcn_mov_dpp((int)w2, 0xB1, 0xf, 0xf, true);
            const unsigned s3 = (unsigned)__builtin_amdgcn_mov_dpp((int)w3, 0xB1, 0xf, 0xf, true);
            const unsigned x0 = b0 ? s1 : w0, x1 = b0 ? w1 : s0, x2 = b0 ? s3 : w2, x3 = b0 ? w3 : s2;
            const unsigned t0 = (unsigned)__builtin_amdgcn_mov_dpp((int)x0, 0x4E, 0xf, 0xf, true);
            const unsigned t1 = (unsigned)__builtin_amdgcn_mov_dpp((int)x1, 0x4E, 0xf, 0xf, true);
            const unsigned t2 = (unsigned)__builtin_amdgcn_mov_dpp((int)x2, 0x4E, 0xf, 0xf, true);
            const unsigned t3 = (unsigned)__builtin_amdgcn_mov_dpp((int)x3, 0x4E, 0xf, 0xf, true);
            struct { unsigned x, y, z, w; } o;
            o.x = b1 ? t2 : x0;
            o.y = b1 ? t3 : x1;
            o.z = b1 ? x2 : t0;
            o.w = b1 ? x3 : t1;
            // non-temporal: the codes are read again by the NEXT stage at the earliest (268 MB at 16384^2, more than the caches hold);
            // measured 3 % faster than a plain store
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u ov = {o.x, o.y, o.z, o.w};
            if (DIAG != 1 || o.x == 0xdeadbeefu) __builtin_nontemporal_store(ov, reinterpret_cast<v4u *>(orow + soff));
            orow += 4 * W;
        }
    }
    for (; r + 2 <= r_end; r += 2) {
        take_row<CPL>(buf[0], lane, zb);
        if (DIAG != 2) issue_row<CPL>(z, r + 1 + NB, H, W, c0, lane, edge_off, buf[0]);
        store_row(finish(d8_row<CPL>(za, zb, pa, pb), r));
        take_row<CPL>(buf[NB - 1], lane, za);
        if (DIAG != 2) issue_row<CPL>(z, r + 2 + NB, H, W, c0, lane, edge_off, buf[NB - 1]);
        store_row(finish(d8_row<CPL>(zb, za, pb, pa), r + 1));
    }
    if (r < r_end) {
        take_row<CPL>(buf[0], lane, zb);
        store_row(finish(d8_row<CPL>(za, zb, pa, pb), r));
    }
    // (only "none" / "some" matters: a plain store of the same value from every wave that saw one, no atomic on one address)
    if (nodir && edges_outward && __any(nnodir != 0u) && lane == 0) *nodir = 1u;
}

// whole strips through d8s_kernel, the ragged last strip (and every raster whose width is not a multiple of 4) through d8_kernel
template <int CPL, int NB, int RPW, int WPS, bool WIDE = true, int DIAG = 0>
int d8_launch(const double *d_z, uint8_t *d_out, int64_t H, int64_t W, int edges_outward, hipStream_t s, int64_t row_off, int64_t Hg,
              unsigned int *d_interior_nodir)
{
    const int64_t nfull = (W % 4 == 0) ? W / (64 * CPL) : 0;
    if (nfull > 0) {
        const unsigned nbx = (unsigned)cdiv(nfull, 4), nby = (unsigned)cdiv(H, RPW);
        const unsigned nblocks = nbx * nby;
        hipLaunchKernelGGL((d8s_kernel<CPL, NB, RPW, WPS, WIDE, DIAG>), dim3(nblocks), dim3(256), 0, s, d_z, d_out, H, W, edges_outward, row_off, Hg,
                           d_interior_nodir, nbx, nblocks, (unsigned)nfull);
        MH_HIP(hipGetLastError());
    }
    const int64_t c_rest = nfull * 64 * CPL;
    if (c_rest < W) {
        constexpr int RPWE = 32;
        const int64_t strip0 = c_rest / STRIP;      // (64 * CPL divides or equals STRIP: c_rest is a multiple of STRIP when CPL == 4 ...
        static_assert(CPL == 4 || CPL == 2, "");
        // ... and with CPL == 2 an odd number of 128-column strips leaves half a 256-column strip: d8_kernel then redoes that half)
        hipLaunchKernelGGL((d8_kernel<RPWE, false, 1>), dim3((unsigned)cdiv(cdiv(W, STRIP) - strip0, 4), (unsigned)cdiv(H, RPWE)), dim3(256), 0, s,
                           d_z, d_out, H, W, edges_outward, row_off, Hg, d_interior_nodir, strip0);
        MH_HIP(hipGetLastError());
    }
    return MHIP_OK;
}

}  // namespace

int d8_dev(const double *d_z, uint8_t *d_out, int64_t H, int64_t W, int edges_outward, hipStream_t s, int64_t row_off,
           int64_t Hg, unsigned int *d_interior_nodir)
{
    if (Hg <= 0) Hg = H;
    // measured with tools/lab/d8lab.hip (16384^2 / 4096^2, HIP events around the launch): 16 rows per wave 0.452 / 0.030 ms,
    // 32: 0.470 / 0.030, 128: 0.466 / 0.065 (too few waves at 4096^2); one or two rows in flight, 4 or 2 columns per lane, 4 to 8
    // waves per SIMD all end within 3 % of each other: the kernel runs at the pace of its loads (without its stores: 0.38 ms =
    // 6.0 TB/s, the read-only peak measured on the same box) plus ~0.07 ms the stores add; its arithmetic alone takes 0.24 ms
    return d8_launch<4, 1, 16, 4>(d_z, d_out, H, W, edges_outward, s, row_off, Hg, d_interior_nodir);
}

}  // namespace mh
