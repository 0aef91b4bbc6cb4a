// noflat_geo.hip -- the no-flats fill (reference fill.py:174-232, speedups/_fill.pyx:72-124) as an INTEGER geodesic
// distance transform inside the level sets of the plain fill.  gfx950 only.
//
// What the reference computes.  Its sweeps lower W from +inf until W = max(dtm, min(W, min(diag nbrs) + diag, min(edge nbrs) +
// short)) everywhere; the limit G is the greatest fixed point, i.e. the one solution of the STRICT equation
//     G[c] = max(dtm[c], min_n fl(G[n] + eps_n))  (interior),  G = dtm (raster border)                                  (*)
// ((*) has one solution: eps > 0 admits no self-supporting cycle).
//
// Structure of the solution.  Let F be the plain fill of the same DEM (fill.py:112-171).  A cell that has a neighbour with a
// lower F, or lies on the raster border, keeps G = dtm = F: a SOURCE.  Every other cell c belongs to a flat of F (a lake
// surface or a natural flat at level V = F[c]) and takes G[c] = min fl(G[n] + eps_n) over its neighbours ON THE SAME LEVEL
// (higher neighbours are out of reach by more than a float32 step, lower ones do not exist): inside a level set, G - V is the
// eps-geodesic distance to the sources of that level set, every step rounded to float64.
// The rounding is what made this look inherently sequential.  But all values of one level set lie in ONE binade of float64
// (V is a float32, the total rise stays far below a float32 step), where the ulp u is a constant and every value is a multiple
// of u: fl(x + eps) = x + rn(eps / u) * u exactly, whatever x.  So  G = V + u * D  with D the chamfer distance for the INTEGER
// weights S = rn(short / u), Dg = rn(diag / u) -- order independent, exact in uint32.  A level whose binade does not give such
// weights (V = 0, denormals, inf, a tie in the rounding, weights out of range) is IRREGULAR: its flat cells are left alone here
// (distance word D_IRR) and get the upper bound F + seed_add in the surface; so does a distance that outgrows the uint32
// headroom.  The caller settles those flats with the float64 relaxation of fill.hip (a "partial" surface).  NaN cells: not for
// this path at all.
// None of this is trusted: as the distances are turned into G, ng_finish_kernel (ng_verify_kernel for a partial surface)
// evaluates (*) at EVERY cell in the reference's own float64 arithmetic.  One mismatch anywhere -> the caller runs the float64
// relaxation from scratch.
//
// Kernels.  ng_first (every tile once, one wavefront per 62 x 62 tile + ring): classifies the window from F -- per cell a 16-bit
// word = same-level adjacency byte | binade class << 8 -- and keeps what the passes want as a block per tile + a header: for a
// window of one or two classes (all but a handful) bit masks -- may the cell move, is it a WALL (a source that a flat cell next to it
// is not adjacent to: see pass_wl), is it of the second class --, for three or more classes the words of both layouts (16 KB); writes
// the start distances (a flat cell next to a wall of its own level starts from its step weight: the seeds).  ng_round (one launch
// per round, tiles from a compacted list): a
// wavefront holds the 64 x 64 window of distances in 64 VGPRs and relaxes it by row-sequential passes (down, up; transposed
// through wave-private LDS: right, left) with DPP neighbours, the rows of a pass written out as text (NG_ROW: 8 instructions in 8
// issue slots); a tile whose edge cells moved marks the neighbouring tiles it
// shares a flat with (one byte per tile, plain stores); ng_compact turns the marks into the next round's list.  Once the rounds
// are small (the tail) they append the woken tiles to the next round's list themselves and the compaction launch goes.
// ng_finish (streaming): G = F + u * D assembled in registers, (*) checked against the eight neighbours, G written.
// ng_assemble / ng_verify: the same in two passes, for a partial surface and for the caller's check after the relaxation.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace mh {

namespace {

constexpr int WN = 64;  // window edge (tile + halo ring)
constexpr int TI = 62;  // tile edge
constexpr uint32_t DINF = 0xE0000000u;     // "not reached yet"; every weight is < 2**28, so DINF + weight does not wrap
constexpr uint32_t WMAX = 1u << 28;
constexpr uint32_t D_IRR = 0xFFFFFFFFu;    // distance word of a flat cell of an irregular level: not part of the transform (see GeoRun::end)
constexpr uint32_t M_NOFLAT = 0xFF00u;     // class 255: not a flat cell (a source)
constexpr uint32_t M_IRR = 0xFD00u;        // a flat cell of a level without integer weights (counted)
constexpr uint32_t M_WALL = 0xFE00u;       // a source that some flat cell next to it is NOT adjacent to (pass_wl)
constexpr int DPP_WF_SL1 = 0x130;          // lane i <- lane i+1
constexpr int DPP_WF_SR1 = 0x138;          // lane i <- lane i-1
#ifndef NG_MAXCYC
#define NG_MAXCYC 1      // local (down, up, right, left) cycles per visit; a visit that is cut off re-queues its own tile
#endif
enum { C_IRREGULAR = 0, C_UNREACHED = 1, C_MISMATCH = 2, C_FATAL = 3, C_STATS = 8 };   // counters[]; C_STATS: 64 x {visits, cycles}

__device__ __forceinline__ uint32_t from_left(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_WF_SR1, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t from_right(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_WF_SL1, 0xf, 0xf, true); }

// float neighbours; a lane without a source lane (lane 0 / 63) reads +inf: never lower, never equal
__device__ __forceinline__ float fleft(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0x7f800000, __float_as_int(v), DPP_WF_SR1, 0xf, 0xf, false));
}
__device__ __forceinline__ float fright(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0x7f800000, __float_as_int(v), DPP_WF_SL1, 0xf, 0xf, false));
}

// biased float64-exponent class of the values just ABOVE V (where the flat's values live), 255: no class
__device__ __forceinline__ uint32_t class_above(float V)
{
    // (selects, no branches: the classification calls this once per row in a loop that is unrolled 64 times)
    const uint32_t b = __float_as_uint(V), ex = (b >> 23) & 0xffu;
    const uint32_t e = ((b >> 31) && (b & 0x7fffffu) == 0u) ? ex - 1u : ex;   // V = -2**k: the values above it are one binade down
    return (ex == 0u || ex == 255u) ? 255u : e;                               // zero, denormal, inf, NaN
}
// ulp of class e (float32 bias 127 -> float64 bias 1023, 52 mantissa bits)
__device__ __forceinline__ double class_ulp(uint32_t e) { return __longlong_as_double((long long)(e + 1023u - 127u - 52u) << 52); }

// ---- one round of tile visits ------------------------------------------------------------------------------------------
struct GeoArgs {
    int64_t H, W;
    int ntr, ntc, nt;
    const float *F;                 // the plain fill
    uint32_t *d;
    uint32_t *blk;                  // [nt][64][64] the tiles' window words as the passes want them (written by the first round)
    uint32_t *hdr;                  // [nt] seams | uniform << 9 | class << 16 | active << 31
    uint8_t *mark;                  // one byte per tile: visit it in the next round
    const int *list;                // this round's tiles (ng_compact_kernel) ...
    const uint32_t *count;          // ... and how many
    const uint32_t *tab;            // [256] S | [256] Dg by class
    unsigned long long *counters;
    int maxcyc;                     // local cycles per visit in this round
    // light rounds (few tiles: the tail, where a launch is worth as much as the visits): no mark bytes and no compaction launch, a
    // woken tile is appended to the next round's list at once.  amark[t] = the last round t was listed for (atomicMax: one entry
    // per round whoever wakes it, and a wake in round r always gives a visit in round r + 1 -- the tile's visit of round r may
    // have run before the waker's stores)
    int append, round_next;
    int *amark;                     // [nt]
    int *list_next;
    uint32_t *count_next;
    int fixed_top, fixed_bot;       // row band: local row 0 / H - 1 is a halo row of the neighbouring band (not a raster border, not mine)
};

// 64 x 64 transpose of 32-bit words through a wave-private LDS scratch [64][65].  The DS instructions are written out: one
// base register + immediate offsets (left to itself the compiler pairs the row writes into ds_write2 with 8-bit offsets, needs a
// new base every four rows and then keeps -- and spills -- all sixteen of them for the next transpose).  `scr_b`: LDS byte
// address of the scratch.  DS operations of one wave execute in order; the explicit waits make the data visible to the VALU.
template <int R> __device__ __forceinline__ void lds_put(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(R) : "memory"); }
template <int R> __device__ __forceinline__ uint32_t lds_get(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(R) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); }
template <int R0, int N> __device__ __forceinline__ void lds_put_rows(uint32_t addr, const uint32_t (&x)[WN])
{
    if constexpr (N > 0) {
        lds_put<R0 * (WN + 1) * 4>(addr, x[R0]);
        lds_put_rows<R0 + 1, N - 1>(addr, x);
    }
}
template <int C0, int N> __device__ __forceinline__ void lds_get_cols(uint32_t addr, uint32_t (&x)[WN])
{
    if constexpr (N > 0) {
        x[C0] = lds_get<C0 * 4>(addr);
        lds_get_cols<C0 + 1, N - 1>(addr, x);
    }
}
__device__ __forceinline__ void transpose32(uint32_t (&x)[WN], uint32_t scr_b, int lane)
{
    lds_put_rows<0, WN>(scr_b + 4u * lane, x);
    lds_wait();
    lds_get_cols<0, WN>(scr_b + 4u * (WN + 1) * lane, x);
    lds_wait();
#pragma unroll
    for (int c = 0; c < WN; ++c) asm volatile("" : "+v"(x[c]));   // uses stay behind the wait
}

struct LaneW {      // classes differ inside the window (it straddles a power of two): weights by the cell's own class
    __device__ __forceinline__ void get(uint32_t mword, const uint32_t *tab_l, uint32_t &s, uint32_t &g) const
    {
        const uint32_t e = (mword >> 8) & 0xffu;
        s = tab_l[e];
        g = tab_l[256 + e];
    }
};

// One row-sequential pass (DOWN: rows 1..62, else 62..1): row r is relaxed from the three cells of the row behind it.
// ni[r]: low byte = INVERTED adjacency of the current layout, bits 8..15 = class.  A cell that must not move (source, halo
// ring, outside the raster) has no adjacency: all its candidates become 0xffffffff.
template <bool DOWN, typename WT>
__device__ __forceinline__ void pass(uint32_t (&d)[WN], uint32_t (&ni)[WN], const WT wt, const uint32_t *tab_l, uint32_t &acc_all,
                                     uint32_t &acc_first, uint32_t &acc_last)
{
    constexpr int first = DOWN ? 1 : TI, dir = DOWN ? 1 : -1;
    constexpr int bC = DOWN ? 0 : 4, bL = DOWN ? 7 : 5, bR = DOWN ? 1 : 3;   // behind-centre, behind-left (lane - 1), behind-right
    // new names for the rows in every pass: what one pass derives from a row (table addresses, mask bits) must not be kept
    // alive for the next one -- 62 extra live registers, i.e. spills inside the row loop
#pragma unroll
    for (int r = 0; r < WN; ++r) asm volatile("" : "+v"(d[r]), "+v"(ni[r]));
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int r = first + dir * i;
        const uint32_t behind = d[r - dir];
        const uint32_t bl = from_left(behind), br = from_right(behind);
        const uint32_t w = ni[r];
        uint32_t S, G;
        wt.get(w, tab_l, S, G);
        const uint32_t c0 = (behind + S) | (uint32_t)__builtin_amdgcn_sbfe((int)w, bC, 1);
        const uint32_t c1 = (bl + G) | (uint32_t)__builtin_amdgcn_sbfe((int)w, bL, 1);
        const uint32_t c2 = (br + G) | (uint32_t)__builtin_amdgcn_sbfe((int)w, bR, 1);
        const uint32_t cu = d[r];
        const uint32_t nv = min(min(c0, c1), min(c2, cu));
        const uint32_t x = nv ^ cu;
        acc_all |= x;
        asm volatile("" : "+v"(acc_all));   // accumulate row by row (a reassociated OR tree keeps all 62 differences alive)
        // row 1 / 62 matter to the neighbouring tile only through cells that are adjacent to something across the seam
        if (r == 1) acc_first |= (w & 0x83u) != 0x83u ? x : 0u;
        if (r == TI) acc_last |= (w & 0x38u) != 0x38u ? x : 0u;
        d[r] = nv;
        __builtin_amdgcn_sched_barrier(0);   // rows in program order: hoisted shifts / mask extractions blow the VGPR budget
    }
}

template <typename WT>
__device__ __forceinline__ void relax(uint32_t (&d)[WN], uint32_t (&ni)[WN], const WT wt, const uint32_t *tab_l, uint32_t scr_b, int lane, int maxcyc,
                                      unsigned &wake, bool &changed, bool &capped, unsigned &cycles)
{
    // wake bits = neighbour k of the 3 x 3 block around the tile (k = 3 * (di + 1) + dj + 1)
    const uint64_t INNER = ((1ull << TI) - 1) << 1, B1 = 2ull, B62 = 1ull << TI;
    capped = true;
#pragma nounroll
    for (int cyc = 0; cyc < maxcyc; ++cyc) {
        uint64_t chg = 0;
        ++cycles;
#pragma nounroll
        for (int half = 0; half < 2; ++half) {
            uint32_t acc_all = 0, acc_first = 0, acc_last = 0;
            pass<true>(d, ni, wt, tab_l, acc_all, acc_first, acc_last);
            pass<false>(d, ni, wt, tab_l, acc_all, acc_first, acc_last);
            const uint64_t all = __ballot(acc_all != 0), fst = __ballot(acc_first != 0), lst = __ballot(acc_last != 0);
            chg |= all;
            // half 0: lane = column, first / last = row 1 / 62.  half 1: lane = row, first / last = column 1 / 62.
            const unsigned e_first = (fst & INNER) ? 1u : 0u, e_last = (lst & INNER) ? 1u : 0u;
            const unsigned l1 = (all & B1) ? 1u : 0u, l62 = (all & B62) ? 1u : 0u;
            const unsigned f1 = (fst & B1) ? 1u : 0u, f62 = (fst & B62) ? 1u : 0u, g1 = (lst & B1) ? 1u : 0u, g62 = (lst & B62) ? 1u : 0u;
            if (half == 0)   // top | bottom | left | right | TL TR BL BR
                wake |= (e_first << 1) | (e_last << 7) | (l1 << 3) | (l62 << 5) | (f1 << 0) | (f62 << 2) | (g1 << 6) | (g62 << 8);
            else
                wake |= (e_first << 3) | (e_last << 5) | (l1 << 1) | (l62 << 7) | (f1 << 0) | (f62 << 6) | (g1 << 2) | (g62 << 8);
            transpose32(d, scr_b, lane);
#pragma unroll
            for (int r = 0; r < WN; ++r) ni[r] = __builtin_amdgcn_alignbit(ni[r], ni[r], 16);   // the other layout's half word
        }
        changed |= chg != 0;
        if (!chg) {
            capped = false;
            break;
        }
    }
}

// ---- a window of ONE class (six of seven tiles): no adjacency bits at all.  A flat cell has no lower neighbour, so a neighbour it is
// NOT adjacent to (another level) is higher, has the flat cell as a lower neighbour and is therefore a source -- distance 0, fixed.
// Such a source reads as "not reached" in the window's registers (a WALL: its candidates never win), and the flat cells of its own
// level that it would have fed start from their step weight instead (the seeds, ng_first).  With the walls in place every candidate
// of a flat cell that can win comes from a cell it is adjacent to: the passes need one bit per cell -- may it move (a regular flat
// cell of this tile's interior) -- instead of three adjacency bits per cell and pass.  Block of such a window, nine words per lane:
// mk[0..1] immovable / mk[2..3] wall bits by register row of the row layout, mk[4..7] the same of the transposed layout,
// mk[8] = bit 0 / 1: row 1 / 62 of this lane is adjacent to something across the seam (row layout), bits 2 / 3: transposed layout.
constexpr int NMK = 9;
// One row of a pass as text: 8 instructions in 8 issue slots.  A DPP operand must not have been written by one of the two instructions in
// front of the DPP instruction (the compiler pads with s_nop; left to itself it spent 13 slots on the row -- and turned the bit test into
// and + compare + select with two more wait states): `b`, the row behind, is written by the v_bfi of the row before, the v_bitop3 and
// the v_add stand between.  O: the row's new value (not one of the inputs), I: its old one, BIT: its bit in the mask word.
#define NG_DPP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define NG_ROW(O, I, B, BIT)                                                                                                                 \
    "v_add_u32 %[u], %[" B "], %[S]\n\t"                                                                                                    \
    "v_add_u32_dpp %[t], %[" B "], %[G] wave_shr:1" NG_DPP                                                                                   \
    "v_add_u32_dpp %[k], %[" B "], %[G] wave_shl:1" NG_DPP                                                                                   \
    "v_min3_u32 %[t], %[u], %[t], %[k]\n\t"                                                                                                 \
    "v_bfe_i32 %[k], %[m], " BIT ", 1\n\t"                                                                                                  \
    "v_min_u32 %[" O "], %[t], %[" I "]\n\t"                                                                                                \
    "v_bfi_b32 %[" O "], %[k], %[" I "], %[" O "]\n\t"                                                                                      \
    "v_bitop3_b32 %[acc], %[" O "], %[acc], %[" I "] bitop3:0xde\n\t"
// rows R, R + dir, .. R + 5 dir from the row behind R (six rows a block: the compiler puts an s_nop in front of every block)
template <int R, int DIR>
__device__ __forceinline__ void rows6_wl(uint32_t (&d)[WN], const uint32_t m, const uint32_t S, const uint32_t G, uint32_t &acc)
{
    static_assert((R >> 5) == ((R + 5 * DIR) >> 5), "one mask word a block");
    uint32_t o0, o1, o2, o3, o4, o5, u, t, k;
    if constexpr (DIR > 0)
        asm volatile(NG_ROW("o0", "i0", "b", "%[bit]") NG_ROW("o1", "i1", "o0", "%[bit]+1") NG_ROW("o2", "i2", "o1", "%[bit]+2")
                     NG_ROW("o3", "i3", "o2", "%[bit]+3") NG_ROW("o4", "i4", "o3", "%[bit]+4") NG_ROW("o5", "i5", "o4", "%[bit]+5")
                     : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [u] "=&v"(u), [t] "=&v"(t),
                       [k] "=&v"(k), [acc] "+v"(acc)
                     : [b] "v"(d[R - DIR]), [i0] "v"(d[R]), [i1] "v"(d[R + DIR]), [i2] "v"(d[R + 2 * DIR]), [i3] "v"(d[R + 3 * DIR]),
                       [i4] "v"(d[R + 4 * DIR]), [i5] "v"(d[R + 5 * DIR]), [S] "v"(S), [G] "v"(G), [m] "v"(m), [bit] "n"(R & 31));
    else
        asm volatile(NG_ROW("o0", "i0", "b", "%[bit]") NG_ROW("o1", "i1", "o0", "%[bit]-1") NG_ROW("o2", "i2", "o1", "%[bit]-2")
                     NG_ROW("o3", "i3", "o2", "%[bit]-3") NG_ROW("o4", "i4", "o3", "%[bit]-4") NG_ROW("o5", "i5", "o4", "%[bit]-5")
                     : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [u] "=&v"(u), [t] "=&v"(t),
                       [k] "=&v"(k), [acc] "+v"(acc)
                     : [b] "v"(d[R - DIR]), [i0] "v"(d[R]), [i1] "v"(d[R + DIR]), [i2] "v"(d[R + 2 * DIR]), [i3] "v"(d[R + 3 * DIR]),
                       [i4] "v"(d[R + 4 * DIR]), [i5] "v"(d[R + 5 * DIR]), [S] "v"(S), [G] "v"(G), [m] "v"(m), [bit] "n"(R & 31));
    d[R] = o0; d[R + DIR] = o1; d[R + 2 * DIR] = o2; d[R + 3 * DIR] = o3; d[R + 4 * DIR] = o4; d[R + 5 * DIR] = o5;
}
// one row on its own (rows 1 and 62: their differences are looked at separately); returns old ^ new
template <int R, int DIR>
__device__ __forceinline__ uint32_t row1_wl(uint32_t (&d)[WN], const uint32_t m, const uint32_t S, const uint32_t G, uint32_t &acc)
{
    uint32_t o0, u, t, k;
    const uint32_t cu = d[R];
    asm volatile("s_nop 1\n\t" NG_ROW("o0", "i0", "b", "%[bit]") "s_nop 1"
                 : [o0] "=&v"(o0), [u] "=&v"(u), [t] "=&v"(t), [k] "=&v"(k), [acc] "+v"(acc)
                 : [b] "v"(d[R - DIR]), [i0] "v"(cu), [S] "v"(S), [G] "v"(G), [m] "v"(m), [bit] "n"(R & 31));
    d[R] = o0;
    return o0 ^ cu;
}
template <int R, int DIR, int N>
__device__ __forceinline__ void blocks_wl(uint32_t (&d)[WN], const uint32_t lo, const uint32_t hi, const uint32_t S, const uint32_t G, uint32_t &acc)
{
    if constexpr (N > 0) {
        rows6_wl<R, DIR>(d, R < 32 ? lo : hi, S, G, acc);
        __builtin_amdgcn_sched_barrier(0);
        blocks_wl<R + 6 * DIR, DIR, N - 1>(d, lo, hi, S, G, acc);
    }
}
template <bool DOWN>
__device__ __forceinline__ void pass_wl(uint32_t (&d)[WN], const uint32_t imm_lo, const uint32_t imm_hi, const uint32_t S, const uint32_t G,
                                        const bool edge_first, const bool edge_last, uint32_t &acc_all, uint32_t &acc_first, uint32_t &acc_last)
{
#pragma unroll
    for (int r = 0; r < WN; ++r) asm volatile("" : "+v"(d[r]));
    uint32_t lo = imm_lo, hi = imm_hi;
    asm volatile("" : "+v"(lo), "+v"(hi));
    if constexpr (DOWN) {
        const uint32_t x1 = row1_wl<1, 1>(d, lo, S, G, acc_all);
        acc_first |= edge_first ? x1 : 0u;
        __builtin_amdgcn_sched_barrier(0);
        blocks_wl<2, 1, 10>(d, lo, hi, S, G, acc_all);              // rows 2 .. 61
        const uint32_t x2 = row1_wl<TI, 1>(d, hi, S, G, acc_all);
        acc_last |= edge_last ? x2 : 0u;
    } else {
        const uint32_t x2 = row1_wl<TI, -1>(d, hi, S, G, acc_all);
        acc_last |= edge_last ? x2 : 0u;
        __builtin_amdgcn_sched_barrier(0);
        blocks_wl<TI - 1, -1, 10>(d, lo, hi, S, G, acc_all);         // rows 61 .. 2
        const uint32_t x1 = row1_wl<1, -1>(d, lo, S, G, acc_all);
        acc_first |= edge_first ? x1 : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void relax_wl(uint32_t (&d)[WN], const uint32_t (&mk)[13], const uint32_t S, const uint32_t G, uint32_t scr_b, int lane, int maxcyc,
                                         unsigned &wake, bool &changed, bool &capped, unsigned &cycles)
{
    const uint64_t INNER = ((1ull << TI) - 1) << 1, B1 = 2ull, B62 = 1ull << TI;
    capped = true;
#pragma nounroll
    for (int cyc = 0; cyc < maxcyc; ++cyc) {
        uint64_t chg = 0;
        ++cycles;
        {   // lane = column, first / last = row 1 / 62
            uint32_t acc_all = 0, acc_first = 0, acc_last = 0;
            pass_wl<true>(d, mk[0], mk[1], S, G, (mk[8] & 1u) != 0u, (mk[8] & 2u) != 0u, acc_all, acc_first, acc_last);
            pass_wl<false>(d, mk[0], mk[1], S, G, (mk[8] & 1u) != 0u, (mk[8] & 2u) != 0u, acc_all, acc_first, acc_last);
            const uint64_t all = __ballot(acc_all != 0), fst = __ballot(acc_first != 0), lst = __ballot(acc_last != 0);
            chg |= all;
            const unsigned e_first = (fst & INNER) ? 1u : 0u, e_last = (lst & INNER) ? 1u : 0u;
            const unsigned l1 = (all & B1) ? 1u : 0u, l62 = (all & B62) ? 1u : 0u;
            const unsigned f1 = (fst & B1) ? 1u : 0u, f62 = (fst & B62) ? 1u : 0u, g1 = (lst & B1) ? 1u : 0u, g62 = (lst & B62) ? 1u : 0u;
            wake |= (e_first << 1) | (e_last << 7) | (l1 << 3) | (l62 << 5) | (f1 << 0) | (f62 << 2) | (g1 << 6) | (g62 << 8);
            transpose32(d, scr_b, lane);
        }
        {   // lane = row, first / last = column 1 / 62
            uint32_t acc_all = 0, acc_first = 0, acc_last = 0;
            pass_wl<true>(d, mk[4], mk[5], S, G, (mk[8] & 4u) != 0u, (mk[8] & 8u) != 0u, acc_all, acc_first, acc_last);
            pass_wl<false>(d, mk[4], mk[5], S, G, (mk[8] & 4u) != 0u, (mk[8] & 8u) != 0u, acc_all, acc_first, acc_last);
            const uint64_t all = __ballot(acc_all != 0), fst = __ballot(acc_first != 0), lst = __ballot(acc_last != 0);
            chg |= all;
            const unsigned e_first = (fst & INNER) ? 1u : 0u, e_last = (lst & INNER) ? 1u : 0u;
            const unsigned l1 = (all & B1) ? 1u : 0u, l62 = (all & B62) ? 1u : 0u;
            const unsigned f1 = (fst & B1) ? 1u : 0u, f62 = (fst & B62) ? 1u : 0u, g1 = (lst & B1) ? 1u : 0u, g62 = (lst & B62) ? 1u : 0u;
            wake |= (e_first << 3) | (e_last << 5) | (l1 << 1) | (l62 << 7) | (f1 << 0) | (f62 << 6) | (g1 << 2) | (g62 << 8);
            transpose32(d, scr_b, lane);
        }
        changed |= chg != 0;
        if (!chg) {
            capped = false;
            break;
        }
    }
}

// ---- a window of TWO classes (a seventh of the benchmark's tiles: it straddles one power of two): the same passes with one more bit per
// cell -- is it of the second class -- that selects the row's weights (11 instructions a row).  Two flat cells next to each other are on
// one level, so of one class: the weights are the target's, whatever the neighbour.  mk[9..10] / mk[11..12]: the class bits of the two layouts.
constexpr int NMK2 = 13;
#define NG_ROW2(O, I, B, BIT)                                                                                                                \
    "v_bfe_i32 %[k], %[c], " BIT ", 1\n\t"                                                                                                  \
    "v_bfi_b32 %[g], %[k], %[G2], %[G1]\n\t"                                                                                                \
    "v_bfi_b32 %[s], %[k], %[S2], %[S1]\n\t"                                                                                                \
    "v_add_u32 %[u], %[" B "], %[s]\n\t"                                                                                                    \
    "v_add_u32_dpp %[t], %[" B "], %[g] wave_shr:1" NG_DPP                                                                                   \
    "v_add_u32_dpp %[k], %[" B "], %[g] wave_shl:1" NG_DPP                                                                                   \
    "v_min3_u32 %[t], %[u], %[t], %[k]\n\t"                                                                                                 \
    "v_bfe_i32 %[k], %[m], " BIT ", 1\n\t"                                                                                                  \
    "v_min_u32 %[" O "], %[t], %[" I "]\n\t"                                                                                                \
    "v_bfi_b32 %[" O "], %[k], %[" I "], %[" O "]\n\t"                                                                                      \
    "v_bitop3_b32 %[acc], %[" O "], %[acc], %[" I "] bitop3:0xde\n\t"
struct TwoWeights {
    uint32_t S1, G1, S2, G2;
};
template <int R, int DIR>
__device__ __forceinline__ void rows6_wl2(uint32_t (&d)[WN], const uint32_t m, const uint32_t c, const TwoWeights w, uint32_t &acc)
{
    static_assert((R >> 5) == ((R + 5 * DIR) >> 5), "one mask word a block");
    uint32_t o0, o1, o2, o3, o4, o5, u, t, k, sw, gw;
    if constexpr (DIR > 0)
        asm volatile(NG_ROW2("o0", "i0", "b", "%[bit]") NG_ROW2("o1", "i1", "o0", "%[bit]+1") NG_ROW2("o2", "i2", "o1", "%[bit]+2")
                     NG_ROW2("o3", "i3", "o2", "%[bit]+3") NG_ROW2("o4", "i4", "o3", "%[bit]+4") NG_ROW2("o5", "i5", "o4", "%[bit]+5")
                     : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [u] "=&v"(u), [t] "=&v"(t),
                       [k] "=&v"(k), [s] "=&v"(sw), [g] "=&v"(gw), [acc] "+v"(acc)
                     : [b] "v"(d[R - DIR]), [i0] "v"(d[R]), [i1] "v"(d[R + DIR]), [i2] "v"(d[R + 2 * DIR]), [i3] "v"(d[R + 3 * DIR]),
                       [i4] "v"(d[R + 4 * DIR]), [i5] "v"(d[R + 5 * DIR]), [S1] "v"(w.S1), [G1] "v"(w.G1), [S2] "v"(w.S2), [G2] "v"(w.G2), [m] "v"(m),
                       [c] "v"(c), [bit] "n"(R & 31));
    else
        asm volatile(NG_ROW2("o0", "i0", "b", "%[bit]") NG_ROW2("o1", "i1", "o0", "%[bit]-1") NG_ROW2("o2", "i2", "o1", "%[bit]-2")
                     NG_ROW2("o3", "i3", "o2", "%[bit]-3") NG_ROW2("o4", "i4", "o3", "%[bit]-4") NG_ROW2("o5", "i5", "o4", "%[bit]-5")
                     : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [u] "=&v"(u), [t] "=&v"(t),
                       [k] "=&v"(k), [s] "=&v"(sw), [g] "=&v"(gw), [acc] "+v"(acc)
                     : [b] "v"(d[R - DIR]), [i0] "v"(d[R]), [i1] "v"(d[R + DIR]), [i2] "v"(d[R + 2 * DIR]), [i3] "v"(d[R + 3 * DIR]),
                       [i4] "v"(d[R + 4 * DIR]), [i5] "v"(d[R + 5 * DIR]), [S1] "v"(w.S1), [G1] "v"(w.G1), [S2] "v"(w.S2), [G2] "v"(w.G2), [m] "v"(m),
                       [c] "v"(c), [bit] "n"(R & 31));
    d[R] = o0; d[R + DIR] = o1; d[R + 2 * DIR] = o2; d[R + 3 * DIR] = o3; d[R + 4 * DIR] = o4; d[R + 5 * DIR] = o5;
}
template <int R, int DIR>
__device__ __forceinline__ uint32_t row1_wl2(uint32_t (&d)[WN], const uint32_t m, const uint32_t c, const TwoWeights w, uint32_t &acc)
{
    uint32_t o0, u, t, k, sw, gw;
    const uint32_t cu = d[R];
    asm volatile("s_nop 1\n\t" NG_ROW2("o0", "i0", "b", "%[bit]") "s_nop 1"
                 : [o0] "=&v"(o0), [u] "=&v"(u), [t] "=&v"(t), [k] "=&v"(k), [s] "=&v"(sw), [g] "=&v"(gw), [acc] "+v"(acc)
                 : [b] "v"(d[R - DIR]), [i0] "v"(cu), [S1] "v"(w.S1), [G1] "v"(w.G1), [S2] "v"(w.S2), [G2] "v"(w.G2), [m] "v"(m), [c] "v"(c),
                   [bit] "n"(R & 31));
    d[R] = o0;
    return o0 ^ cu;
}
template <int R, int DIR, int N>
__device__ __forceinline__ void blocks_wl2(uint32_t (&d)[WN], const uint32_t lo, const uint32_t hi, const uint32_t clo, const uint32_t chi, const TwoWeights w,
                                           uint32_t &acc)
{
    if constexpr (N > 0) {
        rows6_wl2<R, DIR>(d, R < 32 ? lo : hi, R < 32 ? clo : chi, w, acc);
        __builtin_amdgcn_sched_barrier(0);
        blocks_wl2<R + 6 * DIR, DIR, N - 1>(d, lo, hi, clo, chi, w, acc);
    }
}
template <bool DOWN>
__device__ __forceinline__ void pass_wl2(uint32_t (&d)[WN], const uint32_t imm_lo, const uint32_t imm_hi, const uint32_t cls_lo, const uint32_t cls_hi,
                                         const TwoWeights w, const bool edge_first, const bool edge_last, uint32_t &acc_all, uint32_t &acc_first,
                                         uint32_t &acc_last)
{
#pragma unroll
    for (int r = 0; r < WN; ++r) asm volatile("" : "+v"(d[r]));
    uint32_t lo = imm_lo, hi = imm_hi, clo = cls_lo, chi = cls_hi;
    asm volatile("" : "+v"(lo), "+v"(hi), "+v"(clo), "+v"(chi));
    if constexpr (DOWN) {
        const uint32_t x1 = row1_wl2<1, 1>(d, lo, clo, w, acc_all);
        acc_first |= edge_first ? x1 : 0u;
        __builtin_amdgcn_sched_barrier(0);
        blocks_wl2<2, 1, 10>(d, lo, hi, clo, chi, w, acc_all);
        const uint32_t x2 = row1_wl2<TI, 1>(d, hi, chi, w, acc_all);
        acc_last |= edge_last ? x2 : 0u;
    } else {
        const uint32_t x2 = row1_wl2<TI, -1>(d, hi, chi, w, acc_all);
        acc_last |= edge_last ? x2 : 0u;
        __builtin_amdgcn_sched_barrier(0);
        blocks_wl2<TI - 1, -1, 10>(d, lo, hi, clo, chi, w, acc_all);
        const uint32_t x1 = row1_wl2<1, -1>(d, lo, clo, w, acc_all);
        acc_first |= edge_first ? x1 : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void relax_wl2(uint32_t (&d)[WN], const uint32_t (&mk)[NMK2], const TwoWeights w, uint32_t scr_b, int lane, int maxcyc, unsigned &wake,
                                          bool &changed, bool &capped, unsigned &cycles)
{
    const uint64_t INNER = ((1ull << TI) - 1) << 1, B1 = 2ull, B62 = 1ull << TI;
    capped = true;
#pragma nounroll
    for (int cyc = 0; cyc < maxcyc; ++cyc) {
        uint64_t chg = 0;
        ++cycles;
        {   // lane = column, first / last = row 1 / 62
            uint32_t acc_all = 0, acc_first = 0, acc_last = 0;
            pass_wl2<true>(d, mk[0], mk[1], mk[9], mk[10], w, (mk[8] & 1u) != 0u, (mk[8] & 2u) != 0u, acc_all, acc_first, acc_last);
            pass_wl2<false>(d, mk[0], mk[1], mk[9], mk[10], w, (mk[8] & 1u) != 0u, (mk[8] & 2u) != 0u, acc_all, acc_first, acc_last);
            const uint64_t all = __ballot(acc_all != 0), fst = __ballot(acc_first != 0), lst = __ballot(acc_last != 0);
            chg |= all;
            const unsigned e_first = (fst & INNER) ? 1u : 0u, e_last = (lst & INNER) ? 1u : 0u;
            const unsigned l1 = (all & B1) ? 1u : 0u, l62 = (all & B62) ? 1u : 0u;
            const unsigned f1 = (fst & B1) ? 1u : 0u, f62 = (fst & B62) ? 1u : 0u, g1 = (lst & B1) ? 1u : 0u, g62 = (lst & B62) ? 1u : 0u;
            wake |= (e_first << 1) | (e_last << 7) | (l1 << 3) | (l62 << 5) | (f1 << 0) | (f62 << 2) | (g1 << 6) | (g62 << 8);
            transpose32(d, scr_b, lane);
        }
        {   // lane = row, first / last = column 1 / 62
            uint32_t acc_all = 0, acc_first = 0, acc_last = 0;
            pass_wl2<true>(d, mk[4], mk[5], mk[11], mk[12], w, (mk[8] & 4u) != 0u, (mk[8] & 8u) != 0u, acc_all, acc_first, acc_last);
            pass_wl2<false>(d, mk[4], mk[5], mk[11], mk[12], w, (mk[8] & 4u) != 0u, (mk[8] & 8u) != 0u, acc_all, acc_first, acc_last);
            const uint64_t all = __ballot(acc_all != 0), fst = __ballot(acc_first != 0), lst = __ballot(acc_last != 0);
            chg |= all;
            const unsigned e_first = (fst & INNER) ? 1u : 0u, e_last = (lst & INNER) ? 1u : 0u;
            const unsigned l1 = (all & B1) ? 1u : 0u, l62 = (all & B62) ? 1u : 0u;
            const unsigned f1 = (fst & B1) ? 1u : 0u, f62 = (fst & B62) ? 1u : 0u, g1 = (lst & B1) ? 1u : 0u, g62 = (lst & B62) ? 1u : 0u;
            wake |= (e_first << 3) | (e_last << 5) | (l1 << 1) | (l62 << 7) | (f1 << 0) | (f62 << 6) | (g1 << 2) | (g62 << 8);
            transpose32(d, scr_b, lane);
        }
        changed |= chg != 0;
        if (!chg) {
            capped = false;
            break;
        }
    }
}

// ---- a tile that is ONE flat (every cell of the 62 x 62 interior is adjacent to all of its 8 neighbours, one class): no masks --
// the interior of the lakes the tail rounds spend their time in.  Per row: min(left, right of the row behind) + Dg, the cell
// behind + S, min3 with the cell itself; the two ring lanes are put back (they belong to the neighbouring tiles).
#define NG_ROW_OPEN(O, I, B)                                                                                                                 \
    "v_add_u32 %[u], %[" B "], %[S]\n\t"                                                                                                    \
    "v_add_u32_dpp %[t], %[" B "], %[G] wave_shr:1" NG_DPP                                                                                   \
    "v_add_u32_dpp %[k], %[" B "], %[G] wave_shl:1" NG_DPP                                                                                   \
    "v_min3_u32 %[t], %[u], %[t], %[k]\n\t"                                                                                                 \
    "v_min_u32 %[" O "], %[t], %[" I "]\n\t"                                                                                                \
    "v_cndmask_b32_e64 %[" O "], %[" O "], %[" I "], %[ring]\n\t"                                                                            \
    "v_bitop3_b32 %[acc], %[" O "], %[acc], %[" I "] bitop3:0xde\n\t"
template <int R, int DIR>
__device__ __forceinline__ void rows6_open(uint32_t (&d)[WN], const uint64_t ring, const uint32_t S, const uint32_t G, uint32_t &acc)
{
    uint32_t o0, o1, o2, o3, o4, o5, u, t, k;
    asm volatile(NG_ROW_OPEN("o0", "i0", "b") NG_ROW_OPEN("o1", "i1", "o0") NG_ROW_OPEN("o2", "i2", "o1") NG_ROW_OPEN("o3", "i3", "o2")
                 NG_ROW_OPEN("o4", "i4", "o3") NG_ROW_OPEN("o5", "i5", "o4")
                 : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [u] "=&v"(u), [t] "=&v"(t), [k] "=&v"(k),
                   [acc] "+v"(acc)
                 : [b] "v"(d[R - DIR]), [i0] "v"(d[R]), [i1] "v"(d[R + DIR]), [i2] "v"(d[R + 2 * DIR]), [i3] "v"(d[R + 3 * DIR]), [i4] "v"(d[R + 4 * DIR]),
                   [i5] "v"(d[R + 5 * DIR]), [S] "v"(S), [G] "v"(G), [ring] "s"(ring));
    d[R] = o0; d[R + DIR] = o1; d[R + 2 * DIR] = o2; d[R + 3 * DIR] = o3; d[R + 4 * DIR] = o4; d[R + 5 * DIR] = o5;
}
template <int R, int DIR>
__device__ __forceinline__ uint32_t row1_open(uint32_t (&d)[WN], const uint64_t ring, const uint32_t S, const uint32_t G, uint32_t &acc)
{
    uint32_t o0, u, t, k;
    const uint32_t cu = d[R];
    asm volatile("s_nop 1\n\t" NG_ROW_OPEN("o0", "i0", "b") "s_nop 1"
                 : [o0] "=&v"(o0), [u] "=&v"(u), [t] "=&v"(t), [k] "=&v"(k), [acc] "+v"(acc)
                 : [b] "v"(d[R - DIR]), [i0] "v"(cu), [S] "v"(S), [G] "v"(G), [ring] "s"(ring));
    d[R] = o0;
    return o0 ^ cu;
}
template <int R, int DIR, int N>
__device__ __forceinline__ void blocks_open(uint32_t (&d)[WN], const uint64_t ring, const uint32_t S, const uint32_t G, uint32_t &acc)
{
    if constexpr (N > 0) {
        rows6_open<R, DIR>(d, ring, S, G, acc);
        __builtin_amdgcn_sched_barrier(0);
        blocks_open<R + 6 * DIR, DIR, N - 1>(d, ring, S, G, acc);
    }
}
// (the rows as text, like pass_wl: 7 instructions in 7 issue slots; the compiler's version had two s_nop in front of the DPP reads)
template <bool DOWN>
__device__ __forceinline__ void pass_open(uint32_t (&d)[WN], const uint32_t S, const uint32_t G, const bool ring_lane, uint32_t &acc_all, uint32_t &acc_first,
                                          uint32_t &acc_last)
{
#pragma unroll
    for (int r = 0; r < WN; ++r) asm volatile("" : "+v"(d[r]));
    const uint64_t ring = __ballot(ring_lane);
    if constexpr (DOWN) {
        acc_first |= row1_open<1, 1>(d, ring, S, G, acc_all);
        __builtin_amdgcn_sched_barrier(0);
        blocks_open<2, 1, 10>(d, ring, S, G, acc_all);
        acc_last |= row1_open<TI, 1>(d, ring, S, G, acc_all);
    } else {
        acc_last |= row1_open<TI, -1>(d, ring, S, G, acc_all);
        __builtin_amdgcn_sched_barrier(0);
        blocks_open<TI - 1, -1, 10>(d, ring, S, G, acc_all);
        acc_first |= row1_open<1, -1>(d, ring, S, G, acc_all);
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void relax_open(uint32_t (&d)[WN], const uint32_t S, const uint32_t G, uint32_t scr_b, int lane, int maxcyc, unsigned &wake,
                                           bool &changed, bool &capped, unsigned &cycles)
{
    const uint64_t INNER = ((1ull << TI) - 1) << 1, B1 = 2ull, B62 = 1ull << TI;
    const bool ring_lane = (lane == 0) | (lane == WN - 1);
    capped = true;
#pragma nounroll
    for (int cyc = 0; cyc < maxcyc; ++cyc) {
        uint64_t chg = 0;
        ++cycles;
#pragma nounroll
        for (int half = 0; half < 2; ++half) {
            uint32_t acc_all = 0, acc_first = 0, acc_last = 0;
            pass_open<true>(d, S, G, ring_lane, acc_all, acc_first, acc_last);
            pass_open<false>(d, S, G, ring_lane, acc_all, acc_first, acc_last);
            const uint64_t all = __ballot(acc_all != 0), fst = __ballot(acc_first != 0), lst = __ballot(acc_last != 0);
            chg |= all;
            const unsigned e_first = (fst & INNER) ? 1u : 0u, e_last = (lst & INNER) ? 1u : 0u;
            const unsigned l1 = (all & B1) ? 1u : 0u, l62 = (all & B62) ? 1u : 0u;
            const unsigned f1 = (fst & B1) ? 1u : 0u, f62 = (fst & B62) ? 1u : 0u, g1 = (lst & B1) ? 1u : 0u, g62 = (lst & B62) ? 1u : 0u;
            if (half == 0) wake |= (e_first << 1) | (e_last << 7) | (l1 << 3) | (l62 << 5) | (f1 << 0) | (f62 << 2) | (g1 << 6) | (g62 << 8);
            else wake |= (e_first << 3) | (e_last << 5) | (l1 << 1) | (l62 << 7) | (f1 << 0) | (f62 << 6) | (g1 << 2) | (g62 << 8);
            transpose32(d, scr_b, lane);
        }
        changed |= chg != 0;
        if (!chg) {
            capped = false;
            break;
        }
    }
}

// One tile visit.  FIRST (ng_first_kernel, every tile once): the window's words are built from the classification -- ring
// cleared, column layout transposed in, adjacency inverted -- and kept for all later visits as a 16 KB block per tile together
// with a header word (seams, class, "holds a flat cell at all"); distances start from the classification alone.  Later visits
// (ng_round_kernel) load header, block and distances and go straight to the passes.
constexpr uint32_t HDR_ACTIVE = 1u << 31, HDR_UNIFORM = 1u << 9, HDR_OPEN = 1u << 10, HDR_TWO = 1u << 11;   // (HDR_TWO: bits 24..30 = second class - first + 64)
template <bool FIRST>
__device__ __forceinline__ void visit(const GeoArgs &a, int t, const uint32_t *tab_l, uint32_t scr_b, int lane, unsigned &visits, unsigned &cycles)
{
#ifdef NG_PROFILE
    const long long tp0 = __builtin_amdgcn_s_memtime();
#endif
    const int64_t H = a.H, W = a.W;
    const int ti = t / a.ntc, tj = t - ti * a.ntc;
    const int64_t r0 = (int64_t)ti * TI, c0 = (int64_t)tj * TI;
    const int64_t cc = c0 + lane;
    const bool col_in = cc < W;
    // Loads are clamped into the raster: a column >= W reads column W - 1, a row >= H reads row H - 1.  Those are raster BORDER
    // cells: class 255, no adjacency -- to the window they look like sources nobody is adjacent to, which is all "outside the
    // raster" has to mean here.
    const uint32_t lane_c = (uint32_t)(col_in ? lane : W - 1 - c0);

    // buffer addressing: one shared per-lane byte offset + a scalar row offset (no 64-bit address arithmetic in VGPRs)
    const int64_t org = r0 * W + c0;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(a.d + org), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.blk + (int64_t)t * (WN * WN)), 0, WN * WN * 4, 0x00020000);
    int Wi = (int)W;
    asm volatile("" : "+s"(Wi));   // (a new name per tile: or the 64 row offsets r * Wi * 4 are hoisted out of the tile loop and live -- spilled into VGPR lanes -- through all of it)
    const int last_row = (int)(H - 1 - r0 < WN - 1 ? H - 1 - r0 : WN - 1);   // last window row inside the raster
    uint32_t ni[WN], d[WN], mk[NMK2];
    uint32_t hdr;
    if constexpr (FIRST) {
        // ---- classification of the window from the plain fill F (adjacency bits = AGNPS direction codes of common.hpp:
        // U 0, UR 1, R 2, DR 3, D 4, DL 5, L 6, UL 7).  A cell with a lower neighbour, or on the raster border, is a source;
        // every other interior cell is a flat cell: word = same-level adjacency | class << 8.  Class 255 = not a flat cell.
        // The halo ring of the window gets no adjacency (its cells belong to the neighbouring tiles: they never move here);
        // a ring cell that is not KNOWN to be a source from inside the window starts unreached -- an upper bound, which is all
        // the relaxation needs -- and is read from memory in the later rounds.
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)(a.F + org), 0, 0x7fffffff, 0x00020000);
        const bool ring_lane = (lane == 0) | (lane == WN - 1);
        const bool lane_in = (lane >= 1) & (lane <= TI) & (cc < W - 1);          // a column of interior raster cells
        const bool lane_border = (tj == 0 && lane == 0) | (cc >= W - 1);          // raster border column (or its clamped copies)
        const float PINF = __builtin_inff();
        const int border_row = (int)(H - 1 - r0 < 2 * WN ? H - 1 - r0 : 2 * WN);   // window row of the raster's last row (or beyond the window)
        uint32_t lake_any = 0, nirr = 0, nfatal = 0;
        // INNER: the window lies inside the raster and touches none of its border rows / columns (all tiles but the outermost ring): no
        // clamped row offsets, no border tests -- each of them a scalar compare and a live SGPR pair per row of a loop that is short of
        // SGPRs (the compiler parks them in VGPR lanes: 1200 v_readlane / v_writelane in the general version)
        auto classify = [&](auto inner_tag) {
            constexpr bool INNER = decltype(inner_tag)::value;
            const bool lane_in_c = INNER ? ((lane >= 1) & (lane <= TI)) : lane_in;
            float f[WN];
            int so = 0;
#pragma unroll
            for (int r = 0; r < WN; ++r) {
                f[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rf, (int)((INNER ? (uint32_t)lane : lane_c) * 4u), INNER ? r * Wi * 4 : so * 4, 0));
                so = r < last_row ? so + Wi : so;
            }
            float upl = PINF, up = PINF, upr = PINF, cul = fleft(f[0]), cur = fright(f[0]);
            float h_up = PINF, h_cu = fminf(fminf(cul, f[0]), cur);      // minimum of a row's three cells around this column: row r - 1, row r
            // (the class's table entry one row ahead: read in the row before, or every row waits for its LDS round trip)
            uint32_t e_cu = class_above(f[0]), tv_cu = tab_l[e_cu & 0xffu];
#pragma unroll
            for (int r = 0; r < WN; ++r) {
                const float V = f[r], dn = r + 1 < WN ? f[r + 1] : PINF;
                const float dnl = r + 1 < WN ? fleft(dn) : PINF, dnr = r + 1 < WN ? fright(dn) : PINF;
                // (one compare with the minimum of the eight instead of eight compares and seven scalar ORs: the classification is bound by
                // the instructions a wavefront issues; a NaN neighbour is no minimum and was never "lower" either)
                const float h_dn = fminf(fminf(dnl, dn), dnr);
                const bool lower = fminf(fminf(h_up, h_dn), fminf(cul, cur)) < V;
                h_up = h_cu;
                h_cu = h_dn;
                // raster border rows are sources; a band's halo rows are the neighbour's cells (ring-like: never moved here, known
                // only as far as this window can tell); window rows beyond the local raster are nothing
                const bool border = INNER ? false : (lane_border | (ti == 0 && r == 0 && !a.fixed_top) | (r == border_row && !a.fixed_bot) | (r > border_row));
                const bool src = lower | border;
                const bool ring = (r == 0) | (r == WN - 1) | ring_lane | (INNER ? false : (r >= border_row));
                // (no branches on per-lane values here: selects only -- see tools/lint_exec_spills.py)
                const uint32_t adj = (up == V ? 1u : 0u) | (upr == V ? 2u : 0u) | (cur == V ? 4u : 0u) | (dnr == V ? 8u : 0u) |
                                     (dn == V ? 16u : 0u) | (dnl == V ? 32u : 0u) | (cul == V ? 64u : 0u) | (upl == V ? 128u : 0u);
                const uint32_t e = e_cu, tv = tv_cu;
                e_cu = r + 1 < WN ? class_above(dn) : 255u;
                tv_cu = tab_l[e_cu & 0xffu];
                const bool cell = lane_in_c & !border & !ring;     // an interior raster cell of this tile
                const bool flat = cell & !src;
                const bool regular = flat & (adj != 0u) & (e < 253u) & (tv != 0u);     // (253 .. 255: M_IRR, M_WALL, M_NOFLAT)
                const bool nan = cell & (V != V);
                const bool irregular = flat & !regular & !nan;     // a level without integer weights: left to the float64 relaxation
                nirr += irregular ? 1u : 0u;
                nfatal += nan ? 1u : 0u;                            // NaN cells: not for this path at all
                const uint32_t w = regular ? (adj | (e << 8)) : (irregular ? M_IRR : M_NOFLAT);
                lake_any |= w;
                ni[r] = w;
                upl = cul; up = V; upr = cur;
                cul = dnl; cur = dnr;
                __builtin_amdgcn_sched_barrier(0);   // row by row: every comparison is a live SGPR pair until its select has been issued
            }
            // ---- the walls: a cell that some regular flat cell next to it is not adjacent to.  q = the directions in which a flat cell
            // has such a neighbour; a cell collects the bits that point at it from its 8 neighbours (A: one word per row with the bits
            // of the cell itself, of its left and of its right neighbour, by the rows they matter to)
            {
                auto arow = [&](int r) -> uint32_t {
                    if (r < 0 || r >= WN) return 0u;
                    const uint32_t q = (ni[r] & 0xffu) != 0u ? (~ni[r] & 0xffu) : 0u;
                    return (q & 0x11u) | (from_left(q) & 0x0eu) | (from_right(q) & 0xe0u);
                };
                uint32_t Ap = 0u, Ac = arow(0);
#pragma unroll
                for (int r = 0; r < WN; ++r) {
                    const uint32_t An = arow(r + 1);
                    const uint32_t hit = (An & 0x83u) | (Ap & 0x38u) | (Ac & 0x44u);
                    ni[r] = (hit != 0u && (ni[r] & 0xffu) == 0u) ? M_WALL : ni[r];
                    Ap = Ac; Ac = An;
                    asm volatile("" : "+v"(ni[r]), "+v"(Ap), "+v"(Ac));      // (the row is finished here, not after the loop)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- the start distances, straight to memory (the first relaxation is a visit of the next launch): 0 for a source, D_IRR
            // for a flat cell of an irregular level; a regular flat cell starts "not reached" -- or, next to a wall of its own level,
            // from the step that wall would have given it
            {
                auto trow = [&](int r) -> uint32_t {      // wall bits of (r, lane) | (r, lane - 1) << 1 | (r, lane + 1) << 2
                    const uint32_t c = (ni[r] & 0xff00u) == M_WALL ? 1u : 0u;
                    return c | (from_left(c) << 1) | (from_right(c) << 2);
                };
                uint32_t P = trow(0), C = trow(1);
#pragma unroll
                for (int r = 1; r <= TI; ++r) {
                    const uint32_t N = trow(r + 1);
                    const uint32_t nb = (P & 1u) | ((P & 4u) >> 1) | (C & 4u) | ((N & 4u) << 1) | ((N & 1u) << 4) | ((N & 2u) << 4) | ((C & 2u) << 5) | ((P & 2u) << 6);
                    const uint32_t w = ni[r], sb = w & nb;        // (w & 0xff = the adjacency of a regular flat cell, 0 otherwise)
                    const uint32_t e = (w >> 8) & 0xffu;
                    uint32_t seeded = DINF;
                    if (__any(sb != 0u)) {        // (a row without a seed -- two of three -- does not wait for the table)
                        const uint32_t S = tab_l[e], G = tab_l[256 + e];
                        seeded = min((sb & 0x55u) ? S : DINF, (sb & 0xaau) ? G : DINF);
                    }
                    const uint32_t d0 = (w & 0xffu) != 0u ? seeded : ((w & 0xff00u) == M_IRR ? D_IRR : 0u);
                    if ((INNER || r < last_row) && lane_in_c) __builtin_amdgcn_raw_buffer_store_b32(d0, rd, lane * 4, r * Wi * 4, 0);
                    P = C; C = N;
                    asm volatile("" : "+v"(P), "+v"(C));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (ti > 0 && tj > 0 && H - 1 - r0 > WN - 1 && c0 + WN - 1 < W - 1) classify(std::true_type{});
        else classify(std::false_type{});
        {
            uint32_t tot = nirr | (nfatal << 16);      // (at most 62 x 62 of either per window)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) tot += (uint32_t)__shfl_xor((int)tot, o);
            if ((tot & 0xffffu) && lane == 0) atomicAdd(&a.counters[C_IRREGULAR], (unsigned long long)(tot & 0xffffu));
            if ((tot >> 16) && lane == 0) atomicAdd(&a.counters[C_FATAL], (unsigned long long)(tot >> 16));
        }
        // the raster border cells inside this window are sources: their distance 0 is read by every later visit (and by
        // ng_assemble), so somebody has to write it
        if (ti == 0 || tj == 0 || H - 1 - r0 <= WN - 1 || c0 + WN - 1 >= W - 1) {   // (<=: the raster's last row may be the window's last row)
#pragma unroll
            for (int r = 0; r < WN; ++r) {
                const bool brow = (ti == 0 && r == 0 && !a.fixed_top) | (r == last_row && last_row == (int)(H - 1 - r0) && !a.fixed_bot);
                const bool halo = (ti == 0 && r == 0 && a.fixed_top) | (r == last_row && last_row == (int)(H - 1 - r0) && a.fixed_bot);
                const bool bcell = r <= last_row && col_in && !halo && (brow | (tj == 0 && lane == 0) | (cc == W - 1));
                if (bcell) __builtin_amdgcn_raw_buffer_store_b32(0u, rd, lane * 4, r * Wi * 4, 0);
            }
        }
        if (!__any((lake_any & 0xffu) != 0u)) {   // nothing in this tile can move, ever: all its cells are sources (or irregular)
            if (lane == 0) a.hdr[t] = 0u;
            return;
        }
        __builtin_amdgcn_sched_barrier(0);
        // the column layout's words: direction (dr, dc) becomes (dc, dr) = bits 0..6 reversed, bit 7 stays.  Afterwards
        // ni = (this layout's word | the other layout's word << 16) with both adjacency bytes INVERTED for the passes.
        {
            uint32_t tw[WN];
#pragma unroll
            for (int r = 0; r < WN; ++r) {
                const uint32_t adj = ni[r] & 0xffu;
                tw[r] = (__builtin_bitreverse32(adj << 25) | (adj & 0x80u)) | (ni[r] & 0xff00u);
            }
            lds_put_rows<0, WN>(scr_b + 4u * lane, tw);
            lds_wait();
            // read back in two halves: never more than 32 transposed words in flight next to the 64 of this layout
            lds_get_cols<0, 32>(scr_b + 4u * (WN + 1) * lane, tw);
            lds_wait();
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                asm volatile("" : "+v"(tw[r]));
                ni[r] = (ni[r] | (tw[r] << 16)) ^ 0x00ff00ffu;
            }
            lds_get_cols<32, 32>(scr_b + 4u * (WN + 1) * lane, tw);
            lds_wait();
#pragma unroll
            for (int r = 32; r < WN; ++r) {
                asm volatile("" : "+v"(tw[r]));
                ni[r] = (ni[r] | (tw[r] << 16)) ^ 0x00ff00ffu;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // one class in the whole window?  (from the words that stay live anyway: the adjacency byte is inverted by now)
        // (new names for the words in front of every loop over them: what one loop extracts from a word -- its class byte, a compare -- must
        // not be kept for the next loop: 62 live registers each time, i.e. spills)
        auto fresh = [&]() {
#pragma unroll
            for (int r = 0; r < WN; ++r) asm volatile("" : "+v"(ni[r]));
        };
        uint32_t esel = 255u;
        fresh();
#pragma unroll
        for (int r = 1; r <= TI; ++r) esel = (ni[r] & 0xffu) != 0xffu ? (ni[r] >> 8) & 0xffu : esel;
        const uint64_t has = __ballot(esel != 255u);
        const uint32_t eref = (uint32_t)__builtin_amdgcn_readlane((int)esel, (int)__builtin_ctzll(has));   // has != 0: the tile holds a flat cell
        uint32_t mism = 0;
        fresh();
#pragma unroll
        for (int r = 1; r <= TI; ++r) mism |= (ni[r] & 0xffu) != 0xffu ? ((ni[r] >> 8) & 0xffu) ^ eref : 0u;
        // which of the 8 neighbouring tiles share a flat with this one at all?  (bit numbering of `wake`)
        unsigned seams = 1u << 4;
        bool open_tile = false;
        {
            uint32_t l_or = 0xffu;   // AND of the inverted bytes over rows 1..62 = inverted OR of the adjacency
            fresh();
#pragma unroll
            for (int r = 1; r <= TI; ++r) l_or &= ni[r];
            uint32_t any_gap = 0;    // OR of the inverted bytes: a cell that lacks one of its 8 neighbours on its level
#pragma unroll
            for (int r = 1; r <= TI; ++r) any_gap |= ni[r] & 0xffu;
            open_tile = (__ballot(any_gap != 0u) & (((1ull << TI) - 1) << 1)) == 0ull;
            const uint64_t top = __ballot((ni[1] & 0x83u) != 0x83u), bot = __ballot((ni[TI] & 0x38u) != 0x38u);
            const uint64_t lft = __ballot((l_or & 0xe0u) != 0xe0u), rgt = __ballot((l_or & 0x0eu) != 0x0eu);
            const uint64_t INNER = ((1ull << TI) - 1) << 1, B1 = 2ull, B62 = 1ull << TI;
            seams |= (top & INNER) ? 1u << 1 : 0u;
            seams |= (bot & INNER) ? 1u << 7 : 0u;
            seams |= (lft & B1) ? 1u << 3 : 0u;
            seams |= (rgt & B62) ? 1u << 5 : 0u;
            const uint64_t tl = __ballot((ni[1] & 0x80u) == 0u), tr = __ballot((ni[1] & 0x02u) == 0u);
            const uint64_t bl = __ballot((ni[TI] & 0x20u) == 0u), br = __ballot((ni[TI] & 0x08u) == 0u);
            seams |= (tl & B1) ? 1u << 0 : 0u;
            seams |= (tr & B62) ? 1u << 2 : 0u;
            seams |= (bl & B1) ? 1u << 6 : 0u;
            seams |= (br & B62) ? 1u << 8 : 0u;
        }
        const bool uniform = !__any(mism != 0u);
        // exactly two classes?  (the second one as a difference to the first: windows span neighbouring binades)
        uint32_t two = 0, c2v = 0;
        if (!uniform) {
            uint32_t e2 = 255u;
            fresh();
#pragma unroll
            for (int r = 1; r <= TI; ++r) {
                const uint32_t e = (ni[r] >> 8) & 0xffu;
                e2 = ((ni[r] & 0xffu) != 0xffu && e != eref) ? e : e2;
            }
            const uint64_t has2 = __ballot(e2 != 255u);       // (not uniform: some lane holds a second class)
            const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)e2, (int)__builtin_ctzll(has2 ? has2 : 1ull));
            uint32_t m2 = 0;
            fresh();
#pragma unroll
            for (int r = 1; r <= TI; ++r) {
                const uint32_t e = (ni[r] >> 8) & 0xffu;
                m2 |= ((ni[r] & 0xffu) != 0xffu && e != eref && e != c2) ? 1u : 0u;
            }
            const int delta = (int)c2 - (int)eref;
            if (has2 && !__any(m2 != 0u) && delta >= -64 && delta < 64) two = HDR_TWO | ((uint32_t)(delta + 64) << 24);
            c2v = c2;
        }
        hdr = HDR_ACTIVE | seams | (uniform ? HDR_UNIFORM : 0u) | ((uniform && open_tile) ? HDR_OPEN : 0u) | (eref << 16) | two;
        __builtin_amdgcn_sched_barrier(0);
        // the block: packed for a window of one class (relax_pk), the full words otherwise; none for a tile that is one flat
        if ((uniform && !open_tile) || two) {
            uint32_t mk[NMK2] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            fresh();
#pragma unroll
            for (int r = 0; r < WN; ++r) {
                const uint32_t w = ni[r], bit = 1u << (r & 31);
                const int h = r >> 5;
                mk[0 + h] |= (w & 0xffu) == 0xffu ? bit : 0u;                  // (the adjacency bytes are inverted by now: 0xff = none)
                mk[2 + h] |= (w & 0xff00u) == M_WALL ? bit : 0u;
                mk[4 + h] |= (w & 0xff0000u) == 0xff0000u ? bit : 0u;
                mk[6 + h] |= (w >> 24) == (M_WALL >> 8) ? bit : 0u;
                if (two) {        // (flat cells only: the others never move)
                    mk[9 + h] |= ((w & 0xffu) != 0xffu && ((w >> 8) & 0xffu) == c2v) ? bit : 0u;
                    mk[11 + h] |= ((w & 0xff0000u) != 0xff0000u && (w >> 24) == c2v) ? bit : 0u;
                }
            }
            mk[8] = ((ni[1] & 0x83u) != 0x83u ? 1u : 0u) | ((ni[TI] & 0x38u) != 0x38u ? 2u : 0u) |
                    ((ni[1] & 0x830000u) != 0x830000u ? 4u : 0u) | ((ni[TI] & 0x380000u) != 0x380000u ? 8u : 0u);
#pragma unroll
            for (int k = 0; k < NMK; ++k) __builtin_amdgcn_raw_buffer_store_b32(mk[k], rb, lane * 4, k * WN * 4, 0);
            if (two) {
#pragma unroll
                for (int k = NMK; k < NMK2; ++k) __builtin_amdgcn_raw_buffer_store_b32(mk[k], rb, lane * 4, k * WN * 4, 0);
            }
        } else if (!uniform) {
#pragma unroll
            for (int r = 0; r < WN; ++r) __builtin_amdgcn_raw_buffer_store_b32(ni[r], rb, lane * 4, r * WN * 4, 0);
        }
        if (lane == 0) {
            a.hdr[t] = hdr;
            a.mark[t] = 1;       // the first relaxation of the tile is a visit of the next launch like any other
        }
        // (classification and relaxation in one kernel cost more than this extra trip of the distances through memory: the
        // two phases together do not fit the register file and the compiler spills in both)
        return;
    } else {
        // (the distances first: their addresses need nothing but the tile's number, the header -- which block, if any -- is one more
        // round trip that the 64 row loads cover; in the tail rounds a launch is little more than the chain of such round trips)
        const uint32_t hdr_m = a.hdr[t];
        // (a window that lies inside the raster -- every tile but the bottom row's -- takes its rows at fixed multiples of the pitch: the
        // clamped row offsets below cost a dozen scalar instructions and two saved masks per row, again in front of every store)
        if (last_row == WN - 1) {
#pragma unroll
            for (int r = 0; r < WN; ++r) d[r] = __builtin_amdgcn_raw_buffer_load_b32(rd, (int)(lane_c * 4u), r * Wi * 4, 0);
        } else {
            int so = 0;
#pragma unroll
            for (int r = 0; r < WN; ++r) {
                d[r] = __builtin_amdgcn_raw_buffer_load_b32(rd, (int)(lane_c * 4u), so * 4, 0);
                so = r < last_row ? so + Wi : so;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        hdr = __builtin_amdgcn_readfirstlane(hdr_m);
        if (!(hdr & HDR_ACTIVE)) return;   // woken by a neighbour whose flat ends on my ring: nothing of mine can move
        if (hdr & HDR_OPEN) {              // (a tile that is one flat needs no words)
        } else if (hdr & HDR_UNIFORM) {    // one class: bit masks (pass_wl)
#pragma unroll
            for (int k = 0; k < NMK; ++k) mk[k] = __builtin_amdgcn_raw_buffer_load_b32(rb, lane * 4, k * WN * 4, 0);
        } else if (hdr & HDR_TWO) {        // two classes: the class bits as well (pass_wl2)
#pragma unroll
            for (int k = 0; k < NMK2; ++k) mk[k] = __builtin_amdgcn_raw_buffer_load_b32(rb, lane * 4, k * WN * 4, 0);
        } else {
#pragma unroll
            for (int r = 0; r < WN; ++r) ni[r] = __builtin_amdgcn_raw_buffer_load_b32(rb, lane * 4, r * WN * 4, 0);
        }
    }
    ++visits;
    __builtin_amdgcn_sched_barrier(0);
#ifdef NG_PROFILE
    const long long tp1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tp2 = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t eref = (hdr >> 16) & 0xffu;
    unsigned wake = 0;
    bool changed = false, capped = false;
    if (hdr & HDR_OPEN) relax_open(d, tab_l[eref], tab_l[256 + eref], scr_b, lane, a.maxcyc, wake, changed, capped, cycles);
    else if (hdr & HDR_UNIFORM) {
#pragma unroll
        for (int r = 0; r < WN; ++r) d[r] |= (uint32_t)__builtin_amdgcn_sbfe((int)mk[2 + (r >> 5)], r & 31, 1) & DINF;      // the walls (0 in memory)
        relax_wl(d, mk, tab_l[eref], tab_l[256 + eref], scr_b, lane, a.maxcyc, wake, changed, capped, cycles);
        if (changed) {
#pragma unroll
            for (int r = 1; r <= TI; ++r) d[r] &= ~(uint32_t)__builtin_amdgcn_sbfe((int)mk[2 + (r >> 5)], r & 31, 1);
        }
    }
    else if (hdr & HDR_TWO) {
        const uint32_t c2 = (eref + ((hdr >> 24) & 0x7fu) - 64u) & 0xffu;
#pragma unroll
        for (int r = 0; r < WN; ++r) d[r] |= (uint32_t)__builtin_amdgcn_sbfe((int)mk[2 + (r >> 5)], r & 31, 1) & DINF;
        relax_wl2(d, mk, TwoWeights{tab_l[eref], tab_l[256 + eref], tab_l[c2], tab_l[256 + c2]}, scr_b, lane, a.maxcyc, wake, changed, capped, cycles);
        if (changed) {
#pragma unroll
            for (int r = 1; r <= TI; ++r) d[r] &= ~(uint32_t)__builtin_amdgcn_sbfe((int)mk[2 + (r >> 5)], r & 31, 1);
        }
    } else relax(d, ni, LaneW{}, tab_l, scr_b, lane, a.maxcyc, wake, changed, capped, cycles);
    wake &= hdr & 0x1ffu;
#ifdef NG_PROFILE
    const long long tp3 = __builtin_amdgcn_s_memtime();
#endif

    if (changed) {
        const bool lane_ok = (lane >= 1) & (lane <= TI) & (cc < W - 1);
        if (lane_ok) {
            if (last_row == WN - 1) {
#pragma unroll
                for (int r = 1; r <= TI; ++r) __builtin_amdgcn_raw_buffer_store_b32(d[r], rd, lane * 4, r * Wi * 4, 0);
            } else {
#pragma unroll
                for (int r = 1; r <= TI; ++r)
                    if (r < last_row) __builtin_amdgcn_raw_buffer_store_b32(d[r], rd, lane * 4, r * Wi * 4, 0);   // row H - 1 is a border row
            }
        }
    }
    if (capped) wake |= 1u << 4;
    if (wake) {
        const int p = ti + lane / 3 - 1, q = tj + lane % 3 - 1;
        const bool mine = lane < 9 && ((wake >> lane) & 1u) && p >= 0 && p < a.ntr && q >= 0 && q < a.ntc;
        const int tp = p * a.ntc + q;
        if (!FIRST && a.append) {
            bool fresh = false;
            if (mine) fresh = atomicMax(&a.amark[tp], a.round_next) < a.round_next;
            const uint64_t bal = __ballot(fresh);
            if (bal) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.count_next, (uint32_t)__builtin_popcountll(bal));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (fresh) a.list_next[base + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull))] = tp;
            }
        } else if (mine) {
            a.mark[tp] = 1;
        }
    }
#ifdef NG_PROFILE
    if (!FIRST) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long tp4 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            const int o = C_STATS + 128 + ((hdr & HDR_OPEN) ? 8 : 0);
            atomicAdd(&a.counters[o + 0], 1ull);
            atomicAdd(&a.counters[o + 1], (unsigned long long)(tp1 - tp0));
            atomicAdd(&a.counters[o + 2], (unsigned long long)(tp2 - tp1));
            atomicAdd(&a.counters[o + 3], (unsigned long long)(tp3 - tp2));
            atomicAdd(&a.counters[o + 4], (unsigned long long)(tp4 - tp3));
        }
    }
#endif
}

// WPB: wavefronts per workgroup.  Four (68 KB of LDS a workgroup) while the rounds have the chip to themselves; ONE (18.7 KB) for the
// tail rounds when the label branch runs beside them (mhip_ctx_run, MHIP_LABEL_START=1): a 68 KB workgroup of four 242-register
// waves only fits a CU from which half of the labelling's workgroups have gone -- and the dispatcher refills every gap with the
// labelling's small ones first, because THEY fit (no-flats stage 6.8 -> 9.5 ms in round 3 and again in round 4) --, a single wave with
// 18.7 KB fits as soon as one of them leaves.
template <bool FIRST, int WPB = 4>
__device__ __forceinline__ void round_body(const GeoArgs &a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t scr_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(lds + wave * (WN * (WN + 1)));
    uint32_t *tab_l = lds + WPB * (WN * (WN + 1));
    // (the table's two words per thread and the round's tile count in flight together: as a loop, then the count behind the barrier, they
    // were three dependent round trips in front of every launch)
    const int nwaves = (int)gridDim.x * WPB, gw = (int)blockIdx.x * WPB + wave;
    uint32_t tw[512 / (64 * WPB)];
#pragma unroll
    for (int k = 0; k < 512 / (64 * WPB); ++k) tw[k] = a.tab[threadIdx.x + k * 64 * WPB];
    const int n = FIRST ? a.nt : (int)__builtin_amdgcn_readfirstlane((int)*a.count);
    const int t_first = FIRST ? 0 : a.list[gw < a.nt ? gw : a.nt - 1];      // (this wave's first tile, if the round has that many: no fourth trip)
#pragma unroll
    for (int k = 0; k < 512 / (64 * WPB); ++k) tab_l[threadIdx.x + k * 64 * WPB] = tw[k];
    __syncthreads();
    unsigned visits = 0, cycles = 0;
    for (int i = gw; i < n; i += nwaves) {
        const int t = FIRST ? i : __builtin_amdgcn_readfirstlane(i == gw ? t_first : a.list[i]);
        visit<FIRST>(a, t, tab_l, scr_b, lane, visits, cycles);
    }
    if (lane == 0 && visits) {
        unsigned long long *st = a.counters + C_STATS + 2 * (gw & 63);
        atomicAdd(&st[0], (unsigned long long)visits);
        atomicAdd(&st[1], (unsigned long long)cycles);
    }
}
__global__ __launch_bounds__(256, 2) void ng_first_kernel(GeoArgs a) { round_body<true>(a); }
__global__ __launch_bounds__(256, 2) void ng_round_kernel(GeoArgs a) { round_body<false>(a); }
__global__ __launch_bounds__(64, 2) void ng_round_w1_kernel(GeoArgs a) { round_body<false, 1>(a); }

// ---- the marks of one round -> the tile list of the next (clears the marks) --------------------------------------------------
// The mark bytes are read as 64-bit words (the array is padded to a multiple of 8 bytes), one word per thread; the order of the
// list is irrelevant: a workgroup counts its tiles in LDS and takes its range of the list with one global atomic on the round's
// counter (zero before: a round's counter is written by nobody else in mark mode).  One workgroup over the whole array took
// 45 us when most tiles were marked (the first rounds) and 10 us when few were.
__global__ __launch_bounds__(256) void ng_compact_kernel(unsigned long long *mark8, int nwords, int *list, uint32_t *count)
{
    __shared__ uint32_t n_l, base_l;
    if (threadIdx.x == 0) n_l = 0;
    __syncthreads();
    const int w = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const unsigned long long v = w < nwords ? mark8[w] : 0ull;
    unsigned k = 0;
    uint32_t o = 0;
    if (v) {
        mark8[w] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) k += ((v >> (8 * j)) & 0xffull) ? 1u : 0u;
        o = atomicAdd(&n_l, k);
    }
    __syncthreads();
    if (threadIdx.x == 0) base_l = n_l ? atomicAdd(count, n_l) : 0u;
    __syncthreads();
    if (v) {
        o += base_l;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((v >> (8 * j)) & 0xffull) list[o++] = 8 * w + j;
    }
}

// ---- G = F + u * D -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ng_assemble_kernel(const float *__restrict__ F, const uint32_t *__restrict__ d, double *__restrict__ G, int64_t n,
                                                          double seed_add, unsigned long long *counters)
{
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    unsigned bad = 0;
    // a source has distance 0: G = F whatever its class
    auto value = [&](float f, uint32_t dd) {
        if (dd == D_IRR) return (double)f + seed_add;   // a flat cell of an irregular level: an upper bound for the relaxation
        if (dd >= 0x80000000u) {                        // a distance beyond the uint32 headroom (a very long flat in a low binade;
            ++bad;                                      // every smaller distance is exact): the relaxation settles it as well
            return (double)f + seed_add;
        }
        const uint32_t e = class_above(f);
        return dd ? (double)f + (double)dd * class_ulp(e) : (double)f;
    };
    if (i0 + 4 <= n) {
        const float4 f = *reinterpret_cast<const float4 *>(F + i0);
        const uint4 dv = *reinterpret_cast<const uint4 *>(d + i0);
        typedef double __attribute__((ext_vector_type(2))) v2d;
        *reinterpret_cast<v2d *>(G + i0) = v2d{value(f.x, dv.x), value(f.y, dv.y)};
        *reinterpret_cast<v2d *>(G + i0 + 2) = v2d{value(f.z, dv.z), value(f.w, dv.w)};
    } else {
        for (int64_t i = i0; i < n; ++i) G[i] = value(F[i], d[i]);
    }
    if (bad) atomicAdd(&counters[C_UNREACHED], (unsigned long long)bad);
}

// ---- the strict equation (*) at every cell, in the reference's arithmetic (_fill.pyx:107-117) ------------------------------
constexpr int VRB = 16;   // rows per thread
__global__ __launch_bounds__(256) void ng_verify_kernel(const float *__restrict__ dem, const double *__restrict__ G, int64_t H, int64_t W, double sh,
                                                        double dg, int fixed_top, int fixed_bot, unsigned long long *counters)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rb = (int64_t)blockIdx.y * VRB;
    if (c >= W) return;
    const double PINF = __builtin_inf();
    const bool cl = c > 0, cr = c + 1 < W;
    auto row = [&](int64_t r, double (&v)[3]) {
        const bool in = r >= 0 && r < H;
        const double *p = G + (in ? r : 0) * W + c;
        v[0] = (in && cl) ? p[-1] : PINF;
        v[1] = in ? p[0] : PINF;
        v[2] = (in && cr) ? p[1] : PINF;
    };
    double a[3], b[3], n[3];
    row(rb - 1, a);
    row(rb, b);
    unsigned bad = 0;
#pragma unroll 4
    for (int i = 0; i < VRB; ++i) {
        const int64_t r = rb + i;
        if (r >= H) break;
        row(r + 1, n);
        const double own = b[1], dv = (double)dem[r * W + c];
        double want = dv;
        if (r > 0 && r < H - 1 && cl && cr) {
            const double md = fmin(fmin(a[0], a[2]), fmin(n[0], n[2])) + dg;
            const double me = fmin(fmin(a[1], b[0]), fmin(b[2], n[1])) + sh;
            want = fmax(fmin(md, me), dv);
        }
        const bool halo = (r == 0 && fixed_top) || (r == H - 1 && fixed_bot);   // the neighbouring band checks its own rows
        bad += (own == want || halo) ? 0u : 1u;   // NaN anywhere fails too
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a[k] = b[k];
            b[k] = n[k];
        }
    }
    if (bad) atomicAdd(&counters[C_MISMATCH], (unsigned long long)bad);
}

// ---- both at once (the usual case: no irregular level): G is assembled in registers, checked against its eight neighbours and
// written -- the separate verification read all of G back (8 of its 12 bytes per cell).  A wavefront owns 62 columns; lanes 0 and
// 63 carry the neighbouring columns (values only), left / right neighbours come over DPP like in the tile kernels.
constexpr int FRB = 32;   // rows per thread
__device__ __forceinline__ double dleft(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = from_left((uint32_t)b), hi = from_left((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double dright(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = from_right((uint32_t)b), hi = from_right((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// D8: the cell's flow direction as well (flow.terrain_flowdirection with the edges flowing outward, d8.hip) -- the 3 x 3
// neighbourhood of G is in registers here already; as a pass of its own D8 reads all of G back (8 of its 9 bytes per cell).
template <bool D8>
__global__ __launch_bounds__(256) void ng_finish_kernel(const float *__restrict__ F, const uint32_t *__restrict__ d, const float *__restrict__ dem,
                                                        double *__restrict__ G, int64_t H, int64_t W, double sh, double dg, double seed_add, int fixed_top,
                                                        int fixed_bot, unsigned long long *counters, uint8_t *__restrict__ flowdir, unsigned int *nodir)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = ((int64_t)blockIdx.x * 4 + wave) * TI + lane - 1;
    const int64_t rb = (int64_t)blockIdx.y * FRB;
    const bool col_in = c >= 0 && c < W;
    const int64_t cc = c < 0 ? 0 : (c < W ? c : W - 1);
    const bool mine = lane >= 1 && lane <= TI && col_in;                // this lane writes (and checks) its column
    const bool inner_col = c > 0 && c + 1 < W;
    const double PINF = __builtin_inf();
    unsigned unreached = 0, bad = 0, anynodir = 0;
    auto value = [&](int64_t r) -> double {
        if (!(r >= 0 && r < H && col_in)) return PINF;
        const float f = F[r * W + cc];
        const uint32_t dd = d[r * W + cc];
        if (dd == D_IRR) return (double)f + seed_add;                   // (ng_assemble_kernel)
        if (dd >= 0x80000000u) {
            unreached += mine && r >= rb && r < rb + FRB ? 1u : 0u;
            return (double)f + seed_add;
        }
        return dd ? (double)f + (double)dd * class_ulp(class_above(f)) : (double)f;
    };
    double a = value(rb - 1), b = value(rb);
    for (int i = 0; i < FRB; ++i) {
        const int64_t r = rb + i;
        if (r >= H) break;
        const double n = value(r + 1);
        const double al = dleft(a), ar = dright(a), bl = dleft(b), br = dright(b), nl = dleft(n), nr = dright(n);
        if (mine) {
            const double dv = (double)dem[r * W + c];
            double want = dv;
            if (r > 0 && r < H - 1 && inner_col) {
                const double md = fmin(fmin(al, ar), fmin(nl, nr)) + dg;
                const double me = fmin(fmin(a, bl), fmin(br, n)) + sh;
                want = fmax(fmin(md, me), dv);
            }
            const bool halo = (r == 0 && fixed_top) || (r == H - 1 && fixed_bot);   // the neighbouring band checks its own rows
            bad += (b == want || halo) ? 0u : 1u;                                    // NaN anywhere fails too
            G[r * W + c] = b;
            if (D8) {
                unsigned code;
                if (r == 0 || r == H - 1 || c == 0 || c == W - 1) code = edge_code(r, c, H - 1, W - 1);
                else {
                    code = d8_code(b, a, ar, br, nr, n, nl, bl, al);
                    anynodir |= code == 8u ? 1u : 0u;      // an interior cell without a downslope neighbour (d8.hip: the watersheds' fast-path test)
                }
                flowdir[r * W + c] = (uint8_t)code;
            }
        }
        a = b;
        b = n;
    }
    if (unreached) atomicAdd(&counters[C_UNREACHED], (unsigned long long)unreached);
    if (bad) atomicAdd(&counters[C_MISMATCH], (unsigned long long)bad);
    if (D8 && __any(anynodir != 0u) && lane == 0) *nodir = 1u;      // ("none" / "some": a plain store of the same value)
}

// rn(eps / 2**(E - 52)) for the binade class e (E = e - 127) with eps = M * 2**q exactly; 0: no integer weight in (0, 2**28)
uint32_t class_weight(double eps, int e)
{
    if (!(eps > 0.0) || std::isinf(eps)) return 0;
    int q;
    const double fr = std::frexp(eps, &q);                       // eps = fr * 2**q, fr in [0.5, 1)
    const uint64_t M = (uint64_t)std::ldexp(fr, 53);             // 53-bit integer, eps = M * 2**(q - 53)
    const int k = (e - 127 - 52) - (q - 53);                     // eps / u = M >> k
    if (k <= 24 || k > 52) return 0;                             // >= 2**28, or below one ulp
    const uint64_t half = 1ull << (k - 1), rem = M & ((1ull << k) - 1);
    if (rem == half) return 0;                                   // a tie: the rounding would depend on the running value
    const uint64_t w = (M >> k) + (rem > half ? 1 : 0);
    return (w >= 1 && w < WMAX) ? (uint32_t)w : 0;
}

}  // namespace

// ---- the resumable run (a row band refreshes its halo rows between batches) --------------------------------------------
struct GeoRun::Impl {
    DevBuf ws;
    uint32_t *d_tab = nullptr, *d_any = nullptr, *d_hdr = nullptr, *d_blk = nullptr;
    unsigned long long *d_cnt = nullptr;
    uint8_t *d_mark = nullptr;
    int *d_list = nullptr, *d_list2 = nullptr, *d_amark = nullptr;
    int ntr = 0, ntc = 0, round = 0, used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> span;   // around every batch of ng_round_kernel launches
    int span_rounds = 0;
    uint32_t last_count = 0xffffffffu;   // tiles of the last round the host has seen
    bool light = false;          // the rounds from light_from on append to the next round's list themselves
    int light_from = 0;
    int64_t nt = 0;
    size_t lds = 0;
    int maxcyc = NG_MAXCYC;
    bool debug = false;
    unsigned long long irregular = 0;
};
namespace {
constexpr int MAXR = 8192, BATCH = 32, BATCH_HEAD = 16;   // rounds per host read-back: in the tail (a read-back is ~30 us of idle device) / before it
constexpr uint32_t LIGHT_TILES = 6144;     // a round of at most this many tiles appends to the next list (one atomic per wake)
__global__ void ng_mark_row_kernel(uint8_t *mark, int ti, int ntc)
{
    const int tj = blockIdx.x * blockDim.x + threadIdx.x;
    if (tj < ntc) mark[(int64_t)ti * ntc + tj] = 1;
}
}  // namespace

GeoRun::GeoRun() : impl(new Impl) {}
GeoRun::~GeoRun()
{
    for (auto &e : impl->span) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete impl;
}

int GeoRun::launch_rounds(hipStream_t s, int nb)
{
    {   // HIP events around the batch: the device time of the round loop (bench.py: the stage's dominant kernel)
        Impl &mm = *impl;
        hipEvent_t e0, e1;
        MH_HIP(hipEventCreate(&e0));
        MH_HIP(hipEventCreate(&e1));
        mm.span.emplace_back(e0, e1);
        MH_HIP(hipEventRecord(e0, s));
    }
    struct CloseSpan {
        Impl &m;
        hipStream_t s;
        ~CloseSpan() { (void)hipEventRecord(m.span.back().second, s); }
    } close_span{*impl, s};
    Impl &m = *impl;
    GeoArgs a;
    a.H = H; a.W = W; a.ntr = m.ntr; a.ntc = m.ntc; a.nt = (int)m.nt; a.F = filled; a.d = dist; a.tab = m.d_tab; a.counters = m.d_cnt;
    a.blk = m.d_blk; a.hdr = m.d_hdr; a.mark = m.d_mark; a.list = m.d_list; a.fixed_top = fixed_top; a.fixed_bot = fixed_bot;
    a.append = 0; a.round_next = 0; a.amark = m.d_amark; a.list_next = nullptr; a.count_next = nullptr;
    const unsigned grid = (unsigned)std::min<int64_t>((m.nt + 3) / 4, 512);
    for (int k = 0; k < nb; ++k, ++m.round) {
        a.maxcyc = m.maxcyc;
        a.count = m.d_any + m.round;     // tiles of this round (round 0: every tile)
        if (!m.round) {
            hipLaunchKernelGGL(ng_first_kernel, dim3(grid), dim3(256), m.lds, s, a);
            (void)hipEventRecord(m.span.back().first, s);      // (the span counts the rounds after the classification)
            continue;
        }
        // the marks of the last mark-mode round become the list of this one
        if (!m.light || m.round == m.light_from)
            hipLaunchKernelGGL(ng_compact_kernel, dim3((unsigned)(((m.nt + 7) / 8 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<unsigned long long *>(m.d_mark),
                               (int)((m.nt + 7) / 8), m.d_list, m.d_any + m.round);
        if (m.light) {
            const bool odd = ((m.round - m.light_from) & 1) != 0;
            a.append = 1;
            a.round_next = m.round + 1;
            a.list = odd ? m.d_list2 : m.d_list;
            a.list_next = odd ? m.d_list : m.d_list2;
            a.count_next = m.d_any + m.round + 1;
        }
        if (m.light && tail_hook)      // (the label branch runs beside these rounds: one wavefront per workgroup, see round_body)
            hipLaunchKernelGGL(ng_round_w1_kernel, dim3((unsigned)std::min<int64_t>(m.nt, 2048)), dim3(64), (WN * (WN + 1) + 512) * sizeof(uint32_t), s, a);
        else
            hipLaunchKernelGGL(ng_round_kernel, dim3(grid), dim3(256), m.lds, s, a);
    }
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

// The first round: every tile is classified and relaxed once.  *applicable == false: the raster holds flat cells of a level
// without integer weights (or NaNs, or the epsilons give no weights at all): run the float64 relaxation instead.
int GeoRun::begin(hipStream_t s, bool *applicable, bool *active)
{
    Impl &m = *impl;
    static const bool off = [] { const char *e = dev_env("MHIP_NOFLAT"); return e && std::string(e) == "iterative"; }();
    m.debug = dev_env("MHIP_NG_DEBUG") != nullptr;
    *applicable = false;
    *active = false;
    if (off || !filled || H < 3 || W < 3) return MHIP_OK;
    if (!(sh > 0.0) || !(dg > 0.0) || std::isinf(sh) || std::isinf(dg)) return MHIP_OK;
    std::vector<uint32_t> tab(512, 0u);
    bool any_class = false;
    for (int e = 0; e < 255; ++e) {
        const uint32_t S = class_weight(sh, e), G = class_weight(dg, e);
        if (S && G) {
            tab[e] = S;
            tab[256 + e] = G;
            any_class = true;
        }
    }
    if (!any_class) return MHIP_OK;

    const int64_t n = H * W;
    m.ntr = (int)((H - 2 + TI - 1) / TI);
    m.ntc = (int)((W - 2 + TI - 1) / TI);
    m.nt = (int64_t)m.ntr * m.ntc;
    auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t o_d = 0, o_tab = align(o_d + (dist ? 0 : 4 * (size_t)n)), o_cnt = align(o_tab + 2048);
    const size_t o_any = align(o_cnt + 8 * (C_STATS + 128 + 16)), o_mark = align(o_any + 4 * (size_t)(MAXR + BATCH));
    const size_t o_amark = align(o_mark + (size_t)m.nt + 8), o_list = align(o_amark + 4 * (size_t)m.nt), o_list2 = align(o_list + 4 * (size_t)m.nt);
    const size_t o_hdr = align(o_list2 + 4 * (size_t)m.nt), o_blk = align(o_hdr + 4 * (size_t)m.nt);
    MH_TRY(m.ws.alloc(o_blk + 4 * (size_t)m.nt * WN * WN + 256));
    char *b = m.ws.as<char>();
    if (!dist) dist = reinterpret_cast<uint32_t *>(b + o_d);
    m.d_tab = reinterpret_cast<uint32_t *>(b + o_tab);
    m.d_cnt = reinterpret_cast<unsigned long long *>(b + o_cnt);
    m.d_any = reinterpret_cast<uint32_t *>(b + o_any);
    m.d_mark = reinterpret_cast<uint8_t *>(b + o_mark);
    m.d_list = reinterpret_cast<int *>(b + o_list);
    m.d_list2 = reinterpret_cast<int *>(b + o_list2);
    m.d_amark = reinterpret_cast<int *>(b + o_amark);
    m.d_hdr = reinterpret_cast<uint32_t *>(b + o_hdr);
    m.d_blk = reinterpret_cast<uint32_t *>(b + o_blk);
    MH_HIP(hipMemsetAsync(b + o_cnt, 0, o_list - o_cnt, s));
    MH_HIP(hipMemcpyAsync(m.d_tab, tab.data(), 2048, hipMemcpyHostToDevice, s));
    // a band's halo rows belong to the neighbour: "not reached yet" (an upper bound) until the first exchange brings its values
    if (fixed_top) MH_HIP(hipMemsetAsync(dist, 0xE0, 4 * (size_t)W, s));
    if (fixed_bot) MH_HIP(hipMemsetAsync(dist + (H - 1) * W, 0xE0, 4 * (size_t)W, s));

    m.lds = (4 * WN * (WN + 1) + 512) * sizeof(uint32_t);
    {
        static std::mutex mu;
        static bool attr_done[64] = {};
        int dev = 0;
        MH_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            MH_HIP(hipFuncSetAttribute((const void *)ng_round_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m.lds));
            MH_HIP(hipFuncSetAttribute((const void *)ng_first_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m.lds));
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    m.round = 0;
    m.last_count = 0xffffffffu;
    m.used = 0;
    MH_TRY(launch_rounds(s, 1));
    unsigned long long h_c[4] = {0, 0, 0, 0};
    MH_HIP(hipMemcpyAsync(h_c, m.d_cnt, sizeof(h_c), hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));   // (the table upload is complete as well: `tab` may go)
    m.used = 1;
    m.irregular = h_c[C_IRREGULAR];
    if (h_c[C_FATAL] || (m.irregular && !allow_partial)) {
        if (m.debug) fprintf(stderr, "[noflat geodesic] %llu irregular flat cells, %llu NaN cells: float64 relaxation\n", m.irregular, h_c[C_FATAL]);
        return MHIP_OK;
    }
    *applicable = true;
    *active = true;
    return MHIP_OK;
}

// rounds until nothing is marked any more (locally converged: *active = false on return)
int GeoRun::batch(hipStream_t s, bool *active)
{
    Impl &m = *impl;
    // rounds per host read-back; MHIP_NG_BATCH (tests): a small batch makes small rasters reach the self-listing tail rounds too
    const char *eb = dev_env("MHIP_NG_BATCH");
    const int nb_env = eb && atoi(eb) >= 1 && atoi(eb) <= BATCH ? atoi(eb) : 0;
    for (;;) {
        // rounds per host read-back.  A batch is launched blind: the rounds behind the one that finds no tile are no-ops of ~2 us with
        // ~10 us of dispatch gap each -- 28 of them at the end of the 16384^2 benchmark with batches of 32 (round 84 of 112), 0.3 ms.
        // The tile counts fall slowly towards the end (... 531 363 186 100 49 24 9 2): a short batch once the last count is small
        const int nb = nb_env ? nb_env : (!m.light ? BATCH_HEAD : (m.last_count <= 64 ? 8 : (m.last_count <= 512 ? 16 : BATCH)));
        if (m.round + BATCH > MAXR) {
            set_error("no-flats fill (geodesic) did not converge within %d rounds", MAXR);
            return MHIP_ENOTCONV;
        }
        MH_TRY(launch_rounds(s, nb));
        if (tail_hook) tail_hook->fire(s);      // the throughput-bound rounds are behind this point of the stream
        uint32_t h_any[BATCH];
        MH_HIP(hipMemcpyAsync(h_any, m.d_any + (m.round - nb), sizeof(uint32_t) * nb, hipMemcpyDeviceToHost, s));
        MH_HIP(stream_sync(s));
        for (int k = 0; k < nb; ++k) {
            if (!h_any[k]) {   // that round found no marked tile: so did the later launches of the batch
                *active = false;
                m.light = false;   // (a band's halo exchange wakes tiles through the mark bytes: the next round compacts them)
                return MHIP_OK;
            }
            ++m.used;
        }
        m.last_count = h_any[nb - 1];
        if (!m.light && h_any[nb - 1] <= LIGHT_TILES) {   // the tail: from here on the rounds build their lists themselves
            m.light = true;
            m.light_from = m.round;
        }
    }
}

int GeoRun::halo_changed(int side, hipStream_t s)
{
    Impl &m = *impl;
    hipLaunchKernelGGL(ng_mark_row_kernel, dim3((unsigned)((m.ntc + 255) / 256)), dim3(256), 0, s, m.d_mark, side == 0 ? 0 : m.ntr - 1, m.ntc);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

// G = F + u * D over the whole local raster, then the reference's equation at every owned cell.  *ok == false: a cell failed.
int GeoRun::end(hipStream_t s, bool *ok, FillStats *st)
{
    Impl &m = *impl;
    const int64_t n = H * W;
    partial = m.irregular != 0 && allow_partial;
    if (dev_env("MHIP_NG_CORRUPT") && W > 8)   // test hook (tests/test_gpu_noflat_geodesic.py): wrong distances in the middle row -- the check
        MH_HIP(hipMemsetAsync(dist + (H / 2) * W + 1, 0x01, 4 * (size_t)(W - 2), s));   // below has to catch them and send the raster to the relaxation
    d8_done = false;
    const bool with_d8 = !partial && d8_out && d8_nodir && !fixed_top && !fixed_bot;
    if (with_d8)
        hipLaunchKernelGGL(ng_finish_kernel<true>, dim3((unsigned)((W + 4 * TI - 1) / (4 * TI)), (unsigned)((H + FRB - 1) / FRB)), dim3(256), 0, s, filled, dist, dem,
                           out, H, W, sh, dg, seed_add, fixed_top, fixed_bot, m.d_cnt, d8_out, d8_nodir);
    else if (!partial)
        hipLaunchKernelGGL(ng_finish_kernel<false>, dim3((unsigned)((W + 4 * TI - 1) / (4 * TI)), (unsigned)((H + FRB - 1) / FRB)), dim3(256), 0, s, filled, dist, dem,
                           out, H, W, sh, dg, seed_add, fixed_top, fixed_bot, m.d_cnt, (uint8_t *)nullptr, (unsigned int *)nullptr);
    else              // (a partial surface is checked by the caller once the relaxation has settled the irregular flats)
        hipLaunchKernelGGL(ng_assemble_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, filled, dist, out, n, seed_add, m.d_cnt);
    MH_HIP(hipGetLastError());
    unsigned long long h_all[C_STATS + 128 + 16];
    MH_HIP(hipMemcpyAsync(h_all, m.d_cnt, sizeof(h_all), hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    if (allow_partial && h_all[C_UNREACHED]) partial = true;      // (the verification above then reported those cells: ignored)
    if (st) {
        *st = FillStats();
        st->geo_irregular = (int64_t)m.irregular;
        st->geo_unreached = (int64_t)h_all[C_UNREACHED];
        st->geo_mismatch = (int64_t)h_all[C_MISMATCH];
        st->rounds = m.used;
        st->tiles = m.nt;
        st->algorithm = partial ? 3 : 2;
        for (auto &e : m.span) {     // (the stream has been synchronised above)
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) st->hot_ms += ms;
            else (void)hipGetLastError();
        }
        st->hot_launches = m.used > 0 ? m.used - 1 : 0;      // (round 0 is ng_first_kernel)
        for (int k = 0; k < 64; ++k) {
            st->visits += (int64_t)h_all[C_STATS + 2 * k];
            st->cycles += (int64_t)h_all[C_STATS + 2 * k + 1];
        }
    }
    if (m.debug) {
        std::vector<uint32_t> cnt((size_t)m.round);
        MH_HIP(hipMemcpy(cnt.data(), m.d_any, 4 * (size_t)m.round, hipMemcpyDeviceToHost));
        unsigned long long v = 0, cy = 0;
        for (int k = 0; k < 64; ++k) {
            v += h_all[C_STATS + 2 * k];
            cy += h_all[C_STATS + 2 * k + 1];
        }
        fprintf(stderr, "[noflat geodesic] tiles %lld, visits %llu, cycles %llu; tiles per round:", (long long)m.nt, v, cy);
        for (int k = 1; k < m.round && k < 200; ++k) fprintf(stderr, " %u", cnt[k]);
#ifdef NG_PROFILE
        for (int o = 0; o < 2; ++o) {
            const unsigned long long *q = h_all + C_STATS + 128 + 8 * o;
            if (q[0]) fprintf(stderr, "\n[noflat visits, %s] %llu: ticks per visit: setup+issue %.0f, wait for loads %.0f, relax %.0f, store %.0f", o ? "open" : "masked", q[0],
                              (double)q[1] / q[0], (double)q[2] / q[0], (double)q[3] / q[0], (double)q[4] / q[0]);
        }
#endif
        fprintf(stderr, "\n[noflat geodesic] %lld x %lld: rounds %d, unreached %llu, mismatches %llu\n", (long long)H, (long long)W, m.used, h_all[C_UNREACHED],
                h_all[C_MISMATCH]);
    }
    *ok = partial ? true : !(h_all[C_UNREACHED] || h_all[C_MISMATCH]);
    d8_done = with_d8 && !partial && *ok;        // (a surface that still goes through the relaxation gets its directions from d8.hip)
    m.ws.release();
    return MHIP_OK;
}

// the reference's equation at every cell of d_out (the check of GeoRun::end as a function of its own)
int noflat_verify_dev(const float *d_dem, const double *d_out, int64_t H, int64_t W, double sh, double dg, hipStream_t s, bool *ok, int fixed_top,
                      int fixed_bot)
{
    DevBuf cnt;
    MH_TRY(cnt.alloc(8 * (C_STATS + 128 + 16)));
    MH_HIP(hipMemsetAsync(cnt.p, 0, 8 * (C_STATS + 128), s));
    hipLaunchKernelGGL(ng_verify_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)((H + VRB - 1) / VRB)), dim3(256), 0, s, d_dem, d_out, H, W, sh, dg, fixed_top,
                       fixed_bot, cnt.as<unsigned long long>());
    MH_HIP(hipGetLastError());
    unsigned long long bad = 0;
    MH_HIP(hipMemcpyAsync(&bad, cnt.as<unsigned long long>() + C_MISMATCH, 8, hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    *ok = bad == 0;
    return MHIP_OK;
}

// MHIP_OK: d_out holds the no-flats surface, verified -- or, with *partial, exact everywhere but on the flats of irregular levels,
// which hold an upper bound (F + seed_add): the caller relaxes those in float64 and verifies.  MHIP_ELIMIT: not applicable to
// this raster (NaN cells, epsilons without weights, a cell the verification rejects): the caller runs the float64 relaxation.
int fill_noflat_geodesic_dev(const float *d_dem, const float *d_filled, double *d_out, int64_t H, int64_t W, double sh, double dg, double seed_add,
                             hipStream_t s, FillStats *st, bool *partial, StageHook *tail_hook, D8Sink *d8)
{
    GeoRun g;
    g.tail_hook = tail_hook;
    if (d8) { g.d8_out = d8->flowdir; g.d8_nodir = d8->nodir; d8->done = false; }
    g.dem = d_dem; g.filled = d_filled; g.out = d_out; g.H = H; g.W = W; g.sh = sh; g.dg = dg;
    g.allow_partial = partial != nullptr && seed_add == seed_add && seed_add < 1e300;
    g.seed_add = seed_add;
    bool applicable = false, active = false, ok = false;
    MH_TRY(g.begin(s, &applicable, &active));
    if (!applicable) {
        if (st) st->geo_reject = 1;
        return MHIP_ELIMIT;
    }
    while (active) MH_TRY(g.batch(s, &active));
    MH_TRY(g.end(s, &ok, st));
    if (partial) *partial = g.partial;
    if (d8) d8->done = g.d8_done;
    if (!ok && st) st->geo_reject = 2;
    return ok ? MHIP_OK : MHIP_ELIMIT;
}

}  // namespace mh
