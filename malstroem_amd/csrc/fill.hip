// fill.hip -- depression filling on gfx950: fill.fill_terrain (f32) and fill.fill_terrain_no_flats (f64).
//
// Reference semantics (fill.py:112-232, sweeps _fill.pyx:28-124): the greatest fixed point of
//     plain   : W = max(dtm, min(W, 8 neighbours))
//     no-flats: W = max(dtm, min(W, min4(diagonal nbrs) + diag, min4(edge nbrs) + short))
// on interior cells, border cells fixed to dtm, start from +inf.  Both operators are monotone and every
// intermediate state stays >= the fixed point, so ANY chaotic schedule of single-cell updates converges
// to the same bits as the reference's four-direction Gauss-Seidel raster sweeps (single IEEE adds, no
// FMA, no reassociation in the no-flats case).
//
// Device schedule ("tiled iterative sweep"):
//   * the raster is cut into 62x62-cell tiles; ONE WAVEFRONT owns a tile and holds its 64x64 window
//     (tile + 1-cell halo ring) entirely in VGPRs: register r = window row r, lane = window column.
//   * a tile visit runs row-sequential passes (down, up) where the vertical dependency is carried in
//     registers and the horizontal neighbours come from wave_shr/wave_shl DPP shifts; then the window is
//     transposed through a wave-private LDS scratch and the same two passes run along the other axis.
//     Cycles repeat until the tile is locally converged (or a cap is hit).
//   * changed tiles write back and raise "active" flags of the neighbours whose halo they touched; the
//     host launches rounds until no flag is raised (flags are double buffered; a round is one launch).
// Halo lanes/rows stay fixed because their "dem" register is set to their own value (plain) or because
// the update is lane-masked (no-flats).  Raster border cells have W == dem from the start and therefore
// never move; cells outside the raster are +inf and never win a min.
#include "common.hpp"
#include <mutex>
#include <vector>

namespace mh {

namespace {

constexpr int WN = 64;  // window edge (tile + halo)
constexpr int TI = 62;  // tile interior edge
#ifndef MH_MAXCYC
#define MH_MAXCYC 1   // measured on the f64 kernel at 16384^2: 1 -> 18.6 ms, 2 -> 21.4 ms, 3 -> 23.7 ms (a second local
#endif                // cycle mostly re-verifies; a visit that changed something re-queues its tile instead)
#ifndef MH_ROW_BARRIER
#define MH_ROW_BARRIER 1
#endif
constexpr int DPP_WF_SL1 = 0x130;  // lane i <- lane i+1
constexpr int DPP_WF_SR1 = 0x138;  // lane i <- lane i-1

__device__ __forceinline__ float from_left(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WF_SR1, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WF_SL1, 0xf, 0xf, true));
}
__device__ __forceinline__ double from_left(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, DPP_WF_SR1, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, DPP_WF_SR1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, DPP_WF_SL1, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, DPP_WF_SL1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// This translation unit is built with -fno-honor-nans (see Makefile): NaNs are replaced by +inf when a window is loaded,
// so the min/max chains never see one and the compiler can drop the canonicalising v_max it would otherwise put in front
// of every fmin/fmax (measured: no-flats stage 44 -> 27 ms).  NaN tests therefore look at the bits.
__device__ __forceinline__ bool is_nan_bits(float v) { return (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u; }
__device__ __forceinline__ bool is_nan_bits(double v)
{
    return ((unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
}

// ---- one row-sequential pass over the register window -------------------------------------------
// A pass walks the rows in order (DOWN: 1..62, else 62..1).  For the row being updated the three rows
// "behind" (already updated in this pass), "current" and "ahead" contribute; what a row contributes
// through its horizontal neighbours is computed once when the row enters the 3-row window and carried
// in registers: per row 2 fresh horizontal reductions (the row ahead, the updated row) instead of 3.
// Lanes 0 / 63 read a zero from the missing DPP source lane: they are halo lanes whose own results are
// thrown away by the lane mask, and nobody reads a reduction across lanes.
//   updmask: ballot of the lanes that hold an updatable column; `frozen`: the one window row (-1: none) that
//   must not move (a band's halo row inside the window).
// The select is an explicit v_cndmask on an SGPR-pair mask: left to itself the compiler wraps the row in
// an EXEC-masked branch (SALU + s_nop filler on the critical path of every row).
__device__ __forceinline__ float sel_mask(float if_clear, float if_set, uint64_t mask)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}
__device__ __forceinline__ double sel_mask(double if_clear, double if_set, uint64_t mask)
{
    int lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(if_clear)), "v"(__double2loint(if_set)), "s"(mask));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(if_clear)), "v"(__double2hiint(if_set)), "s"(mask));
    return __hiloint2double(hi, lo);
}
// min(x, left(x), right(x)) as two v_min_f32_dpp (the opaque asm keeps the compiler from forming a v_min3, which
// cannot carry a DPP modifier and would need two extra v_mov_dpp)
__device__ __forceinline__ float hmin3(float x)
{
    float t = fminf(from_left(x), x);
    asm volatile("" : "+v"(t));
    return fminf(from_right(x), t);
}
__device__ __forceinline__ double hmin2(double x) { return fmin(from_left(x), from_right(x)); }

template <bool DOWN>
__device__ __forceinline__ void pass_plain(float (&w)[WN], const float (&d)[WN], uint64_t updmask, int frozen, uint64_t &any)
{
    constexpr int first = DOWN ? 1 : TI, dir = DOWN ? 1 : -1;
    asm volatile("" : "+s"(frozen));   // recompute the row masks in every pass: 62 live SGPR pairs would be spilled
    // new names for the rows: otherwise the horizontal reductions of the previous pass are kept alive (CSE) for the
    // rows that did not change since -- 62 extra live registers, i.e. spills inside the row loop
#pragma unroll
    for (int r = 0; r < WN; ++r) asm volatile("" : "+v"(w[r]));
    float h_behind = hmin3(w[first - dir]);
    float h_cur = hmin3(w[first]);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int r = first + dir * i;
        const float cu = w[r];
        const float h_ahead = hmin3(w[r + dir]);
        float nv = fmaxf(fminf(fminf(h_behind, h_cur), h_ahead), d[r]);
        nv = sel_mask(cu, nv, r == frozen ? 0ull : updmask);
        any |= __ballot(__float_as_uint(nv) != __float_as_uint(cu));
        w[r] = nv;
        h_behind = hmin3(nv);
        h_cur = h_ahead;
#if MH_ROW_BARRIER
        __builtin_amdgcn_sched_barrier(0);  // keep rows in program order: hoisted DPP shifts blow the VGPR budget
#endif
    }
}

// no-flats f64: min4(diagonals)+diag, min4(edges)+short, self; single IEEE adds (_fill.pyx:107-117).
template <bool DOWN>
__device__ __forceinline__ void pass_noflat(double (&w)[WN], const float (&d)[WN], uint64_t updmask, int frozen, double sh,
                                            double dg, uint64_t &any)
{
    constexpr int first = DOWN ? 1 : TI, dir = DOWN ? 1 : -1;
    asm volatile("" : "+s"(frozen));
#pragma unroll
    for (int r = 0; r < WN; ++r) asm volatile("" : "+v"(w[r]));   // see pass_plain
    double h_behind = hmin2(w[first - dir]);
    double h_cur = hmin2(w[first]);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int r = first + dir * i;
        const double cu = w[r], behind = w[r - dir], ahead = w[r + dir];
        const double h_ahead = hmin2(ahead);
        const double md = __dadd_rn(fmin(h_behind, h_ahead), dg);
        const double me = __dadd_rn(fmin(fmin(behind, ahead), h_cur), sh);
        double nv = fmax(fmin(fmin(md, me), cu), (double)d[r]);
        nv = sel_mask(cu, nv, r == frozen ? 0ull : updmask);
        any |= __ballot(nv != cu);
        w[r] = nv;
        h_behind = hmin2(nv);
        h_cur = h_ahead;
#if MH_ROW_BARRIER
        __builtin_amdgcn_sched_barrier(0);  // keep rows in program order: hoisted DPP shifts blow the VGPR budget
#endif
    }
}

// wave shifts that deliver `fill` to the lane without a source lane (lane 0 / lane 63).  NOTE: a DPP read of a lane
// that is disabled in EXEC returns 0 (bound_ctrl) -- never put a wave shift inside a divergent `?:` or `if`.
__device__ __forceinline__ float from_left_or(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), DPP_WF_SR1, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_right_or(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), DPP_WF_SL1, 0xf, 0xf, false));
}
__device__ __forceinline__ double from_left_or(double v, double fill)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), DPP_WF_SR1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), DPP_WF_SR1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right_or(double v, double fill)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), DPP_WF_SL1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), DPP_WF_SL1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// ---- halo probe: which cells of the halo rows 0 / 63 (they belong to neighbouring tiles) would drop if their owner
// revisited them now, given this tile's interior rows 1 / 62?  Only those neighbours need to be re-queued.  The halo
// values in the window may be stale, i.e. too high, so the test errs on the side of re-queueing.
__device__ __forceinline__ void probe_plain(const float (&w)[WN], const float (&d)[WN], bool upd, uint64_t &drop_first, uint64_t &drop_last)
{
    const float INF = __builtin_inff();
    const float a = upd ? w[1] : INF, b = upd ? w[TI] : INF;
    const float al = from_left_or(a, INF), ar = from_right_or(a, INF), bl = from_left_or(b, INF), br = from_right_or(b, INF);
    const float ma = fminf(fminf(al, ar), a), mb = fminf(fminf(bl, br), b);
    drop_first = __ballot(fmaxf(d[0], fminf(w[0], ma)) < w[0]);
    drop_last = __ballot(fmaxf(d[WN - 1], fminf(w[WN - 1], mb)) < w[WN - 1]);
}
__device__ __forceinline__ void probe_noflat(const double (&w)[WN], const float (&d)[WN], bool upd, double sh, double dg,
                                             uint64_t &drop_first, uint64_t &drop_last)
{
    const double INF = __builtin_inf();
    const double a = upd ? w[1] : INF, b = upd ? w[TI] : INF;
    const double al = from_left_or(a, INF), ar = from_right_or(a, INF), bl = from_left_or(b, INF), br = from_right_or(b, INF);
    const double ca = fmin(__dadd_rn(a, sh), __dadd_rn(fmin(al, ar), dg));
    const double cb = fmin(__dadd_rn(b, sh), __dadd_rn(fmin(bl, br), dg));
    drop_first = __ballot(fmax((double)d[0], fmin(w[0], ca)) < w[0]);
    drop_last = __ballot(fmax((double)d[WN - 1], fmin(w[WN - 1], cb)) < w[WN - 1]);
}

// ---- 64x64 transpose of 32-bit words through a wave-private LDS scratch [64][65] ------------------
__device__ __forceinline__ void transpose32(uint32_t (&x)[WN], uint32_t *scr, int lane)
{
#pragma unroll
    for (int r = 0; r < WN; ++r) scr[r * (WN + 1) + lane] = x[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int c = 0; c < WN; ++c) x[c] = scr[lane * (WN + 1) + c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void transpose(float (&x)[WN], uint32_t *scr, int lane)
{
    uint32_t t[WN];
#pragma unroll
    for (int r = 0; r < WN; ++r) t[r] = __float_as_uint(x[r]);
    transpose32(t, scr, lane);
#pragma unroll
    for (int r = 0; r < WN; ++r) x[r] = __uint_as_float(t[r]);
}
__device__ __forceinline__ void transpose(double (&x)[WN], uint32_t *scr, int lane)
{
    uint32_t lo[WN], hi[WN];
#pragma unroll
    for (int r = 0; r < WN; ++r) {
        lo[r] = (uint32_t)__double2loint(x[r]);
        hi[r] = (uint32_t)__double2hiint(x[r]);
    }
    transpose32(lo, scr, lane);
    transpose32(hi, scr, lane);
#pragma unroll
    for (int r = 0; r < WN; ++r) x[r] = __hiloint2double((int)hi[r], (int)lo[r]);
}

// columns 1 and TI of the window (lane = row) without a full transpose: rows go to the scratch, two strided reads back
__device__ __forceinline__ void extract_cols(const float (&x)[WN], uint32_t *scr, int lane, float &c1, float &c62)
{
#pragma unroll
    for (int r = 0; r < WN; ++r) scr[r * (WN + 1) + lane] = __float_as_uint(x[r]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    c1 = __uint_as_float(scr[lane * (WN + 1) + 1]);
    c62 = __uint_as_float(scr[lane * (WN + 1) + TI]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void extract_cols(const double (&x)[WN], uint32_t *scr, int lane, double &c1, double &c62)
{
    int lo1, lo62, hi1, hi62;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int r = 0; r < WN; ++r) scr[r * (WN + 1) + lane] = (uint32_t)(half ? __double2hiint(x[r]) : __double2loint(x[r]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int a1 = (int)scr[lane * (WN + 1) + 1], a62 = (int)scr[lane * (WN + 1) + TI];
        if (half) { hi1 = a1; hi62 = a62; } else { lo1 = a1; lo62 = a62; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    c1 = __hiloint2double(hi1, lo1);
    c62 = __hiloint2double(hi62, lo62);
}

template <typename WT> struct Inf;
template <> struct Inf<float> { static __device__ __forceinline__ float v() { return __builtin_inff(); } };
template <> struct Inf<double> { static __device__ __forceinline__ double v() { return __builtin_inf(); } };

// ---- one round: a resident grid of workgroups pulls the active MACRO TILES from sharded worklists ------------
// The scheduling unit is a macro tile = 2 x 2 tiles = one workgroup of four wavefronts, one tile each.  Inside a visit
// the four waves iterate: one local cycle each, then they publish their tiles' edge rows / columns in LDS, refresh
// the halos that face a sibling from there, and repeat until no sibling changed (or a cap): information crosses the
// macro tile at LDS speed and one HBM load / store / worklist round trip is amortised over several cycles.
// Round k consumes list[k&1][shard][..] (count[k][shard] entries) and appends the macro tiles whose halo it changed to
// list[(k+1)&1][shard'] where shard' = id % NSHARD; mark[][] de-duplicates appends (a macro tile clears its own mark
// when it is visited).  Block g starts pulling at shard g % NSHARD and walks the other shards when its own runs dry,
// so the work is balanced while every queue head / append counter only sees 1/NSHARD of the traffic (one shared
// word saturates at ~90 atomics per microsecond on this chip).  sum(count[k+1][*]) == 0  <=>  converged.
constexpr int NSHARD = 8;
constexpr int STAT_CHANGED = 140;   // stats word: visits (after the initialising round) that changed their tile
#ifndef MH_MAXIT
#define MH_MAXIT 2
#endif
constexpr int MAXIT = MH_MAXIT;   // sibling-exchange iterations per macro visit (measured: 2 is best for the f32 fill)
constexpr int MAXCYC = MH_MAXCYC; // local cycles per visit of a single tile (MT == 1)
enum { E_ROW1 = 0, E_ROW62 = 1, E_COL1 = 2, E_COL62 = 3 };
struct RoundArgs {
    int64_t H, Wd;
    int ntr, ntc;               // tiles
    int mtr, mtc;               // macro tiles
    int shard_cap;              // list capacity per shard
    int *list_cur, *list_nxt;   // [NSHARD][shard_cap]
    unsigned int *mark_cur, *mark_nxt;
    unsigned int *count_cur, *count_nxt, *head;   // [NSHARD] each, per round
    double sh, dg, seed_add;
    unsigned long long *stats;  // [64][2] sharded {visits, local cycles}
    // row-band mode: local row 0 / H-1 is a HALO row owned by the neighbouring band: never updated here, not a raster
    // border (its start value is +inf or the seed, not dem); the host refreshes it between rounds
    int fixed_top, fixed_bot;
};
enum { INIT_NONE = 0, INIT_INF = 1, INIT_SEED = 2 };

// Development aid (-DMH_PROFILE_VISIT): per-wave time stamps around the phases of a visit, summed in registers and
// added to stats[128..] when the wave leaves; finish() prints the averages.
#ifdef MH_PROFILE_VISIT
#define MH_STAMP(var) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long var = __builtin_amdgcn_s_memtime()
struct Prof { long long load, pass, tr, store, push; };   // per-lane (VGPR) accumulators: the kernels have no SGPRs to spare
#define MH_PROF_ARG , Prof &pf
#define MH_PROF_PASS , pf
#else
#define MH_STAMP(var)
#define MH_PROF_ARG
#define MH_PROF_PASS
#endif

// One macro-tile visit; called by all four waves of the block (it contains block barriers).  `X`: the exchange area
// [2 parities][4 waves][4 edges][64] of WT in LDS.
// MT == 2: as described above.  MT == 1: the scheduling unit is a single tile and a single wavefront (no barriers, no
// exchange, up to MAXCYC local cycles) -- the f64 kernel runs one wave per SIMD, where a block that waits for its slowest
// sibling leaves whole SIMDs idle (measured: no-flats 20.4 ms with MT == 1, 24.5-31 ms with MT == 2).
template <typename WT, bool NOFLAT, int INIT, int MT>
__device__ __forceinline__ void visit_macro(const RoundArgs &a, const float *__restrict__ dem, const float *__restrict__ seed,
                                            WT *__restrict__ W, int macro, unsigned flags, uint32_t *scr, WT *X, int lane, int wave,
                                            unsigned &visits, unsigned &cycles MH_PROF_ARG)
{
    MH_STAMP(tp0);
    const int64_t H = a.H, Wd = a.Wd;
    const int ntr = a.ntr, ntc = a.ntc;
    constexpr bool first_round = INIT != INIT_NONE;  // compile time: keeps the 128 row loads straight-line
    const int mi = macro / a.mtc, mj = macro - mi * a.mtc;
    constexpr bool MACRO = MT == 2;
    const int qi = MACRO ? wave >> 1 : 0, qj = MACRO ? wave & 1 : 0;
    const int ti = MT * mi + qi, tj = MT * mj + qj;
    const bool has_tile = !MACRO || (ti < ntr && tj < ntc);   // wave-uniform; a wave without a tile only keeps the barriers company
    const int64_t r0 = (int64_t)ti * TI, c0 = (int64_t)tj * TI;  // raster coords of window (0,0)
    const int64_t cc = c0 + lane;
    const bool col_in = cc < Wd;
    // lane predicates are written with bitwise & : a short-circuit && on per-lane values becomes a divergent branch, and a
    // divergent region is where this compiler may drop a spill of a full-wave value (tools/lint_exec_spills.py)
    const bool upd = (lane >= 1) & (lane <= TI);
    const WT INF = Inf<WT>::v();

    // ---- stage-in: all 128 row loads are issued back to back (scalar row base + one shared per-lane offset keeps
    // them off the VGPR address budget); a visit's latency is dominated by this one round trip to HBM/L2.
    WT w[WN];
    float d[WN];
    if (has_tile) {
        // Unconditional loads from clamped (always in-raster) addresses + selects: no exec-masked branches, so all
        // 128 loads are in flight together.  Row pointers advance incrementally in SGPRs (opaque asm keeps the
        // compiler from hoisting 64 row offsets out of the persistent tile loop and spilling them).
        const int lane_c = (int)(cc < Wd ? lane : Wd - 1 - c0);
        typedef const float __attribute__((address_space(1))) *gfp;   // keep "global" through the opaque asm
        typedef const WT __attribute__((address_space(1))) *gwp;
        gfp dp = (gfp)(dem + (r0 * Wd + c0));
        gfp sp = (gfp)(seed + (r0 * Wd + c0));
        gwp wp = (gwp)(W + (r0 * Wd + c0));
        const bool col_border = (cc == 0) || (cc == Wd - 1);
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            const int64_t rr = r0 + r;
            const bool in = col_in & (rr < H);
            float dv = dp[lane_c];
            WT wv;
            if constexpr (INIT == INIT_NONE) {
                wv = wp[lane_c];
                wv = in ? wv : INF;
            } else {
                const bool border = (rr == 0 && !a.fixed_top) || (rr == H - 1 && !a.fixed_bot) || col_border;
                if constexpr (INIT == INIT_SEED) {  // rigorous upper bound of the fixed point: see fill_noflat_dev
                    const float fv = sp[lane_c];
                    wv = in ? (border ? (WT)dv : (WT)((double)fv + a.seed_add)) : INF;
                } else {
                    wv = (in && border) ? (WT)dv : INF;  // fill.py:102-109 _initialize_filled
                }
            }
            dv = in ? dv : __builtin_inff();
            if (rr + 1 < H) {  // wave-uniform: stay on the last raster row once we run off the bottom
                dp += Wd;
                sp += Wd;
                wp += Wd;
            }
            asm volatile("" : "+s"(dp), "+s"(sp), "+s"(wp));
            // NaN handling: a NaN neighbour never wins `a <= b` in the reference (_fill.pyx:22) and a NaN dem cell
            // is never updated (`fv > NaN` is false): both behave like +inf inside the window.
            if (is_nan_bits(dv)) dv = __builtin_inff();
            if (is_nan_bits(wv)) wv = INF;
            w[r] = wv;
            d[r] = dv;
        }
    } else {
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            w[r] = INF;
            d[r] = __builtin_inff();
        }
    }

    MH_STAMP(tp1);
    // ---- local solve: (down, up) passes in the row layout, transpose, (down, up) = (right, left), transpose
    // window rows that may be updated (all but a band's halo rows); after the transpose rows <-> lanes swap roles
    uint64_t rowok = ~0ull;
    if (a.fixed_top && r0 == 0) rowok &= ~1ull;
    if (a.fixed_bot && H - 1 - r0 < WN) rowok &= ~(1ull << (H - 1 - r0));
    const bool upd_t = upd & (((rowok >> lane) & 1ull) != 0);
    const uint64_t updmask = has_tile ? __ballot(upd) : 0ull, updmask_t = has_tile ? __ballot(upd_t) : 0ull;
    // rows 0 / 63 are never updated by a pass; a frozen row in between can only be a band's bottom halo row
    const int frozen = (a.fixed_bot && H - 1 - r0 < WN - 1) ? (int)(H - 1 - r0) : -1;
    uint64_t anyAll = 0, topN = 0, botN = 0, leftT = 0, rightT = 0;  // change mask; halo probes
    bool capped = true;   // block-uniform: the sibling exchange was cut off while something still moved
    int ncyc = 0;
    auto edge = [&](int parity, int wv, int e) { return X + ((parity * 4 + wv) * 4 + e) * WN; };
    if (MACRO && !has_tile) {  // my published edges never win a min
        for (int e = 0; e < 4; ++e) {
            edge(0, wave, e)[lane] = INF;
            edge(1, wave, e)[lane] = INF;
        }
    }
    bool moved = true;   // did my previous cycle change anything?
#pragma nounroll
    for (int it = 0; it < (MACRO ? MAXIT : MAXCYC); ++it) {
        const int cur = it & 1, prev = cur ^ 1;
        uint64_t chg = 0;
        // A cycle can only change something if my last cycle did, or if a sibling's edge moved between its last two
        // publications (parity `cur` still holds the one before last).  The corner cells of the diagonal sibling
        // arrive through the halo lanes of those edges.
        bool need = has_tile;
        if (MACRO && it == 0) need = has_tile && ((flags >> wave) & 1u);   // nobody asked for this tile: its window is consistent
        if (MACRO && has_tile && it >= 2 && !moved) {
            const WT *v1 = edge(prev, qi ? wave - 2 : wave + 2, qi ? E_ROW62 : E_ROW1), *v0 = edge(cur, qi ? wave - 2 : wave + 2, qi ? E_ROW62 : E_ROW1);
            const WT *h1 = edge(prev, qj ? wave - 1 : wave + 1, qj ? E_COL62 : E_COL1), *h0 = edge(cur, qj ? wave - 1 : wave + 1, qj ? E_COL62 : E_COL1);
            need = __any(v1[lane] != v0[lane] || h1[lane] != h0[lane]) != 0;
        }
        if (MACRO) __builtin_amdgcn_s_barrier();   // everybody has compared parity `cur` before anybody overwrites it
        if (need) {
            ++ncyc;
            // the two halves share one copy of the pass code (the unrolled passes are most of the kernel: two copies
            // overflow the instruction cache, measured +10 % on the f64 kernel)
#pragma nounroll
            for (int half = 0; half < 2; ++half) {
                // halos that face a sibling: its edge as of the end of the previous iteration (min: never raise a cell);
                // half 0 = row layout / vertical sibling, half 1 = column layout / horizontal sibling
                if (MACRO && it > 0) {
                    const int q = half ? qj : qi, step = half ? 1 : 2;
                    if (q == 1) {
                        const WT x = edge(prev, wave - step, half ? E_COL62 : E_ROW62)[lane];
                        w[0] = x < w[0] ? x : w[0];
                    } else {
                        const WT x = edge(prev, wave + step, half ? E_COL1 : E_ROW1)[lane];
                        w[WN - 1] = x < w[WN - 1] ? x : w[WN - 1];
                    }
                }
                uint64_t any = 0, p0, p1;
                if constexpr (NOFLAT) {
                    pass_noflat<true>(w, d, half ? updmask_t : updmask, half ? -1 : frozen, a.sh, a.dg, any);
                    pass_noflat<false>(w, d, half ? updmask_t : updmask, half ? -1 : frozen, a.sh, a.dg, any);
                    probe_noflat(w, d, half ? upd_t : upd, a.sh, a.dg, p0, p1);
                } else {
                    pass_plain<true>(w, d, half ? updmask_t : updmask, half ? -1 : frozen, any);
                    pass_plain<false>(w, d, half ? updmask_t : updmask, half ? -1 : frozen, any);
                    probe_plain(w, d, half ? upd_t : upd, p0, p1);
                }
                chg |= any;
                if (half == 0) {
                    topN = p0;
                    botN = p1;
                } else {
                    // a band's halo rows are frozen here: their owner is the neighbouring band, nobody to re-queue locally
                    leftT = p0 & rowok;
                    rightT = p1 & rowok;
                    if (MACRO) {
                        edge(cur, wave, E_COL1)[lane] = w[1];
                        edge(cur, wave, E_COL62)[lane] = w[TI];
                    }
                }
                transpose(w, scr, lane);
                transpose(d, scr, lane);
                if (MACRO && half == 1) {
                    edge(cur, wave, E_ROW1)[lane] = w[1];
                    edge(cur, wave, E_ROW62)[lane] = w[TI];
                }
            }
            anyAll |= chg;
        } else if (MACRO && has_tile) {
            if (it == 0) {   // not asked for: publish the edges of the window as loaded
                WT c1, c62;
                extract_cols(w, scr, lane, c1, c62);
                edge(cur, wave, E_COL1)[lane] = c1;
                edge(cur, wave, E_COL62)[lane] = c62;
                edge(cur, wave, E_ROW1)[lane] = w[1];
                edge(cur, wave, E_ROW62)[lane] = w[TI];
            } else {
                for (int e = 0; e < 4; ++e) edge(cur, wave, e)[lane] = edge(prev, wave, e)[lane];   // unchanged
            }
        }
        moved = chg != 0;
        // one barrier per iteration: publishes parity `cur`, and nobody writes parity `prev` before everybody read it
        if (MACRO ? __syncthreads_or(chg != 0) == 0 : chg == 0) {
            capped = false;
            break;
        }
    }

    const bool changed = anyAll != 0;
    MH_STAMP(tp2);
    visits += has_tile ? 1 : 0;
    cycles += ncyc + ((changed & has_tile & !first_round) ? (1u << 20) : 0u);   // bits 20..: visits that changed their tile (certify())
    // ---- stage-out: interior cells that are not raster border cells (those never move)
    if (has_tile && (changed || first_round)) {
        const bool lane_ok = upd & col_in & (cc != 0) & (cc != Wd - 1);
        typedef WT __attribute__((address_space(1))) *gwsp;
        gwsp wst = (gwsp)(W + ((r0 + 1) * Wd + c0));
#pragma unroll
        for (int r = 1; r <= TI; ++r) {
            const int64_t rr = r0 + r;
            if (rr < H - 1 && lane_ok) wst[lane] = w[r];  // rr >= 1 always; H-1 is a border row
            wst += Wd;
            asm volatile("" : "+s"(wst));
        }
    }
    if (first_round && has_tile) {
        // raster border cells are written once, straight from dem, wherever they sit in the window (halo included);
        // a band's halo rows get their start value (+inf / seed) until the neighbour's first exchange arrives
#pragma unroll 2
        for (int r = 0; r < WN; ++r) {
            const int64_t rr = r0 + r;
            if (rr < H && col_in) {
                const bool colb = (cc == 0 || cc == Wd - 1);
                const bool fixed = (a.fixed_top && rr == 0) || (a.fixed_bot && rr == H - 1);
                if (colb || ((rr == 0 || rr == H - 1) && !fixed)) {
                    W[rr * Wd + cc] = (WT)dem[rr * Wd + cc];
                } else if (fixed) {
                    WT v = Inf<WT>::v();
                    if constexpr (INIT == INIT_SEED) v = (WT)((double)seed[rr * Wd + cc] + a.seed_add);
                    W[rr * Wd + cc] = v;
                }
            }
        }
    }
    MH_STAMP(tp3);
    if (has_tile) {
        // which neighbours would be lowered by this tile's current state?  (a pure function of the window, so it is
        // evaluated on every visit: a visit that changed nothing still re-queues a neighbour whose cells lag behind)
        // N layout masks are indexed by column, T layout by row.
        // Lane k (k < 9) owns neighbour tile k, so the returning atomics of all directions are in flight together.
        // topN/botN: halo-row cells (bit = column 0..63) that would drop; leftT/rightT: halo-column cells (bit = row).
        // Bits 1..62 belong to the edge neighbour, bits 0 / 63 are the corner cells of the diagonal neighbours.
        // A neighbour tile inside my own macro tile is a sibling.  Its probe bit counts like any other neighbour's: a sibling has seen
        // my EDGE cells unless the exchange was capped, but my corner cell reaches the DIAGONAL sibling only through the halo lanes
        // of the two edge siblings, one iteration later -- and the loop ends after ONE quiet iteration.  (Until round 3 a
        // sibling was re-queued only when capped: a corner that dropped in iteration 0 of a visit whose iteration 1 was quiet
        // never reached the diagonal sibling's corner cell -- the one cell of 1.07 G of DESIGN 4.2; tools/fill_protocol_model.py
        // reproduces it.)
        const uint64_t INNER = ((1ull << TI) - 1) << 1, C0 = 1ull, C63 = 1ull << (WN - 1);
        const bool top = (topN & INNER) != 0, bot = (botN & INNER) != 0, left = (leftT & INNER) != 0, right = (rightT & INNER) != 0;
        const bool tl = ((topN | leftT) & C0) != 0, tr = (topN & C63) || (rightT & C0);
        const bool bl = (botN & C0) || (leftT & C63), br = ((botN | rightT) & C63) != 0;
        const unsigned bits = (tl ? 1u : 0u) | (top ? 2u : 0u) | (tr ? 4u : 0u) | (left ? 8u : 0u) | (capped ? 16u : 0u) |
                              (right ? 32u : 0u) | (bl ? 64u : 0u) | (bot ? 128u : 0u) | (br ? 256u : 0u);
        if (lane < 9) {
            const int p = ti + lane / 3 - 1, q = tj + lane % 3 - 1;
            if (p >= 0 && p < ntr && q >= 0 && q < ntc) {
                const int t = MACRO ? (p >> 1) * a.mtc + (q >> 1) : p * ntc + q, sh = t % NSHARD;
                const bool want = (((bits >> lane) & 1u) != 0) | (t == macro && capped);
                // the mark word carries one bit per tile of the macro tile: which of them has a reason to run
                if (want && atomicOr(&a.mark_nxt[t], MACRO ? 1u << ((p & 1) * 2 + (q & 1)) : 1u) == 0u)
                    a.list_nxt[(size_t)sh * a.shard_cap + atomicAdd(&a.count_nxt[sh], 1u)] = t;
            }
        }
    }
#ifdef MH_PROFILE_VISIT
    MH_STAMP(tp4);
    pf.load += tp1 - tp0; pf.pass += tp2 - tp1; pf.store += tp3 - tp2; pf.push += tp4 - tp3;
#endif
}

template <typename WT, bool NOFLAT, int INIT, int MT>
__global__ __launch_bounds__(256, NOFLAT ? 1 : 2) void fill_round_kernel(RoundArgs a, const float *__restrict__ dem,
                                                                         const float *__restrict__ seed, WT *__restrict__ W)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ int s_idx, s_macro, s_shard;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform => scalar addressing
    uint32_t *scr = lds + wave * (WN * (WN + 1));
    WT *X = reinterpret_cast<WT *>(lds + 4 * (WN * (WN + 1)));           // [2][4][4][64] edge exchange
    const int nmt = a.mtr * a.mtc;
    const int g = (int)blockIdx.x;
    unsigned visits = 0, cycles = 0;
#ifdef MH_PROFILE_VISIT
    long long vz = 0;
    asm volatile("" : "+v"(vz));   // opaque per-lane zero
    Prof pf{vz, vz, vz, vz, vz};
    const long long tk0 = __builtin_amdgcn_s_memtime() + vz;
#endif
    if constexpr (MT == 1) {
        // single tiles: every wavefront schedules itself
        const int64_t gw = (int64_t)g * 4 + wave;
        if constexpr (INIT != INIT_NONE) {
            const int64_t nwaves = (int64_t)gridDim.x * 4;
            for (int64_t t = gw; t < nmt; t += nwaves)
                visit_macro<WT, NOFLAT, INIT, MT>(a, dem, seed, W, (int)t, 1u, scr, X, lane, wave, visits, cycles MH_PROF_PASS);
        } else {
            int sh = (int)(gw % NSHARD);
            for (int pass = 0; pass < 4; ++pass) {
                const unsigned int n = a.count_cur[sh];
                for (;;) {
                    unsigned int i = 0;
                    if (lane == 0) i = atomicAdd(&a.head[sh], 1u);
                    i = __builtin_amdgcn_readfirstlane(i);
                    if (i >= n) break;
                    const int tile = a.list_cur[(size_t)sh * a.shard_cap + i];
                    if (lane == 0) a.mark_cur[tile] = 0u;
                    visit_macro<WT, NOFLAT, INIT, MT>(a, dem, seed, W, tile, 1u, scr, X, lane, wave, visits, cycles MH_PROF_PASS);
                }
                bool more = false;
                if (lane < NSHARD) more = __hip_atomic_load(&a.head[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.count_cur[lane];
                const uint64_t left = __ballot(more);
                if (!left) break;
                int want = (int)(gw % __builtin_popcountll(left));   // spread the helpers over the shards that still have work
                uint64_t mm = left;
                while (want--) mm &= mm - 1;
                sh = __builtin_ctzll(mm);
            }
        }
    } else if constexpr (INIT != INIT_NONE) {
        // first round: every macro tile, statically strided (uniform work)
        for (int t = g; t < nmt; t += (int)gridDim.x) {
            unsigned flags = 0xfu;
            if constexpr (INIT == INIT_INF) {
                // a window without a raster border cell (nor a band halo row) starts all +inf and cannot move before a
                // sibling or a neighbour offers something finite: its first cycle is skipped (the window is still stored)
                flags = 0u;
                const int mi = t / a.mtc, mj = t - mi * a.mtc;
                for (int q = 0; q < 4; ++q) {
                    const int64_t r0 = (int64_t)(2 * mi + (q >> 1)) * TI, c0 = (int64_t)(2 * mj + (q & 1)) * TI;
                    const bool interior = r0 > 0 && r0 + WN - 1 < a.H - 1 && c0 > 0 && c0 + WN - 1 < a.Wd - 1;
                    if (!interior) flags |= 1u << q;
                }
            }
            visit_macro<WT, NOFLAT, INIT, MT>(a, dem, seed, W, t, flags, scr, X, lane, wave, visits, cycles MH_PROF_PASS);
            __syncthreads();   // the exchange area is reused by the next visit
        }
    } else {
        // own shard first; then one vector look at all (head, count) pairs picks the shards that still hold work
        // (a stale head only costs one wasted pop), so an idle block leaves after ~2 memory round trips
        int sh = g % NSHARD;
        for (int pass = 0; pass < 4; ++pass) {
            const unsigned int n = a.count_cur[sh];
            for (;;) {
                if (threadIdx.x == 0) {
                    const unsigned int i = atomicAdd(&a.head[sh], 1u);
                    int m = -1;
                    if (i < n) {
                        m = a.list_cur[(size_t)sh * a.shard_cap + i];
                        s_idx = (int)a.mark_cur[m];   // nobody else touches this round's marks
                        a.mark_cur[m] = 0u;
                    }
                    s_macro = m;
                }
                __syncthreads();
                const int m = s_macro;
                if (m < 0) break;
                visit_macro<WT, NOFLAT, INIT, MT>(a, dem, seed, W, m, (unsigned)s_idx, scr, X, lane, wave, visits, cycles MH_PROF_PASS);
                __syncthreads();   // s_macro and the exchange area are reused
            }
            if (wave == 0) {
                bool more = false;
                if (lane < NSHARD) more = __hip_atomic_load(&a.head[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.count_cur[lane];
                const uint64_t left = __ballot(more);
                int nxt = -1;
                if (left) {  // spread the helpers: pick the (g mod popcount)-th shard that still has work
                    int want = g % __builtin_popcountll(left);
                    uint64_t mm = left;
                    while (want--) mm &= mm - 1;
                    nxt = __builtin_ctzll(mm);
                }
                if (lane == 0) s_shard = nxt;
            }
            __syncthreads();
            sh = s_shard;
            __syncthreads();
            if (sh < 0) break;
        }
    }
    if (lane == 0 && visits) {
        unsigned long long *sh = a.stats + 2 * ((g * 4 + wave) & 63);
        atomicAdd(&sh[0], (unsigned long long)visits);
        atomicAdd(&sh[1], (unsigned long long)(cycles & 0xfffffu));
        if (cycles >> 20) atomicAdd(&a.stats[STAT_CHANGED], (unsigned long long)(cycles >> 20));
#ifdef MH_PROFILE_VISIT
        unsigned long long *pr = a.stats + 128;
        atomicAdd(&pr[0], (unsigned long long)pf.load); atomicAdd(&pr[1], (unsigned long long)pf.pass);
        atomicAdd(&pr[2], (unsigned long long)pf.tr); atomicAdd(&pr[3], (unsigned long long)pf.store);
        atomicAdd(&pr[4], (unsigned long long)pf.push); atomicAdd(&pr[5], (unsigned long long)visits);
        atomicAdd(&pr[6], (unsigned long long)(__builtin_amdgcn_s_memtime() - tk0));
#endif
    }
}

template <typename WT> __global__ void copy_dem_kernel(const float *dem, WT *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (WT)dem[i];
}

constexpr int MAX_ROUNDS = 1 << 15;
constexpr int MAX_BATCH = 32;   // rounds per host check: FillRun::rounds_per_batch (even, <= MAX_BATCH)

// marks every macro tile of macro row `ti` active for the round that is launched next (band mode: a halo row changed)
__global__ void activate_tile_row_kernel(int ti, int ntc, int shard_cap, int *list, unsigned int *mark, unsigned int *count)
{
    const int tj = blockIdx.x * blockDim.x + threadIdx.x;
    if (tj >= ntc) return;
    const int t = ti * ntc + tj, sh = t % NSHARD;
    if (atomicExch(&mark[t], 0xfu) == 0u) list[(size_t)sh * shard_cap + atomicAdd(&count[sh], 1u)] = t;
}

// marks every macro tile active for the round that is launched next (certify(): one full sweep over the raster)
__global__ void activate_all_kernel(int nt, int shard_cap, int *list, unsigned int *mark, unsigned int *count)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const int sh = t % NSHARD;
    if (atomicExch(&mark[t], 0xfu) == 0u) list[(size_t)sh * shard_cap + atomicAdd(&count[sh], 1u)] = t;
}

}  // namespace

// ---- resumable fill: begin() runs the initialising round, batch() runs BATCH rounds between host checks ---------
struct FillRun::Impl {
    DevBuf ws;
    int *lists = nullptr;
    unsigned int *marks = nullptr, *count = nullptr, *head = nullptr;
    unsigned long long *d_stats = nullptr;
    int ntr = 0, ntc = 0, mtr = 0, mtc = 0, shard_cap = 0, round = 0, rounds_used = 0;
    int64_t nt = 0;    // macro tiles (the scheduling unit)
    int64_t ntiles = 0;
    size_t list_elems = 0;
    bool trivial = false;  // no interior cell
};

FillRun::FillRun() : impl(new Impl) {}
FillRun::~FillRun() { delete impl; }

template <typename WT, bool NOFLAT>
static int fill_launch(FillRun &f, int round, int init, hipStream_t s)
{
    FillRun::Impl &m = *f.impl;
    constexpr int MT = NOFLAT ? 1 : 2;
    const size_t lds = 4 * WN * (WN + 1) * sizeof(uint32_t) + (MT == 2 ? 2 * 4 * 4 * WN * sizeof(WT) : 0);   // transposes + edge exchange
    const int resident_blocks = 256 * (NOFLAT ? 1 : 2);  // as many blocks as the chip holds at this kernel's occupancy
    auto k_none = fill_round_kernel<WT, NOFLAT, INIT_NONE, MT>;
    auto k_inf = fill_round_kernel<WT, NOFLAT, INIT_INF, MT>;
    auto k_seed = fill_round_kernel<WT, NOFLAT, NOFLAT ? INIT_SEED : INIT_INF, MT>;
    // > 64 KB of dynamic LDS is an opt-in PER DEVICE: one flag per device (and per instantiation of this function), set
    // under a lock -- band contexts on several GPUs launch from several host threads
    {
        static std::mutex mu;
        static bool attr_done[64] = {};
        int dev = 0;
        MH_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            MH_HIP(hipFuncSetAttribute((const void *)k_none, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            MH_HIP(hipFuncSetAttribute((const void *)k_inf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            MH_HIP(hipFuncSetAttribute((const void *)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    RoundArgs a;
    a.H = f.H; a.Wd = f.W; a.ntr = m.ntr; a.ntc = m.ntc; a.mtr = m.mtr; a.mtc = m.mtc; a.shard_cap = m.shard_cap;
    a.list_cur = m.lists + (size_t)(round & 1) * m.list_elems; a.list_nxt = m.lists + (size_t)((round + 1) & 1) * m.list_elems;
    a.mark_cur = m.marks + (size_t)(round & 1) * m.nt; a.mark_nxt = m.marks + (size_t)((round + 1) & 1) * m.nt;
    a.count_cur = m.count + (size_t)round * NSHARD; a.count_nxt = m.count + (size_t)(round + 1) * NSHARD;
    a.head = m.head + (size_t)round * NSHARD;
    a.sh = f.sh; a.dg = f.dg; a.seed_add = f.seed_add; a.stats = m.d_stats;
    a.fixed_top = f.fixed_top; a.fixed_bot = f.fixed_bot;
    const int64_t want = MT == 2 ? m.nt : cdiv(m.nt, 4);
    const unsigned grid = (unsigned)(want < resident_blocks ? want : resident_blocks);
    auto kern = init == INIT_NONE ? k_none : (init == INIT_SEED ? k_seed : k_inf);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, f.dem, f.seed ? f.seed : f.dem, reinterpret_cast<WT *>(f.out));
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

static int fill_launch_any(FillRun &f, int round, int init, hipStream_t s)
{
    return f.noflat ? fill_launch<double, true>(f, round, init, s) : fill_launch<float, false>(f, round, init, s);
}

// workspace of a run whose `out` already holds an upper bound of the fixed point: no initialising round; certify() finds the
// tiles that can still move and batch() relaxes them
int FillRun::attach(hipStream_t s)
{
    attach_only = true;
    bool active = false;
    const int rc = begin(s, &active);
    attach_only = false;
    return rc;
}

int FillRun::begin(hipStream_t s, bool *active)
{
    Impl &m = *impl;
    m.round = 0;
    m.rounds_used = 0;
    *active = false;
    m.trivial = H < 3 || W < 3;
    if (m.trivial && attach_only) return MHIP_OK;
    if (m.trivial) {  // no interior cell: filled == dem (fill.py:102-109 with an empty sweep area)
        const int64_t n = H * W;
        if (noflat) hipLaunchKernelGGL((copy_dem_kernel<double>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, dem, reinterpret_cast<double *>(out), n);
        else hipLaunchKernelGGL((copy_dem_kernel<float>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, dem, reinterpret_cast<float *>(out), n);
        MH_HIP(hipGetLastError());
        return MHIP_OK;
    }
    m.ntr = (int)cdiv(H - 2, TI);
    m.ntc = (int)cdiv(W - 2, TI);
    const int mt = noflat ? 1 : 2;   // scheduling unit: single tiles for the f64 kernel (fill_launch)
    m.mtr = (m.ntr + mt - 1) / mt;
    m.mtc = (m.ntc + mt - 1) / mt;
    m.ntiles = (int64_t)m.ntr * m.ntc;
    m.nt = (int64_t)m.mtr * m.mtc;
    // workspace: lists[2][NSHARD][cap] | marks[2][nt] | count[MAX_ROUNDS+1][NSHARD] | head[MAX_ROUNDS+1][NSHARD] | stats[64][2]
    auto align16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    m.shard_cap = (int)cdiv(m.nt, NSHARD) + 1;
    m.list_elems = (size_t)NSHARD * m.shard_cap;
    const size_t off_marks = align16(m.list_elems * 2 * 4), off_count = align16(off_marks + (size_t)m.nt * 2 * 4);
    const size_t off_head = align16(off_count + (size_t)(MAX_ROUNDS + 1) * NSHARD * 4);
    const size_t off_stats = align16(off_head + (size_t)(MAX_ROUNDS + 1) * NSHARD * 4);
    MH_TRY(m.ws.alloc(off_stats + 64 * 16 + 128));   // + profile words (MH_PROFILE_VISIT builds)
    MH_HIP(hipMemsetAsync(m.ws.as<char>() + off_marks, 0, off_stats + 64 * 16 + 128 - off_marks, s));
    m.lists = m.ws.as<int>();
    m.marks = reinterpret_cast<unsigned int *>(m.ws.as<char>() + off_marks);
    m.count = reinterpret_cast<unsigned int *>(m.ws.as<char>() + off_count);
    m.head = reinterpret_cast<unsigned int *>(m.ws.as<char>() + off_head);
    m.d_stats = reinterpret_cast<unsigned long long *>(m.ws.as<char>() + off_stats);

    if (attach_only) return MHIP_OK;
    MH_TRY(fill_launch_any(*this, 0, seed ? INIT_SEED : INIT_INF, s));
    m.round = 1;
    m.rounds_used = 1;
    unsigned int h_cnt[NSHARD];
    MH_HIP(hipMemcpyAsync(h_cnt, m.count + NSHARD, sizeof(h_cnt), hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    unsigned int t = 0;
    for (int k = 0; k < NSHARD; ++k) t += h_cnt[k];
    *active = t != 0;
    return MHIP_OK;
}

int FillRun::batch(hipStream_t s, bool *active)
{
    Impl &m = *impl;
    *active = false;
    if (m.trivial) return MHIP_OK;
    const int BATCH = rounds_per_batch < 2 ? 2 : (rounds_per_batch > MAX_BATCH ? MAX_BATCH : rounds_per_batch & ~1);
    if (m.round + BATCH >= MAX_ROUNDS) {
        set_error("fill did not converge within %d rounds", MAX_ROUNDS);
        return MHIP_ENOTCONV;
    }
    unsigned int h_cnt[MAX_BATCH * NSHARD];
    for (int b = 0; b < BATCH; ++b) MH_TRY(fill_launch_any(*this, m.round + b, INIT_NONE, s));
    MH_HIP(hipMemcpyAsync(h_cnt, m.count + (size_t)(m.round + 1) * NSHARD, sizeof(unsigned int) * BATCH * NSHARD, hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    int used = BATCH;
    bool still = true;
    for (int b = 0; b < BATCH; ++b) {
        unsigned int t = 0;
        for (int k = 0; k < NSHARD; ++k) t += h_cnt[b * NSHARD + k];
        if (t == 0) {  // round (round + b) appended nothing: converged, the later launches were no-ops
            used = b + 1;
            still = false;
            break;
        }
    }
    // every launched round index is spent (its queue heads have been advanced, even by no-op launches), so the next
    // batch -- possibly re-armed by activate_row() -- starts on fresh counters; BATCH is even: list parity is kept
    m.round += BATCH;
    m.rounds_used += used;
    *active = still;
    return MHIP_OK;
}

int FillRun::activate_row(int side, hipStream_t s)
{
    Impl &m = *impl;
    if (m.trivial) return MHIP_OK;
    const int ti = side == 0 ? 0 : m.mtr - 1;
    hipLaunchKernelGGL(activate_tile_row_kernel, dim3((unsigned)cdiv(m.mtc, 256)), dim3(256), 0, s, ti, m.mtc, m.shard_cap,
                       m.lists + (size_t)(m.round & 1) * m.list_elems, m.marks + (size_t)(m.round & 1) * m.nt,
                       m.count + (size_t)m.round * NSHARD);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

// Certification.  The worklist schedule visits a tile again only when a neighbour's probe says its halo would drop; a wake-up
// lost anywhere (observed: ONE cell of 1.07 G left too high on a 16384 x 65536 raster filled as four concurrent bands) would
// leave a state that is not a fixed point.  certify() streams over the raster once (a 3 x 3 stencil: 8 / 12 bytes per cell)
// and queues the tile of every cell that could still drop -- W[c] > max(dtm[c], candidate from its neighbours) -- then runs
// the schedule to local convergence again.  No such cell anywhere proves the state is a fixed point reached from above,
// i.e. the reference's result.
namespace {
template <typename WT, bool NOFLAT>
__global__ __launch_bounds__(256) void verify_kernel(const float *__restrict__ dem, const WT *__restrict__ W, int64_t H, int64_t Wd, double sh,
                                                     double dg, int fixed_top, int fixed_bot, int mt, int mtc, int shard_cap, int *list,
                                                     unsigned int *mark, unsigned int *count)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = (int64_t)blockIdx.y + 1;
    if (c < 1 || c >= Wd - 1 || r >= H - 1) return;   // interior cells only (band halo rows are local rows 0 / H-1: never updated here)
    (void)fixed_top;
    (void)fixed_bot;
    auto at = [&](int64_t rr, int64_t cc) -> WT {
        const WT v = W[rr * Wd + cc];
        return is_nan_bits(v) ? Inf<WT>::v() : v;        // a NaN neighbour never wins a minimum (bit test: -fno-honor-nans)
    };
    const WT own = W[r * Wd + c];
    const float dv = dem[r * Wd + c];
    if (is_nan_bits(own) || is_nan_bits(dv)) return;      // NaN cells never move
    const WT me = fmin(fmin(at(r - 1, c), at(r + 1, c)), fmin(at(r, c - 1), at(r, c + 1)));
    const WT md = fmin(fmin(at(r - 1, c - 1), at(r - 1, c + 1)), fmin(at(r + 1, c - 1), at(r + 1, c + 1)));
    WT cand;
    if constexpr (NOFLAT) cand = fmin(__dadd_rn((double)md, dg), __dadd_rn((double)me, sh));
    else cand = fmin(md, me);
    if (own > fmax((WT)dv, cand)) {                      // the cell would still drop: queue its tile
        const int ti = (int)((r - 1) / TI), tj = (int)((c - 1) / TI);
        const int t = (ti / mt) * mtc + tj / mt, shd = t % NSHARD;
        const unsigned bit = mt == 2 ? 1u << ((ti & 1) * 2 + (tj & 1)) : 1u;
        if (atomicOr(&mark[t], bit) == 0u) list[(size_t)shd * shard_cap + atomicAdd(&count[shd], 1u)] = t;
    }
}
}  // namespace

int FillRun::certify(hipStream_t s, bool *changed)
{
    Impl &m = *impl;
    *changed = false;
    if (m.trivial) return MHIP_OK;
    const dim3 grid((unsigned)cdiv(W, 256), (unsigned)(H - 2));
    int *list = m.lists + (size_t)(m.round & 1) * m.list_elems;
    unsigned int *mark = m.marks + (size_t)(m.round & 1) * m.nt, *cnt = m.count + (size_t)m.round * NSHARD;
    const int mt = noflat ? 1 : 2;
    if (noflat)
        hipLaunchKernelGGL((verify_kernel<double, true>), grid, dim3(256), 0, s, dem, reinterpret_cast<const double *>(out), H, W, sh, dg, fixed_top,
                           fixed_bot, mt, m.mtc, m.shard_cap, list, mark, cnt);
    else
        hipLaunchKernelGGL((verify_kernel<float, false>), grid, dim3(256), 0, s, dem, reinterpret_cast<const float *>(out), H, W, sh, dg, fixed_top,
                           fixed_bot, mt, m.mtc, m.shard_cap, list, mark, cnt);
    MH_HIP(hipGetLastError());
    unsigned int h_cnt[NSHARD];
    MH_HIP(hipMemcpyAsync(h_cnt, cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    unsigned int queued = 0;
    for (int k = 0; k < NSHARD; ++k) queued += h_cnt[k];
    if (!queued) return MHIP_OK;
    *changed = true;
    bool active = true;
    while (active) MH_TRY(batch(s, &active));
    return MHIP_OK;
}

int FillRun::finish(hipStream_t s, FillStats *st)
{
    Impl &m = *impl;
    if (st) {
        *st = FillStats();
        if (!m.trivial) {
            unsigned long long h_stats[128];
            MH_HIP(hipMemcpyAsync(h_stats, m.d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, s));
            MH_HIP(stream_sync(s));
            st->rounds = m.rounds_used;
#ifdef MH_PROFILE_VISIT
            {
                unsigned long long pr[8];
                MH_HIP(hipMemcpy(pr, m.d_stats + 128, sizeof(pr), hipMemcpyDeviceToHost));
                const double v = (double)(pr[5] ? pr[5] : 1);
                fprintf(stderr, "[visit profile %s] visits=%llu ticks/visit: load=%.0f passes=%.0f transposes=%.0f store=%.0f push=%.0f | wave busy ticks/visit=%.0f\n",
                        noflat ? "noflat" : "plain", pr[5], pr[0] / v, pr[1] / v, pr[2] / v, pr[3] / v, pr[4] / v, pr[6] / v);
                // active tiles per round
                std::vector<unsigned int> cnt((size_t)(m.round + 1) * NSHARD);
                MH_HIP(hipMemcpy(cnt.data(), m.count, cnt.size() * 4, hipMemcpyDeviceToHost));
                const long edges[6] = {256, 1024, 2048, 4096, 16384, 1L << 40};
                long nr[6] = {0}, nt_[6] = {0};
                for (int r = 1; r <= m.round; ++r) {
                    long t = 0;
                    for (int k = 0; k < NSHARD; ++k) t += cnt[(size_t)r * NSHARD + k];
                    if (!t) continue;
                    for (int e = 0; e < 6; ++e) if (t < edges[e]) { nr[e]++; nt_[e] += t; break; }
                }
                fprintf(stderr, "[rounds by active tiles] <256: %ld (%ld) <1024: %ld (%ld) <2048: %ld (%ld) <4096: %ld (%ld) <16384: %ld (%ld) more: %ld (%ld)\n",
                        nr[0], nt_[0], nr[1], nt_[1], nr[2], nt_[2], nr[3], nt_[3], nr[4], nt_[4], nr[5], nt_[5]);
            }
#endif
            for (int k = 0; k < 64; ++k) {
                st->visits += (int64_t)h_stats[2 * k];
                st->cycles += (int64_t)h_stats[2 * k + 1];
            }
            st->tiles = m.ntiles;
        }
    }
    m.ws.release();
    return MHIP_OK;
}

static int fill_run_to_convergence(FillRun &f, hipStream_t s, FillStats *st)
{
    bool active = false;
    MH_TRY(f.begin(s, &active));
    while (active) MH_TRY(f.batch(s, &active));
    // until a full sweep changes nothing.  (MHIP_FILL_NOCERTIFY, development: the raw schedule, for the tests that pin its protocol)
    if (!dev_env("MHIP_FILL_NOCERTIFY"))
        for (bool changed = true; changed;) MH_TRY(f.certify(s, &changed));
    return f.finish(s, st);
}


// fill.fill_terrain on one raster.  The exact tiled priority-flood (pflood.hip) is the default; the iterative tile schedule
// of this file takes over when a tile exceeds one of the flood's capacities, for rasters without an interior, and when
// MHIP_FILL=iterative is set.  d_depths (optional): filled - dem, written by the flood's last pass (*depths_done = true) --
// the iterative path leaves it to the caller.
int fill_plain_dev(const float *d_dem, float *d_out, int64_t H, int64_t W, hipStream_t s, FillStats *st, float *d_depths, bool *depths_done)
{
    const bool force_iter = [] { const char *e = dev_env("MHIP_FILL"); return e && std::string(e) == "iterative"; }();   // (development: engine selection for A/B runs and tests)
    if (depths_done) *depths_done = false;
    if (!force_iter && H >= 3 && W >= 3) {
        bool violated = false;
        const int rc = fill_plain_pflood_dev(d_dem, d_out, d_depths, H, W, s, st, &violated);
        if (rc == MHIP_OK && !violated) {
            if (depths_done) *depths_done = d_depths != nullptr;
            if (st) st->algorithm = 1;
            return MHIP_OK;
        }
        if (rc == MHIP_OK) {
            // the flood's surface failed its proof (check.hip): it is an upper bound of the result all the same, so the iterative
            // schedule starts from it -- certify() queues the tiles that can still move -- instead of from +inf
            FillStats st_flood = st ? *st : FillStats();
            FillRun h;
            h.noflat = false; h.dem = d_dem; h.out = d_out; h.H = H; h.W = W;
            MH_TRY(h.attach(s));
            for (bool changed = true; changed;) MH_TRY(h.certify(s, &changed));
            FillStats st_rel;
            MH_TRY(h.finish(s, &st_rel));
            if (st) {
                *st = st_flood;
                st->algorithm = 4;     // flood + repair
                st->rounds += st_rel.rounds;
                st->visits += st_rel.visits;
                st->cycles += st_rel.cycles;
            }
            return MHIP_OK;         // (the depths are left to the caller: *depths_done stays false)
        }
        if (rc != MHIP_ELIMIT) return rc;
    }
    FillRun f;
    f.noflat = false; f.dem = d_dem; f.out = d_out; f.H = H; f.W = W;
    return fill_run_to_convergence(f, s, st);
}

// No-flats fill.  With `d_filled` (the plain fill F of the same DEM) the iteration starts from the pointwise
// upper bound U = F + K instead of +inf, K = 1.01 * ncells * diag:  along a simple path that realises F[c] the
// no-flats recurrence v <- max(dtm, fl(v + eps)) grows by at most eps*(1 + 2**-10) per step over the running
// maximum (<= F[c]), a simple path has at most ncells steps and eps <= diag, hence G <= U.  The greatest fixed
// point G is the only fixed point X with G <= X (Knaster-Tarski), and the monotone chaotic iteration started
// at any U >= G stays >= G and is dominated by the iteration started at +inf, so it stops exactly at G: the
// seeded schedule returns the same bits as the reference's start from +inf, in far fewer rounds.
void noflat_seed(FillRun &f, const float *d_filled, double sh, double dg, int64_t ncells_global)
{
    const bool seedable = d_filled && sh >= 0.0 && dg >= sh && dg == dg && dg < 1e300;
    f.seed = seedable ? d_filled : nullptr;
    f.seed_add = 1.01 * (double)ncells_global * dg;
}

int fill_noflat_dev(const float *d_dem, double *d_out, int64_t H, int64_t W, double sh, double dg, hipStream_t s,
                    FillStats *st, const float *d_filled, StageHook *tail_hook, D8Sink *d8)
{
    if (d8) d8->done = false;
    struct FireAtExit {     // whatever path is taken, the hook fires (at the latest when the stage is done)
        StageHook *h;
        hipStream_t s;
        ~FireAtExit() { if (h) h->fire(s); }
    } fire_at_exit{tail_hook, s};
    // the integer geodesic transform of noflat_geo.hip first; it hands back MHIP_ELIMIT for what it does not cover
    bool partial = false;
    const double seed_add = 1.01 * (double)(H * W) * dg;      // see noflat_seed()
    const int rc = fill_noflat_geodesic_dev(d_dem, d_filled, d_out, H, W, sh, dg, seed_add, s, st, &partial, tail_hook, d8);
    if (tail_hook) tail_hook->fire(s);      // (not applicable: the relaxation below is throughput bound all the way)
    if (rc != MHIP_ELIMIT && !(rc == MHIP_OK && partial)) return rc;
    if (rc == MHIP_OK) {
        // hybrid: exact everywhere but on the flats of irregular levels (a sea at elevation 0, ...), which hold the upper bound
        // F + seed_add.  The relaxation settles them -- certify() queues exactly the tiles that can still move -- and the
        // reference's equation is checked at every cell afterwards; a failure falls through to the relaxation from scratch.
        FillStats st_geo = st ? *st : FillStats();
        FillRun h;
        h.noflat = true; h.dem = d_dem; h.out = d_out; h.H = H; h.W = W; h.sh = sh; h.dg = dg;
        MH_TRY(h.attach(s));
        int sweeps = 0;
        for (bool changed = true; changed; ++sweeps) MH_TRY(h.certify(s, &changed));
        FillStats st_rel;
        MH_TRY(h.finish(s, &st_rel));
        bool ok = false;
        MH_TRY(noflat_verify_dev(d_dem, d_out, H, W, sh, dg, s, &ok));
        if (dev_env("MHIP_NG_DEBUG"))
            fprintf(stderr, "[noflat hybrid] relaxation: %d certification sweeps, %d rounds, %lld visits; verified: %d\n", sweeps, st_rel.rounds,
                    (long long)st_rel.visits, (int)ok);
        if (ok) {
            if (st) {
                *st = st_geo;
                st->algorithm = 3;
                st->rounds += st_rel.rounds;
                st->visits += st_rel.visits;
                st->cycles += st_rel.cycles;
            }
            return MHIP_OK;
        }
    }
    FillStats st_geo = st ? *st : FillStats();      // (why the transform handed the raster back: diagnostics)
    FillRun f;
    f.noflat = true; f.dem = d_dem; f.out = d_out; f.H = H; f.W = W; f.sh = sh; f.dg = dg;
    noflat_seed(f, d_filled, sh, dg, H * W);
    const int rc_f = fill_run_to_convergence(f, s, st);
    if (st) {
        st->geo_reject = st_geo.geo_reject ? st_geo.geo_reject : (rc == MHIP_OK ? 2 : 1);
        st->geo_irregular = st_geo.geo_irregular; st->geo_unreached = st_geo.geo_unreached; st->geo_mismatch = st_geo.geo_mismatch;
    }
    return rc_f;
}

}  // namespace mh
