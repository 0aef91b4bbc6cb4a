// pflood.hip -- fill.fill_terrain (reference fill.py:112-171, sweeps _fill.pyx:28-70) as an exact tiled priority-flood.
//
// The reference's result is the greatest fixed point of W = max(dtm, min(W, 8 nbrs)) with the raster border fixed to dtm,
// i.e. F[c] = min over paths c -> raster border of the maximum elevation on the path (a minimax path).  The iterative
// schedule of fill.hip needs O(drainage path length / tile) rounds over the raster; this file computes the same bits in a
// fixed number of raster passes (after Barnes, "Parallel priority-flood depression filling for trillion cell digital
// elevation models", 2016, restated for LDS):
//
//   K1 pf_tile_kernel     one 64 x 64 window (62 x 62 owned cells + ring) per workgroup, all in LDS:
//                         steepest-descent pointers (plateaus of equal cells are merged first) -> pointer doubling ->
//                         BASINS (one per local pit); min pass height between adjacent basins (LDS hash); label-correcting
//                         on the basin graph gives every basin its tile-local spill level V (minimax to the window ring /
//                         the raster border) and the SEED it drains to (a ring cell that is a local pit, or OCEAN = the
//                         raster border).  Out: basin slot per owned cell (u16), per-tile basin table (V, seed), the seed
//                         of every ring cell, and the min spill elevation between pairs of seeds.
//   K2 pf_ring_kernel /   a ring cell of tile T is an owned cell of a neighbouring tile T': both seeds it drains to are
//      pf_link2_kernel    joined by an edge of weight W_T'[cell]; de-duplicated per tile.
//   K3 pf_pack_kernel /   minimax distance of every seed to OCEAN over (spill edges + links): block visits (4 x 4 tiles) taken from
//      pf_solve_queue_    a queue inside ONE launch; the block's relaxations are packed once and held in registers during a visit,
//      kernel             the levels of its 6 x 6 region of tiles in LDS (pf_solve_kernel: the same visits as one launch per round).
//   K4 pf_final_kernel /  level of a basin = max(V, L[seed]);  F[c] = max(dem[c], level[basin[c]]), depths = F - dem.
//      pf_apply_kernel
//
// Why it is exact (tested bit for bit against the oracle on every existing fill case): with V[c] := max(dem[c], V[basin(c)])
// (the fill of the tile alone, ring cells fixed) every cell c has a path to its seed s whose maximum is <= V[c]; an
// optimal raster path from c to the border has maximum >= V[a] for every cell a on it (its tail from a must reach a's
// window ring); so the minimax distance L over the graph whose edges are (seed(a), seed(b), max(V[a], V[b])) for adjacent
// cells a, b -- plus the identification of a ring cell's two seeds -- satisfies F[c] = max(V[c], L[seed(c)]) in both
// directions.  Only comparisons and copies of float32 values are involved: no rounding anywhere.
//
// Capacity limits (basins / basin pairs / seed pairs / links per tile) raise an overflow flag and the caller falls back to
// the iterative schedule (fill.hip), which has none; row bands with halo rows use the iterative schedule as well.
#include "common.hpp"
#include <algorithm>
#include <vector>

namespace mh {

namespace {

constexpr int WN = 64, TI = 62, NC = WN * WN;
constexpr int NBMAX = 1024;        // basins per tile
constexpr int HE = 2048;           // basin-pair hash: words per half of the hash's memory ...
constexpr int HEU = 2032;          // ... and entries in use (the last 16 words of the second half are the workgroup's scalars)
constexpr int SE = 512;            // seed-pair hash entries
constexpr int SPMAX = 192;         // spill edges stored per tile (unused entries hold ~0)
constexpr int LMAX = 256;          // links stored per tile (unused entries hold ~0)
constexpr int NSMAX = 128;         // seeds per tile (compact indices; OCEAN stays 255)
constexpr int LH = 1024;           // link hash entries
constexpr uint32_t KINV = 0xFFFFFFFFu;   // key of a cell outside the raster (above +inf)
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr int OCEAN = 255, NOLAB = 254;
constexpr uint8_t C_VALID = 1, C_BORDER = 2, C_RING = 4;

struct PfArgs {
    int64_t H, W;
    int ntr, ntc;
    const float *dem;
    uint16_t *bslot;        // [H * W]
    uint32_t *tabV;         // [ntiles * NBMAX] spill level keys, then (pf_final_kernel) final level keys
    uint8_t *tabL;          // [ntiles * NBMAX] seed of the basin (compact index inside the tile, or OCEAN)
    int *tileNB;            // [ntiles]
    uint8_t *ringLab;       // [ntiles * 256] seed (compact index) of every ring cell, by ring position
    unsigned long long *spill;  // [ntiles * SPMAX]  (la << 40 | lb << 32 | w)
    int *tileNS;            // [ntiles]
    unsigned long long *links;  // [ntiles * LMAX]   (myLab << 48 | dir << 40 | nbrLab << 32 | w)
    int *tileNL;            // [ntiles] links of the tile = tileNL0 links between tiles + the band's halo links
    int *tileNL0;           // [ntiles]
    unsigned long long *ringRec;   // [ntiles * 256] per ring position of a tile: its own seed << 40 | the owner tile's seed << 32 | the cell's fill level there (pf_ring_kernel)
    int fixed_top, fixed_bot;   // row band: local row 0 / H - 1 is a halo row of the neighbouring band (a ring row, not a raster border)
    uint32_t *Lv;           // [ntiles * NSMAX] minimax level of every seed (keys)
    unsigned int *flags;    // [0]: overflow
    uint32_t *mm;           // [ntiles * 8 * 2] per wavefront of pf_tile_kernel: smallest | largest elevation key of its strip (all ones: a NaN)
    unsigned long long *prof;   // -DPF_PROFILE builds: clock ticks per phase of pf_tile_kernel, summed over the tiles
    int stop;                   // -DPF_PHASES builds: pf_tile_kernel returns after phase `stop` - 1 (timing launches only)
};

#ifdef PF_PROFILE
#define PF_STAMP(i)                                                               \
    do {                                                                          \
        __syncthreads();                                                          \
        if (threadIdx.x == 0) {                                                   \
            const long long now_ = __builtin_amdgcn_s_memtime();                  \
            atomicAdd(&a.prof[i], (unsigned long long)(now_ - pf_t0_));           \
            pf_t0_ = now_;                                                        \
        }                                                                         \
    } while (0)
#elif defined(PF_PHASES)
#define PF_STAMP(i)                       \
    do {                                  \
        if ((a.stop & 0xff) == (i) + 1) return;    \
    } while (0)
#else
#define PF_STAMP(i)
#endif

__device__ __forceinline__ uint32_t dem_key(float v)
{
    if (v != v) return f32_key(__builtin_inff());   // a NaN cell never wins a comparison (_fill.pyx:22): like +inf
    return f32_key(v + 0.0f);                        // -0.0 -> +0.0: the two compare equal in the reference
}

__device__ __forceinline__ int ring_pos(int wr, int wc)
{
    if (wr == 0) return wc;
    if (wr == WN - 1) return WN + wc;
    if (wc == 0) return 2 * WN + wr - 1;
    return 2 * WN + (WN - 2) + wr - 1;
}
// inverse: window coordinates of ring position p (0..251)
__device__ __forceinline__ void ring_cell(int p, int &wr, int &wc)
{
    if (p < WN) { wr = 0; wc = p; }
    else if (p < 2 * WN) { wr = WN - 1; wc = p - WN; }
    else if (p < 2 * WN + WN - 2) { wr = p - 2 * WN + 1; wc = 0; }
    else { wr = p - (2 * WN + WN - 2) + 1; wc = WN - 1; }
}

// ---- K1 -----------------------------------------------------------------------------------------------------------------
constexpr int NT = 512;            // threads per tile: eight wavefronts, a 8-row x 1-column strip of the window per thread
constexpr int CPT = NC / NT;       // cells per thread
constexpr int EPT = HE / NT;       // (compacted) basin pairs per thread in the label-correcting loop

struct WinGeom {
    int64_t r0, c0, H, W;
    int fixed_top, fixed_bot;
    bool inner;      // the window touches no border row / column of the raster: a cell's class is its position in the window
};
// class of window cell ci: valid / raster border / window ring (the last two exclude each other)
__device__ __forceinline__ uint8_t cell_class(const WinGeom &g, int ci)
{
    const int wr = ci >> 6, wc = ci & 63;
    // (all windows but the outermost ring of them: no 64-bit compares and branches per cell)
    if (g.inner) return (wr == 0 || wr == WN - 1 || wc == 0 || wc == WN - 1) ? (uint8_t)(C_VALID | C_RING) : (uint8_t)C_VALID;
    const int64_t rr = g.r0 + wr, cc = g.c0 + wc;
    if (rr >= g.H || cc >= g.W) return 0;
    // (a band's halo rows sit on window ring rows -- PfRun::begin checks the alignment -- and are ring cells like any other)
    if ((rr == 0 && !g.fixed_top) || (rr == g.H - 1 && !g.fixed_bot) || cc == 0 || cc == g.W - 1) return C_VALID | C_BORDER;
    if (wr == 0 || wr == WN - 1 || wc == 0 || wc == WN - 1) return C_VALID | C_RING;
    return C_VALID;
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void pf_tile_kernel(PfArgs a)
{
    // LDS plan: 40 KB to the byte -- FOUR workgroups = 32 wavefronts per CU (and <= 64 VGPRs, see the launch attribute); all
    // phases but the pair fold are bound by LDS / barrier latency, not by LDS bandwidth, so residency is what pays (measured with
    // one / two / three workgroups per CU: 11.9 / 6.9 / 4.6 ms; until round 4 the plan was 49 KB: three).
    //   zk   16 KB  elevation keys of the window                      -> after S5: the compacted basin pairs (ek | ew)
    //   ptr   8 KB  steepest-descent pointer -> pit (root) of a cell  -> S4: basin slot of a root -> S5: basin slot of a cell
    //   X    16 KB  S2b: plateau ids | plateau drains (16 bit each)
    //               S4:  bkey (8 KB: level | seed of every basin) + btype (1 KB)   -- parked in the tile's tables in global
    //                    memory while the pair fold needs the room (the basins' kinds are rebuilt from the roots' classes) --
    //               S5:  the basin-pair hash (keys | values, HEU = 2032 of 2048 entries each)
    //               label-correcting, S6: bkey + btype again, seed map, seed-pair hash
    //               the last 16 words: the workgroup's scalars (S2b keeps its plateau ids in `ptr` and leaves them alone)
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[NC + NC / 2 + 2 * HE];
    static_assert(sizeof(uint32_t) * (NC + NC / 2 + 2 * HE) == 40960, "four workgroups per CU: 160 KB / 4");
    uint32_t *const zk = lds_all;
    uint16_t *const ptr = reinterpret_cast<uint16_t *>(lds_all + NC);
    uint32_t *const hkv = lds_all + NC + NC / 2;                                   // X
    unsigned long long *const bkey = reinterpret_cast<unsigned long long *>(hkv); // [NBMAX], X[0 .. 2047]
    uint8_t *const btype = reinterpret_cast<uint8_t *>(hkv + 2 * NBMAX);          // [NBMAX], X[2048 .. 2303]; 1: interior pit (level to be found), 0: ring pit or raster border (fixed)
    int *const s_scan = reinterpret_cast<int *>(hkv + 2 * HE - 16);               // [NT / 64]
    int &s_cnt = *reinterpret_cast<int *>(hkv + 2 * HE - 8), &s_ne = *reinterpret_cast<int *>(hkv + 2 * HE - 7);
    int *const s_or = reinterpret_cast<int *>(hkv + 2 * HE - 6);                   // [3]: wg_or below
    static_assert(2 * NBMAX + NBMAX / 4 + 64 + 256 + 2 * SE <= 2 * HE - 16 && HEU <= HE - 16 && NT / 64 <= 8, "X holds every phase's arrays in front of the scalars");

    const int t = threadIdx.x, wc = t & 63, q = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile = blockIdx.x, ti = tile / a.ntc, tj = tile - ti * a.ntc;
    WinGeom g;
    g.r0 = (int64_t)ti * TI; g.c0 = (int64_t)tj * TI; g.H = a.H; g.W = a.W; g.fixed_top = a.fixed_top; g.fixed_bot = a.fixed_bot;
    g.inner = g.r0 > 0 && g.c0 > 0 && g.r0 + WN - 1 < g.H - 1 && g.c0 + WN - 1 < g.W - 1;      // (wave-uniform)
    const int64_t W = a.W;
    uint32_t *hk = hkv, *hv = hkv + HE;
    static_assert(2 * HE == NC, "plateau ids / drains (one 16-bit word per cell each) share the hash arrays");
    const int wr0 = q * CPT;
    // "does any thread of the workgroup say yes", one barrier per call like __syncthreads_or -- whose library implementation keeps
    // 256 bytes of LDS of its own, the 256 bytes that decide between three and four workgroups per CU.  Three flag words take
    // turns: call n raises s_or[n % 3] in front of its barrier, reads it behind, and thread 0 clears the word of call n + 2 there
    // (its last readers left before call n's barrier, its next writers come behind call n + 1's).
    int or_turn = 0;
    auto wg_or = [&](bool p) -> bool {
        if (__any(p) && wc == 0) s_or[or_turn] = 1;
        __syncthreads();
        const bool r = s_or[or_turn] != 0;
        if (t == 0) s_or[or_turn == 0 ? 2 : or_turn - 1] = 0;
        or_turn = or_turn == 2 ? 0 : or_turn + 1;
        return r;
    };
    if (t == 0) { s_cnt = 0; s_ne = 0; s_or[0] = 0; s_or[1] = 0; s_or[2] = 0; }      // (in front of the barrier behind S1)
#ifdef PF_PROFILE
    long long pf_t0_ = __builtin_amdgcn_s_memtime();
#endif
    // ---- S1: window -> LDS.  Buffer addressing: one per-lane byte offset + a scalar row offset (eight 64-bit addresses per thread
    // cost the registers of a third resident workgroup); rows / columns outside the raster read a clamped cell and become KINV
    const int64_t cc = g.c0 + wc;
    const int Wi = (int)W;
    const int64_t row_b = g.r0 + wr0 < g.H ? g.r0 + wr0 : g.H - 1;                  // wave-uniform
    const int nrow_in = (int)(g.H - row_b < CPT ? g.H - row_b : CPT);              // rows of the strip inside the raster (>= 1)
    const int lane_b = (int)(cc < W ? wc : W - 1 - g.c0);
    const __amdgpu_buffer_rsrc_t rdem = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dem + row_b * W + g.c0), 0, 0x7fffffff, 0x00020000);
    uint32_t clsw = 0;    // the classes of the strip's cells, four bits each (one register instead of CPT)
    auto cls = [&](int k) -> uint8_t { return (uint8_t)((clsw >> (4 * k)) & 7u); };
    {
        int so = 0;
        // the DEM's smallest and largest elevation ride along (minimum_safe_short_and_diag, fill.py:235-250, wants them before the
        // no-flats fill: a separate pass over the DEM next to this kernel cost it 0.4 ms): keys of the cells inside the raster,
        // all ones for a NaN (np.amax / np.amin propagate it)
        uint32_t klo = 0xffffffffu, khi = 0u;
        // (all of the strip's loads first: issued between the cells' classification -- 64-bit compares and branches -- each was waited
        // for before the next went out, eight round trips in a row)
        float vq[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            vq[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdem, lane_b * 4, so * 4, 0));
            so = k + 1 < nrow_in ? so + Wi : so;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int wr = wr0 + k;
            const uint8_t c = cell_class(g, wr * WN + wc);
            clsw |= (uint32_t)c << (4 * k);
            const float v = vq[k];
            zk[wr * WN + wc] = c ? dem_key(v) : KINV;
            const uint32_t kv = v != v ? 0xffffffffu : f32_key(v);
            klo = c && kv < klo ? kv : klo;
            khi = c && kv > khi ? kv : khi;
        }
        if (a.mm) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t lo2 = (uint32_t)__shfl_xor((int)klo, o), hi2 = (uint32_t)__shfl_xor((int)khi, o);
                klo = lo2 < klo ? lo2 : klo;
                khi = hi2 > khi ? hi2 : khi;
            }
            if (wc == 0) {
                a.mm[((size_t)tile * (NT / 64) + q) * 2] = klo;
                a.mm[((size_t)tile * (NT / 64) + q) * 2 + 1] = khi;
            }
        }
    }
    __syncthreads();
    PF_STAMP(0);

    // ---- S2: steepest descent pointer of every cell (lowest neighbour if strictly lower, ties -> lowest index)
    auto ld = [&](int wr, int c) -> uint32_t { return (wr < 0 || wr >= WN || c < 0 || c >= WN) ? KINV : zk[wr * WN + c]; };
    unsigned eqmask = 0, lowmask = 0;
    {
        uint32_t up[3], mid[3], dn[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            up[j] = ld(wr0 - 1, wc - 1 + j);
            mid[j] = ld(wr0, wc - 1 + j);
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int wr = wr0 + k, ci = wr * WN + wc;
#pragma unroll
            for (int j = 0; j < 3; ++j) dn[j] = ld(wr + 1, wc - 1 + j);
            const uint32_t own = mid[1];
            uint32_t best = up[0];
            int bi = ci - WN - 1;
            if (up[1] < best) { best = up[1]; bi = ci - WN; }
            if (up[2] < best) { best = up[2]; bi = ci - WN + 1; }
            if (mid[0] < best) { best = mid[0]; bi = ci - 1; }
            if (mid[2] < best) { best = mid[2]; bi = ci + 1; }
            if (dn[0] < best) { best = dn[0]; bi = ci + WN - 1; }
            if (dn[1] < best) { best = dn[1]; bi = ci + WN; }
            if (dn[2] < best) { best = dn[2]; bi = ci + WN + 1; }
            const bool valid = own != KINV;
            const bool eq = valid && (up[0] == own || up[1] == own || up[2] == own || mid[0] == own || mid[2] == own || dn[0] == own ||
                                      dn[1] == own || dn[2] == own);
            const bool lower = valid && best < own;
            if (eq) eqmask |= 1u << k;
            if (lower) lowmask |= 1u << k;
            ptr[ci] = (uint16_t)((lower && !(cls(k) & C_BORDER)) ? bi : ci);   // a raster border cell is a root by decree (OCEAN)
#pragma unroll
            for (int j = 0; j < 3; ++j) { up[j] = mid[j]; mid[j] = dn[j]; }
        }
    }
    // ---- S2b: plateaus (connected equal cells) drain through ANY member that has a lower neighbour (or is a border cell).
    // Rare on float terrain, the rule on integer-valued DEMs.  Two 16-bit words per cell, each written by its own thread
    // only: the plateau id (smallest member index) and the plateau's drain (smallest candidate index), both found by min
    // propagation over equal neighbours with a jump through the current id.
    if (wg_or(eqmask != 0)) {
        // (the ids live in the pointer array -- a thread keeps the pointers of its own cells in registers meanwhile --, the drains
        // in the first half of X: the workgroup's scalars at the end of X stay valid)
        uint16_t *comp = ptr, *drain = reinterpret_cast<uint16_t *>(hkv);
        constexpr uint16_t NODRAIN = 0xFFFF;
        uint16_t keep[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int ci = (wr0 + k) * WN + wc;
            keep[k] = ptr[ci];
            comp[ci] = (uint16_t)ci;
            uint16_t cand = NODRAIN;
            if ((eqmask >> k) & 1u) {
                if (cls(k) & C_BORDER) cand = (uint16_t)ci;
                else if ((lowmask >> k) & 1u) cand = keep[k];       // still the steepest-descent target
            }
            drain[ci] = cand;
        }
        __syncthreads();
        for (int it = 0; it < 4 * NC; ++it) {
            bool ch = false;
#pragma unroll 1
            for (int k = 0; k < CPT; ++k) {
                if (!((eqmask >> k) & 1u)) continue;
                const int wr = wr0 + k, ci = wr * WN + wc;
                const uint32_t own = zk[ci];
                uint16_t m = comp[ci], dmin = drain[ci];
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int rr = wr + dr, c2 = wc + dc;
                        if ((dr | dc) == 0 || rr < 0 || rr >= WN || c2 < 0 || c2 >= WN) continue;
                        if (zk[rr * WN + c2] == own) {
                            m = min(m, comp[rr * WN + c2]);
                            dmin = min(dmin, drain[rr * WN + c2]);
                        }
                    }
                m = min(m, comp[m]);
                dmin = min(dmin, drain[m]);
                if (m < comp[ci] || dmin < drain[ci]) {
                    comp[ci] = m;
                    drain[ci] = dmin;
                    ch = true;
                }
            }
            if (!wg_or(ch)) break;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {      // (own entries only: comp[ci] IS ptr[ci])
            const int ci = (wr0 + k) * WN + wc;
            if (!((eqmask >> k) & 1u) || (cls(k) & C_BORDER)) {
                ptr[ci] = keep[k];
                continue;
            }
            const uint16_t root = comp[ci];
            if (root != (uint16_t)ci) ptr[ci] = root;
            else ptr[ci] = drain[ci] != NODRAIN ? drain[ci] : (uint16_t)ci;
        }
        __syncthreads();
    }
    for (int i = t; i < 2 * HE - 16; i += NT) hkv[i] = EMPTY;   // (the pair hash is cleared again behind S4: bkey / btype use this memory first)
    __syncthreads();
    PF_STAMP(1);

    // ---- S3: pointer doubling -> ptr[c] = the pit (root) c drains to
    for (int it = 0; it < 16; ++it) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int ci = (wr0 + k) * WN + wc;
            const uint16_t p = ptr[ci], pp = ptr[p];
            if (pp != p) {
                ptr[ci] = pp;
                ch = true;
            }
        }
        if (!wg_or(ch)) break;
    }
    PF_STAMP(2);

    // ---- S4: number the roots (basin slots); the slot of a root replaces its (self) pointer
    int nroot = 0;
    unsigned rootmask = 0;
    uint16_t myroot[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int ci = (wr0 + k) * WN + wc;
        myroot[k] = ptr[ci];
        if (cls(k) && myroot[k] == (uint16_t)ci) {
            rootmask |= 1u << k;
            ++nroot;
        }
    }
    // (the roots are remembered in the spare bit of the class nibbles: behind the pair fold the basins' kinds are rebuilt from them)
#pragma unroll
    for (int k = 0; k < CPT; ++k) clsw |= ((rootmask >> k) & 1u) << (4 * k + 3);
    int incl = nroot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (wc >= o) incl += v;
    }
    if (wc == 63) s_scan[q] = incl;
    __syncthreads();          // also: every thread has read its pointers
    int base = incl - nroot, NB = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        if (w < q) base += s_scan[w];
        NB += s_scan[w];
    }
    if (NB > NBMAX) {         // block-uniform
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        if (!((rootmask >> k) & 1u)) continue;
        const int wr = wr0 + k, ci = wr * WN + wc, s = base++;
        ptr[ci] = (uint16_t)s;
        unsigned long long key = ~0ull;
        uint8_t ty = 1;
        if (cls(k) & C_BORDER) { key = ((unsigned long long)zk[ci] << 32) | OCEAN; ty = 0; }
        else if (cls(k) & C_RING) { key = ((unsigned long long)zk[ci] << 32) | (unsigned)ring_pos(wr, wc); ty = 0; }
        bkey[s] = key;
        btype[s] = ty;
    }
    __syncthreads();
    uint16_t myslot[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) myslot[k] = cls(k) ? ptr[myroot[k]] : (uint16_t)0xFFFF;
    __syncthreads();
    // ring cells inside an interior-pit basin are outlets of that basin at their own elevation; and every cell's basin slot
    // replaces its pointer
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int wr = wr0 + k;
        ptr[wr * WN + wc] = myslot[k];
        if (!(cls(k) & C_RING)) continue;
        const int s = myslot[k];
        if (btype[s]) atomicMin(&bkey[s], ((unsigned long long)zk[wr * WN + wc] << 32) | (unsigned)ring_pos(wr, wc));
    }
    __syncthreads();
    // the pair fold needs X for its hash: (level, seed) of the basins wait in the tile's own tables in global memory (the level key
    // as it is, the seed -- a ring position, OCEAN, or the 0xff.. of "not reached" -- as a byte), a thread's two basins each
    for (int sI = t; sI < NB; sI += NT) {
        const unsigned long long k64 = bkey[sI];
        a.tabV[(size_t)tile * NBMAX + sI] = (uint32_t)(k64 >> 32);
        a.tabL[(size_t)tile * NBMAX + sI] = (uint8_t)k64;
    }
    __syncthreads();
    for (int i = t; i < 2 * NBMAX + NBMAX / 4; i += NT) hkv[i] = EMPTY;    // (the rest of the hash has been EMPTY since S2b)
    __syncthreads();
    PF_STAMP(3);
    // ---- S5: min pass height between adjacent basins.  Neighbour slots / elevations of the strip come in one batch of
    // independent LDS reads; the four pairs of a row go to the hash together (independent atomics in flight).  A pair is
    // skipped when the same thread (row above) or the lane to the left has just offered the same pair at a weight at least
    // as low -- adjacent cells mostly straddle the same two basins, and same-address LDS atomics serialise.
    bool overflow = false;
    {
        constexpr uint16_t NONE = 0xFFFF;
        // Phase A: which neighbour pairs (right, down-left, down, down-right of a cell) straddle two basins.  Only 29 % of them do,
        // and only those are folded below (folding every pair, candidate or not, was 47 instructions x 32 pairs per thread: the one
        // VALU-bound phase of the kernel).  For this phase a thread owns a block of 4 rows x 2 columns of the wavefront's 8 x 64
        // strip, not its 8 x 1 column: the turns of phase B are set by the lane with the most candidates, and a lane on a basin
        // boundary that runs down its column held ~3 per row.  Bit 8 r + 4 c + d of `cmask`: row r, column c of the block, pair d.
        const int bc0 = 2 * (wc & 31), br0 = wr0 + 4 * (wc >> 5);
        uint32_t cmask = 0;
        {
            // slots of the block's rows br0 .. br0 + 4, columns bc0 - 1 .. bc0 + 2 (NONE outside the window)
            auto slot = [&](int row, int col) -> uint32_t {
                const bool in = row < WN && col >= 0 && col < WN;
                const uint32_t v = ptr[(row < WN ? row : WN - 1) * WN + (col < 0 ? 0 : (col < WN ? col : WN - 1))];
                return in ? v : (uint32_t)NONE;
            };
            uint32_t up[4], dn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) up[j] = slot(br0, bc0 - 1 + j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dn[j] = slot(br0 + r + 1, bc0 - 1 + j);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const uint32_t sa = up[1 + c];
                    const uint32_t nb[4] = {up[2 + c], dn[c], dn[1 + c], dn[2 + c]};
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        cmask |= ((sa != NONE) & (nb[d] != NONE) & (nb[d] != sa)) ? 1u << (8 * r + 4 * c + d) : 0u;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) up[j] = dn[j];
            }
        }
        // LDS atomics are the expensive instruction here (several cycles per active lane, more when lanes share a slot): a
        // thread first folds its 4 * CPT candidates into a register set of distinct pairs (adjacent cells mostly straddle the
        // same two basins) and only the set goes to the hash; a candidate that finds the set full goes there directly
        constexpr int NCK = 4;
        uint32_t ck[NCK], cw[NCK];
#pragma unroll
        for (int j = 0; j < NCK; ++j) { ck[j] = EMPTY; cw[j] = EMPTY; }
#ifdef PF_PROFILE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long pf_s5a = __builtin_amdgcn_s_memtime();
        unsigned pf_ncand = 0, pf_nlive = 0, pf_nslow = 0;
#endif
        auto hash_insert = [&](uint32_t key, uint32_t w) {
            unsigned hh = __umulhi(key * 2654435761u, (unsigned)HEU);
#pragma nounroll
            for (int probe = 0; probe < 64; ++probe) {
                const uint32_t pv = atomicCAS(&hk[hh], EMPTY, key);
                if (pv == EMPTY || pv == key) {
                    atomicMin(&hv[hh], w);
                    return true;
                }
                hh = hh + 1 == (unsigned)HEU ? 0u : hh + 1;
            }
            return false;
        };
        // Phase B: one candidate per lane and turn until the slowest lane of the wavefront is through (9 candidates per lane on
        // average, but a lane on a basin boundary that runs down its strip holds ~3 per row: pair phase 1.63 -> 1.51 ms only;
        // prefetching the next turn's operands changes nothing -- the turns are VALU, not latency)
#pragma nounroll
        while (__any(cmask != 0u)) {
            const bool cand = cmask != 0u;
            const int bpos = cand ? __builtin_ctz(cmask) : 0;
            cmask &= cmask - 1u;                                     // (0 stays 0)
            const int dd = bpos & 3;
            const int ci = (br0 + (bpos >> 3)) * WN + bc0 + ((bpos >> 2) & 1);
            const int ni_ = min(ci + (dd == 0 ? 1 : WN - 2 + dd), NC - 1);   // right | down-left, down, down-right
            const uint32_t sa = ptr[ci], sb = ptr[ni_];
            const uint32_t w = max(zk[ci], zk[ni_]);
            const uint32_t key = cand ? (min(sa, sb) << 10 | max(sa, sb)) : EMPTY;
            bool placed = !cand;
#pragma unroll
            for (int j = 0; j < NCK; ++j) {
                const bool hit = !placed && (ck[j] == key || ck[j] == EMPTY);
                cw[j] = hit ? min(cw[j], w) : cw[j];
                ck[j] = hit ? key : ck[j];
                placed = placed || hit;
            }
#ifdef PF_PROFILE
            pf_ncand += cand;
#endif
            if (!placed) {   // set full (rare)
#ifdef PF_PROFILE
                ++pf_nslow;
#endif
                if (!hash_insert(key, w)) overflow = true;
            }
        }
        {
            uint32_t prev[NCK];
            unsigned h[NCK];
#pragma unroll
            for (int j = 0; j < NCK; ++j) {
                h[j] = __umulhi(ck[j] * 2654435761u, (unsigned)HEU);
                prev[j] = EMPTY;
#ifdef PF_PHASES
                if (a.stop & 0x100) continue;
#endif
                if (ck[j] != EMPTY) prev[j] = atomicCAS(&hk[h[j]], EMPTY, ck[j]);
            }
#pragma unroll
            for (int j = 0; j < NCK; ++j) {
                if (ck[j] == EMPTY) continue;
#ifdef PF_PROFILE
                ++pf_nlive;
#endif
#ifdef PF_PHASES
                if (a.stop & 0x100) { if (ck[j] + cw[j] == 12345u) overflow = true; continue; }
                if (a.stop & 0x200) continue;
#endif
                if (prev[j] == EMPTY || prev[j] == ck[j]) atomicMin(&hv[h[j]], cw[j]);
                else if (!hash_insert(ck[j], cw[j])) overflow = true;
            }
        }
#ifdef PF_PROFILE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t == 0) {
            atomicAdd(&a.prof[14], (unsigned long long)(pf_s5a - pf_t0_));
            atomicAdd(&a.prof[15], (unsigned long long)(__builtin_amdgcn_s_memtime() - pf_s5a));
        }
        atomicAdd(&a.prof[16], (unsigned long long)pf_ncand);
        atomicAdd(&a.prof[17], (unsigned long long)pf_nlive);
        atomicAdd(&a.prof[18], (unsigned long long)pf_nslow);
#endif
    }
    if (wg_or(overflow)) {
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
    PF_STAMP(4);
    // the basins' (level, seed) come back from the tile's tables -- each thread reads what it wrote itself --: the loads go out HERE,
    // in front of the compaction, and are used behind it (two dependent round trips to memory otherwise sit between two barriers)
    static_assert(NBMAX <= 2 * NT, "a thread parks at most two basins");
    uint32_t pk_hi[2] = {0xffffffffu, 0xffffffffu}, pk_lo[2] = {0xffu, 0xffu};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sI = t + u * NT;
        if (sI < NB) {
            pk_hi[u] = __hip_atomic_load(&a.tabV[(size_t)tile * NBMAX + sI], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pk_lo[u] = __hip_atomic_load(&a.tabL[(size_t)tile * NBMAX + sI], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // compact the pairs into the memory of the elevation keys (every thread holds what it still needs in registers)
    uint32_t *ek = zk, *ew = zk + HE;
    for (int h = t; h < HEU; h += NT) {
        const uint32_t key = hk[h];
        if (key == EMPTY) continue;
        const int i = atomicAdd(&s_ne, 1);
        ek[i] = key;
        ew[i] = hv[h];
    }
    __syncthreads();
    const int NE = s_ne;
    // X is free again: bkey / btype move back in; the basins' kinds from the roots -- a basin whose root is a ring cell or a raster
    // border cell is fixed, every other is an interior pit
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sI = t + u * NT;
        if (sI < NB) {
            const uint32_t hi = pk_hi[u], lo = pk_lo[u];
            bkey[sI] = ((unsigned long long)hi << 32) | ((hi == 0xffffffffu && lo == 0xffu) ? 0xffffffffull : (unsigned long long)lo);
            btype[sI] = 1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPT; ++k)
        if (((clsw >> (4 * k + 3)) & 1u) && (cls(k) & (C_BORDER | C_RING))) btype[ptr[(wr0 + k) * WN + wc]] = 0;
    __syncthreads();
    PF_STAMP(5);
    // ---- label-correcting on the basin graph: (level, seed) of every interior-pit basin
    {
        int e_sa[EPT], e_sb[EPT];
        unsigned long long e_w[EPT];
        unsigned live = 0;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int e = t + j * NT;
            e_sa[j] = e_sb[j] = 0;
            e_w[j] = 0;
            if (e < NE) {
                const uint32_t key = ek[e];
                e_sa[j] = (int)(key >> 10);
                e_sb[j] = (int)(key & 1023u);
                e_w[j] = ew[e];
                if (btype[e_sa[j]]) live |= 1u << (2 * j);
                if (btype[e_sb[j]]) live |= 2u << (2 * j);
            }
        }
        for (int it = 0; it < 4 * NBMAX; ++it) {
            bool ch = false;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                if (!((live >> (2 * j)) & 3u)) continue;
                const unsigned long long ka = bkey[e_sa[j]], kb = bkey[e_sb[j]];
                if ((live >> (2 * j)) & 1u) {
                    const unsigned long long hi = max(e_w[j], kb >> 32), cand = (hi << 32) | (kb & 0xffffffffull);
                    if (cand < ka) { atomicMin(&bkey[e_sa[j]], cand); ch = true; }
                }
                if ((live >> (2 * j)) & 2u) {
                    const unsigned long long hi = max(e_w[j], ka >> 32), cand = (hi << 32) | (ka & 0xffffffffull);
                    if (cand < kb) { atomicMin(&bkey[e_sb[j]], cand); ch = true; }
                }
            }
            if (!wg_or(ch)) break;
        }
    }
    PF_STAMP(6);
    // ---- S6: outputs.  The seeds in use (ring positions) get compact indices 0 .. NS-1 inside the tile; OCEAN stays 255.
    // The hash arrays are free again: seed map | in-use flags | seed-pair hash
    uint8_t *cmap = reinterpret_cast<uint8_t *>(hkv + 2 * NBMAX + NBMAX / 4);   // [256]   (behind bkey and btype)
    uint32_t *used = hkv + 2 * NBMAX + NBMAX / 4 + 64;                            // [256]
    uint32_t *sk = used + 256, *sv = sk + SE;                                      // [SE] each
    for (int i = t; i < 256; i += NT) used[i] = 0u;
    for (int i = t; i < SE; i += NT) { sk[i] = EMPTY; sv[i] = EMPTY; }
    __syncthreads();
    for (int sI = t; sI < NB; sI += NT) {
        const uint32_t lab = (uint32_t)(bkey[sI] & 0xffu);
        if (lab != (uint32_t)OCEAN) used[lab] = 1u;
    }
    __syncthreads();
    if (t < 256) {
        const bool u = used[t] != 0u;
        const unsigned long long m = __ballot(u);
        if (wc == 0) s_scan[q] = __builtin_popcountll(m);
        const int within = __builtin_popcountll(m & ((1ull << wc) - 1ull));
        used[t] = u ? (uint32_t)within | 0x100u : 0u;       // bit 8: in use; low bits: rank inside the wavefront
    }
    __syncthreads();
    const int NS = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    if (t < 256) {
        int off = 0;
        for (int w2 = 0; w2 < (t >> 6); ++w2) off += s_scan[w2];
        cmap[t] = (used[t] & 0x100u) ? (uint8_t)(off + (int)(used[t] & 0xffu)) : (uint8_t)NOLAB;
        if (t == OCEAN) cmap[t] = (uint8_t)OCEAN;
    }
    __syncthreads();
    if (NS > NSMAX) {   // block-uniform
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
    for (int sI = t; sI < NB; sI += NT) {
        a.tabV[(size_t)tile * NBMAX + sI] = (uint32_t)(bkey[sI] >> 32);
        a.tabL[(size_t)tile * NBMAX + sI] = cmap[(int)(bkey[sI] & 0xffu)];
    }
    if (t == 0) a.tileNB[tile] = NB;
    const __amdgpu_buffer_rsrc_t rslot = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bslot + (g.r0 + wr0) * W + g.c0), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int wr = wr0 + k;
        const uint8_t c = cls(k);
        if (!c || (c & C_BORDER)) continue;
        const int sI = ptr[wr * WN + wc];
        if (c & C_RING) a.ringLab[(size_t)tile * 256 + ring_pos(wr, wc)] = cmap[(int)(bkey[sI] & 0xffu)];
        else __builtin_amdgcn_raw_buffer_store_b16((short)sI, rslot, wc * 2, k * Wi * 2, 0);   // (an owned cell: inside the raster)
    }
    PF_STAMP(7);
    // min spill elevation between pairs of seeds
    bool ov2 = false;
    for (int e = t; e < NE; e += NT) {
        const uint32_t key = ek[e];
        const int sa = (int)(key >> 10), sb = (int)(key & 1023u);
        const unsigned long long ka = bkey[sa], kb = bkey[sb];
        if ((ka & 0xffu) == (kb & 0xffu)) continue;
        const uint32_t la = cmap[(int)(ka & 0xffu)], lb = cmap[(int)(kb & 0xffu)];
        const uint32_t ww = max(ew[e], max((uint32_t)(ka >> 32), (uint32_t)(kb >> 32)));
        const uint32_t k2 = la < lb ? (la << 8 | lb) : (lb << 8 | la);
        unsigned hh = (k2 * 2654435761u) >> 23;
        bool done = false;
#pragma nounroll
        for (int probe = 0; probe < 64; ++probe) {
            const uint32_t prev = atomicCAS(&sk[hh], EMPTY, k2);
            if (prev == EMPTY || prev == k2) {
                atomicMin(&sv[hh], ww);
                done = true;
                break;
            }
            hh = (hh + 1) & (SE - 1);
        }
        if (!done) ov2 = true;
    }
    __syncthreads();
    for (int h = t; h < SE; h += NT) {
        if (sk[h] == EMPTY) continue;
        const int i = atomicAdd(&s_cnt, 1);
        if (i < SPMAX) a.spill[(size_t)tile * SPMAX + i] = ((unsigned long long)sk[h] << 32) | sv[h];
        else ov2 = true;
    }
    if (wg_or(ov2)) {
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
    for (int i = s_cnt + t; i < SPMAX; i += NT) a.spill[(size_t)tile * SPMAX + i] = ~0ull;
    if (t == 0) a.tileNS[tile] = s_cnt;
    PF_STAMP(8);
#ifdef PF_PROFILE
    if (t == 0) {
        atomicAdd(&a.prof[9], (unsigned long long)NB);
        atomicAdd(&a.prof[10], (unsigned long long)NE);
        atomicAdd(&a.prof[13], (unsigned long long)s_cnt);
        atomicAdd(&a.prof[11], (unsigned long long)NS);
    }
#endif
}

// ---- K2: the two seeds of a ring cell (its own tile's and its owner's) are joined at the owner's fill level of the cell.
// In two passes since round 4.  Until then one kernel gathered, per tile, the ring cells of all NINE windows around it (a cell
// concerns a tile as the window's tile or as its owner): 2 x 252 useful entries out of 9 x 252 candidates, nine dependent gather
// chains per thread, the ring COLUMNS a sector per cell -- 0.60 ms, 12 B of traffic per raster cell.  Split: (A) every tile resolves
// its OWN ring once (one entry per thread, one gather chain: slot -> owner's tables) into a record per ring position; (B) a tile's
// links are its own 252 records + the 252 records of its eight neighbours whose cells it owns -- four contiguous runs of 62 and
// four corners, all coalesced -- through the same LDS hash as before.  Same links, bit for bit (the tests that pin the flood pin them);
// 0.60 -> 0.54 ms (fill 7.51 -> 7.45 ms, same box, alternating): pass A is still one chain of three dependent gathers per ring cell.
__global__ __launch_bounds__(256) void pf_ring_kernel(PfArgs a)
{
    const int t = threadIdx.x;
    const int tile = blockIdx.x, ti = tile / a.ntc, tj = tile - ti * a.ntc;
    unsigned long long rec = ~0ull;
    if (t < 252) {
        int wr, wc;
        ring_cell(t, wr, wc);
        const int64_t r = (int64_t)ti * TI + wr, c = (int64_t)tj * TI + wc;
        if (!(r <= 0 || r >= a.H - 1 || c <= 0 || c >= a.W - 1)) {          // inside the raster and not a raster border cell
            const int oi = (int)((r - 1) / TI), oj = (int)((c - 1) / TI);     // the tile that owns the cell
            const int ot = oi * a.ntc + oj;
            const uint32_t labX = a.ringLab[(size_t)tile * 256 + t];
            const int so = min((int)a.bslot[r * a.W + c], NBMAX - 1);          // (a tile that gave up on a capacity wrote no slots: stay inside the tables)
            const uint32_t labO = a.tabL[(size_t)ot * NBMAX + so];
            const uint32_t w = max(dem_key(a.dem[r * a.W + c]), a.tabV[(size_t)ot * NBMAX + so]);
            rec = ((unsigned long long)labX << 40) | ((unsigned long long)labO << 32) | w;
        }
    }
    a.ringRec[(size_t)tile * 256 + t] = rec;
}

__global__ __launch_bounds__(256) void pf_link2_kernel(PfArgs a)
{
    __shared__ uint32_t lk[LH], lw[LH];
    __shared__ int s_cnt;
    const int t = threadIdx.x;
    const int tile = blockIdx.x, ti = tile / a.ntc, tj = tile - ti * a.ntc;
    for (int i = t; i < LH; i += 256) { lk[i] = EMPTY; lw[i] = EMPTY; }
    if (t == 0) s_cnt = 0;
    // this thread's two entries: one of the tile's own ring (e < 252), one of a neighbour's ring cell the tile owns
    unsigned long long rec[2] = {~0ull, ~0ull};
    int dirs[2] = {4, 4};
    bool own[2] = {true, false};
    if (t < 252) {
        rec[0] = a.ringRec[(size_t)tile * 256 + t];
        int wr, wc;
        ring_cell(t, wr, wc);
        const int di = wr == 0 ? -1 : (wr == WN - 1 ? 1 : 0), dj = wc == 0 ? -1 : (wc == WN - 1 ? 1 : 0);     // where the cell's owner lies
        dirs[0] = (di + 1) * 3 + (dj + 1);
        // the neighbour (di2, dj2) and the ring position of its window that is a cell of MINE
        const int q = t;
        int di2, dj2, p;
        if (q < 62) { di2 = -1; dj2 = 0; p = WN + 1 + q; }                               // above: its bottom ring row, columns 1 .. 62
        else if (q < 124) { di2 = 1; dj2 = 0; p = 1 + (q - 62); }                         // below: its top ring row
        else if (q < 186) { di2 = 0; dj2 = -1; p = 2 * WN + (WN - 2) + (q - 124); }       // left: its right ring column, rows 1 .. 62
        else if (q < 248) { di2 = 0; dj2 = 1; p = 2 * WN + (q - 186); }                   // right: its left ring column
        else if (q == 248) { di2 = -1; dj2 = -1; p = WN + WN - 1; }                       // the four corners
        else if (q == 249) { di2 = -1; dj2 = 1; p = WN; }
        else if (q == 250) { di2 = 1; dj2 = -1; p = WN - 1; }
        else { di2 = 1; dj2 = 1; p = 0; }
        const int xi = ti + di2, xj = tj + dj2;
        if (xi >= 0 && xi < a.ntr && xj >= 0 && xj < a.ntc) rec[1] = a.ringRec[(size_t)(xi * a.ntc + xj) * 256 + p];
        dirs[1] = (di2 + 1) * 3 + (dj2 + 1);
    }
    __syncthreads();
    bool ov = false;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (rec[u] == ~0ull) continue;
        const uint32_t labX = (uint32_t)(rec[u] >> 40) & 0xffu, labO = (uint32_t)(rec[u] >> 32) & 0xffu, w = (uint32_t)rec[u];
        const uint32_t mylab = own[u] ? labX : labO, nlab = own[u] ? labO : labX;
        if (mylab == (uint32_t)OCEAN) continue;                            // OCEAN's level is fixed
        const uint32_t key = mylab << 16 | (uint32_t)dirs[u] << 8 | nlab;
        unsigned h = (key * 2654435761u) >> 22;
        bool done = false;
        for (int probe = 0; probe < 64; ++probe) {
            const uint32_t prev = atomicCAS(&lk[h], EMPTY, key);
            if (prev == EMPTY || prev == key) {
                atomicMin(&lw[h], w);
                done = true;
                break;
            }
            h = (h + 1) & (LH - 1);
        }
        if (!done) ov = true;
    }
    __syncthreads();
    for (int h = t; h < LH; h += 256) {
        if (lk[h] == EMPTY) continue;
        const int i = atomicAdd(&s_cnt, 1);
        if (i < LMAX) a.links[(size_t)tile * LMAX + i] = ((unsigned long long)lk[h] << 32) | lw[h];
        else ov = true;
    }
    if (__syncthreads_or(ov)) {
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
    for (int i = s_cnt + t; i < LMAX; i += 256) a.links[(size_t)tile * LMAX + i] = ~0ull;
    if (t == 0) {
        a.tileNL[tile] = s_cnt;
        a.tileNL0[tile] = s_cnt;
    }
    // start values of the solve: +inf for every seed (OCEAN is not stored: its level is below everything)
    if (t < NSMAX) a.Lv[(size_t)tile * NSMAX + t] = EMPTY;
}

// ---- K3: one round of the seed-graph solve; a workgroup visits a BLOCK of BT x BT tiles --------------------------------------
// The levels L of the block's seeds and of the ring of tiles around it (a (BT+2)^2 region) sit in LDS, the block's edges (links
// and spill edges, as directed relaxations dst <- max(w, src)) in registers; the block iterates to ITS fixed point there, with
// the ring frozen, and writes its levels back.  Information crosses BT tiles per round (the per-tile version of this kernel
// needed ~225 rounds at 16384^2 and was bound by the latency of ~25 small dependent global accesses per visit).
// Worklist: a list of blocks per round, double buffered; a block that lowered a seed with a link into a neighbouring block
// appends that block to the next round's list (one atomicExch on its mark word + one atomicAdd on the list's counter, a few
// thousand per round).  Levels move with plain loads / stores: a value written in this round may or may not be seen by
// a neighbour visited in the same round (it is only ever lower: harmless), and that neighbour is visited again in the next
// round -- a new launch -- where it is seen for sure.
constexpr int BT = 4;                      // tiles per block edge
constexpr int RT = BT + 2;                 // region edge (block + ring)
constexpr int EMAX = 6144;                 // directed relaxations held in LDS per block (the rest is streamed from memory)
struct SolveArgs {
    PfArgs a;
    unsigned int *mark_cur, *mark_nxt;     // [nbr * nbc] "is in the list of this / the next round"
    const int *list_cur;                   // the blocks of this round ...
    const unsigned int *count_cur;         // ... and how many
    int *list_nxt;
    unsigned int *count_nxt;
    unsigned long long *visits;
    int nbr, nbc;                          // blocks
    int first;                             // 1: every block
    unsigned long long *eblk;              // [nbr * nbc][EMAX] the directed relaxations of every block, packed once (pf_pack_kernel)
    int row0;                              // pf_pack_kernel: first block row of the launch
    int *ecount;                           // [nbr * nbc] used words of a block's packed relaxations
};

constexpr int ST = 512;                   // threads per block visit

// The graph does not change during the solve: the directed relaxations of a block (its 16 tiles' links and spill edges, with the
// positions of their two seeds in the block's 6 x 6 region of levels) are packed ONCE into EMAX words per block, unused words ~0.
// A visit then loads its relaxations with the same round trip as the levels -- decoding them from the per-tile tables took three
// dependent round trips and 48 KB of LDS at every one of the ~9 visits of a block.  (A row band re-packs its first and last
// block rows after a halo exchange changed their links.)
__global__ __launch_bounds__(ST) void pf_pack_kernel(SolveArgs sa)
{
    __shared__ int s_seg[BT * BT + 1], s_nl[BT * BT];
    const PfArgs &a = sa.a;
    const int t = threadIdx.x;
    const int bi = sa.row0 + (int)blockIdx.y, bj = (int)blockIdx.x, blk = bi * sa.nbc + bj;
    unsigned long long *E = sa.eblk + (size_t)blk * EMAX;
    for (int i = t; i < EMAX; i += ST) E[i] = ~0ull;
    if (t < BT * BT) {
        const int p = bi * BT + t / BT, q = bj * BT + t % BT;
        int nl = 0, ns = 0;
        if (p < a.ntr && q < a.ntc) {     // (zero for a tile that gave up on a capacity: PfRun::begin clears the counts)
            nl = min(max(a.tileNL[p * a.ntc + q], 0), LMAX);
            ns = min(max(a.tileNS[p * a.ntc + q], 0), SPMAX);
        }
        s_nl[t] = nl;
        s_seg[t + 1] = nl + 2 * ns;
    }
    __syncthreads();      // (also orders the ~0 fill before the entries below: same workgroup, same addresses)
    if (t == 0) {
        s_seg[0] = 0;
        for (int k = 0; k < BT * BT; ++k) s_seg[k + 1] += s_seg[k];
    }
    __syncthreads();
    if (s_seg[BT * BT] > EMAX) {   // block-uniform; more relaxations than a visit holds (6144 ~ 16 x 380): tell the caller to fall back
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
    if (t == 0 && sa.ecount) sa.ecount[blk] = s_seg[BT * BT];
    for (int idx = t; idx < BT * BT * (LMAX + SPMAX); idx += ST) {
        const int bt = idx / (LMAX + SPMAX), e0 = idx - bt * (LMAX + SPMAX);
        const int p = bi * BT + bt / BT, q = bj * BT + bt % BT;
        if (p >= a.ntr || q >= a.ntc) continue;
        const int tile = p * a.ntc + q;
        const int nl = s_nl[bt], seg = s_seg[bt], ns = (s_seg[bt + 1] - seg - nl) / 2;
        const int base = ((bt / BT + 1) * RT + bt % BT + 1) * NSMAX;     // this tile's levels in the region
        if (e0 < LMAX) {
            const int e = e0;
            if (e >= nl) continue;
            const unsigned long long v = a.links[(size_t)tile * LMAX + e];
            const uint32_t key = (uint32_t)(v >> 32);
            const int mylab = (int)(key >> 16), dir = (int)((key >> 8) & 0xffu), nlab = (int)(key & 0xffu);
            const int src = nlab == OCEAN ? 0xFFFF : ((bt / BT + dir / 3) * RT + bt % BT + dir % 3) * NSMAX + nlab;
            E[seg + e] = ((unsigned long long)(base + mylab) << 48) | ((unsigned long long)src << 32) | (uint32_t)v;
        } else {
            const int e = e0 - LMAX;
            if (e >= ns) continue;
            const unsigned long long v = a.spill[(size_t)tile * SPMAX + e];
            const uint32_t key = (uint32_t)(v >> 32);
            const int la = (int)(key >> 8), lb = (int)(key & 0xffu);
            const int ia = la == OCEAN ? 0xFFFF : base + la, ib = lb == OCEAN ? 0xFFFF : base + lb;
            E[seg + nl + 2 * e] = la != OCEAN ? ((unsigned long long)ia << 48) | ((unsigned long long)ib << 32) | (uint32_t)v : ~0ull;
            E[seg + nl + 2 * e + 1] = lb != OCEAN ? ((unsigned long long)ib << 48) | ((unsigned long long)ia << 32) | (uint32_t)v : ~0ull;
        }
    }
}

__global__ __launch_bounds__(ST) __attribute__((amdgpu_waves_per_eu(6, 6))) void pf_solve_kernel(SolveArgs sa)
{
    __shared__ uint32_t L[RT * RT * NSMAX];            // 18 KB: levels of the region, tile (ri, rj) at (ri * RT + rj) * NSMAX
    __shared__ uint32_t Lold[BT * BT * NSMAX];         //  8 KB: the block's levels as loaded
    __shared__ int s_wake;
    const PfArgs &a = sa.a;
    const int t = threadIdx.x;
    // a resident grid walks the round's list: launching one workgroup per block of the raster (4489 x 1024 threads at 16384^2,
    // nearly all of them only to find their block inactive) cost more than the visits themselves
    const unsigned int nwork = sa.first ? (unsigned int)(sa.nbr * sa.nbc) : *sa.count_cur;
    for (unsigned int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    __syncthreads();                                   // the LDS arrays of the previous visit are free
    const int blk = sa.first ? (int)wi : sa.list_cur[wi], bi = blk / sa.nbc, bj = blk - bi * sa.nbc;
    int tv = t;                                        // (a new name per visit: see pf_solve_queue_body -- the words' region coordinates stay out of scratch)
    asm volatile("" : "+v"(tv));
    if (t == 0) {
        sa.mark_cur[blk] = 0u;                         // this buffer is appended to again two rounds from now
        s_wake = 0;
    }
#ifdef PF_PROFILE
    long long pk0 = __builtin_amdgcn_s_memtime(), pk1 = 0, pk2 = 0;
    int pk_it = 0;
#endif
    // ---- levels of the region (a thread's nine words in flight together, next to the relaxations: as a rolled loop with its bounds
    // branch every word was loaded, waited for and written to LDS before the next one went out -- nine round trips per visit)
    {
        constexpr int NLV = RT * RT * NSMAX / ST;
        static_assert(RT * RT * NSMAX % ST == 0, "whole words per thread");
        uint32_t lv[NLV];
#pragma unroll
        for (int u = 0; u < NLV; ++u) {
            const int i = tv + u * ST;
            const int rt = i / NSMAX, k = i - rt * NSMAX;
            const int p = bi * BT + rt / RT - 1, q = bj * BT + rt % RT - 1;
            const bool ok = p >= 0 && p < a.ntr && q >= 0 && q < a.ntc;
            lv[u] = EMPTY;
            if (ok) lv[u] = a.Lv[(size_t)(p * a.ntc + q) * NSMAX + k];
        }
#pragma unroll
        for (int u = 0; u < NLV; ++u) {       // (the block's own levels also as loaded, in front of the barrier: see pf_solve_queue_body)
            const int i = tv + u * ST;
            const int rt = i / NSMAX, k = i - rt * NSMAX, ri = rt / RT, rj = rt - ri * RT;
            L[i] = lv[u];
            if (ri >= 1 && ri <= BT && rj >= 1 && rj <= BT) Lold[((ri - 1) * BT + rj - 1) * NSMAX + k] = lv[u];
        }
    }
    // ---- the block's relaxations: registers for the whole visit (EMAX / ST = 12 per thread); in flight together with the levels
    unsigned long long er[EMAX / ST];
#pragma unroll
    for (int k = 0; k < EMAX / ST; ++k) er[k] = sa.eblk[(size_t)blk * EMAX + t + k * ST];
    __syncthreads();
#ifdef PF_PROFILE
    pk1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- the block's fixed point (only the levels are read from LDS)
    for (int it = 0; it < BT * BT * NSMAX; ++it) {
#ifdef PF_PROFILE
        ++pk_it;
#endif
        bool ch = false;
#pragma unroll
        for (int k = 0; k < EMAX / ST; ++k) {
            const unsigned long long r = er[k];
            if (r == ~0ull) continue;
            const int dst = (int)(r >> 48), src = (int)((r >> 32) & 0xffffu);
            const uint32_t v = max((uint32_t)r, src == 0xFFFF ? 0u : L[src]);
            if (v < L[dst]) {
                atomicMin(&L[dst], v);
                ch = true;
            }
        }
        if (!__syncthreads_or(ch)) break;
    }
#ifdef PF_PROFILE
    pk2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- write back; wake the neighbouring blocks that hold a link to a seed whose level dropped
    bool moved = false;
    for (int i = tv; i < BT * BT * NSMAX; i += ST) {
        const int bt = i / NSMAX, k = i - bt * NSMAX;
        const int p = bi * BT + bt / BT, q = bj * BT + bt % BT;
        const uint32_t v = L[((bt / BT + 1) * RT + bt % BT + 1) * NSMAX + k];
        if (v < Lold[i] && p < a.ntr && q < a.ntc) {
            a.Lv[(size_t)(p * a.ntc + q) * NSMAX + k] = v;
            moved = true;
        }
    }
    if (__syncthreads_or(moved)) {
        unsigned wake = 0;
#pragma unroll
        for (int k = 0; k < EMAX / ST; ++k) {
            const unsigned long long r = er[k];
            const int dst = (int)(r >> 48), src = (int)((r >> 32) & 0xffffu);
            if (src == 0xFFFF) continue;                                       // also the ~0 placeholders
            const int st = src / NSMAX, si = st / RT, sj = st % RT;             // region tile of the source
            if (si >= 1 && si <= BT && sj >= 1 && sj <= BT) continue;          // inside the block
            const int dt = dst / NSMAX, di_ = dt / RT - 1, dj_ = dt % RT - 1;   // block-local tile of the destination
            if (L[dst] < Lold[(di_ * BT + dj_) * NSMAX + dst % NSMAX]) {
                const int wi2 = si == 0 ? 0 : (si == RT - 1 ? 2 : 1), wj = sj == 0 ? 0 : (sj == RT - 1 ? 2 : 1);
                wake |= 1u << (wi2 * 3 + wj);
            }
        }
        if (wake) atomicOr(&s_wake, (int)wake);
        __syncthreads();
        wake = (unsigned)s_wake;
        if (t < 9 && ((wake >> t) & 1u)) {
            const int p = bi + t / 3 - 1, q = bj + t % 3 - 1;
            if (p >= 0 && p < sa.nbr && q >= 0 && q < sa.nbc) {
                const int nb = p * sa.nbc + q;
                if (atomicExch(&sa.mark_nxt[nb], 1u) == 0u) sa.list_nxt[atomicAdd(sa.count_nxt, 1u)] = nb;
            }
        }
    }
    if (t == 0) atomicAdd(sa.visits, 1ull);
#ifdef PF_PROFILE
    if (t == 0) {
        const long long pk3 = __builtin_amdgcn_s_memtime();
        atomicAdd(&a.prof[19], (unsigned long long)(pk1 - pk0));
        atomicAdd(&a.prof[21], (unsigned long long)(pk2 - pk1));
        atomicAdd(&a.prof[22], (unsigned long long)(pk3 - pk2));
        atomicAdd(&a.prof[23], (unsigned long long)pk_it);
    }
#endif
    }   // the round's list
}

// ---- K3 as ONE launch: the same block visits, taken from a queue instead of from a list per round ---------------------------------
// The rounds above are a Jacobi schedule: 53 launches at 16384^2 (33 of ~70 us while the levels travel from the raster border
// inward one block per launch, then a chain of ~30 us launches with a handful of blocks each), 2.85 ms with most of the chip idle.
// Here a resident grid takes block visits from ONE queue in global memory: a visit that lowered a seed its neighbour holds a link
// to appends that neighbour (once: the mark word) and the next free workgroup visits it at once -- a hop costs a visit, not a
// launch.  What crosses workgroups inside the launch:
//   * levels: lowered with agent-scope atomic MINs (performed at the memory side: visible to every XCD), drained (`s_waitcnt vmcnt(0)` in every wave, workgroup
//     barrier) BEFORE the wake-ups; read with sc1 loads (never L1) AFTER the visit's own mark was cleared by an agent-scope
//     atomic -- a level that changes after it was read finds the mark cleared and queues the block again.  A stale read can only
//     be an older, HIGHER level: the solve stays an upper bound, and the flood's result is proven cell by cell afterwards
//     (check.hip) whatever this kernel did;
//   * the queue: tickets from two agent-scope counters, a slot carries (generation | block) so that nobody resets slots; a visit
//     hands out its wake-up tickets (`tail`) BEFORE it counts itself `finished`, so "finished == tail" -- `finished` read first,
//     then `tail`: every ticket below that value was done when the first load returned, nobody was left to hand out another --
//     is only ever true when the solve is over.
// Every spin is bounded: a workgroup that waited PFQ_SPIN_LIMIT polls raises `abort`, everybody leaves and the host runs the
// rounds above instead.  No workgroup waits for another to be resident (a ticket beyond the tail is only waited for while
// finished < tail, i.e. while a RUNNING visit may still produce it).
struct PfQueue {                   // (every counter on a line of its own: same-line atomics queue up behind each other at ~12 ns apiece --
    unsigned int head;             //  with head, tail and a pending count in one line the 45 k visits of the benchmark waited 2 ms for them)
    unsigned int pad0[63];         // tickets taken
    unsigned int tail;             // tickets handed out
    unsigned int pad1[63];
    unsigned int finished;         // visits completed: the solve is over when finished == tail (read in THAT order: see the kernel)
    unsigned int pad2[63];
    unsigned int abort;            // a spin ran out, or the generations did
    unsigned int cap_mask;         // slots - 1 (a power of two >= 2 * blocks + grid)
    unsigned int flags[2];         // the flood's flags (PfArgs::flags: [0] a capacity overflowed, [1] the proof failed) ...
    unsigned int mm[2];            // ... and the DEM's smallest | largest elevation key (pf_minmax_kernel): they live HERE so that ONE copy
    unsigned int pad3[58];         // of the header brings everything the host wants to know after the solve (six 4-byte copies cost 25 us each)
    unsigned int slots[1];         // [cap]: (ticket / cap + 1) << 20 | block
};
static_assert(offsetof(PfQueue, slots) == 1024, "header = four lines of 256 bytes");
constexpr unsigned int PFQ_SPIN_LIMIT = 1u << 21;    // polls of ~0.3-1 us each
constexpr int PFQ_BLK_BITS = 20;

__device__ __forceinline__ unsigned int ld_sc1(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: the blocks that hold a tile on the outline of the block grid (the only ones with an edge to OCEAN -- the raster border,
// or a band's halo rows), in ring order; mode 1: the block rows `row_a` and `row_b` (a band whose halo links changed).  A block
// nobody ever wakes holds no seed that a level could reach: its levels stay at +inf, as they would in the rounds.
__global__ void pf_queue_init_kernel(PfQueue *q, unsigned int *mark, int nbr, int nbc, int mode, int row_a, int row_b, unsigned int cap_mask)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int n, blk;
    if (mode == 0) {
        const int h = nbr, w = nbc;
        n = (h == 1 || w == 1) ? h * w : 2 * (w - 1) + 2 * (h - 1);
        if (i >= n) return;
        int bi, bj;
        if (h == 1 || w == 1) { bi = w == 1 ? i : 0; bj = w == 1 ? 0 : i; }
        else if (i < w - 1) { bi = 0; bj = i; }
        else if (i < (w - 1) + (h - 1)) { bi = i - (w - 1); bj = w - 1; }
        else if (i < 2 * (w - 1) + (h - 1)) { bi = h - 1; bj = w - 1 - (i - (w - 1) - (h - 1)); }
        else { bi = h - 1 - (i - 2 * (w - 1) - (h - 1)); bj = 0; }
        blk = bi * nbc + bj;
    } else {
        const int two = row_b != row_a ? 2 : 1;
        n = two * nbc;
        if (i >= n) return;
        blk = (i < nbc ? row_a : row_b) * nbc + (i < nbc ? i : i - nbc);
    }
    mark[blk] = 1u;
    q->slots[i] = (1u << PFQ_BLK_BITS) | (unsigned int)blk;
    if (i == 0) {
        q->head = 0u;
        q->tail = (unsigned int)n;
        q->finished = 0u;
        q->abort = 0u;
        q->cap_mask = cap_mask;
    }
}

__device__ __forceinline__ void pf_solve_queue_body(const SolveArgs &sa, PfQueue *q, const int *__restrict__ ecount)
{
    __shared__ uint32_t L[RT * RT * NSMAX + 4];        // 18 KB: levels of the region, tile (ri, rj) at (ri * RT + rj) * NSMAX; L[LSENT] = 0: OCEAN,
    constexpr int LSENT = RT * RT * NSMAX;             // and the destination of a placeholder (nothing is ever below 0): no tests in the sweeps
    __shared__ uint32_t Lold[BT * BT * NSMAX];         //  8 KB: the block's levels as loaded
    __shared__ int s_wake, s_blk, s_or[3];
    const PfArgs &a = sa.a;
    const int t = threadIdx.x;
    // (one barrier per "does any thread say yes", three flag words taking turns: see pf_tile_kernel; the library's
    // __syncthreads_or is a reduction through LDS with two)
    int or_turn = 0;
    auto wg_or = [&](bool p) -> bool {
        if (__any(p) && (t & 63) == 0) s_or[or_turn] = 1;
        __syncthreads();
        const bool r = s_or[or_turn] != 0;
        if (t == 0) s_or[or_turn == 0 ? 2 : or_turn - 1] = 0;
        or_turn = or_turn == 2 ? 0 : or_turn + 1;
        return r;
    };
    if (t == 0) { s_or[0] = 0; s_or[1] = 0; s_or[2] = 0; L[LSENT] = 0u; }      // (in front of the first barrier of the loop below)
    const unsigned int cap_mask = q->cap_mask, cap_shift = (unsigned int)__builtin_popcount(cap_mask);
    for (;;) {
        __syncthreads();                               // the LDS arrays (and s_blk) of the previous visit are free
        if (t == 0) {
            int blk = -1;
            const unsigned int ticket = atomicAdd(&q->head, 1u);
            const unsigned int gen = (ticket >> cap_shift) + 1u;
            if (gen >= (1u << (32 - PFQ_BLK_BITS)) - 1u) atomicExch(&q->abort, 1u);       // (never: ~4000 x the blocks in visits)
            else {
                const unsigned int *slot = &q->slots[ticket & cap_mask];
                for (unsigned int n = 0;; ++n) {
                    const unsigned int v = ld_sc1(slot);
                    if ((v >> PFQ_BLK_BITS) == gen) { blk = (int)(v & ((1u << PFQ_BLK_BITS) - 1u)); break; }
                    if ((n & 7u) == 7u) {
                        const unsigned int fin = ld_sc1(&q->finished);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // `finished` first, THEN `tail`
                        if (fin == ld_sc1(&q->tail) || ld_sc1(&q->abort)) break;
                    }
                    if (n > PFQ_SPIN_LIMIT) { atomicExch(&q->abort, 1u); break; }
                    if (n < 64u) __builtin_amdgcn_s_sleep(8);
                    else __builtin_amdgcn_s_sleep(48);
                }
            }
            // mark word of a block: bit 0 = queued, or (while bit 1 is up) woken again; bit 1 = a visit of it is running.  Returned
            // before the barrier below: a level that changes from now on finds bit 0 clear and sets it
            if (blk >= 0) (void)atomicExch(&sa.mark_cur[blk], 2u);
            s_blk = blk;
            s_wake = 0;
        }
        __syncthreads();
        const int blk = s_blk;
        if (blk < 0) return;                           // (block-uniform) the solve is over, or was called off
        const int bi = blk / sa.nbc, bj = blk - bi * sa.nbc;
        // (a new name for the thread's number per visit: or the region coordinates of its nine words -- a dozen values that only depend
        // on it -- are computed once in front of the visit loop and live in SCRATCH through it: every one of the nine level loads then
        // waited for a scratch reload of its address first, nine dependent round trips per visit)
        int tv = t;
        asm volatile("" : "+v"(tv));
        // ---- levels of the region: sc1 loads (another workgroup of this launch may have written them), all in flight together
        {
            constexpr int NLV = RT * RT * NSMAX / ST;
            uint32_t lv[NLV];
#pragma unroll
            for (int u = 0; u < NLV; ++u) {
                const int i = tv + u * ST;
                const int rt = i / NSMAX, k = i - rt * NSMAX;
                const int p = bi * BT + rt / RT - 1, qq = bj * BT + rt % RT - 1;
                const bool ok = p >= 0 && p < a.ntr && qq >= 0 && qq < a.ntc;
                lv[u] = EMPTY;
                if (ok) lv[u] = ld_sc1(&a.Lv[(size_t)(p * a.ntc + qq) * NSMAX + k]);
            }
            // (the block's own levels a second time, as loaded: what a seed is compared with at the end of the visit.  Written HERE, in
            // front of the barrier the sweeps start behind -- as a pass of its own between that barrier and the sweeps, a thread that
            // was through with its copies lowered a level before another thread had copied it: the drop was never written back, nobody
            // was woken, the proof failed -- seen in one step of six when the kernel was built for 64 registers, whose spills skew the
            // waves; the rounds' kernel had the same window since round 2)
#pragma unroll
            for (int u = 0; u < NLV; ++u) {
                const int i = tv + u * ST;
                const int rt = i / NSMAX, k = i - rt * NSMAX, ri = rt / RT, rj = rt - ri * RT;
                L[i] = lv[u];
                if (ri >= 1 && ri <= BT && rj >= 1 && rj <= BT) Lold[((ri - 1) * BT + rj - 1) * NSMAX + k] = lv[u];
            }
        }
        // ---- the block's relaxations (read-only during the solve): only the used part of the packed array
        const int ne = ecount[blk];
        unsigned long long er[EMAX / ST];
#pragma unroll
        for (int k = 0; k < EMAX / ST; ++k) {
            er[k] = ~0ull;
            if (t + k * ST < ne) er[k] = sa.eblk[(size_t)blk * EMAX + t + k * ST];
        }
        // (once per visit instead of two tests per relaxation and sweep: OCEAN as a source and both ends of a placeholder -> the word that holds 0)
#pragma unroll
        for (int k = 0; k < EMAX / ST; ++k) {
            const unsigned long long r = er[k];
            const unsigned src = (unsigned)(r >> 32) & 0xffffu;
            const unsigned long long dst = r == ~0ull ? (unsigned long long)LSENT : r >> 48;
            er[k] = (dst << 48) | ((unsigned long long)(src == 0xFFFFu ? (unsigned)LSENT : src) << 32) | (r & 0xffffffffull);
        }
        __syncthreads();
        // ---- the block's fixed point (only the levels are read from LDS)
        int dbg_sweeps = 0;
        for (int it = 0; it < BT * BT * NSMAX; ++it) {
            bool ch = false;
            ++dbg_sweeps;
#pragma unroll
            for (int k = 0; k < EMAX / ST; ++k) {
                const unsigned long long r = er[k];
                const int dst = (int)(r >> 48), src = (int)((r >> 32) & 0xffffu);
                const uint32_t v = max((uint32_t)r, L[src]);
                if (v < L[dst]) {
                    atomicMin(&L[dst], v);
                    ch = true;
                }
            }
            if (!wg_or(ch)) break;
        }
        // ---- write back (write-through), drain, then wake the neighbouring blocks that hold a link to a seed whose level dropped
        bool moved = false;
        for (int i = tv; i < BT * BT * NSMAX; i += ST) {
            const int bt = i / NSMAX, k = i - bt * NSMAX;
            const int p = bi * BT + bt / BT, qq = bj * BT + bt % BT;
            const uint32_t v = L[((bt / BT + 1) * RT + bt % BT + 1) * NSMAX + k];
            if (v < Lold[i] && p < a.ntr && qq < a.ntc) {
                // an agent-scope MIN, not a store: a block can be in two visits at once (woken again while a visit of it is still
                // running, taken by another workgroup), and the one that started from the older levels would put a seed back UP
                // over what the other has just written -- nobody is woken for that, and the flood's proof fails (seen: 4 of 8 runs)
                atomicMin(&a.Lv[(size_t)(p * a.ntc + qq) * NSMAX + k], v);
                moved = true;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every writing wave, before the barrier the wake-ups sit behind
        const bool dbg_moved = wg_or(moved);
        if (dbg_moved) {
            unsigned wake = 0;
#pragma unroll
            for (int k = 0; k < EMAX / ST; ++k) {
                const unsigned long long r = er[k];
                const int dst = (int)(r >> 48), src = (int)((r >> 32) & 0xffffu);
                if (src == LSENT) continue;                                        // OCEAN, and the placeholders
                const int st = src / NSMAX, si = st / RT, sj = st % RT;             // region tile of the source
                if (si >= 1 && si <= BT && sj >= 1 && sj <= BT) continue;          // inside the block
                const int dt = dst / NSMAX, di_ = dt / RT - 1, dj_ = dt % RT - 1;   // block-local tile of the destination
                if (L[dst] < Lold[(di_ * BT + dj_) * NSMAX + dst % NSMAX]) {
                    const int wi2 = si == 0 ? 0 : (si == RT - 1 ? 2 : 1), wj = sj == 0 ? 0 : (sj == RT - 1 ? 2 : 1);
                    wake |= 1u << (wi2 * 3 + wj);
                }
            }
            if (wake) atomicOr(&s_wake, (int)wake);
            __syncthreads();
            wake = (unsigned)s_wake;
            // one ticket range for all of this visit's wake-ups (a returning add: the tail is up before this visit counts as finished)
            bool fresh = false;
            int nb = 0;
            if (t < 9 && ((wake >> t) & 1u)) {
                const int p = bi + t / 3 - 1, qq = bj + t % 3 - 1;
                if (p >= 0 && p < sa.nbr && qq >= 0 && qq < sa.nbc) {
                    nb = p * sa.nbc + qq;
                    fresh = atomicOr(&sa.mark_cur[nb], 1u) == 0u;      // (2: a visit of it is running -- that visit queues the block again when it ends)
                }
            }
            if (t < 64) {
                const unsigned long long bal = __ballot(fresh);
                if (bal) {
                    unsigned int base = 0;
                    if (t == 0) base = atomicAdd(&q->tail, (unsigned int)__builtin_popcountll(bal));
                    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                    if (fresh) {
                        const unsigned int tk = base + (unsigned int)__builtin_popcountll(bal & ((1ull << t) - 1ull));
                        st_sc1(&q->slots[tk & cap_mask], (((tk >> cap_shift) + 1u) << PFQ_BLK_BITS) | (unsigned int)nb);
                    }
                }
            }
        }
        __syncthreads();
        if (sa.first == 2 && t == 0) {      // (MHIP_PF_DEBUG: visits that lowered a seed, sweeps)
            if (s_wake >= 0 && dbg_moved) atomicAdd(&q->pad2[0], 1u);
            atomicAdd(&q->pad2[1], (unsigned int)dbg_sweeps);
        }
        if (t == 0) {
            // woken while this visit ran (by a level that may have changed after it was loaded): once more, through the queue -- one
            // visit of a block at a time, none of them lost
            if (atomicAnd(&sa.mark_cur[blk], ~2u) & 1u) {
                const unsigned int tk = atomicAdd(&q->tail, 1u);
                st_sc1(&q->slots[tk & cap_mask], (((tk >> cap_shift) + 1u) << PFQ_BLK_BITS) | (unsigned int)blk);
            }
            atomicAdd(&q->finished, 1u);
        }
    }
}

// three workgroups per CU at up to 80 VGPRs.  (Four at 64 -- the loads' address arithmetic then spills a dozen words once per visit --
// were measured: fill 7.96 against 7.53 ms on the same box, tools/lab/qtest.sh; not kept.)
__global__ __launch_bounds__(ST) __attribute__((amdgpu_waves_per_eu(6, 6))) void pf_solve_queue_kernel(SolveArgs sa, PfQueue *q, const int *__restrict__ ecount)
{
    pf_solve_queue_body(sa, q, ecount);
}

// ---- K4: final level of every basin, then the raster ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void pf_final_kernel(PfArgs a)
{
    const int tile = blockIdx.x;
    const int nb = a.tileNB[tile];
    for (int s = threadIdx.x; s < nb; s += 256) {
        const size_t i = (size_t)tile * NBMAX + s;
        const int lab = a.tabL[i];
        a.tabV[i] = max(a.tabV[i], lab == OCEAN ? 0u : a.Lv[(size_t)tile * NSMAX + lab]);
    }
}

__global__ __launch_bounds__(256) void pf_apply_kernel(PfArgs a, float *__restrict__ filled, float *__restrict__ depths)
{
    // four cells per lane: one 16-byte load of the DEM, one 8-byte load of the basin slots, 16-byte stores (4-byte stores
    // cost 3.5x the bytes in write traffic here: partial lines).  Rows whose width is not a multiple of 4 take the scalar tail.
    const int64_t W = a.W, H = a.H;
    const int64_t groups = (W + 3) / 4;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups * H) return;
    const int64_t r = g / groups, c4 = (g - r * groups) * 4;
    if ((r == 0 && a.fixed_top) || (r == H - 1 && a.fixed_bot)) return;   // a band's halo rows belong to the neighbour
    const bool rowb = r == 0 || r == H - 1;
    const int ti = (int)((r - 1) / TI);
    const int64_t i0 = r * W + c4;
    const bool vec = (W & 3) == 0;     // then every group is whole and 16-byte aligned
    float d[4];
    uint16_t sl[4];
    if (vec) {
        const float4 dv = *reinterpret_cast<const float4 *>(a.dem + i0);
        const ushort4 sv = *reinterpret_cast<const ushort4 *>(a.bslot + i0);
        d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
        sl[0] = sv.x; sl[1] = sv.y; sl[2] = sv.z; sl[3] = sv.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = c4 + k < W;
            d[k] = in ? a.dem[i0 + k] : 0.0f;
            sl[k] = in ? a.bslot[i0 + k] : (uint16_t)0;
        }
    }
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c4 + k;
        f[k] = d[k];
        if (!(rowb || c == 0 || c >= W - 1)) {
            const int tj = (int)((c - 1) / TI);
            const uint32_t lev = a.tabV[(size_t)(ti * a.ntc + tj) * NBMAX + sl[k]];
            f[k] = key_f32(max(dem_key(d[k]), lev));
        }
    }
    if (vec) {
        *reinterpret_cast<float4 *>(filled + i0) = make_float4(f[0], f[1], f[2], f[3]);
        if (depths) *reinterpret_cast<float4 *>(depths + i0) = make_float4(f[0] - d[0], f[1] - d[1], f[2] - d[2], f[3] - d[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c4 + k < W) {
                filled[i0 + k] = f[k];
                if (depths) depths[i0 + k] = f[k] - d[k];
            }
    }
}

// ---- K4 fused with the run-time proof (check.hip) for rasters whose width is a multiple of 256 ---------------------------------
// pf_apply_kernel writes F and the separate check reads dem + F back (8 B/cell, 0.36 ms at 16384^2).  Here one wavefront walks a
// 256-column strip downwards, computes F of a row from dem + basin slots (as pf_apply_kernel does), keeps the horizontal minima of
// three rows in registers and evaluates the reference's update at every cell of the middle row before the row leaves: the check
// costs no memory traffic (the two rows above and below a row block are computed twice: 2 of 64).  A band's halo rows hold the
// neighbour's surface: they are loaded, not computed.
namespace {
constexpr int DPP_SR1 = 0x138, DPP_SL1 = 0x130;
__device__ __forceinline__ float pf_lane_left(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_SR1, 0xf, 0xf, true)); }
__device__ __forceinline__ float pf_lane_right(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_SL1, 0xf, 0xf, true)); }
__device__ __forceinline__ float pf_inf_if_nan(float v) { return v != v ? __builtin_inff() : v; }

struct PfRow {
    float f[4], d[4], h[4];     // surface, dem, min of the surface over columns c - 1, c, c + 1 (NaN counts as +inf)
};

// a row in two steps, so that the streaming loads of row r + 2 are in flight while the level tables of row r + 1 are gathered:
// pf_row_load issues dem + basin slots (+ the strip's neighbour column on lanes 0 / 63), pf_row_finish gathers the levels
struct PfRaw {
    float4 d;            // dem
    ushort4 s;           // basin slots (computed rows)
    float4 f;            // the surface as stored (a band's halo row)
    float ed;            // lanes 0 / 63: dem (or stored surface) of the neighbour column
    uint16_t es;         //               and its basin slot
    int64_t r;
    bool loaded;
};

__device__ __forceinline__ void pf_row_load(const PfArgs &a, const float *__restrict__ filled, int64_t rr, int64_t c0, int lane, PfRaw &o)
{
    const int64_t r = rr < 0 ? 0 : (rr >= a.H ? a.H - 1 : rr);
    o.r = r;
    o.loaded = (r == 0 && a.fixed_top) || (r == a.H - 1 && a.fixed_bot);     // (wave-uniform)
    const int64_t i0 = r * a.W + c0 + lane * 4;
    o.d = *reinterpret_cast<const float4 *>(a.dem + i0);
    o.s = make_ushort4(0, 0, 0, 0);
    o.f = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o.loaded) o.f = *reinterpret_cast<const float4 *>(filled + i0);
    else o.s = *reinterpret_cast<const ushort4 *>(a.bslot + i0);
    o.ed = 0.0f;
    o.es = 0;
    if (lane == 0 || lane == 63) {
        int64_t c = lane == 0 ? c0 - 1 : c0 + 256;
        c = c < 0 ? 0 : (c > a.W - 1 ? a.W - 1 : c);
        o.ed = o.loaded ? filled[r * a.W + c] : a.dem[r * a.W + c];
        if (!o.loaded) o.es = a.bslot[r * a.W + c];
    }
}

__device__ __forceinline__ void pf_row_finish(const PfArgs &a, const PfRaw &w, int64_t c0, int lane, PfRow &o)
{
    const int64_t r = w.r;
    o.d[0] = w.d.x; o.d[1] = w.d.y; o.d[2] = w.d.z; o.d[3] = w.d.w;
    const bool rowb = r == 0 || r == a.H - 1;
    const int ti = (int)((r - 1) / TI);
    float e = w.ed;
    if (w.loaded) {
        o.f[0] = w.f.x; o.f[1] = w.f.y; o.f[2] = w.f.z; o.f[3] = w.f.w;
    } else {
        const uint16_t sl[4] = {w.s.x, w.s.y, w.s.z, w.s.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t c = c0 + lane * 4 + k;
            o.f[k] = o.d[k];
            if (!(rowb || c == 0 || c >= a.W - 1)) {
                const int tj = (int)((c - 1) / TI);
                o.f[k] = key_f32(max(dem_key(o.d[k]), a.tabV[(size_t)(ti * a.ntc + tj) * NBMAX + sl[k]]));
            }
        }
        if (lane == 0 || lane == 63) {
            int64_t c = lane == 0 ? c0 - 1 : c0 + 256;
            c = c < 0 ? 0 : (c > a.W - 1 ? a.W - 1 : c);
            if (!(rowb || c == 0 || c >= a.W - 1))
                e = key_f32(max(dem_key(e), a.tabV[(size_t)(ti * a.ntc + (int)((c - 1) / TI)) * NBMAX + w.es]));
        }
    }
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = pf_inf_if_nan(o.f[k]);
    e = pf_inf_if_nan(e);
    const float l0 = pf_lane_left(g[3]), r0 = pf_lane_right(g[0]);
    const float l = lane == 0 ? e : l0, rt = lane == 63 ? e : r0;
    o.h[0] = fminf(fminf(l, g[0]), g[1]);
    o.h[1] = fminf(fminf(g[0], g[1]), g[2]);
    o.h[2] = fminf(fminf(g[1], g[2]), g[3]);
    o.h[3] = fminf(fminf(g[2], g[3]), rt);
}

constexpr int PF_RPW = 32;     // rows per wavefront of the last pass (64: 16 wavefronts per SIMD in two batches of eight -- a tail; 32: -0.09 ms)
__global__ __launch_bounds__(256) void pf_apply_check_kernel(PfArgs a, float *__restrict__ filled, float *__restrict__ depths, unsigned nbx,
                                                            unsigned int *flag)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned by = blockIdx.x / nbx, bx = blockIdx.x - by * nbx;
    const int64_t c0 = ((int64_t)bx * 4 + wave) * 256;
    if (c0 >= a.W) return;
    const int64_t r_begin = (int64_t)by * PF_RPW, r_end = r_begin + PF_RPW < a.H ? r_begin + PF_RPW : a.H;
    PfRow up, mid, dn;
    PfRaw w0, w1, w2;
    pf_row_load(a, filled, r_begin - 1, c0, lane, w0);
    pf_row_load(a, filled, r_begin, c0, lane, w1);
    pf_row_load(a, filled, r_begin + 1, c0, lane, w2);
    pf_row_finish(a, w0, c0, lane, up);
    pf_row_finish(a, w1, c0, lane, mid);
    bool bad = false;
    for (int64_t r = r_begin; r < r_end; ++r) {
        w1 = w2;
        pf_row_load(a, filled, r + 2, c0, lane, w2);      // in flight while row r + 1 gathers its levels
        pf_row_finish(a, w1, c0, lane, dn);
        const bool row_border = r == 0 || r == a.H - 1;
        const bool halo = (r == 0 && a.fixed_top) || (r == a.H - 1 && a.fixed_bot);   // a band's halo row belongs to the neighbour
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + lane * 4 + j;
            const float own = mid.f[j], d = mid.d[j];
            if (row_border || c == 0 || c == a.W - 1) {
                bad |= !halo && __float_as_uint(own) != __float_as_uint(d);
            } else {
                const float m9 = fminf(fminf(up.h[j], mid.h[j]), dn.h[j]);
                bad |= ((own > d) & (m9 < own)) | (own < d);
            }
        }
        if (!halo) {
            const int64_t i0 = r * a.W + c0 + lane * 4;
            *reinterpret_cast<float4 *>(filled + i0) = make_float4(mid.f[0], mid.f[1], mid.f[2], mid.f[3]);
            if (depths) *reinterpret_cast<float4 *>(depths + i0) = make_float4(mid.f[0] - mid.d[0], mid.f[1] - mid.d[1], mid.f[2] - mid.d[2], mid.f[3] - mid.d[3]);
        }
        up = mid;
        mid = dn;
    }
    if (__any(bad) && lane == 0) *flag = 1u;
}
}  // namespace

// ---- row bands ---------------------------------------------------------------------------------------------------------------
// A halo row holds the neighbouring band's CURRENT estimate of its filled edge row (an upper bound of the final surface that only
// ever drops; +inf before the first exchange).  To this band a halo cell is a ring cell whose level is that estimate: for the
// seed s its ring position drains to,  L[s] <= estimate  -- a link from s to OCEAN with that weight.  The halo links of a tile
// follow its tileNL0 links between tiles and are rebuilt whenever a halo row changed; the solve kernel does not know the
// difference.  What this band owes its neighbours are the estimates of its own edge rows (pf_edge_rows_kernel).
__global__ __launch_bounds__(128) void pf_halo_links_kernel(PfArgs a, const float *__restrict__ filled)
{
    __shared__ uint32_t minw[NSMAX];
    __shared__ int s_n;
    const int t = threadIdx.x;
    const int tj = blockIdx.x % a.ntc, side = blockIdx.x / a.ntc;          // side 0: tile row 0, side 1: the last tile row
    if (side == 1 && a.ntr == 1) return;                                    // one tile row: side 0 does both halo rows
    const int ti = side == 0 ? 0 : a.ntr - 1;
    const int tile = ti * a.ntc + tj;
    if (t < NSMAX) minw[t] = EMPTY;
    if (t == 0) s_n = 0;
    __syncthreads();
    const int64_t c0 = (int64_t)tj * TI;
    for (int hs = 0; hs < 2; ++hs) {                                        // top halo row (ring row 0), bottom halo row (ring row 63)
        if (hs == 0 ? !(a.fixed_top && ti == 0) : !(a.fixed_bot && ti == a.ntr - 1)) continue;
        const int64_t r = hs == 0 ? 0 : a.H - 1;
        if (t < WN) {
            const int64_t c = c0 + t;
            if (c > 0 && c < a.W - 1) {                                     // raster border columns are OCEAN to K1 already
                const uint32_t lab = a.ringLab[(size_t)tile * 256 + ring_pos(hs == 0 ? 0 : WN - 1, t)];
                if (lab < (uint32_t)NSMAX) atomicMin(&minw[lab], dem_key(filled[r * a.W + c]));
            }
        }
    }
    __syncthreads();
    const int nl0 = a.tileNL0[tile];
    if (t < NSMAX && minw[t] != EMPTY && minw[t] < f32_key(__builtin_inff())) {
        const int i = nl0 + atomicAdd(&s_n, 1);
        if (i < LMAX) a.links[(size_t)tile * LMAX + i] = ((unsigned long long)((uint32_t)t << 16 | 4u << 8 | (uint32_t)OCEAN) << 32) | minw[t];
        else atomicOr(a.flags, 1u);
    }
    __syncthreads();
    const int n = min(nl0 + s_n, LMAX);
    for (int i = n + t; i < LMAX; i += 128) a.links[(size_t)tile * LMAX + i] = ~0ull;
    if (t == 0) a.tileNL[tile] = n;
}

// current estimate of the filled surface on local row r (an owned edge row): max(dem, V[basin], L[seed of the basin])
__global__ __launch_bounds__(256) void pf_edge_rows_kernel(PfArgs a, float *__restrict__ filled, int64_t r)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.W) return;
    const float d = a.dem[r * a.W + c];
    float f = d;
    if (c > 0 && c < a.W - 1) {
        const int tile = (int)((r - 1) / TI) * a.ntc + (int)((c - 1) / TI);
        const size_t i = (size_t)tile * NBMAX + a.bslot[r * a.W + c];
        const int lab = a.tabL[i];
        const uint32_t lev = max(a.tabV[i], lab == OCEAN ? 0u : a.Lv[(size_t)tile * NSMAX + lab]);
        const uint32_t k = max(dem_key(d), lev);
        f = k >= f32_key(__builtin_inff()) ? __builtin_inff() : key_f32(k);   // a seed nobody has reached yet: +inf
    }
    filled[r * a.W + c] = f;
}

__global__ void pf_append_row_kernel(unsigned int *mark, int *list, unsigned int *count, int first, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && atomicExch(&mark[first + i], 1u) == 0u) list[atomicAdd(count, 1u)] = first + i;
}
__global__ void pf_fill_f32_kernel(float *p, int64_t n, float v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

// the per-wavefront elevation keys of pf_tile_kernel -> out[0] = smallest, out[1] = largest (all ones: the DEM holds a NaN)
__global__ __launch_bounds__(256) void pf_minmax_kernel(const uint32_t *__restrict__ mm, int64_t npairs, uint32_t *out)
{
    __shared__ uint32_t lo_l[4], hi_l[4];
    uint32_t klo = 0xffffffffu, khi = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * 256) {
        const uint2 v = reinterpret_cast<const uint2 *>(mm)[i];
        klo = v.x < klo ? v.x : klo;
        khi = v.y > khi ? v.y : khi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo2 = (uint32_t)__shfl_xor((int)klo, o), hi2 = (uint32_t)__shfl_xor((int)khi, o);
        klo = lo2 < klo ? lo2 : klo;
        khi = hi2 > khi ? hi2 : khi;
    }
    if ((threadIdx.x & 63) == 0) { lo_l[threadIdx.x >> 6] = klo; hi_l[threadIdx.x >> 6] = khi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { klo = lo_l[k] < klo ? lo_l[k] : klo; khi = hi_l[k] > khi ? hi_l[k] : khi; }
        atomicMin(&out[0], klo);
        atomicMax(&out[1], khi);
    }
}

// ---- the resumable run -------------------------------------------------------------------------------------------------------
struct PfRun::Impl {
    DevBuf ws;
    PfArgs a;
    unsigned int *mark = nullptr;      // [2][nslots]
    int *list = nullptr;               // [2][nslots]
    unsigned int *any = nullptr;       // [round]: blocks in the list of that round
    unsigned long long *visits = nullptr;
    unsigned long long *eblk = nullptr;   // [nslots][EMAX] packed relaxations (pf_pack_kernel)
    int *ecount = nullptr;                // [nslots] used words of each
    PfQueue *queue = nullptr;             // the visit queue of pf_solve_queue_kernel
    size_t queue_bytes = 0;
    unsigned int queue_cap = 0;
    unsigned long long queue_visits = 0;
    bool rounds_ran = false;              // pf_solve_kernel (the fall-back / A-B partner) counted visits on the device
    bool solved_once = false;             // the first solve starts from the outline of the block grid, later ones from the band's first / last block rows
    int64_t ntiles = 0;
    size_t nslots = 0;
    int nbr = 0, nbc = 0, round = 0, launches = 0;
    bool halo_dirty = false;
    uint32_t *mmout = nullptr;          // [2] smallest | largest elevation key of the raster (pf_minmax_kernel)
    uint32_t h_mm[2] = {0xffffffffu, 0u};
    bool mm_valid = false;
    hipEvent_t k1_e0 = nullptr, k1_e1 = nullptr;    // around pf_tile_kernel (the stage's largest launch)
    ~Impl()
    {
        if (k1_e0) (void)hipEventDestroy(k1_e0);
        if (k1_e1) (void)hipEventDestroy(k1_e1);
    }
};
namespace {
constexpr int PF_MAXR = 1 << 14, PF_BATCH = 32;
}

PfRun::PfRun() : impl(new Impl) {}
PfRun::~PfRun() { delete impl; }

// the relaxations of the block rows [row0, row0 + nrows) into their packed form
int PfRun::pack(hipStream_t s, int row0, int nrows)
{
    Impl &m = *impl;
    SolveArgs sa = {};
    sa.a = m.a;
    sa.nbr = m.nbr;
    sa.nbc = m.nbc;
    sa.eblk = m.eblk;
    sa.ecount = m.ecount;
    sa.row0 = row0;
    hipLaunchKernelGGL(pf_pack_kernel, dim3((unsigned)m.nbc, (unsigned)nrows), dim3(ST), 0, s, sa);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

// the seed-graph solve: ONE launch that takes block visits from a queue (pf_solve_queue_kernel); the rounds below are the
// fall-back when that launch calls itself off (a spin that ran out) and the A/B partner (MHIP_PF_SOLVE=rounds, development)
int PfRun::solve(hipStream_t s)
{
    Impl &m = *impl;
    std::vector<unsigned int> h_cnt(PF_BATCH + 1);
    unsigned int h_flag = 0;
    static const bool use_rounds = [] { const char *e = dev_env("MHIP_PF_SOLVE"); return e && std::string(e) == "rounds"; }();
    if (!use_rounds) {
        const int mode = m.solved_once ? 1 : 0;
        MH_HIP(hipMemsetAsync(m.queue, 0, offsetof(PfQueue, abort), s));                     // head | tail | finished (+ the debug counters)
        MH_HIP(hipMemsetAsync(m.queue->slots, 0, (size_t)m.queue_cap * 4, s));
        const int n_init = mode == 0 ? ((m.nbr == 1 || m.nbc == 1) ? m.nbr * m.nbc : 2 * (m.nbc - 1) + 2 * (m.nbr - 1)) : 2 * m.nbc;
        hipLaunchKernelGGL(pf_queue_init_kernel, dim3((unsigned)cdiv(n_init, 256)), dim3(256), 0, s, m.queue, m.mark, m.nbr, m.nbc, mode, 0, m.nbr - 1, m.queue_cap - 1u);
        SolveArgs sa = {};
        sa.a = m.a;
        sa.mark_cur = m.mark;
        sa.visits = m.visits;
        sa.nbr = m.nbr;
        sa.nbc = m.nbc;
        sa.eblk = m.eblk;
        sa.first = dev_env("MHIP_PF_DEBUG") ? 2 : 0;
        hipLaunchKernelGGL(pf_solve_queue_kernel, dim3((unsigned)std::min<size_t>(m.nslots, 768)), dim3(ST), 0, s, sa, m.queue, (const int *)m.ecount);   // three resident per CU (512 / 640 workgroups: slower, round 4)
        MH_HIP(hipGetLastError());
        m.launches += 2;
        unsigned int h_hdr[256 * 4 / 4];          // the queue's header: ONE copy
        MH_HIP(hipMemcpyAsync(h_hdr, m.queue, sizeof(h_hdr), hipMemcpyDeviceToHost, s));
        MH_HIP(stream_sync(s));
        const PfQueue *hq = reinterpret_cast<const PfQueue *>(h_hdr);
        const unsigned int h_q[4] = {hq->head, hq->tail, hq->finished, hq->abort};
        h_flag = hq->flags[0];
        if (!m.mm_valid) { m.h_mm[0] = hq->mm[0]; m.h_mm[1] = hq->mm[1]; }
        m.mm_valid = true;
        m.solved_once = true;
        if (h_flag) return MHIP_ELIMIT;
        if (dev_env("MHIP_PF_DEBUG")) {
            unsigned int h_d[2] = {0, 0};
            MH_HIP(hipMemcpy(h_d, m.queue->pad2, 8, hipMemcpyDeviceToHost));
            fprintf(stderr, "[pf_solve queue] tickets taken %u, handed out %u, visits finished %u (lowered a seed: %u, LDS sweeps %u), abort %u\n", h_q[0], h_q[1], h_q[2],
                    h_d[0], h_d[1], h_q[3]);
        }
        m.queue_visits += h_q[2];
        if (!h_q[3] && h_q[2] == h_q[1]) return MHIP_OK;
        // called off (never seen): the levels are upper bounds of the solution all the same -- the rounds finish the job
        MH_HIP(hipMemsetAsync(m.mark, 0, m.nslots * 2 * 4, s));
        m.round = 0;
        MH_HIP(hipMemsetAsync(m.any, 0, (size_t)(PF_MAXR + 2) * 4, s));
    }
    m.rounds_ran = true;
    for (;;) {
        if (m.round + PF_BATCH + 1 >= PF_MAXR) {
            set_error("priority-flood seed graph did not converge within %d rounds", PF_MAXR);
            return MHIP_ENOTCONV;
        }
        for (int k = 0; k < PF_BATCH; ++k) {
            const int r = m.round + k;
            SolveArgs sa;
            sa.a = m.a;
            sa.mark_cur = m.mark + (size_t)(r & 1) * m.nslots;
            sa.mark_nxt = m.mark + (size_t)((r + 1) & 1) * m.nslots;
            sa.list_cur = m.list + (size_t)(r & 1) * m.nslots;
            sa.list_nxt = m.list + (size_t)((r + 1) & 1) * m.nslots;
            sa.count_cur = m.any + r;
            sa.count_nxt = m.any + r + 1;
            sa.visits = m.visits;
            sa.nbr = m.nbr;
            sa.nbc = m.nbc;
            sa.first = r == 0;
            sa.eblk = m.eblk;
            sa.row0 = 0;
            hipLaunchKernelGGL(pf_solve_kernel, dim3((unsigned)std::min<size_t>(m.nslots, 768)), dim3(ST), 0, s, sa);   // three resident per CU
        }
        m.launches += PF_BATCH;
        MH_HIP(hipGetLastError());
        MH_HIP(hipMemcpyAsync(h_cnt.data(), m.any + m.round + 1, sizeof(unsigned int) * PF_BATCH, hipMemcpyDeviceToHost, s));
        MH_HIP(hipMemcpyAsync(&h_flag, m.a.flags, 4, hipMemcpyDeviceToHost, s));
        if (!m.mm_valid) MH_HIP(hipMemcpyAsync(m.h_mm, m.mmout, 8, hipMemcpyDeviceToHost, s));
        MH_HIP(stream_sync(s));
        m.mm_valid = true;
        if (h_flag) return MHIP_ELIMIT;
        if (dev_env("MHIP_PF_DEBUG")) {
            fprintf(stderr, "[pf_solve] rounds %d..%d appended work:", m.round, m.round + PF_BATCH - 1);
            for (int k = 0; k < PF_BATCH; ++k) fprintf(stderr, " %u", h_cnt[k]);
            fprintf(stderr, "\n");
        }
        m.round += PF_BATCH;     // (even: the parity of the active-byte buffers is kept)
        for (int k = 0; k < PF_BATCH; ++k)
            if (h_cnt[k] == 0) {   // that round appended nothing: converged; the launches after it were no-ops
                m.launches -= PF_BATCH - (k + 1);
                return MHIP_OK;
            }
    }
}

int PfRun::publish_edges(hipStream_t s)
{
    Impl &m = *impl;
    const unsigned g = (unsigned)cdiv(W, 256);
    if (fixed_top) hipLaunchKernelGGL(pf_edge_rows_kernel, dim3(g), dim3(256), 0, s, m.a, out, (int64_t)1);
    if (fixed_bot) hipLaunchKernelGGL(pf_edge_rows_kernel, dim3(g), dim3(256), 0, s, m.a, out, H - 2);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

// tiles, links, the local solve (with the halo rows at +inf: nothing known about the neighbours yet) and the band's first edge
// rows.  MHIP_ELIMIT: a capacity was exceeded, or a band whose bottom halo row does not sit on a window ring row.
int PfRun::begin(hipStream_t s)
{
    Impl &m = *impl;
    if (H < 3 || W < 3) return MHIP_ELIMIT;
    if (fixed_bot && (H - 2) % TI != 0) return MHIP_ELIMIT;
    const int ntr = (int)cdiv(H - 2, TI), ntc = (int)cdiv(W - 2, TI);
    const int64_t ntiles = (int64_t)ntr * ntc;
    const size_t n = (size_t)(H * W);
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    // one workspace: bslot | tabV | tabL | tileNB | ringLab | spill | tileNS | links | tileNL | tileNL0 | Lv | active bytes | counters
    size_t off = 0;
    const size_t o_bslot = off; off = al(off + n * 2);
    const size_t o_tabV = off; off = al(off + (size_t)ntiles * NBMAX * 4);
    const size_t o_tabL = off; off = al(off + (size_t)ntiles * NBMAX);
    const size_t o_ring = off; off = al(off + (size_t)ntiles * 256);
    const size_t o_spill = off; off = al(off + (size_t)ntiles * SPMAX * 8);
    const size_t o_links = off; off = al(off + (size_t)ntiles * LMAX * 8);
    const size_t o_rrec = off; off = al(off + (size_t)ntiles * 256 * 8);
    const size_t o_lv = off; off = al(off + (size_t)ntiles * NSMAX * 4);
    // solve worklist: one active byte per block of BT x BT tiles, double buffered
    m.nbr = (int)cdiv(ntr, BT);
    m.nbc = (int)cdiv(ntc, BT);
    m.nslots = (size_t)m.nbr * m.nbc;
    m.ntiles = ntiles;
    // cleared before every run, from here ...
    const size_t o_nb = off; off = al(off + (size_t)ntiles * 4);
    const size_t o_ns = off; off = al(off + (size_t)ntiles * 4);
    const size_t o_nl = off; off = al(off + (size_t)ntiles * 4);
    const size_t o_nl0 = off; off = al(off + (size_t)ntiles * 4);
    const size_t o_ecnt = off; off = al(off + m.nslots * 4);
    const size_t o_act = off; off = al(off + m.nslots * 2 * 4);
    const size_t o_list = off; off = al(off + m.nslots * 2 * 4);
    const size_t o_cnt = off; off = al(off + (size_t)(PF_MAXR + 2) * 4 + 64 + 32 * 8);
    // ... to here
    const size_t o_eblk = off; off = al(off + m.nslots * EMAX * 8);
    const size_t o_mm = off; off = al(off + (size_t)ntiles * (NT / 64) * 2 * 4 + 64);
    // visit queue: a power of two of slots >= twice the blocks (a block is queued at most once) + the resident grid
    m.queue_cap = 1024;
    while (m.queue_cap < 2 * m.nslots + 1024) m.queue_cap <<= 1;
    m.queue_bytes = sizeof(PfQueue) + (size_t)m.queue_cap * 4;
    const size_t o_queue = off; off = al(off + m.queue_bytes);
    MH_TRY(m.ws.alloc(off));
    char *b = m.ws.as<char>();
    MH_HIP(hipMemsetAsync(b + o_nb, 0, o_eblk - o_nb, s));               // per-tile / per-block counts, active bytes, per-round words, flags, visits
    MH_HIP(hipMemsetAsync(b + o_ring, NOLAB, (size_t)ntiles * 256, s));
    // a tile that exceeds a capacity raises the overflow flag and returns at once; the kernels behind it in the queue still run
    // (the flag is read with the solve's results): its counts must read "nothing here", not what the pool block held before --
    // found by the poisoned pool (MHIP_DEVELOPER): pf_pack_kernel turned a stale tileNS into an offset 12 GB in front of its array
    // (the counts sit in front of the worklist words: one clear for all of them, above)
    PfArgs &a = m.a;
    a.H = H; a.W = W; a.ntr = ntr; a.ntc = ntc; a.dem = dem; a.fixed_top = fixed_top; a.fixed_bot = fixed_bot; a.stop = 0;
    a.bslot = reinterpret_cast<uint16_t *>(b + o_bslot);
    a.tabV = reinterpret_cast<uint32_t *>(b + o_tabV);
    a.tabL = reinterpret_cast<uint8_t *>(b + o_tabL);
    a.tileNB = reinterpret_cast<int *>(b + o_nb);
    a.ringLab = reinterpret_cast<uint8_t *>(b + o_ring);
    a.spill = reinterpret_cast<unsigned long long *>(b + o_spill);
    a.tileNS = reinterpret_cast<int *>(b + o_ns);
    a.links = reinterpret_cast<unsigned long long *>(b + o_links);
    a.ringRec = reinterpret_cast<unsigned long long *>(b + o_rrec);
    a.tileNL = reinterpret_cast<int *>(b + o_nl);
    a.tileNL0 = reinterpret_cast<int *>(b + o_nl0);
    a.Lv = reinterpret_cast<uint32_t *>(b + o_lv);
    m.any = reinterpret_cast<unsigned int *>(b + o_cnt);      // [PF_MAXR + 2] "round r was handed work"
    m.visits = reinterpret_cast<unsigned long long *>(m.any + (PF_MAXR + 2) + 2);
    a.prof = m.visits + 1;   // 24 words (inside the zeroed tail of the workspace)
    m.mark = reinterpret_cast<unsigned int *>(b + o_act);
    m.list = reinterpret_cast<int *>(b + o_list);
    m.eblk = reinterpret_cast<unsigned long long *>(b + o_eblk);
    a.mm = reinterpret_cast<uint32_t *>(b + o_mm);
    m.ecount = reinterpret_cast<int *>(b + o_ecnt);
    m.queue = reinterpret_cast<PfQueue *>(b + o_queue);
    a.flags = m.queue->flags;             // (in the queue's header: see PfQueue)
    m.mmout = m.queue->mm;
    MH_HIP(hipMemsetAsync(&m.queue->abort, 0, 256, s));
    m.solved_once = false;
    m.queue_visits = 0;
    m.rounds_ran = false;
    m.mm_valid = false;
    {
        const uint32_t init[2] = {0xffffffffu, 0u};
        MH_HIP(hipMemcpyAsync(m.mmout, init, sizeof(init), hipMemcpyHostToDevice, s));
    }
    m.round = 0;
    m.halo_dirty = false;

    if (fixed_top) hipLaunchKernelGGL(pf_fill_f32_kernel, dim3((unsigned)cdiv(W, 256)), dim3(256), 0, s, out, W, __builtin_inff());
    if (fixed_bot) hipLaunchKernelGGL(pf_fill_f32_kernel, dim3((unsigned)cdiv(W, 256)), dim3(256), 0, s, out + (H - 1) * W, W, __builtin_inff());
#ifdef PF_PHASES
    {   // cumulative cost of the phases: the kernel cut short after each of them (development builds)
        static const char *names[9] = {"load", "descent+plateaus", "doubling", "slots", "pairs", "compact", "label-correcting", "outputs", "spill"};
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        fprintf(stderr, "[pf_tile phases] cumulative ms:");
        for (int st = 1; st <= 9; ++st) {
            PfArgs b = a;
            b.stop = st;
            hipLaunchKernelGGL(pf_tile_kernel, dim3((unsigned)ntiles), dim3(NT), 0, s, b);
            hipEventRecord(e0, s);
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(pf_tile_kernel, dim3((unsigned)ntiles), dim3(NT), 0, s, b);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            fprintf(stderr, " %s=%.2f", names[st - 1], ms / 3);
        }
        for (int var = 1; var <= 2; ++var) {
            PfArgs b = a;
            b.stop = 5 | (var << 8);
            hipLaunchKernelGGL(pf_tile_kernel, dim3((unsigned)ntiles), dim3(NT), 0, s, b);
            hipEventRecord(e0, s);
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(pf_tile_kernel, dim3((unsigned)ntiles), dim3(NT), 0, s, b);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            fprintf(stderr, " | pairs %s=%.2f", var == 1 ? "without the flush atomics" : "with the CAS only", ms / 3);
        }
        fprintf(stderr, "\n");
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        MH_HIP(hipMemsetAsync(a.flags, 0, 4, s));
    }
#endif
    if (!m.k1_e0) {
        MH_HIP(hipEventCreate(&m.k1_e0));
        MH_HIP(hipEventCreate(&m.k1_e1));
    }
    MH_HIP(hipEventRecord(m.k1_e0, s));
    hipLaunchKernelGGL(pf_tile_kernel, dim3((unsigned)ntiles), dim3(NT), 0, s, a);
    MH_HIP(hipEventRecord(m.k1_e1, s));
    hipLaunchKernelGGL(pf_minmax_kernel, dim3(64), dim3(256), 0, s, a.mm, ntiles * (NT / 64), m.mmout);
    {
        hipLaunchKernelGGL(pf_ring_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, a);
        hipLaunchKernelGGL(pf_link2_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, a);
    }
    MH_TRY(pack(s, 0, m.nbr));
    MH_HIP(hipGetLastError());
    m.launches = 5;         // pf_tile, pf_minmax, pf_ring, pf_link2, pf_pack
    MH_TRY(solve(s));
    return publish_edges(s);
}

int PfRun::halo_changed(int, hipStream_t)
{
    impl->halo_dirty = true;
    return MHIP_OK;
}

// after a halo row changed: its links again, the blocks of the band's first / last tile rows, the solve, the edge rows
int PfRun::batch(hipStream_t s)
{
    Impl &m = *impl;
    if (!m.halo_dirty) return MHIP_OK;
    m.halo_dirty = false;
    hipLaunchKernelGGL(pf_halo_links_kernel, dim3((unsigned)(2 * m.a.ntc)), dim3(128), 0, s, m.a, (const float *)out);
    // the blocks of the first and of the last tile row go into the list of the round that comes next
    unsigned int *mark = m.mark + (size_t)(m.round & 1) * m.nslots;
    int *list = m.list + (size_t)(m.round & 1) * m.nslots;
    hipLaunchKernelGGL(pf_append_row_kernel, dim3((unsigned)cdiv(m.nbc, 256)), dim3(256), 0, s, mark, list, m.any + m.round, 0, m.nbc);
    if (m.nbr > 1)
        hipLaunchKernelGGL(pf_append_row_kernel, dim3((unsigned)cdiv(m.nbc, 256)), dim3(256), 0, s, mark, list, m.any + m.round, (m.nbr - 1) * m.nbc, m.nbc);
    MH_TRY(pack(s, 0, 1));                                  // the halo links of the first / last tile row changed
    if (m.nbr > 1) MH_TRY(pack(s, m.nbr - 1, 1));
    MH_HIP(hipGetLastError());
    MH_TRY(solve(s));
    return publish_edges(s);
}

// the final level of every basin, then the raster (a band's halo rows stay as the neighbour left them)
int PfRun::finish(hipStream_t s, float *d_depths, FillStats *st, bool *violated)
{
    Impl &m = *impl;
    hipLaunchKernelGGL(pf_final_kernel, dim3((unsigned)m.ntiles), dim3(256), 0, s, m.a);
    // The run-time proof (check.hip): K3 is a worklist schedule (marks + list appends wake neighbouring blocks); a wake lost there, or
    // a fault in any of K1..K4, leaves a surface that is not a fixed point.  Rasters whose width is a multiple of 256 take the fused
    // kernel (the check rides on the rows K4 computes); every other raster K4 + one streaming pass over dem + filled (8 B per cell).
    unsigned int h_viol = 0;
    const bool poke = violated && dev_env("MHIP_PF_CORRUPT") != nullptr && H > 8 && W > 8;    // test hook: one interior cell raised after the flood
    const bool fused = violated && !poke && W % 256 == 0;
    if (violated) MH_HIP(hipMemsetAsync(m.a.flags + 1, 0, 4, s));
    if (fused) {
        const unsigned nbx = (unsigned)(W / 1024 + (W % 1024 ? 1 : 0)), nby = (unsigned)cdiv(H, PF_RPW);
        hipLaunchKernelGGL(pf_apply_check_kernel, dim3(nbx * nby), dim3(256), 0, s, m.a, out, d_depths, nbx, m.a.flags + 1);
    } else {
        const int64_t groups = ((W + 3) / 4) * H;
        hipLaunchKernelGGL(pf_apply_kernel, dim3((unsigned)cdiv(groups, 256)), dim3(256), 0, s, m.a, out, d_depths);
    }
    MH_HIP(hipGetLastError());
    m.launches += 2;
    if (violated) {
        if (poke) hipLaunchKernelGGL(pf_fill_f32_kernel, dim3(1), dim3(1), 0, s, out + (H / 2) * W + W / 2, (int64_t)1, 3.0e38f);
        if (!fused) {
            MH_TRY(fill_check_f32_dev(dem, out, H, W, fixed_top, fixed_bot, s, m.a.flags + 1));
            m.launches += 1;
        }
        MH_HIP(hipMemcpyAsync(&h_viol, m.a.flags + 1, 4, hipMemcpyDeviceToHost, s));
    }
#ifdef PF_PROFILE
    {
        unsigned long long h_prof[24];
        MH_HIP(hipMemcpyAsync(h_prof, m.a.prof, sizeof(h_prof), hipMemcpyDeviceToHost, s));
        MH_HIP(stream_sync(s));
        const char *names[9] = {"load", "descent+plateaus", "doubling", "slots", "pairs", "compact", "label-correcting", "outputs", "spill"};
        double tot = 0;
        for (int i = 0; i < 9; ++i) tot += (double)h_prof[i];
        fprintf(stderr, "[pf_tile profile] ticks/tile:");
        for (int i = 0; i < 9; ++i) fprintf(stderr, " %s=%.0f", names[i], (double)h_prof[i] / (double)m.ntiles);
        fprintf(stderr, " | total=%.0f basins/tile=%.1f pairs/tile=%.1f seeds/tile=%.1f links/tile=%.1f spill/tile=%.1f\n", tot / (double)m.ntiles,
                (double)h_prof[9] / (double)m.ntiles, (double)h_prof[10] / (double)m.ntiles, (double)h_prof[11] / (double)m.ntiles,
                (double)h_prof[12] / (double)m.ntiles, (double)h_prof[13] / (double)m.ntiles);
        {
            unsigned long long hv_ = 0;
            (void)hipMemcpy(&hv_, m.visits, 8, hipMemcpyDeviceToHost);
            const double nv = (double)(hv_ ? hv_ : 1);
            fprintf(stderr, "[pf_solve, per visit] load ticks=%.0f links=%.0f spill loop=%.0f (%.1f iterations) store+push=%.0f | visits=%llu\n", (double)h_prof[19] / nv,
                    (double)h_prof[20] / nv, (double)h_prof[21] / nv, (double)h_prof[23] / nv, (double)h_prof[22] / nv, hv_);
        }
        fprintf(stderr, "[pf_tile S5, wave 0] preload ticks=%.0f insert ticks=%.0f | per tile: candidates=%.0f live=%.0f slow-path=%.0f\n", (double)h_prof[14] / (double)m.ntiles,
                (double)h_prof[15] / (double)m.ntiles, (double)h_prof[16] / (double)m.ntiles, (double)h_prof[17] / (double)m.ntiles, (double)h_prof[18] / (double)m.ntiles);
    }
#endif
    if (st) {
        unsigned long long h_vis = 0;
        if (m.rounds_ran) MH_HIP(hipMemcpyAsync(&h_vis, m.visits, 8, hipMemcpyDeviceToHost, s));      // (the queue solve's visits came with its header)
        MH_HIP(stream_sync(s));
        *st = FillStats();
        st->rounds = m.launches;
        st->visits = (int64_t)(h_vis + m.queue_visits);
        st->cycles = 0;
        st->tiles = m.ntiles;
        st->algorithm = 1;
        if (m.k1_e0 && hipEventElapsedTime(&st->hot_ms, m.k1_e0, m.k1_e1) == hipSuccess) st->hot_launches = 1;
        else (void)hipGetLastError();
        if (m.mm_valid) {
            st->have_minmax = true;
            st->dem_nan = m.h_mm[1] == 0xffffffffu;
            st->dem_min = key_f32(m.h_mm[0]);
            st->dem_max = st->dem_nan ? 0.0f : key_f32(m.h_mm[1]);
        }
    } else {
        MH_HIP(stream_sync(s));   // the workspace goes back to the pool now
    }
    if (violated) *violated = h_viol != 0;
    m.ws.release();
    return MHIP_OK;
}


// Exact tiled priority-flood on one raster.  Returns MHIP_ELIMIT (without touching d_out) when a per-tile capacity was
// exceeded: the caller then runs the iterative schedule.
int fill_plain_pflood_dev(const float *d_dem, float *d_out, float *d_depths, int64_t H, int64_t W, hipStream_t s, FillStats *st, bool *violated)
{
    PfRun f;
    f.dem = d_dem; f.out = d_out; f.H = H; f.W = W;
    MH_TRY(f.begin(s));
    return f.finish(s, d_depths, st, violated);
}

}  // namespace mh
