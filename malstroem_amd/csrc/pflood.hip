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
//   K2 pf_link_kernel     a ring cell of tile T is an owned cell of a neighbouring tile T': both seeds it drains to are
//                         joined by an edge of weight W_T'[cell]; de-duplicated per tile.
//   K3 pf_solve_kernel    minimax distance of every seed to OCEAN over (spill edges + links): tile worklist rounds like
//                         fill.hip's, but a visit touches ~2 KB instead of a 48 KB window.
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
#include <vector>

namespace mh {

namespace {

constexpr int WN = 64, TI = 62, NC = WN * WN;
constexpr int NBMAX = 1024;        // basins per tile
constexpr int HE = 2048;           // basin-pair hash entries
constexpr int SE = 512;            // seed-pair hash entries
constexpr int SPMAX = 256;         // spill edges stored per tile
constexpr int LMAX = 384;          // links stored per tile
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
    uint8_t *tabL;          // [ntiles * NBMAX] seed labels
    int *tileNB;            // [ntiles]
    uint8_t *ringLab;       // [ntiles * 256]
    unsigned long long *spill;  // [ntiles * SPMAX]  (la << 40 | lb << 32 | w)
    int *tileNS;            // [ntiles]
    unsigned long long *links;  // [ntiles * LMAX]   (myLab << 48 | dir << 40 | nbrLab << 32 | w)
    int *tileNL;            // [ntiles]
    uint32_t *Lv;           // [ntiles * 256] minimax level of every seed (keys)
    unsigned int *flags;    // [0]: overflow
};

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
__global__ __launch_bounds__(256) void pf_tile_kernel(PfArgs a)
{
    __shared__ uint32_t zk[NC];
    __shared__ uint16_t ptr[NC];
    __shared__ uint32_t aux[NC];          // plateau ids, then slot of a root
    __shared__ uint8_t cls[NC];
    __shared__ uint32_t hkv[2 * HE];      // basin-pair hash: keys | values (also: the plateau drains, one word per cell)
    __shared__ unsigned long long bkey[NBMAX];
    __shared__ uint16_t broot[NBMAX];
    __shared__ uint8_t btype[NBMAX];      // 1: interior pit (level to be found), 0: ring pit or raster border (fixed)
    __shared__ uint32_t sk[SE], sv[SE];
    __shared__ int s_scan[8];
    __shared__ int s_cnt;

    const int t = threadIdx.x, wc = t & 63, q = t >> 6;
    const int tile = blockIdx.x, ti = tile / a.ntc, tj = tile - ti * a.ntc;
    const int64_t r0 = (int64_t)ti * TI, c0 = (int64_t)tj * TI;
    const int64_t H = a.H, W = a.W;
    uint32_t *hk = hkv, *hv = hkv + HE, *drn = hkv;
    static_assert(2 * HE == NC, "the plateau drains reuse the hash arrays: one word per window cell");

    // ---- S1: window -> LDS
    const int64_t cc = c0 + wc;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int wr = q * 16 + k;
        const int64_t rr = r0 + wr;
        const bool valid = rr < H && cc < W;
        uint32_t key = KINV;
        uint8_t c = 0;
        if (valid) {
            key = dem_key(a.dem[rr * W + cc]);
            c = C_VALID;
            if (rr == 0 || rr == H - 1 || cc == 0 || cc == W - 1) c |= C_BORDER;
            else if (wr == 0 || wr == WN - 1 || wc == 0 || wc == WN - 1) c |= C_RING;
        }
        zk[wr * WN + wc] = key;
        cls[wr * WN + wc] = c;
    }
    for (int i = t; i < HE; i += 256) { hk[i] = EMPTY; hv[i] = EMPTY; }
    for (int i = t; i < SE; i += 256) { sk[i] = EMPTY; sv[i] = EMPTY; }
    if (t == 0) s_cnt = 0;
    __syncthreads();

    // ---- S2: steepest descent pointer of every cell (lowest neighbour if strictly lower, ties -> lowest index)
    auto ld = [&](int wr, int c) -> uint32_t { return (wr < 0 || wr >= WN || c < 0 || c >= WN) ? KINV : zk[wr * WN + c]; };
    unsigned eqmask = 0, lowmask = 0;
    {
        uint32_t up[3], mid[3], dn[3];
        const int wr0 = q * 16;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            up[j] = ld(wr0 - 1, wc - 1 + j);
            mid[j] = ld(wr0, wc - 1 + j);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
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
            const bool border = (cls[ci] & C_BORDER) != 0;
            ptr[ci] = (uint16_t)((lower && !border) ? bi : ci);   // a raster border cell is a root by decree (OCEAN)
#pragma unroll
            for (int j = 0; j < 3; ++j) { up[j] = mid[j]; mid[j] = dn[j]; }
        }
    }
    // ---- S2b: plateaus (connected equal cells) drain through ANY member that has a lower neighbour (or is a border cell)
    if (__syncthreads_or(eqmask != 0)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) aux[(q * 16 + k) * WN + wc] = (uint32_t)((q * 16 + k) * WN + wc);
        __syncthreads();
        for (int it = 0; it < 4 * NC; ++it) {   // min-index propagation over equal neighbours with pointer jumping
            bool ch = false;
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                if (!((eqmask >> k) & 1u)) continue;
                const int wr = q * 16 + k, ci = wr * WN + wc;
                const uint32_t own = zk[ci];
                uint32_t m = aux[ci];
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int rr = wr + dr, c2 = wc + dc;
                        if ((dr | dc) == 0 || rr < 0 || rr >= WN || c2 < 0 || c2 >= WN) continue;
                        if (zk[rr * WN + c2] == own) m = min(m, aux[rr * WN + c2]);
                    }
                m = min(m, aux[m]);
                if (m < aux[ci]) {
                    atomicMin(&aux[ci], m);
                    ch = true;
                }
            }
            if (!__syncthreads_or(ch)) break;
        }
        for (int i = t; i < NC; i += 256) drn[i] = EMPTY;   // hk + hv are re-initialised below
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
            if (!((eqmask >> k) & 1u)) continue;
            const int ci = (q * 16 + k) * WN + wc;
            if (cls[ci] & C_BORDER) atomicMin(&drn[aux[ci]], (uint32_t)ci);
            else if ((lowmask >> k) & 1u) atomicMin(&drn[aux[ci]], (uint32_t)ptr[ci]);   // still the steepest-descent target
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
            if (!((eqmask >> k) & 1u)) continue;
            const int ci = (q * 16 + k) * WN + wc;
            if (cls[ci] & C_BORDER) continue;
            const uint32_t root = aux[ci];
            if (root != (uint32_t)ci) ptr[ci] = (uint16_t)root;
            else ptr[ci] = (uint16_t)(drn[ci] != EMPTY ? drn[ci] : (uint32_t)ci);
        }
        __syncthreads();
        for (int i = t; i < HE; i += 256) { hk[i] = EMPTY; hv[i] = EMPTY; }
    }
    __syncthreads();

    // ---- S3: pointer doubling -> ptr[c] = the pit (root) c drains to
    for (int it = 0; it < 16; ++it) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ci = (q * 16 + k) * WN + wc;
            const uint16_t p = ptr[ci], pp = ptr[p];
            if (pp != p) {
                ptr[ci] = pp;
                ch = true;
            }
        }
        if (!__syncthreads_or(ch)) break;
    }

    // ---- S4: number the roots (basin slots)
    int nroot = 0;
    unsigned rootmask = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ci = (q * 16 + k) * WN + wc;
        if ((cls[ci] & C_VALID) && ptr[ci] == (uint16_t)ci) {
            rootmask |= 1u << k;
            ++nroot;
        }
    }
    int incl = nroot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (wc >= o) incl += v;
    }
    if (wc == 63) s_scan[q] = incl;
    __syncthreads();
    int base = incl - nroot;
    for (int w = 0; w < q; ++w) base += s_scan[w];
    const int NB = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    const bool too_many = NB > NBMAX;
    if (!too_many) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (!((rootmask >> k) & 1u)) continue;
            const int wr = q * 16 + k, ci = wr * WN + wc, s = base++;
            aux[ci] = (uint32_t)s;
            broot[s] = (uint16_t)ci;
            const uint8_t c = cls[ci];
            unsigned long long key = ~0ull;
            uint8_t ty = 1;
            if (c & C_BORDER) { key = ((unsigned long long)zk[ci] << 32) | OCEAN; ty = 0; }
            else if (c & C_RING) { key = ((unsigned long long)zk[ci] << 32) | (unsigned)ring_pos(wr, wc); ty = 0; }
            bkey[s] = key;
            btype[s] = ty;
        }
    }
    __syncthreads();
    bool overflow = too_many;
    if (!too_many) {
        // ring cells inside an interior-pit basin are outlets of that basin at their own elevation
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
            const int wr = q * 16 + k, ci = wr * WN + wc;
            if (!(cls[ci] & C_RING)) continue;
            const int s = (int)aux[ptr[ci]];
            if (btype[s]) atomicMin(&bkey[s], ((unsigned long long)zk[ci] << 32) | (unsigned)ring_pos(wr, wc));
        }
        // ---- S5: min pass height between adjacent basins
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
            const int wr = q * 16 + k, ci = wr * WN + wc;
            const uint32_t own = zk[ci];
            if (own == KINV) continue;
            const uint16_t ra = ptr[ci];
            const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, -1, 0, 1};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int rr = wr + dr[d], c2 = wc + dc[d];
                if (rr >= WN || c2 < 0 || c2 >= WN) continue;
                const int cj = rr * WN + c2;
                const uint32_t other = zk[cj];
                if (other == KINV) continue;
                const uint16_t rb = ptr[cj];
                if (ra == rb) continue;
                const uint32_t sa = aux[ra], sb = aux[rb];
                const uint32_t key = sa < sb ? (sa << 10 | sb) : (sb << 10 | sa);
                const uint32_t w = max(own, other);
                unsigned h = (key * 2654435761u) >> 21;
                bool done = false;
                for (int probe = 0; probe < 64; ++probe) {
                    const uint32_t prev = atomicCAS(&hk[h], EMPTY, key);
                    if (prev == EMPTY || prev == key) {
                        atomicMin(&hv[h], w);
                        done = true;
                        break;
                    }
                    h = (h + 1) & (HE - 1);
                }
                if (!done) overflow = true;
            }
        }
    }
    if (__syncthreads_or(overflow)) {
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
    // ---- label-correcting on the basin graph: (level, seed) of every interior-pit basin
    for (int it = 0; it < 4 * NBMAX; ++it) {
        bool ch = false;
        for (int h = t; h < HE; h += 256) {
            const uint32_t key = hk[h];
            if (key == EMPTY) continue;
            const int sa = (int)(key >> 10), sb = (int)(key & 1023u);
            const unsigned long long w = hv[h];
            const unsigned long long ka = bkey[sa], kb = bkey[sb];
            if (btype[sa]) {
                const unsigned long long hi = max(w, kb >> 32), cand = (hi << 32) | (kb & 0xffffffffull);
                if (cand < ka) { atomicMin(&bkey[sa], cand); ch = true; }
            }
            if (btype[sb]) {
                const unsigned long long hi = max(w, ka >> 32), cand = (hi << 32) | (ka & 0xffffffffull);
                if (cand < kb) { atomicMin(&bkey[sb], cand); ch = true; }
            }
        }
        if (!__syncthreads_or(ch)) break;
    }
    // ---- S6: outputs
    for (int s = t; s < NB; s += 256) {
        a.tabV[(size_t)tile * NBMAX + s] = (uint32_t)(bkey[s] >> 32);
        a.tabL[(size_t)tile * NBMAX + s] = (uint8_t)(bkey[s] & 0xffu);
    }
    if (t == 0) a.tileNB[tile] = NB;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int wr = q * 16 + k, ci = wr * WN + wc;
        const uint8_t c = cls[ci];
        if (!(c & C_VALID) || (c & C_BORDER)) continue;
        const int s = (int)aux[ptr[ci]];
        if (c & C_RING) a.ringLab[(size_t)tile * 256 + ring_pos(wr, wc)] = (uint8_t)(bkey[s] & 0xffu);
        else a.bslot[(r0 + wr) * W + cc] = (uint16_t)s;
    }
    // min spill elevation between pairs of seeds
    bool ov2 = false;
    for (int h = t; h < HE; h += 256) {
        const uint32_t key = hk[h];
        if (key == EMPTY) continue;
        const int sa = (int)(key >> 10), sb = (int)(key & 1023u);
        const unsigned long long ka = bkey[sa], kb = bkey[sb];
        const uint32_t la = (uint32_t)(ka & 0xffu), lb = (uint32_t)(kb & 0xffu);
        if (la == lb) continue;
        const uint32_t ww = max(hv[h], max((uint32_t)(ka >> 32), (uint32_t)(kb >> 32)));
        const uint32_t k2 = la < lb ? (la << 8 | lb) : (lb << 8 | la);
        unsigned hh = (k2 * 2654435761u) >> 23;
        bool done = false;
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
    for (int h = t; h < SE; h += 256) {
        if (sk[h] == EMPTY) continue;
        const int i = atomicAdd(&s_cnt, 1);
        if (i < SPMAX) a.spill[(size_t)tile * SPMAX + i] = ((unsigned long long)sk[h] << 32) | sv[h];
        else ov2 = true;
    }
    if (__syncthreads_or(ov2)) {
        if (t == 0) atomicOr(a.flags, 1u);
        return;
    }
    if (t == 0) a.tileNS[tile] = s_cnt;
}

// ---- K2: the two seeds of a ring cell (its own tile's and its owner's) are joined at the owner's fill level of the cell
__global__ __launch_bounds__(256) void pf_link_kernel(PfArgs a)
{
    __shared__ uint32_t lk[LH], lw[LH];
    __shared__ int s_cnt;
    const int t = threadIdx.x;
    const int tile = blockIdx.x, ti = tile / a.ntc, tj = tile - ti * a.ntc;
    for (int i = t; i < LH; i += 256) { lk[i] = EMPTY; lw[i] = EMPTY; }
    if (t == 0) s_cnt = 0;
    __syncthreads();
    bool ov = false;
    // ring cells of the 3 x 3 tiles around (and including) this one; an entry concerns this tile when it is the cell's
    // window tile (x == 4) or its owner
    for (int e = t; e < 9 * 252; e += 256) {
        const int x = e / 252, p = e - x * 252;
        const int xi = ti + x / 3 - 1, xj = tj + x % 3 - 1;
        if (xi < 0 || xi >= a.ntr || xj < 0 || xj >= a.ntc) continue;
        int wr, wc;
        ring_cell(p, wr, wc);
        const int64_t r = (int64_t)xi * TI + wr, c = (int64_t)xj * TI + wc;
        if (r <= 0 || r >= a.H - 1 || c <= 0 || c >= a.W - 1) continue;   // outside the raster or a raster border cell
        const int oi = (int)((r - 1) / TI), oj = (int)((c - 1) / TI);       // the tile that owns the cell
        const int xt = xi * a.ntc + xj, ot = oi * a.ntc + oj;
        if (x == 4 ? false : ot != tile) continue;
        const uint32_t labX = a.ringLab[(size_t)xt * 256 + p];
        const int so = a.bslot[r * a.W + c];
        const uint32_t labO = a.tabL[(size_t)ot * NBMAX + so];
        const uint32_t w = max(dem_key(a.dem[r * a.W + c]), a.tabV[(size_t)ot * NBMAX + so]);
        uint32_t mylab, nlab;
        int di, dj;
        if (x == 4) { mylab = labX; nlab = labO; di = oi - ti; dj = oj - tj; }
        else { mylab = labO; nlab = labX; di = xi - ti; dj = xj - tj; }
        if (mylab == (uint32_t)OCEAN) continue;                            // OCEAN's level is fixed
        const uint32_t dir = (uint32_t)((di + 1) * 3 + (dj + 1));
        const uint32_t key = mylab << 16 | dir << 8 | nlab;
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
    if (t == 0) a.tileNL[tile] = s_cnt;
    // start values of the solve: +inf for every seed, OCEAN below everything
    a.Lv[(size_t)tile * 256 + t] = t == OCEAN ? 0u : EMPTY;
}

// ---- K3: one round of the seed-graph solve; a wavefront visits a tile -------------------------------------------------
struct SolveArgs {
    PfArgs a;
    int *list_cur, *list_nxt;
    unsigned int *mark_nxt;
    unsigned int *count_cur, *count_nxt, *head;
    unsigned long long *visits;
    int first;    // 1: every tile (statically strided), no list
};

__global__ __launch_bounds__(256) void pf_solve_kernel(SolveArgs sa)
{
    __shared__ uint32_t Ls[4][256];
    const PfArgs &a = sa.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *L = Ls[wave];
    const int ntiles = a.ntr * a.ntc;
    const int gw = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    unsigned nvis = 0;
    const unsigned n = sa.first ? (unsigned)ntiles : *sa.count_cur;
    for (;;) {
        unsigned i;
        if (sa.first) {
            i = (unsigned)gw + nvis * (unsigned)nwaves;
        } else {
            i = 0;
            if (lane == 0) i = atomicAdd(sa.head, 1u);
            i = __builtin_amdgcn_readfirstlane(i);
        }
        if (i >= n) break;
        const int tile = sa.first ? (int)i : sa.list_cur[i];
        ++nvis;
        const int ti = tile / a.ntc, tj = tile - ti * a.ntc;
        uint32_t old[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            old[k] = __hip_atomic_load(&a.Lv[(size_t)tile * 256 + lane * 4 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L[lane * 4 + k] = old[k];
        }
        __builtin_amdgcn_wave_barrier();
        const int nl = a.tileNL[tile], ns = a.tileNS[tile];
        for (int e = lane; e < nl; e += 64) {
            const unsigned long long v = a.links[(size_t)tile * LMAX + e];
            const uint32_t key = (uint32_t)(v >> 32), w = (uint32_t)v;
            const int mylab = (int)(key >> 16), dir = (int)((key >> 8) & 0xffu), nlab = (int)(key & 0xffu);
            uint32_t ln = 0u;
            if (nlab != OCEAN) {
                const int nt = (ti + dir / 3 - 1) * a.ntc + (tj + dir % 3 - 1);
                ln = __hip_atomic_load(&a.Lv[(size_t)nt * 256 + nlab], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            atomicMin(&L[mylab], max(w, ln));
        }
        __builtin_amdgcn_wave_barrier();
        for (int it = 0; it < 256; ++it) {   // spill edges inside the tile, to a local fixed point
            bool ch = false;
            for (int e = lane; e < ns; e += 64) {
                const unsigned long long v = a.spill[(size_t)tile * SPMAX + e];
                const uint32_t key = (uint32_t)(v >> 32), w = (uint32_t)v;
                const int la = (int)(key >> 8), lb = (int)(key & 0xffu);
                const uint32_t va = L[la], vb = L[lb];
                const uint32_t ca = max(w, vb), cb = max(w, va);
                if (la != OCEAN && ca < va) { atomicMin(&L[la], ca); ch = true; }
                if (lb != OCEAN && cb < vb) { atomicMin(&L[lb], cb); ch = true; }
            }
            __builtin_amdgcn_wave_barrier();
            if (!__any(ch)) break;
        }
        bool changed = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t v = L[lane * 4 + k];
            if (v < old[k]) {
                __hip_atomic_store(&a.Lv[(size_t)tile * 256 + lane * 4 + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                changed = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (__any(changed) && lane < 9 && lane != 4) {
            const int p = ti + lane / 3 - 1, qq = tj + lane % 3 - 1;
            if (p >= 0 && p < a.ntr && qq >= 0 && qq < a.ntc) {
                const int nt = p * a.ntc + qq;
                if (atomicExch(&sa.mark_nxt[nt], 1u) == 0u) sa.list_nxt[atomicAdd(sa.count_nxt, 1u)] = nt;
            }
        }
    }
    if (lane == 0 && nvis) atomicAdd(sa.visits, (unsigned long long)nvis);
}

__global__ __launch_bounds__(256) void pf_clear_marks_kernel(const int *list, const unsigned int *count, unsigned int *mark)
{
    const unsigned n = *count;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) mark[list[i]] = 0u;
}

// ---- K4: final level of every basin, then the raster ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void pf_final_kernel(PfArgs a)
{
    const int tile = blockIdx.x;
    const int nb = a.tileNB[tile];
    for (int s = threadIdx.x; s < nb; s += 256) {
        const size_t i = (size_t)tile * NBMAX + s;
        a.tabV[i] = max(a.tabV[i], a.Lv[(size_t)tile * 256 + a.tabL[i]]);
    }
}

__global__ __launch_bounds__(256) void pf_apply_kernel(PfArgs a, float *__restrict__ filled, float *__restrict__ depths)
{
    const int64_t W = a.W, H = a.H;
    const int64_t groups = (W + 3) / 4;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups * H) return;
    const int64_t r = g / groups, c4 = (g - r * groups) * 4;
    const bool rowb = r == 0 || r == H - 1;
    const int ti = (int)((r - 1) / TI);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c4 + k;
        if (c >= W) break;
        const int64_t i = r * W + c;
        const float d = a.dem[i];
        float f = d;
        if (!(rowb || c == 0 || c == W - 1)) {
            const int tj = (int)((c - 1) / TI);
            const uint32_t lev = a.tabV[(size_t)(ti * a.ntc + tj) * NBMAX + a.bslot[i]];
            f = key_f32(max(dem_key(d), lev));
        }
        filled[i] = f;
        if (depths) depths[i] = f - d;
    }
}

}  // namespace

// Exact tiled priority-flood.  Returns MHIP_ELIMIT (without touching d_out) when a per-tile capacity was exceeded: the
// caller then runs the iterative schedule.
int fill_plain_pflood_dev(const float *d_dem, float *d_out, float *d_depths, int64_t H, int64_t W, hipStream_t s, FillStats *st)
{
    const int ntr = (int)cdiv(H - 2, TI), ntc = (int)cdiv(W - 2, TI);
    const int64_t ntiles = (int64_t)ntr * ntc;
    const size_t n = (size_t)(H * W);
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    // one workspace: bslot | tabV | tabL | tileNB | ringLab | spill | tileNS | links | tileNL | Lv | lists[2] | marks[2] | counters
    constexpr int MAXR = 1 << 14;
    size_t off = 0;
    const size_t o_bslot = off; off = al(off + n * 2);
    const size_t o_tabV = off; off = al(off + (size_t)ntiles * NBMAX * 4);
    const size_t o_tabL = off; off = al(off + (size_t)ntiles * NBMAX);
    const size_t o_nb = off; off = al(off + (size_t)ntiles * 4);
    const size_t o_ring = off; off = al(off + (size_t)ntiles * 256);
    const size_t o_spill = off; off = al(off + (size_t)ntiles * SPMAX * 8);
    const size_t o_ns = off; off = al(off + (size_t)ntiles * 4);
    const size_t o_links = off; off = al(off + (size_t)ntiles * LMAX * 8);
    const size_t o_nl = off; off = al(off + (size_t)ntiles * 4);
    const size_t o_lv = off; off = al(off + (size_t)ntiles * 256 * 4);
    const size_t o_list = off; off = al(off + (size_t)ntiles * 2 * 4);
    const size_t o_mark = off; off = al(off + (size_t)ntiles * 2 * 4);
    const size_t o_cnt = off; off = al(off + (size_t)(MAXR + 2) * 4 * 2 + 64);
    DevBuf ws;
    MH_TRY(ws.alloc(off));
    char *b = ws.as<char>();
    MH_HIP(hipMemsetAsync(b + o_mark, 0, off - o_mark, s));               // marks, counters, heads, flags, visits
    MH_HIP(hipMemsetAsync(b + o_ring, NOLAB, (size_t)ntiles * 256, s));
    PfArgs a;
    a.H = H; a.W = W; a.ntr = ntr; a.ntc = ntc; a.dem = d_dem;
    a.bslot = reinterpret_cast<uint16_t *>(b + o_bslot);
    a.tabV = reinterpret_cast<uint32_t *>(b + o_tabV);
    a.tabL = reinterpret_cast<uint8_t *>(b + o_tabL);
    a.tileNB = reinterpret_cast<int *>(b + o_nb);
    a.ringLab = reinterpret_cast<uint8_t *>(b + o_ring);
    a.spill = reinterpret_cast<unsigned long long *>(b + o_spill);
    a.tileNS = reinterpret_cast<int *>(b + o_ns);
    a.links = reinterpret_cast<unsigned long long *>(b + o_links);
    a.tileNL = reinterpret_cast<int *>(b + o_nl);
    a.Lv = reinterpret_cast<uint32_t *>(b + o_lv);
    unsigned int *count = reinterpret_cast<unsigned int *>(b + o_cnt);    // [MAXR + 2]
    unsigned int *head = count + (MAXR + 2);                               // [MAXR + 2]
    a.flags = head + (MAXR + 2);
    unsigned long long *visits = reinterpret_cast<unsigned long long *>(a.flags + 2);
    int *lists = reinterpret_cast<int *>(b + o_list);
    unsigned int *marks = reinterpret_cast<unsigned int *>(b + o_mark);

    hipLaunchKernelGGL(pf_tile_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pf_link_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, a);
    MH_HIP(hipGetLastError());
    int launches = 2, round = 0;
    const unsigned grid = (unsigned)(cdiv(ntiles, 4) < 1024 ? cdiv(ntiles, 4) : 1024);
    auto launch_round = [&](int r) {
        SolveArgs sa;
        sa.a = a;
        sa.list_cur = lists + (size_t)(r & 1) * ntiles;
        sa.list_nxt = lists + (size_t)((r + 1) & 1) * ntiles;
        sa.mark_nxt = marks + (size_t)((r + 1) & 1) * ntiles;
        sa.count_cur = count + r;
        sa.count_nxt = count + r + 1;
        sa.head = head + r;
        sa.visits = visits;
        sa.first = r == 0;
        hipLaunchKernelGGL(pf_solve_kernel, dim3(grid), dim3(256), 0, s, sa);
        // the marks of the list this round consumed must be clear before the round after next appends to them again
        if (r > 0) hipLaunchKernelGGL(pf_clear_marks_kernel, dim3(64), dim3(256), 0, s, sa.list_cur, sa.count_cur, marks + (size_t)(r & 1) * ntiles);
    };
    constexpr int BATCH = 32;
    std::vector<unsigned int> h_cnt(BATCH + 1);
    unsigned int h_flag = 0;
    bool done = false;
    while (!done) {
        if (round + BATCH + 1 >= MAXR) {
            set_error("priority-flood seed graph did not converge within %d rounds", MAXR);
            return MHIP_ENOTCONV;
        }
        for (int k = 0; k < BATCH; ++k) launch_round(round + k);
        launches += BATCH;
        MH_HIP(hipGetLastError());
        MH_HIP(hipMemcpyAsync(h_cnt.data(), count + round + 1, sizeof(unsigned int) * BATCH, hipMemcpyDeviceToHost, s));
        MH_HIP(hipMemcpyAsync(&h_flag, a.flags, 4, hipMemcpyDeviceToHost, s));
        MH_HIP(hipStreamSynchronize(s));
        if (h_flag) return MHIP_ELIMIT;
        for (int k = 0; k < BATCH; ++k)
            if (h_cnt[k] == 0) {   // round (round + k) appended nothing: converged; the launches after it were no-ops
                done = true;
                launches -= BATCH - (k + 1);
                break;
            }
        round += BATCH;
    }
    hipLaunchKernelGGL(pf_final_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, a);
    const int64_t groups = ((W + 3) / 4) * H;
    hipLaunchKernelGGL(pf_apply_kernel, dim3((unsigned)cdiv(groups, 256)), dim3(256), 0, s, a, d_out, d_depths);
    MH_HIP(hipGetLastError());
    launches += 2;
    if (st) {
        unsigned long long h_vis = 0;
        MH_HIP(hipMemcpyAsync(&h_vis, visits, 8, hipMemcpyDeviceToHost, s));
        MH_HIP(hipStreamSynchronize(s));
        *st = FillStats();
        st->rounds = launches;
        st->visits = (int64_t)h_vis;
        st->cycles = 0;
        st->tiles = ntiles;
    } else {
        MH_HIP(hipStreamSynchronize(s));   // the workspace goes back to the pool when this function returns
    }
    return MHIP_OK;
}

}  // namespace mh
