// watershed.hip -- watershed labelling by pointer jumping on the D8 flow forest (gfx950).
//
// Reference: flow.watersheds_from_labels (flow.py:398-412): for every raster EDGE cell run the upstream DFS
// assign_watersheds_upstream (_flow.pyx:276-403, python flow.py:367-395) carrying the "downstream label";
// unlabelled cells take it, labelled cells replace it.  Order independent restatement (SURVEY.md 8a row W,
// fuzz-verified against the oracle): an unlabelled cell X receives the label of the FIRST labelled cell Y on its
// downstream path provided Y's own downstream path (Y included) contains an edge cell; otherwise X keeps
// `unassigned`.  Labelled cells never change.
//
// Fast path (every flow path leaves the raster, i.e. no interior NODIR cell -- always true for D8 on a no-flats
// surface with edges flowing outward): P[c] = next cell downstream, labelled cells are fixed points, cells that
// leave the raster are dead ends; log2(path length) rounds of P[c] = P[P[c]].
// General path (interior sinks / inward edges): a second jump structure Q over ALL cells accumulates
// "an edge cell lies on my downstream path" so that labelled terminals without an edge cell are ignored.
// Flow cycles (on which the reference does not terminate when they contain an edge cell) never resolve and
// keep `unassigned`; the number of rounds is capped at 40 (> log2 of any int32-indexable path).
#include <algorithm>

#include <type_traits>

#include "common.hpp"

namespace mh {
namespace {

constexpr int32_t NONE = 0x7fffffff;
constexpr uint32_t QFLAG = 0x80000000u, QMASK = 0x7fffffffu, QTERM = 0x7fffffffu;

__device__ __forceinline__ int64_t downstream(const uint8_t *fd, int64_t i, int64_t H, int64_t W)
{
    const unsigned code = fd[i];
    if (code > 7u) return -1;
    const int64_t r = i / W, c = i - r * W;
    const int64_t nr = r + dir_dr((int)code), nc = c + dir_dc((int)code);
    if (nr < 0 || nr >= H || nc < 0 || nc >= W) return -1;
    return nr * W + nc;
}

// (16 cells per thread: a byte per thread made this count a 1.7 ms pass over 268 M cells)
__global__ __launch_bounds__(256) void ws_count_interior_nodir(const uint8_t *__restrict__ fd, int64_t H, int64_t W,
                                                              unsigned int *count)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16, n = H * W;
    unsigned hits = 0;
    if (i0 < n) {
        int64_t r = i0 / W, c = i0 - r * W;
        uint8_t b[16];
        const bool whole = i0 + 16 <= n;
        if (whole) {      // (i0 is a multiple of 16 whatever W is: the rows are walked below)
            *reinterpret_cast<uint4 *>(b) = *reinterpret_cast<const uint4 *>(fd + i0);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = i0 + k < n ? fd[i0 + k] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            hits += (b[k] > 7u && r > 0 && r < H - 1 && c > 0 && c < W - 1) ? 1u : 0u;
            if (++c == W) {
                c = 0;
                ++r;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o);
    if (hits && (threadIdx.x & 63) == 0) atomicAdd(count, hits);
}

__global__ __launch_bounds__(256) void ws_init_kernel(const uint8_t *__restrict__ fd, const int32_t *__restrict__ lab,
                                                     int32_t *__restrict__ P, uint32_t *__restrict__ Q, int64_t H, int64_t W,
                                                     int32_t unassigned)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int64_t d = downstream(fd, i, H, W);
    P[i] = lab[i] != unassigned ? (int32_t)i : (d < 0 ? NONE : (int32_t)d);
    if (Q) {
        const int64_t r = i / W, c = i - r * W;
        const bool edge = r == 0 || r == H - 1 || c == 0 || c == W - 1;
        Q[i] = (edge ? QFLAG : 0u) | (d < 0 ? QTERM : (uint32_t)d);
    }
}

// One launch performs up to HOPS pointer jumps per cell (P[i] <- P[P[i]] repeatedly): the raster is read once per launch
// while the path length covered grows by 2**HOPS.  Racing updates are benign: every value a thread can observe in
// P[] / Q[] is a cell further down the same flow path (or the final terminal).  A cell whose pointer has reached a
// labelled cell stores it with the DONE bit: later launches then cost one coalesced read for it, no gather.
constexpr int HOPS = 5;
constexpr int32_t DONE = (int32_t)0x80000000;
// `open`: set when some pointer is still on its way after this launch (the host stops as soon as a launch leaves it 0;
// pointers on a flow cycle stay open for ever: the launch count is capped)
__global__ __launch_bounds__(256) void ws_jump_kernel(int32_t *P, uint32_t *Q, int64_t n, unsigned int *open)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ch = false;
    int32_t t = P[i];
    if (t >= 0 && t != NONE && t != (int32_t)i) {
        const int32_t t0 = t;
#pragma unroll
        for (int h = 0; h < HOPS; ++h) {
            const int32_t pt = P[t];
            if (pt == t) {  // t is a labelled fixed point: resolved
                t |= DONE;
                break;
            }
            t = pt;         // skip over t (pt may be NONE, or already carry DONE)
            if (t < 0 || t == NONE) break;
        }
        if (t != t0) P[i] = t;
        ch = t >= 0 && t != NONE;   // neither at a labelled cell nor out of the raster yet
    }
    if (Q) {
        uint32_t q = Q[i];
        const uint32_t q0 = q;
#pragma unroll
        for (int h = 0; h < HOPS; ++h) {
            const uint32_t idx = q & QMASK;
            if (idx == QTERM) break;
            const uint32_t qt = Q[idx];
            q = ((q | qt) & QFLAG) | (qt & QMASK);
        }
        if (q != q0) Q[i] = q;
        ch |= (q & QMASK) != QTERM;
    }
    if (ch) *open = 1u;
}

// ---- fast path, tile first --------------------------------------------------------------------------------------------------
// One workgroup resolves the pointers of a 64 x 64 tile in LDS (pull-only jumping on 16-bit tile-local indices: no atomics):
// afterwards an unlabelled cell points at the labelled cell of ITS TILE it drains to (DONE), out of the raster (NONE), or at the
// first cell of its path outside the tile -- an ENTRY cell on the perimeter of a neighbouring tile.  Only perimeter cells are
// jumped through global memory (6 % of the raster, and one hop now crosses a tile); the final pass takes one more hop for the
// cells that still point at an entry.
constexpr int WT = 64;
// The perimeter cells of the tiles -- the only cells a path can ENTER a tile at -- also live in a compact array, 256 slots per tile
// (1 KB: the jumps over the entry cells gather there instead of in the raster, where a tile's left / right columns cost a
// sector per cell: 8.7 B per raster cell for 6 % of the cells).  slot: top row, bottom row, left column, right column.
__device__ __forceinline__ int ws_perim_slot(int lr, int lc)
{
    if (lr == 0) return lc;
    if (lr == WT - 1) return WT + lc;
    if (lc == 0) return 2 * WT + (lr - 1);
    if (lc == WT - 1) return 2 * WT + (WT - 2) + (lr - 1);
    return -1;
}
// the node of an entry cell; -1 for a cell inside its tile (a pointer caught in a flow cycle: it never resolves)
__device__ __forceinline__ int64_t ws_node_of(int32_t cell, uint32_t W, int ntc)
{
    const uint32_t r = (uint32_t)cell / W, c = (uint32_t)cell - r * W;
    const int slot = ws_perim_slot((int)(r & 63u), (int)(c & 63u));
    return slot < 0 ? -1 : (int64_t)((r >> 6) * (uint32_t)ntc + (c >> 6)) * 256 + slot;
}
// `pc` (optional): the pour-point candidates of the tile on the way (common.hpp: PourCandDev) -- a cell whose downstream cell does
// not carry its own label (or which has none).  Whether the downstream cell is labelled is in `val` already.
__global__ __launch_bounds__(256) void ws_tile_kernel(const uint8_t *__restrict__ fd, const int32_t *__restrict__ lab, int32_t *__restrict__ P, int64_t H,
                                                     int64_t W, int ntc, int32_t unassigned, PourCandDev pc, int32_t *__restrict__ Pn)
{
    __shared__ uint16_t ptr[WT * WT];
    // of a terminal, one byte (the global index it stands for is put together when it is needed: 12 instead of 24 KB of LDS, the
    // table no longer limits the resident workgroups): V_LAB labelled (its own index), V_NONE the path ends unlabelled, else the
    // direction 0..7 in which the path leaves the tile -- the entry cell is that neighbour
    __shared__ uint8_t val[WT * WT];
    constexpr uint8_t V_LAB = 8, V_NONE = 9;
    const int ti = blockIdx.x / ntc, tj = blockIdx.x - ti * ntc;
    const int64_t r0 = (int64_t)ti * WT, c0 = (int64_t)tj * WT;
    // global I/O: a thread owns 16 consecutive cells of a tile row (one 16-byte load of the directions, four of the labels, four
    // 16-byte stores); the jumping in between walks the table with consecutive lanes on consecutive cells
    const int lr = threadIdx.x >> 2, lc0 = (threadIdx.x & 3) * 16;
    const int64_t r = r0 + lr, cbase = c0 + lc0;
    const bool vec = (W & 15) == 0 && r < H && cbase + 16 <= W;     // whole, 16-byte aligned group
    uint8_t code[16];
    int32_t lb[16];
    if (vec) {
        const uint4 cv = *reinterpret_cast<const uint4 *>(fd + r * W + cbase);
        memcpy(code, &cv, 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int4 lv = *reinterpret_cast<const int4 *>(lab + r * W + cbase + 4 * q);
            lb[4 * q] = lv.x; lb[4 * q + 1] = lv.y; lb[4 * q + 2] = lv.z; lb[4 * q + 3] = lv.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool in = r < H && cbase + k < W;
            code[k] = in ? fd[r * W + cbase + k] : (uint8_t)8;
            lb[k] = in ? lab[r * W + cbase + k] : unassigned;
        }
    }
    uint32_t lmask = 0;      // my labelled cells
    // (INNER: the tile touches no border of the raster -- every cell and every downstream cell is a raster cell: no 64-bit bounds
    // tests per cell, local coordinates only)
    auto init_cells = [&](auto inner_tag) {
        constexpr bool INNER = decltype(inner_tag)::value;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int li = lr * WT + lc0 + k;
            const int64_t c = cbase + k;
            uint16_t p = (uint16_t)li;
            uint8_t v = V_NONE;
            if (INNER || (r < H && c < W)) {
                if (lb[k] != unassigned) {
                    lmask |= 1u << k;
                    v = V_LAB;                              // labelled: a fixed point
                } else if (code[k] <= 7u) {
                    const int lr2 = lr + dir_dr((int)code[k]), lc2 = lc0 + k + dir_dc((int)code[k]);
                    bool in_raster = true;
                    if (!INNER) {
                        const int64_t nr = r0 + lr2, nc = c0 + lc2;
                        in_raster = nr >= 0 && nr < H && nc >= 0 && nc < W;
                    }
                    if (in_raster) {
                        if (lr2 >= 0 && lr2 < WT && lc2 >= 0 && lc2 < WT) p = (uint16_t)(lr2 * WT + lc2);
                        else v = code[k];                   // leaves the tile: the path continues at that entry cell
                    }
                }
            }
            ptr[li] = p;
            val[li] = v;
        }
    };
    if (r0 > 0 && c0 > 0 && r0 + WT < H && c0 + WT < W) init_cells(std::true_type{});
    else init_cells(std::false_type{});
    __syncthreads();
    if (pc.mask0) {
        // (labels and directions are NOT kept in registers across the barrier above -- 32 registers that cost the kernel a third of
        // its resident workgroups: the directions are read again, 16 bytes the tile has just pulled through the cache; "labelled"
        // is a bit per cell; a label's value is only needed for the few candidates)
        uint32_t cw[4] = {0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
        if (vec) {
            const uint4 cv = *reinterpret_cast<const uint4 *>(fd + r * W + cbase);
            cw[0] = cv.x; cw[1] = cv.y; cw[2] = cv.z; cw[3] = cv.w;
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (r < H && cbase + k < W) cw[k >> 2] = (cw[k >> 2] & ~(0xffu << (8 * (k & 3)))) | ((uint32_t)fd[r * W + cbase + k] << (8 * (k & 3)));
        }
        const uint32_t Wu = (uint32_t)W, Hu = (uint32_t)H, r0u = (uint32_t)r0, c0u = (uint32_t)c0;
        uint32_t m0 = 0, ml = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (!(r < H && cbase + k < W)) continue;
            const unsigned cd = (cw[k >> 2] >> (8 * (k & 3))) & 0xffu;
            const bool mylab = (lmask >> k) & 1u;
            bool same = false;         // the downstream cell carries my label (unlabelled: it is unlabelled as well)
            if (cd <= 7u) {
                const int lr2 = lr + dir_dr((int)cd), lc2 = lc0 + k + dir_dc((int)cd);
                const uint32_t nr = r0u + (uint32_t)lr2, nc = c0u + (uint32_t)lc2;      // (wraps to a huge value left of / above the raster)
                if (nr < Hu && nc < Wu) {
                    const uint32_t gd = nr * Wu + nc;
                    bool labelled_d;
                    if (lr2 >= 0 && lr2 < WT && lc2 >= 0 && lc2 < WT) labelled_d = val[lr2 * WT + lc2] == V_LAB;
                    else labelled_d = lab[gd] != unassigned;
                    if (pc.components) same = labelled_d == mylab;
                    else same = (labelled_d ? lab[gd] : unassigned) == (mylab ? lab[(uint32_t)(r * W + cbase + k)] : unassigned);
                }
            }
            if (!same) {
                if (!mylab) m0 |= 1u << k;
                else ml |= 1u << k;
            }
        }
        pc.mask0[(int64_t)blockIdx.x * 256 + threadIdx.x] = (uint16_t)m0;
        // the labelled candidates of the tile, contiguous in the list: counts -> offsets over the workgroup, one global atomic
        __shared__ uint32_t wsum[4], base_s;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint32_t mine = (uint32_t)__builtin_popcount(ml);
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            const bool fits = tot <= POUR_TILE_CAP;
            if (!fits) pc.flags[0] = 1u;      // (the caller then runs the general pass over accumulation + labels)
            pc.tile_cnt[blockIdx.x] = fits ? tot : 0u;
            base_s = fits ? blockIdx.x * POUR_TILE_CAP : 0xffffffffu;
        }
        __syncthreads();
        if (base_s != 0xffffffffu) {
            uint32_t o = base_s + incl - mine;
            for (int w = 0; w < wave; ++w) o += wsum[w];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if ((ml >> k) & 1u) {
                    const uint32_t cell = (uint32_t)(r * W + cbase + k);
                    pc.list[o++] = make_uint2(cell, (uint32_t)lab[cell]);
                }
        }
    }
    for (int round = 0; round < 13; ++round) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int li = k * 256 + (int)threadIdx.x;
            const uint16_t p = ptr[li], q = ptr[p];
            if (q != p) {
                ptr[li] = q;        // racing readers see p or q: both further down the same path
                ch = true;
            }
        }
        if (!__syncthreads_or(ch)) break;       // (WgVote measured slower here: 1.58 -> 1.77 ms)
    }
    int32_t out[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int li = lr * WT + lc0 + k;
        const int t = ptr[li];
        const int64_t gt = (r0 + (t >> 6)) * W + c0 + (t & 63);
        if (ptr[t] != t) out[k] = (int32_t)gt;              // a flow cycle inside the tile: never resolves
        else {
            const unsigned vb = val[t];
            int32_t v = NONE;
            if (vb == V_LAB) v = (int32_t)gt;
            else if (vb <= 7u) v = (int32_t)((r0 + (t >> 6) + dir_dr((int)vb)) * W + c0 + (t & 63) + dir_dc((int)vb));
            out[k] = (vb == V_LAB && t != li) ? (v | DONE) : v;   // v: own index (a labelled cell keeps P == self), NONE, or an entry cell
        }
        const int slot = ws_perim_slot(lr, lc0 + k);
        if (slot >= 0) Pn[(int64_t)blockIdx.x * 256 + slot] = (r < H && cbase + k < W) ? out[k] : NONE;
    }
    if (vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<int4 *>(P + r * W + cbase + 4 * q) = make_int4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (r < H && cbase + k < W) P[r * W + cbase + k] = out[k];
    }
}

// the jump of ws_jump_kernel for the perimeter cells of the tiles only, on their compact array: a node that still points at an
// entry cell takes over what that entry's node points at.  A labelled node holds its own cell.
__global__ __launch_bounds__(256) void ws_jump_perimeter_kernel(int32_t *Pn, int64_t W, int ntc, int64_t nnodes, unsigned int *open)
{
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nnodes || (x & 255) >= 4 * WT - 4) return;
    int32_t t = Pn[x];
    if (t < 0 || t == NONE) return;
    if (ws_node_of(t, (uint32_t)W, ntc) == x) return;      // labelled: a fixed point
    const int32_t t0 = t;
#pragma unroll
    for (int h = 0; h < HOPS; ++h) {
        const int64_t nd = ws_node_of(t, (uint32_t)W, ntc);
        if (nd < 0) {          // a flow cycle inside that tile: final as it is (the cell stays unassigned)
            if (t != t0) Pn[x] = t;
            return;
        }
        const int32_t pt = Pn[nd];
        if (pt == t) {
            t |= DONE;
            break;
        }
        t = pt;
        if (t < 0 || t == NONE) break;
    }
    if (t != t0) Pn[x] = t;
    if (t >= 0 && t != NONE) *open = 1u;
}

// final pass of the fast path: a cell that still points at an entry cell takes that cell's (resolved) pointer
// `src` != `lab`: out of place -- every cell of `lab` is written (its own label unless it takes one from downstream), which saves
// the caller the copy of the label raster it would otherwise start from
__global__ __launch_bounds__(256) void ws_assign_hop_kernel(const int32_t *__restrict__ P, const int32_t *__restrict__ Pn, const int32_t *src, int32_t *lab, int64_t n,
                                                           int64_t W, int ntc)
{
    // four cells per thread: 16-byte loads of P (and of src out of place), one 16-byte store out of place
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    const bool copy = src != lab;
    auto target = [&](int64_t i, int32_t p) -> int32_t {     // the cell whose label cell i takes, or -1: it keeps its own
        if (p == NONE || p == (int32_t)i) return -1;         // flows out unlabelled, or labelled
        if (p >= 0) {                                         // an entry cell: what its node has been resolved to
            const int64_t nd = ws_node_of(p, (uint32_t)W, ntc);
            if (nd < 0) return -1;                            // (a flow cycle inside the tile)
            const int32_t q = Pn[nd];
            if (q == p) p = q | DONE;                         // the entry cell is labelled itself
            else if (q < 0) p = q;                            // resolved through the entry cell
            else return -1;                                   // NONE, or a flow cycle: stays unassigned
        }
        return p & ~DONE;
    };
    if (i0 + 4 <= n && copy) {
        // (out of place, the usual call: the own labels, the four entry nodes and then the four labels each in flight together -- cell by
        // cell the look-ups were four + four round trips in a row)
        const int4 p4 = *reinterpret_cast<const int4 *>(P + i0);
        int4 v = *reinterpret_cast<const int4 *>(src + i0);
        const int32_t pq[4] = {p4.x, p4.y, p4.z, p4.w};
        int64_t nd[4];
        int32_t q[4], t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool hop = pq[k] >= 0 && pq[k] != NONE && pq[k] != (int32_t)(i0 + k);
            nd[k] = hop ? ws_node_of(pq[k], (uint32_t)W, ntc) : -1;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = nd[k] >= 0 ? Pn[nd[k]] : NONE;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int32_t pp = pq[k];
            t[k] = -1;
            if (pp == NONE || pp == (int32_t)(i0 + k)) continue;           // flows out unlabelled, or labelled
            if (pp >= 0) {                                                 // an entry cell: what its node has been resolved to
                if (nd[k] < 0) continue;                                   // (a flow cycle inside the tile)
                if (q[k] == pp) pp = q[k] | DONE;                          // the entry cell is labelled itself
                else if (q[k] < 0) pp = q[k];                              // resolved through the entry cell
                else continue;                                             // NONE, or a flow cycle: stays unassigned
            }
            t[k] = pp & ~DONE;
        }
        int32_t lv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lv[k] = src[t[k] >= 0 ? t[k] : 0];
        if (t[0] >= 0) v.x = lv[0];
        if (t[1] >= 0) v.y = lv[1];
        if (t[2] >= 0) v.z = lv[2];
        if (t[3] >= 0) v.w = lv[3];
        *reinterpret_cast<int4 *>(lab + i0) = v;
    } else if (i0 + 4 <= n) {
        const int4 p = *reinterpret_cast<const int4 *>(P + i0);
        const int32_t t0 = target(i0, p.x), t1 = target(i0 + 1, p.y), t2 = target(i0 + 2, p.z), t3 = target(i0 + 3, p.w);
        if (copy) {
            int4 v = *reinterpret_cast<const int4 *>(src + i0);
            if (t0 >= 0) v.x = src[t0];
            if (t1 >= 0) v.y = src[t1];
            if (t2 >= 0) v.z = src[t2];
            if (t3 >= 0) v.w = src[t3];
            *reinterpret_cast<int4 *>(lab + i0) = v;
        } else {      // in place: a labelled cell is never written
            if (t0 >= 0) lab[i0] = src[t0];
            if (t1 >= 0) lab[i0 + 1] = src[t1];
            if (t2 >= 0) lab[i0 + 2] = src[t2];
            if (t3 >= 0) lab[i0 + 3] = src[t3];
        }
    } else {
        for (int64_t i = i0; i < n; ++i) {
            const int32_t t = target(i, P[i]);
            if (t >= 0) lab[i] = src[t];
            else if (copy) lab[i] = src[i];
        }
    }
}

__global__ __launch_bounds__(256) void ws_assign_kernel(const int32_t *__restrict__ P, const uint32_t *__restrict__ Q,
                                                       int32_t *lab, int64_t n, int32_t unassigned)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = P[i];
    if (p >= 0) return;                        // labelled cell, flows out unlabelled, or never resolved (flow cycle)
    const int32_t t = p & ~DONE;
    if (Q && !(Q[t] & QFLAG)) return;          // labelled terminal whose path never meets an edge cell
    lab[i] = lab[t];                           // lab[t] is an input label: labelled cells are never written
}

// band mode: halo rows become terminals carrying pseudo labels -(1+col) (top) / -(1+W+col) (bottom)
__global__ void ws_pseudo_kernel(int32_t *ws, int64_t H, int64_t W, int top, int bottom)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    if (top) ws[c] = -(int32_t)(1 + c);
    if (bottom) ws[(H - 1) * W + c] = -(int32_t)(1 + W + c);
}
// (four cells per thread where the raster allows 16-byte accesses; the store only where a pseudo label was resolved)
__global__ __launch_bounds__(256) void ws_neg_lut_kernel(int32_t *lab, int64_t n, const int32_t *__restrict__ lut, int64_t nlut)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    auto look = [&](int32_t v) -> int32_t { return (v < 0 && (int64_t)(-(int64_t)v - 1) < nlut) ? lut[-(int64_t)v - 1] : v; };
    if (i0 + 4 <= n) {
        const int4 v = *reinterpret_cast<const int4 *>(lab + i0);
        if ((v.x | v.y | v.z | v.w) >= 0) return;      // no negative value among the four
        *reinterpret_cast<int4 *>(lab + i0) = make_int4(look(v.x), look(v.y), look(v.z), look(v.w));
    } else {
        for (int64_t i = i0; i < n; ++i) lab[i] = look(lab[i]);
    }
}

}  // namespace

int band_pseudo_labels_dev(int32_t *d_ws, int64_t H, int64_t W, int top, int bottom, hipStream_t s)
{
    hipLaunchKernelGGL(ws_pseudo_kernel, dim3((unsigned)cdiv(W, 256)), dim3(256), 0, s, d_ws, H, W, top, bottom);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

int negative_lut_dev(int32_t *d_lab, int64_t n, const int32_t *d_lut, int64_t nlut, hipStream_t s)
{
    hipLaunchKernelGGL(ws_neg_lut_kernel, dim3((unsigned)cdiv(cdiv(n, 4), 256)), dim3(256), 0, s, d_lab, n, d_lut, nlut);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

namespace {
// label 0: the largest of the tiles' keys
__global__ __launch_bounds__(1024) void pour_key0_kernel(const unsigned long long *__restrict__ tile_key0, int64_t ntiles, unsigned long long *key)
{
    unsigned long long best = 0;
    for (int64_t t = (int64_t)blockIdx.x * 1024 + threadIdx.x; t < ntiles; t += (int64_t)gridDim.x * 1024) best = tile_key0[t] > best ? tile_key0[t] : best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o);
        best = ob > best ? ob : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(&key[0], best);
}

__global__ __launch_bounds__(256) void pour_finish_kernel(const unsigned long long *__restrict__ key, int64_t nrec, int64_t W, mhip_index_record *rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    mhip_index_record r;
    const unsigned long long k = key[i];
    if (k == 0) {  // no cell (a real key never is 0: its low word is 0xffffffff - cell > 0): label_max_index's initial record
        r.value = -__builtin_inf();
        r.row = -1;
        r.col = -1;
    } else {
        const uint64_t p = 0xffffffffu - (uint32_t)k;
        r.value = (double)(uint32_t)(k >> 32);
        r.row = (int64_t)(p / (uint64_t)W);
        r.col = (int64_t)(p % (uint64_t)W);
    }
    rec[i] = r;
}

}  // namespace

int pour_finish_dev(unsigned long long *d_key, const unsigned long long *d_tile_key0, int64_t ntiles, int64_t nlab, int64_t W, mhip_index_record *d_rec,
                    hipStream_t s)
{
    hipLaunchKernelGGL(pour_key0_kernel, dim3((unsigned)std::min<int64_t>(cdiv(ntiles, 1024), 64)), dim3(1024), 0, s, d_tile_key0, ntiles, d_key);
    hipLaunchKernelGGL(pour_finish_kernel, dim3((unsigned)cdiv(nlab + 1, 256)), dim3(256), 0, s, d_key, nlab + 1, W, d_rec);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

namespace {
// watersheds_dev never leaves the thread that waits for its pour-point candidates without an answer
struct PourNotify {
    PourLink *pl;
    bool done = false;
    void operator()(int v)
    {
        if (pl && pl->notify && !done) pl->notify(pl->arg, v);
        done = true;
    }
    ~PourNotify() { (*this)(0); }
};
}  // namespace

int watersheds_dev(const uint8_t *d_fd, int32_t *d_labels, int64_t H, int64_t W, int32_t unassigned, hipStream_t s, bool band_mode,
                   const unsigned int *d_known_interior_nodir, const int32_t *d_src, PourLink *pour)
{
    PourNotify notify{pour};
    // d_src (optional): the label raster to start from when d_labels does not hold a copy of it yet -- the fast path then reads
    // the labels there and writes every cell of d_labels; the general path makes the copy first
    const int32_t *src = d_src ? d_src : d_labels;
    const int64_t n = H * W;
    if (n >= (int64_t)NONE - 1) {
        set_error("watersheds: %lld cells exceed the int32 index domain", (long long)n);
        return MHIP_ELIMIT;
    }
    const unsigned grid = (unsigned)cdiv(n, 256);
    DevBuf P, Q, flags;
    MH_TRY(P.alloc(4 * (size_t)n));
    MH_TRY(flags.alloc(sizeof(unsigned int) * 4));
    MH_HIP(hipMemsetAsync(flags.p, 0, sizeof(unsigned int) * 4, s));
    unsigned int *d_cnt = flags.as<unsigned int>(), *d_changed = d_cnt + 2;
    unsigned int interior_nodir = 0;
    // flow directions of unknown origin: count; the D8 kernel leaves the number next to its result
    if (!d_known_interior_nodir) hipLaunchKernelGGL(ws_count_interior_nodir, dim3((unsigned)cdiv(cdiv(n, 16), 256)), dim3(256), 0, s, d_fd, H, W, d_cnt);
    MH_HIP(hipMemcpyAsync(&interior_nodir, d_known_interior_nodir ? d_known_interior_nodir : d_cnt, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    uint32_t *q = nullptr;
    if (interior_nodir && band_mode) {
        set_error("watersheds on a row band need every flow path to leave the raster (interior NODIR cell found)");
        return MHIP_EINVAL;
    }
    // The fast path takes "the first labelled cell downstream" without asking whether that cell's own path meets an edge cell: true
    // when every path leaves the raster, i.e. no interior NODIR cell AND no flow cycle.  Directions the library computed itself
    // (d_known_interior_nodir: D8 descends strictly, edges flow outward) have no cycles; a raster of unknown origin may (a labelled
    // cell on a cycle labels nothing in the reference): it takes the general path.  (Row bands always bring their own D8.)
    if (interior_nodir || (!d_known_interior_nodir && !band_mode)) {
        MH_TRY(Q.alloc(4 * (size_t)n));
        q = Q.as<uint32_t>();
    }
    if (!q) {
        const int64_t ntr = cdiv(H, WT), ntc = cdiv(W, WT), ntiles = ntr * ntc;
        const bool cand = pour && pour->dev.mask0 && unassigned == 0 && n < 0xffffffffll;
        DevBuf Pn;
        MH_TRY(Pn.alloc(4 * 256 * (size_t)ntiles));
        hipLaunchKernelGGL(ws_tile_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, d_fd, src, P.as<int32_t>(), H, W, (int)ntc, unassigned,
                           cand ? pour->dev : PourCandDev(), Pn.as<int32_t>());
        if (cand) {
            MH_HIP(hipEventRecord(pour->ev, s));
            notify(1);
        } else {
            notify(0);
        }
        constexpr int MAX_ROUNDS = 40;   // x 5 hops over entry cells; flow cycles end here
        for (int round = 0; round < MAX_ROUNDS;) {
            const int k = round == 0 ? 2 : 1;
            MH_HIP(hipMemsetAsync(d_changed, 0, 8, s));
            for (int j = 0; j < k; ++j)
                hipLaunchKernelGGL(ws_jump_perimeter_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, Pn.as<int32_t>(), W, (int)ntc, ntiles * 256, d_changed + j);
            unsigned int h[2] = {0, 0};
            MH_HIP(hipMemcpyAsync(h, d_changed, 8, hipMemcpyDeviceToHost, s));
            MH_HIP(stream_sync(s));
            round += k;
            if (!h[k - 1]) break;
        }
        hipLaunchKernelGGL(ws_assign_hop_kernel, dim3((unsigned)cdiv(cdiv(n, 4), 256)), dim3(256), 0, s, P.as<int32_t>(), Pn.as<int32_t>(), src, d_labels, n, W,
                           (int)ntc);
        MH_HIP(hipGetLastError());
        MH_HIP(stream_sync(s));
        return MHIP_OK;
    }
    if (src != d_labels) MH_HIP(hipMemcpyAsync(d_labels, src, 4 * (size_t)n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(ws_init_kernel, dim3(grid), dim3(256), 0, s, d_fd, d_labels, P.as<int32_t>(), q, H, W, unassigned);
    // two launches (paths up to 2**10 cells), then one at a time; every launch has its own "still open" flag
    constexpr int MAX_ROUNDS = 32;   // x 5 hops each: paths up to 2**160 cells; flow cycles end here
    for (int round = 0; round < MAX_ROUNDS;) {
        const int k = round == 0 ? 2 : 1;
        MH_HIP(hipMemsetAsync(d_changed, 0, 8, s));
        for (int j = 0; j < k; ++j)
            hipLaunchKernelGGL(ws_jump_kernel, dim3(grid), dim3(256), 0, s, P.as<int32_t>(), q, n, d_changed + j);
        unsigned int h[2] = {0, 0};
        MH_HIP(hipMemcpyAsync(h, d_changed, 8, hipMemcpyDeviceToHost, s));
        MH_HIP(stream_sync(s));
        round += k;
        if (!h[k - 1]) break;
    }
    hipLaunchKernelGGL(ws_assign_kernel, dim3(grid), dim3(256), 0, s, P.as<int32_t>(), q, d_labels, n, unassigned);
    MH_HIP(hipGetLastError());
    MH_HIP(stream_sync(s));
    return MHIP_OK;
}

}  // namespace mh
