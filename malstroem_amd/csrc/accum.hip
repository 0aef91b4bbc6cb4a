// accum.hip -- D8 flow accumulation: tile-local topological walks in LDS + a perimeter graph (gfx950).
//
// Reference: flow.accumulated_flow (_flow.pyx:256-273, python flow.py:344-364) with
// trace_accumulated_flow (_flow.pyx:225-247): accum[c] = 1 + sum(accum[n]) over in-raster neighbours n whose
// flow direction points at c (flowdir[n] == (dir(c->n)+4)%8, codes > 7 never flow, _flow.pyx:212-222).
// Cells on a flow cycle, and everything downstream of one, stay 0.  Values are integers < 2**53, so any
// summation order is bit-exact in float64.  A NODIR cell (code > 7) receives but does not forward (the
// reference leaves that step undefined).
//
// A single global "last arriver continues" walk needs one returning 64-bit global atomic per cell (measured:
// ~5.4 G atomics/s, 50 ms at 16384^2).  Instead (after Barnes 2017, restated for LDS):
//   phase 1  per 64x64 tile, in LDS: the tile-local accumulation (external inflow ignored) gives every cell's
//            tile-local sum; for each perimeter cell we publish {local sum, resolved?, leaves-the-tile?} and for each
//            ENTRY cell (has an upstream neighbour outside the tile) the perimeter cell where its local path exits.
//   phase 2  global, on perimeter cells only (~1/16 of the raster): exit cell x forwards F(x) = local(x) + all flux
//            routed through x to the entry cell it flows into, and on to that entry's exit: the same packed
//            64-bit "pending count | sum" walk as before, but over ~4 % of the cells.
//   phase 3  per tile, in LDS again: the same accumulation with every entry cell pre-loaded with its external
//            inflow produces the final values, written once as float64.
// Tile-local accumulation = pointer doubling on the in-tile flow forest (a Kahn walk was measured chain-bound: the
// longest in-tile flow path, ~260 cells on fBm terrain, at ~570 cycles per step with 0.3 of 64 lanes busy).  With
// A_k[w] the 2**k-th downstream cell of w (SENT once the path has left the tile / ended) and S_k[v] the sum over the
// cells at most 2**k - 1 steps upstream of v:   S_{k+1}[A_k[w]] += S_k[w]  for every w,  A_{k+1}[w] = A_k[A_k[w]].
// ~log2(longest path) rounds of independent LDS atomics instead of a dependent chain.  A and S share one 64-bit LDS
// word, so the push and the read of A_k[A_k[w]] are ONE returning atomic add.  Bit 63 of S is a taint flag
// (a cell that can never be resolved: unknown band halo, entry whose inflow never arrives, flow cycle); it travels
// downstream with the sums and turns the result into the reference's 0.
#include "common.hpp"
#include <type_traits>

namespace mh {
namespace {

constexpr int AT = 64;                // tile edge
constexpr int PERIM = 4 * AT - 4;     // perimeter cells of a tile
constexpr int NODE_STRIDE = 256;      // perimeter slots per tile in the global node arrays
constexpr int FS = AT + 32;           // LDS row stride of the flow-direction window: column c sits at byte c + 16, so the
                                      // 16-byte chunks of a row land 16-byte aligned (ring columns at 15 and AT + 16)
constexpr int WOFF = 16;
constexpr uint64_t SRC = 1ull << 63;      // phase-2 node word: source flag
// Tile state in LDS, two arrays (one 64-bit word per cell pushed to with a RETURNING 64-bit atomic was the first design: the LDS
// atomic unit was the bound, and a returning 64-bit atomic is its most expensive operation):
//   S[i]  running sum, bit 31 / 63 = taint.  32-bit when no sum can pass 2**31 (phase 1 outside the final band pass: at most
//         the 4096 cells of the tile), else 64-bit with sums below 2**38 (accum_dev refuses larger rasters)
//   P[i]  A (2**k-th downstream cell, SENT13 = none) | R << 13 (last in-tile cell reached so far; phase 1 only: 16-bit words
//         without R in phase 3).  Written by the owner of the cell between two barriers only, so the push phase reads it plainly.
constexpr uint64_t TAINT = 1ull << 63;    // never resolved (64-bit sums; TAINT32 in the 32-bit ones)
constexpr uint32_t TAINT32 = 1u << 31;
constexpr int R_SHIFT = 13;
constexpr uint32_t A_MASK = 0x1fffu, R_MASK = 0xfffu, SENT13 = A_MASK;
constexpr uint16_t NO_EXIT = 0xffffu;
#ifndef ACC_ATN
#define ACC_ATN 256
#endif
constexpr int ATN = ACC_ATN;              // threads per tile
constexpr int CPT = AT * AT / ATN;        // cells per thread
constexpr int MAX_DOUBLINGS = 12;         // 2**12 = cells of a tile >= any simple path
// phase-2 node word: an exit cell can be fed by every entry of its tile (hundreds), so the pending field is wider
constexpr int G_SHIFT = 44;
constexpr uint64_t G_ONE = 1ull << G_SHIFT, G_SUM = G_ONE - 1, G_PEND = 0x7ffffull;

enum : uint8_t { F_EXIT = 1, F_RESOLVED = 2 };
// The external in-degree of a perimeter cell (upstream neighbours outside its tile, at most 5) is PUSHED by the tiles the flux
// comes from: an exit cell adds DEXT_ONE to the `arrived` word of the cell it flows into (phase 1; ~80 atomics per tile).  Reading
// it -- the window's ring of neighbour codes, whose two columns cost a sector per cell: 4 B fetched per cell for 1 B of flow
// directions -- is gone.  Low half of the word: the deliveries of phase 2.  A cell with an external in-degree is an ENTRY.
constexpr uint32_t DEXT_ONE = 1u << 16, ARRIVED_MASK = 0xffffu;

__device__ __forceinline__ bool flows_into(unsigned code, int k_from_me) { return code <= 7u && code == (unsigned)((k_from_me + 4) & 7); }

__device__ __forceinline__ int perim_slot(int r, int c)
{
    if (r == 0) return c;
    if (r == AT - 1) return AT + c;
    if (c == 0) return 2 * AT + (r - 1);
    if (c == AT - 1) return 2 * AT + (AT - 2) + (r - 1);
    return -1;
}
__device__ __forceinline__ void perim_cell(int p, int &r, int &c)
{
    if (p < AT) { r = 0; c = p; }
    else if (p < 2 * AT) { r = AT - 1; c = p - AT; }
    else if (p < 2 * AT + (AT - 2)) { r = p - 2 * AT + 1; c = 0; }
    else { r = p - 2 * AT - (AT - 2) + 1; c = AT - 1; }
}

struct Nodes {           // global perimeter-node arrays, index = tile * NODE_STRIDE + slot
    uint64_t *gstate;    // phase 2 walk state: pending | F
    uint64_t *inflow;    // per ENTRY cell: sum of delivered flux
    uint32_t *arrived;   // per ENTRY cell: number of deliveries | external in-degree << 16 (zeroed before phase 1)
    int32_t *next;       // per EXIT cell: node index of the exit its flux continues to (-1: none)
    int32_t *dst;        // per EXIT cell: node index of the entry cell it flows into
    uint16_t *exit_of;   // per ENTRY cell: slot of the exit of its tile-local path (NO_EXIT: ends inside / leaves raster)
    uint8_t *flags;
    int32_t *halo_first;  // band mode, boundary pass only (else nullptr): per halo cell (top row: [0, W), bottom row: [W, 2W))
                          // where its path goes in its first tile, see accum_tile_kernel
    int32_t *bexit;       // ... and per entry node: side * W + column of the cell from which its tile-local path flows into a halo row
};

#ifdef MH_PROFILE_ACCUM   // development aid: per-phase ticks of thread 0 of every block
__device__ unsigned long long g_accum_prof[2][8];
#define MH_ASTAMP(k) do { if (threadIdx.x == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_accum_prof[FINAL][k], (unsigned long long)(t_ - tprev_)); tprev_ = t_; } } while (0)
#else
#define MH_ASTAMP(k)
#endif

// ---- the tile kernel (phase 1 when FINAL == false, phase 3 when FINAL == true) ------------------------------
// WIDE: 64-bit sums (a row band's final pass, where the halo cells are sources of the neighbouring band's flux -- sums of the
// whole raster -- and any raster of 2**31 cells or more; everything else stays below 2**31 in every sum)
// STOREL (phase 1 of a raster that is no row band and has fewer than 2**31 cells): the tile-local sums are kept -- 16-bit
// words (sum <= 4096 | taint << 15) at the start of each tile row's own segment of `out` -- for accum_final_walk_kernel.
template <bool FINAL, bool WIDE, bool STOREL = false>
// Row-band mode: local row 0 / H-1 may be a HALO row owned by the neighbouring band.  Its cells carry the neighbour's
// final value in `out` (> 0: known, acts as a source of that much flux; <= 0: not known yet, blocks everything below it);
// they never receive and are never written here.
#ifdef ACC_WAVES
__attribute__((amdgpu_waves_per_eu(ACC_WAVES, ACC_WAVES)))
#endif
__global__ __launch_bounds__(ATN, WIDE ? 1 : 4) void accum_tile_kernel(const uint8_t *__restrict__ fd, double *__restrict__ out, int64_t H,
                                                        int64_t W, int ntc, Nodes nd, int fixed_top, int fixed_bot, int halo_zero)
{
    using sum_t = typename std::conditional<WIDE, unsigned long long, uint32_t>::type;
    using ptr_t = typename std::conditional<FINAL, uint16_t, uint32_t>::type;
    constexpr sum_t TAINT_S = WIDE ? (sum_t)TAINT : (sum_t)TAINT32;
    auto halo_row = [&](int64_t rr) { return (fixed_top && rr == 0) || (fixed_bot && rr == H - 1); };
    __shared__ sum_t S[AT * AT];
    __shared__ ptr_t P[AT * AT];
    const int tile = blockIdx.x;
    const int ti = tile / ntc, tj = tile - ti * ntc;
    const int64_t r0 = (int64_t)ti * AT, c0 = (int64_t)tj * AT;
    const int tid = threadIdx.x;
#ifdef MH_PROFILE_ACCUM
    long long tprev_ = __builtin_amdgcn_s_memtime();
#endif

    // flow-direction window incl. the 1-cell ring; outside the raster = NODIR (never flows, never receives).
    // Per window row: four 16-byte chunks (the tile's own columns) + the two ring bytes.
    __shared__ __attribute__((aligned(16))) uint8_t win[(AT + 2) * FS];
    __shared__ uint64_t inflow_l[FINAL ? NODE_STRIDE : 1];
    __shared__ uint32_t arrived_l[FINAL ? NODE_STRIDE : 1];
    if (FINAL && tid < NODE_STRIDE) {  // what phase 2 delivered to my perimeter cells: one coalesced read per array
        inflow_l[tid] = nd.inflow[(int64_t)tile * NODE_STRIDE + tid];
        arrived_l[tid] = nd.arrived[(int64_t)tile * NODE_STRIDE + tid];
    }
    const bool wide = (W % 16) == 0 && c0 + AT <= W;   // c0 is a multiple of 64: chunks are 16-byte aligned in global memory
    // (the loads of a thread's items first, then the LDS writes: one round trip instead of one per item)
    constexpr int NWQ = ((AT + 2) * 6 + ATN - 1) / ATN;
    uint4 wv[NWQ];
    uint8_t wb[NWQ];
#pragma unroll
    for (int u = 0; u < NWQ; ++u) {
        const int q = tid + u * ATN;
        const int wr = q / 6, k = q - wr * 6;
        const int64_t rr = r0 + wr - 1;
        const bool row_in = q < (AT + 2) * 6 && wr >= 1 && wr <= AT && rr < H;   // (the ring stays NODIR: nothing reads the neighbours' codes)
        wv[u] = make_uint4(0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u);
        wb[u] = 8;
        if (k < 4) {
            if (row_in) {
                if (wide) wv[u] = *reinterpret_cast<const uint4 *>(fd + rr * W + c0 + 16 * k);
                else {
                    uint8_t b[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int64_t cc = c0 + 16 * k + t;
                        b[t] = cc < W ? fd[rr * W + cc] : (uint8_t)8;
                    }
                    memcpy(&wv[u], b, 16);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NWQ; ++u) {
        const int q = tid + u * ATN;
        if (q >= (AT + 2) * 6) continue;
        const int wr = q / 6, k = q - wr * 6;
        if (k < 4) *reinterpret_cast<uint4 *>(&win[wr * FS + WOFF + 16 * k]) = wv[u];
        else win[wr * FS + (k == 4 ? WOFF - 1 : WOFF + AT)] = wb[u];
    }
    __syncthreads();
    MH_ASTAMP(0);

    // external in-degree of the perimeter cells (upstream neighbours outside the tile): pushed by those neighbours' tiles in
    // phase 1, read here by the final pass only
    __shared__ uint8_t dext_l[NODE_STRIDE];
    if (tid < NODE_STRIDE) dext_l[tid] = FINAL ? (uint8_t)(arrived_l[tid] >> 16) : (uint8_t)0;
    __syncthreads();

    // my CPT cells: i = tid + ATN j (consecutive lanes = consecutive LDS words)
    sum_t sreg[CPT];      // my cells' sums as of the last barrier (what I push)
    uint32_t preg[CPT];   // ... and their A | R << 13
    // (a tile that lies inside the raster and holds no halo row -- all but the last row / column of tiles -- needs none of the bounds
    // and halo tests below: a dozen 64-bit compares per cell)
    auto init_cells = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int i = tid + ATN * j;
        const int r = i / AT, c = i - r * AT;
        const bool inside = FULL || ((r0 + r) < H && (c0 + c) < W);
        const int slot = perim_slot(r, c);
        unsigned deg_ext = slot >= 0 ? dext_l[slot] : 0u;
        sum_t v = inside ? (sum_t)1 : TAINT_S;   // not a raster cell: nothing flows into it (its neighbours see NODIR... it has none)
        const bool halo = !FULL && inside && halo_row(r0 + r);
        if (halo) {  // the neighbouring band's cell: known (> 0) = a source of that much flux, else it blocks its path
            // boundary pass of the band protocol (halo_zero): a known source of NO flux -- the local sums then are the band's own
            // contribution, and the halo cell's path is traced to where it leaves the band (accum_band_exit_kernel)
            const double ext = halo_zero ? 0.0 : out[(r0 + r) * W + c0 + c];
            v = halo_zero ? (sum_t)0 : (ext > 0.0 ? (sum_t)(unsigned long long)ext : TAINT_S);   // (not halo_zero: a WIDE launch)
            deg_ext = 0;
        }
        if (FINAL && deg_ext && inside) {
            if ((arrived_l[slot] & ARRIVED_MASK) == deg_ext) v += (sum_t)inflow_l[slot];
            else v |= TAINT_S;  // some upstream flux never arrives (flow cycle upstream): stays unresolved => 0
        }
        // downstream cell: inside the tile, the raster and the band, else the path ends here
        const unsigned code = win[(r + 1) * FS + c + WOFF];
        uint32_t nx = SENT13;
        if (inside && code <= 7u) {
            const int nr = r + dir_dr((int)code), nc = c + dir_dc((int)code);
            if (nr >= 0 && nr < AT && nc >= 0 && nc < AT && (FULL || ((r0 + nr) < H && (c0 + nc) < W && !halo_row(r0 + nr))))
                nx = (uint32_t)(nr * AT + nc);
        }
        const uint32_t pw = nx | ((uint32_t)i << R_SHIFT);
        S[i] = v;
        P[i] = (ptr_t)pw;       // (phase 3 keeps A only)
        sreg[j] = v;
        preg[j] = pw;
    }
    };
    if (r0 + AT <= H && c0 + AT <= W && !fixed_top && !fixed_bot) init_cells(std::true_type{});
    else init_cells(std::false_type{});
    __syncthreads();
    MH_ASTAMP(1);

    bool more = true;
    for (int round = 0; round < MAX_DOUBLINGS && more; ++round) {
        uint32_t got[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const uint32_t a = preg[j] & A_MASK;
            got[j] = 0;
            if (a != SENT13) {
                // push my sum to A_k[me] (no value comes back) and read A_k, R_k of that cell
                const sum_t v = sreg[j];
                atomicAdd(&S[a], (sum_t)(v & ~TAINT_S));
                if (v & TAINT_S) atomicOr(&S[a], TAINT_S);
                got[j] = P[a];
            }
        }
        __syncthreads();   // every push of this round is done; nobody but its owner touches a word until the next barrier
        bool mine = false;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int i = tid + ATN * j;
            if ((preg[j] & A_MASK) != SENT13) {
                const uint32_t w = FINAL ? (got[j] & A_MASK) : got[j];   // A_{k+1} | R_{k+1}
                P[i] = (ptr_t)w;
                preg[j] = w;
                sreg[j] = S[i];                                           // S_{k+1}
                mine |= (w & A_MASK) != SENT13;
            }
        }
        more = __syncthreads_or(mine) != 0;   // (a one-barrier vote through three LDS flags -- common.hpp: WgVote -- measured slower here: 3.61 -> 3.83 ms)
    }
    if (more) {
        // still walking after 2**12 steps: the path has entered a flow cycle, and A is a cell ON the cycle; over all
        // such paths these ancestors cover every cell of the cycle (a rotation of the cycle is onto)
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const uint32_t a = preg[j] & A_MASK;
            if (a != SENT13) atomicOr(&S[a], TAINT_S);
        }
        __syncthreads();
    }
    MH_ASTAMP(2);

    if (FINAL) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int i = tid + ATN * j;
            const int r = i / AT, c = i - r * AT;
            if ((r0 + r) < H && (c0 + c) < W && !halo_row(r0 + r)) {
                const sum_t s = S[i];
                out[(r0 + r) * W + c0 + c] = (s & TAINT_S) ? 0.0 : (double)(s & ~TAINT_S);
            }
        }
        MH_ASTAMP(3);
        return;
    }

    if (STOREL) {   // (a wave writes one tile row per step: 256 contiguous bytes)
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int i = tid + ATN * j;
            const int r = i / AT, c = i - r * AT;
            if ((r0 + r) < H && (c0 + c) < W)     // (at most the 4096 cells of the tile: 13 bits + the taint)
                reinterpret_cast<uint16_t *>(out + (r0 + r) * W + c0)[c] = (uint16_t)((S[i] & 0x7fffu) | ((S[i] & TAINT_S) ? 0x8000u : 0u));
        }
    }
    // boundary pass: where does the path of every halo cell of this tile go?  -1: it ends inside the band (or never enters it),
    // <= -2: it leaves the band again from the cell  -2 - value = side * W + column,  >= 0: it leaves the tile through that node
    auto band_side = [&](int64_t gr) { return (fixed_top && gr == 0) ? (int64_t)0 : W; };
    if (nd.halo_first) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int i = tid + ATN * j;
            const int r = i / AT, c = i - r * AT;
            if (!((r0 + r) < H && (c0 + c) < W && halo_row(r0 + r))) continue;
            const uint32_t pw = P[i];
            int32_t first = -1;
            const unsigned own = win[(r + 1) * FS + c + WOFF];
            const bool enters = own <= 7u && dir_dr((int)own) != 0 && (r0 + r + dir_dr((int)own)) >= 0 && (r0 + r + dir_dr((int)own)) < H &&
                                (c0 + c + dir_dc((int)own)) >= 0 && (c0 + c + dir_dc((int)own)) < W && !halo_row(r0 + r + dir_dr((int)own));
            if (enters && (pw & A_MASK) == SENT13) {
                const int last = (int)((pw >> R_SHIFT) & R_MASK);
                const int pr = last / AT, pc = last - pr * AT;
                const unsigned cd = win[(pr + 1) * FS + pc + WOFF];
                if (cd <= 7u) {
                    const int nr = pr + dir_dr((int)cd), nc = pc + dir_dc((int)cd);
                    const int64_t gr = r0 + nr, gc = c0 + nc;
                    const bool in_raster = gr >= 0 && gr < H && gc >= 0 && gc < W;
                    const bool out_of_tile = nr < 0 || nr >= AT || nc < 0 || nc >= AT;
                    if (in_raster && last != i && halo_row(gr)) first = (int32_t)(-2 - (band_side(gr) + c0 + pc));
                    else if (in_raster && out_of_tile) first = tile * NODE_STRIDE + perim_slot(pr, pc);
                }
            }
            nd.halo_first[band_side(r0 + r) + c0 + c] = first;
        }
    }
    // phase 1: publish the perimeter
    if (tid < PERIM) {
        int r, c;
        perim_cell(tid, r, c);
        const int64_t node = (int64_t)tile * NODE_STRIDE + tid;
        const bool inside = (r0 + r) < H && (c0 + c) < W;
        const sum_t s = S[r * AT + c];
        const uint32_t pw = P[r * AT + c];
        const bool resolved = inside && !(s & TAINT_S);
        const unsigned code = win[(r + 1) * FS + c + WOFF];
        uint8_t fl = resolved ? F_RESOLVED : 0;
        int32_t dst = -1;
        if (inside && code <= 7u) {
            const int nr = r + dir_dr((int)code), nc = c + dir_dc((int)code);
            const int64_t gr = r0 + nr, gc = c0 + nc;
            if ((nr < 0 || nr >= AT || nc < 0 || nc >= AT) && gr >= 0 && gr < H && gc >= 0 && gc < W) {
                fl |= F_EXIT;
                const int t2 = (int)(gr / AT) * ntc + (int)(gc / AT);
                dst = t2 * NODE_STRIDE + perim_slot((int)(gr % AT), (int)(gc % AT));
            }
        }
        // a possible entry cell (whether a neighbouring tile flows into it is only known when every tile has pushed its exits):
        // follow its tile-local path to the cell where it leaves the tile
        const bool entry = inside && !halo_row(r0 + r);
        uint16_t ex = NO_EXIT;
        int32_t bex = -1;      // band mode: my path ends by flowing into a halo row from this cell of the first / last owned row
        if (entry) {
            // the last in-tile cell of my path (doubling above); the path leaves through it iff that cell flows into a
            // raster cell outside the tile.  A path that ends in a sink, leaves the raster, continues in the neighbouring
            // band or runs into a flow cycle has no exit.
            if ((pw & A_MASK) == SENT13) {
                const int last = (int)((pw >> R_SHIFT) & R_MASK);
                const int pr = last / AT, pc = last - pr * AT;
                const unsigned cd = win[(pr + 1) * FS + pc + WOFF];
                if (cd <= 7u) {
                    const int nr = pr + dir_dr((int)cd), nc = pc + dir_dc((int)cd);
                    const int64_t gr = r0 + nr, gc = c0 + nc;
                    if ((nr < 0 || nr >= AT || nc < 0 || nc >= AT) && gr >= 0 && gr < H && gc >= 0 && gc < W) ex = (uint16_t)perim_slot(pr, pc);
                    if (gr >= 0 && gr < H && gc >= 0 && gc < W && halo_row(gr)) bex = (int32_t)(band_side(gr) + c0 + pc);
                }
            }
        }
        if (nd.bexit) nd.bexit[node] = bex;
        nd.flags[node] = fl;
        nd.dst[node] = dst;
        nd.exit_of[node] = ex;
        nd.gstate[node] = ((resolved ? 0ull : 1ull) << G_SHIFT) | (unsigned long long)(s & ~TAINT_S);  // unresolved: blocks itself forever
        nd.next[node] = -1;
        if (dst >= 0) atomicAdd(&nd.arrived[dst], DEXT_ONE);      // (inflow / arrived: zeroed by the host before this launch)
    }
    MH_ASTAMP(3);
}

// ---- phase 3 without a second doubling (one context, fewer than 2**31 cells) ------------------------------------------------
// final[c] = local[c] + the external inflow of every ENTRY cell whose tile-local path runs through c.  Entries are few (60 of a
// tile's 252 perimeter cells on the benchmark terrain) and their paths short (38 cells on average, the longest of a tile 70), and
// they cover a sixth of the cells: every entry WALKS its path and adds its inflow -- one non-returning LDS add and one pointer
// read per step, ~2300 steps per tile on one or two wavefronts -- where the doubling pushed 5 x 4096 sums.  The local sums come
// from phase 1 (STOREL), the pointers from the tile's own flow directions (no ring needed: the external in-degree of an entry is
// in its node flags).  A cell that phase 1 left tainted ends every walk that reaches it (all of its downstream cells are
// tainted as well); an entry whose inflow never arrived taints its path.
// `pc.key` (optional): the pour points on the way (common.hpp: PourCandDev) -- the final sums of the tile's candidate cells go
// into the keys of their labels.
// Row bands (fixed_top / fixed_bot: local row 0 / H - 1 is a halo row): the BOUNDARY pass only, where the halo cells are sources of no
// flux -- phase 1 has given them the local sum 0 and cut every path that flows into a halo row; here they start no walk, end every
// walk that would step onto them and are not written (their words in `out` are scratch until the exchange fills the halo rows).
__global__ __launch_bounds__(ATN) void accum_final_walk_kernel(const uint8_t *__restrict__ fd, double *__restrict__ out, int64_t H, int64_t W, int ntc, Nodes nd,
                                                              PourCandDev pc, int fixed_top, int fixed_bot)
{
    auto halo_row = [&](int64_t rr) { return (fixed_top && rr == 0) || (fixed_bot && rr == H - 1); };
    __shared__ __attribute__((aligned(16))) uint32_t S[AT * AT];
    __shared__ __attribute__((aligned(16))) uint16_t P[AT * AT];
    __shared__ uint32_t wl_cell[NODE_STRIDE], wl_add[NODE_STRIDE];
    __shared__ uint32_t wl_n;
    const int tile = blockIdx.x;
    const int ti = tile / ntc, tj = tile - ti * ntc;
    const int64_t r0 = (int64_t)ti * AT, c0 = (int64_t)tj * AT;
    const int tid = threadIdx.x;
    if (tid == 0) wl_n = 0;
    // node data of my perimeter cell first (the longest dependent chain of the kernel starts here)
    uint64_t inflow = 0;
    uint32_t arrived = 0;
    if (tid < PERIM) {
        const int64_t node = (int64_t)tile * NODE_STRIDE + tid;
        inflow = nd.inflow[node];
        arrived = nd.arrived[node];
    }
    const bool wide = (W % 16) == 0 && c0 + AT <= W;
    // local sums and pointers: a lane takes four cells of a row (16 lanes per row: 256 B of sums, 64 B of directions), 16 rows
    // per step of the workgroup; LDS in 16- / 8-byte writes
    constexpr int NQ = AT * 16 / ATN;      // steps of the workgroup over the tile's 64 rows (16 lanes per row)
    static_assert(AT * 16 % ATN == 0, "whole steps");
    // (all loads of the tile first: a step's loads were waited for before the next step's went out -- four round trips in a row)
    uint2 lq[NQ];
    uint32_t fq[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = tid + u * ATN, r = q >> 4, c4 = (q & 15) * 4;
        const int64_t rr = r0 + r;
        lq[u] = make_uint2(0x80008000u, 0x80008000u);
        fq[u] = 0x08080808u;
        if (rr < H && wide) {
            lq[u] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(out + rr * W + c0) + c4);
            fq[u] = *reinterpret_cast<const uint32_t *>(fd + rr * W + c0 + c4);
        }
    }
    // (FULL: the tile lies inside the raster -- no bounds tests per cell)
    auto build = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = tid + u * ATN, r = q >> 4, c4 = (q & 15) * 4;
        const int64_t rr = r0 + r;
        uint32_t l[4];
        uint8_t b[4];
        if (FULL || (rr < H && wide)) {
            const uint2 v = lq[u];
            l[0] = v.x & 0xffffu; l[1] = v.x >> 16; l[2] = v.y & 0xffffu; l[3] = v.y >> 16;
            const uint32_t f = fq[u];
            memcpy(b, &f, 4);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t cc = c0 + c4 + t;
                const bool in = rr < H && cc < W;
                l[t] = in ? reinterpret_cast<const uint16_t *>(out + rr * W + c0)[c4 + t] : 0x8000u;
                b[t] = in ? fd[rr * W + cc] : (uint8_t)8;
            }
        }
        uint16_t nx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            l[t] = (l[t] & 0x7fffu) | ((l[t] & 0x8000u) ? TAINT32 : 0u);
            const int c = c4 + t;
            const unsigned code = b[t];
            nx[t] = (uint16_t)SENT13;
            if (code <= 7u && !(l[t] & TAINT32)) {
                const int nr = r + dir_dr((int)code), nc = c + dir_dc((int)code);
                if (nr >= 0 && nr < AT && nc >= 0 && nc < AT && (FULL || ((r0 + nr) < H && (c0 + nc) < W && !halo_row(r0 + nr)))) nx[t] = (uint16_t)(nr * AT + nc);
            }
        }
        *reinterpret_cast<uint4 *>(&S[r * AT + c4]) = make_uint4(l[0], l[1], l[2], l[3]);
        *reinterpret_cast<uint2 *>(&P[r * AT + c4]) = make_uint2((uint32_t)nx[0] | ((uint32_t)nx[1] << 16), (uint32_t)nx[2] | ((uint32_t)nx[3] << 16));
    }
    };
    if (wide && r0 + AT <= H && !(fixed_top && ti == 0) && !(fixed_bot && r0 + AT >= H)) build(std::true_type{});
    else build(std::false_type{});
    __syncthreads();
    // the walkers: every entry with its inflow (bit 31: the inflow never arrived -- the path is tainted)
    unsigned dext = arrived >> 16;
    if (tid < PERIM && dext && (fixed_top || fixed_bot)) {      // (a band's halo cell is no entry, whatever flowed at it)
        int r, c;
        perim_cell(tid, r, c);
        if (halo_row(r0 + r)) dext = 0;
    }
    if (tid < PERIM && dext) {
        int r, c;
        perim_cell(tid, r, c);
        const uint32_t k = atomicAdd(&wl_n, 1u);
        wl_cell[k] = (uint32_t)(r * AT + c);
        wl_add[k] = (arrived & ARRIVED_MASK) == dext ? (uint32_t)inflow : TAINT32;
    }
    __syncthreads();
    const uint32_t nw = wl_n;
    if ((uint32_t)tid < nw) {
        uint32_t a = wl_cell[tid];
        const uint32_t v = wl_add[tid];
        if (v & TAINT32) {
            do {
                atomicOr(&S[a], TAINT32);
                a = P[a];
            } while (a != SENT13);
        } else {
            do {
                atomicAdd(&S[a], v);
                a = P[a];
            } while (a != SENT13);
        }
    }
    __syncthreads();
    if (pc.key) {
        auto key_of = [&](int lr, int lc, bool *bad) -> unsigned long long {
            const uint32_t sv = S[lr * AT + lc];
            if (sv & TAINT32) *bad = true;      // (an unresolved cell: the caller runs the general pass)
            const uint32_t cell = (uint32_t)((r0 + lr) * W + c0 + lc);
            return ((unsigned long long)((sv & TAINT32) ? 0u : sv) << 32) | (unsigned long long)(0xffffffffu - cell);
        };
        bool bad = false;
        // unlabelled candidates: one key for the tile
        unsigned long long best = 0;
        {
            uint32_t m = pc.mask0[(int64_t)tile * 256 + tid];
            const int lr = tid >> 2, cb = (tid & 3) * 16;
            while (m) {
                const int k = __builtin_ctz(m);
                m &= m - 1;
                const unsigned long long kk = key_of(lr, cb + k, &bad);
                best = kk > best ? kk : best;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ob = __shfl_xor(best, o);
            best = ob > best ? ob : best;
        }
        __shared__ unsigned long long best_w[ATN / 64];
        if ((tid & 63) == 0) best_w[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < ATN / 64; ++w) best = best_w[w] > best ? best_w[w] : best;
            pc.tile_key0[tile] = best;
        }
        // labelled candidates: a handful per label
        const uint32_t off = (uint32_t)tile * POUR_TILE_CAP, cnt = pc.tile_cnt[tile];
        for (uint32_t i = tid; i < cnt; i += ATN) {
            const uint2 e = pc.list[off + i];
            const uint32_t gr = e.x / (uint32_t)W, gc = e.x - gr * (uint32_t)W;
            if (e.y > pc.nlab) {
                bad = true;
                continue;
            }
            atomicMax(&pc.key[e.y], key_of((int)(gr - (uint32_t)r0), (int)(gc - (uint32_t)c0), &bad));
        }
        if (bad) pc.flags[1] = 1u;
    }
    uint32_t anyt = 0;     // an unresolved cell anywhere (a label may consist of a flow cycle alone: no candidate, and its record is (0, first cell))
    for (int q = tid; q < AT * 16; q += ATN) {
        const int r = q >> 4, c4 = (q & 15) * 4;
        const int64_t rr = r0 + r;
        if (rr >= H || halo_row(rr)) continue;
        typedef double __attribute__((ext_vector_type(2))) v2d;
        const uint4 sv = *reinterpret_cast<const uint4 *>(&S[r * AT + c4]);
        const uint32_t sq[4] = {sv.x, sv.y, sv.z, sv.w};
        if (wide) {
            anyt |= sq[0] | sq[1] | sq[2] | sq[3];
            *reinterpret_cast<v2d *>(out + rr * W + c0 + c4) = v2d{(sq[0] & TAINT32) ? 0.0 : (double)sq[0], (sq[1] & TAINT32) ? 0.0 : (double)sq[1]};
            *reinterpret_cast<v2d *>(out + rr * W + c0 + c4 + 2) = v2d{(sq[2] & TAINT32) ? 0.0 : (double)sq[2], (sq[3] & TAINT32) ? 0.0 : (double)sq[3]};
        } else {
            for (int t = 0; t < 4; ++t)
                if (c0 + c4 + t < W) {
                    anyt |= sq[t];
                    out[rr * W + c0 + c4 + t] = (sq[t] & TAINT32) ? 0.0 : (double)sq[t];
                }
        }
    }
    if (pc.key && (anyt & TAINT32)) pc.flags[1] = 1u;
}

// ---- phase 2: the perimeter graph -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void accum_link_kernel(Nodes nd, int64_t nnodes)
{
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nnodes || (x % NODE_STRIDE) >= PERIM) return;
    if (!(nd.flags[x] & F_EXIT)) return;
    const int32_t e = nd.dst[x];
    const uint16_t ex = nd.exit_of[e];
    if (ex == NO_EXIT) return;
    const int32_t nx = (e / NODE_STRIDE) * NODE_STRIDE + ex;
    nd.next[x] = nx;
    atomicAdd(reinterpret_cast<unsigned long long *>(&nd.gstate[nx]), (unsigned long long)G_ONE);
}

__global__ __launch_bounds__(256) void accum_mark_kernel(Nodes nd, int64_t nnodes)
{
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nnodes || (x % NODE_STRIDE) >= PERIM) return;
    if ((nd.flags[x] & F_EXIT) && ((nd.gstate[x] >> G_SHIFT) & G_PEND) == 0) nd.gstate[x] |= SRC;
}

__global__ __launch_bounds__(256) void accum_graph_walk_kernel(Nodes nd, int64_t nnodes)
{
    int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nnodes || (x % NODE_STRIDE) >= PERIM) return;
    const uint64_t s0 = nd.gstate[x];
    if (!(s0 & SRC)) return;
    uint64_t total = s0 & G_SUM;
    // the walk is a chain of dependent round trips (a river crosses hundreds of tiles): the links of the next node are fetched
    // while its pending count comes back, one round trip per step instead of two
    int32_t e = nd.dst[x], nx = nd.next[x];
    for (;;) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&nd.inflow[e]), (unsigned long long)total);
        atomicAdd(&nd.arrived[e], 1u);
        if (nx < 0) break;
        const int32_t e2 = nd.dst[nx], nx2 = nd.next[nx];
        const uint64_t delta = total - G_ONE;
        const uint64_t now = atomicAdd(reinterpret_cast<unsigned long long *>(&nd.gstate[nx]), (unsigned long long)delta) + delta;
        if ((now >> G_SHIFT) & G_PEND) break;
        total = now & G_SUM;
        x = nx;
        e = e2;
        nx = nx2;
    }
}

}  // namespace

namespace {
// Band protocol, boundary pass: follow every halo cell's path over the perimeter graph to the cell where it leaves the band
// again.  exit_map[k] (k as in Nodes::halo_first) = side * W + column of that EXIT cell (side 0: first owned row, 1: last owned
// row), or -1 when the flux of the halo cell stays inside the band (sink, raster border, flow cycle).
__global__ __launch_bounds__(256) void accum_band_exit_kernel(Nodes nd, int64_t H, int64_t W, int ntc, int64_t ntiles, int fixed_top, int fixed_bot,
                                                             int32_t *exit_map)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * W) return;
    int32_t res = -1;
    const bool present = k < W ? fixed_top != 0 : fixed_bot != 0;
    int32_t x = present ? nd.halo_first[k] : -1;
    if (x <= -2) res = -2 - x;
    for (int64_t step = 0; x >= 0 && step < ntiles * PERIM + 4; ++step) {     // x: an F_EXIT node; a simple path visits an exit node once (a meander crosses a tile's outline many times): longer = a flow cycle
        const int32_t d = nd.dst[x];
        int dr_, dc_, xr, xc;
        perim_cell(d % NODE_STRIDE, dr_, dc_);
        const int64_t dgr = (int64_t)((d / NODE_STRIDE) / ntc) * AT + dr_;
        if ((fixed_top && dgr == 0) || (fixed_bot && dgr == H - 1)) {     // x flows into a halo row (x is an owned cell: halo -> halo never gets here)
            perim_cell(x % NODE_STRIDE, xr, xc);
            res = (int32_t)((dgr == 0 && fixed_top ? 0 : W) + (int64_t)((x / NODE_STRIDE) % ntc) * AT + xc);
            break;
        }
        const uint16_t ex = nd.exit_of[d];
        if (ex == NO_EXIT) {
            res = nd.bexit[d];
            break;
        }
        x = (d / NODE_STRIDE) * NODE_STRIDE + ex;
    }
    exit_map[k] = res;
}
}  // namespace

namespace {
// ---- row bands, second pass as a DELTA over the first (round 4) -------------------------------------------------------------------
// After the boundary pass a band holds the accumulation of its OWN cells (halo cells as sources of nothing); the neighbours' final
// values X[h] of the halo cells then arrive, and  final[c] = own[c] + sum of X[h] over the halo cells h whose flow path runs through c.
// The band used to accumulate a second time from scratch with the halo cells as sources (64-bit sums, both tile passes and the graph
// walk again: 52 of 175 ms per step at 4 bands of 32768^2).  The perimeter graph of the first pass already says where every halo
// cell's path goes: (A) one thread per halo cell follows its path from tile to tile and adds X[h] to the entry cells it passes
// (the `inflow` words, cleared first) -- a few steps per halo cell, rivers merge but are walked separately --, marking the tiles it
// touches; (B) a workgroup per TOUCHED tile lets every such entry (and every halo cell of the tile that flows into it) walk its
// in-tile path and add its share, on the tile's own sums in LDS, and writes the tile back.  Tiles no halo flux reaches -- most of a
// band -- are not read at all.  An unknown halo value (<= 0: a flow cycle upstream in the neighbouring band) or a path that does
// not end (a cycle across tiles) raises `flag`: the caller then runs the full second pass.
__global__ __launch_bounds__(256) void accum_delta_graph_kernel(Nodes nd, const uint8_t *__restrict__ fd, const double *__restrict__ out, int64_t H, int64_t W,
                                                               int ntc, int64_t ntiles, int fixed_top, int fixed_bot, uint8_t *touched, unsigned int *flag)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * W) return;
    const int side = k < W ? 0 : 1;
    if (side == 0 ? !fixed_top : !fixed_bot) return;
    const int64_t col = k - (int64_t)side * W, row = side == 0 ? 0 : H - 1;
    auto halo_row = [&](int64_t rr) { return (fixed_top && rr == 0) || (fixed_bot && rr == H - 1); };
    const unsigned code = fd[row * W + col];
    if (code > 7u || dir_dr((int)code) == 0) return;
    const int64_t tr = row + dir_dr((int)code), tc = col + dir_dc((int)code);
    if (tr < 0 || tr >= H || tc < 0 || tc >= W || halo_row(tr)) return;      // the cell does not flow into the band
    const double X = out[row * W + col];
    if (!(X > 0.0)) {                 // not known (a flow cycle upstream, in the neighbouring band): everything below it stays unresolved
        *flag = 1u;
        return;
    }
    const unsigned long long D = (unsigned long long)X;
    touched[(row / AT) * ntc + col / AT] = 1;
    int32_t x = nd.halo_first[k];
    int64_t step = 0;
    const int64_t cap = ntiles * PERIM + 4;                     // a simple path visits an exit node once: longer = a flow cycle across tiles
    for (; x >= 0 && step < cap; ++step) {                      // x: the exit node of the tile the path is in
        const int32_t e = nd.dst[x];
        int er, ec;
        perim_cell(e % NODE_STRIDE, er, ec);
        const int64_t egr = (int64_t)((e / NODE_STRIDE) / ntc) * AT + er;
        if (halo_row(egr)) break;                               // the path leaves the band: its flux is the neighbour's business
        atomicAdd(reinterpret_cast<unsigned long long *>(&nd.inflow[e]), D);
        touched[e / NODE_STRIDE] = 1;
        const uint16_t ex = nd.exit_of[e];
        if (ex == NO_EXIT) break;
        x = (e / NODE_STRIDE) * NODE_STRIDE + ex;
    }
    if (x >= 0 && step >= cap) *flag = 1u;
}

__global__ __launch_bounds__(ATN) void accum_delta_tile_kernel(const uint8_t *__restrict__ fd, double *__restrict__ out, int64_t H, int64_t W, int ntc, Nodes nd,
                                                              int fixed_top, int fixed_bot, const uint8_t *__restrict__ touched)
{
    const int tile = blockIdx.x;
    if (!touched[tile]) return;                                  // (block-uniform)
    __shared__ unsigned long long S[AT * AT];                    // own sum (0: unresolved -- stays 0 and ends every walk) + the shares added
    __shared__ uint16_t P[AT * AT];
    __shared__ uint32_t wl_cell[PERIM + AT * 2];
    __shared__ unsigned long long wl_add[PERIM + AT * 2];
    __shared__ uint32_t wl_n;
    const int ti = tile / ntc, tj = tile - ti * ntc;
    const int64_t r0 = (int64_t)ti * AT, c0 = (int64_t)tj * AT;
    const int tid = threadIdx.x;
    auto halo_row = [&](int64_t rr) { return (fixed_top && rr == 0) || (fixed_bot && rr == H - 1); };
    if (tid == 0) wl_n = 0;
    for (int i = tid; i < AT * AT; i += ATN) {
        const int r = i / AT, c = i - r * AT;
        const int64_t rr = r0 + r, cc = c0 + c;
        const bool inside = rr < H && cc < W;
        const bool owned = inside && !halo_row(rr);
        const double own = owned ? out[rr * W + cc] : 0.0;
        S[i] = own > 0.0 ? (unsigned long long)own : 0ull;
        uint16_t nx = (uint16_t)SENT13;
        const unsigned code = inside ? fd[rr * W + cc] : 8u;
        if (owned && code <= 7u) {
            const int nr = r + dir_dr((int)code), nc = c + dir_dc((int)code);
            if (nr >= 0 && nr < AT && nc >= 0 && nc < AT && (r0 + nr) < H && (c0 + nc) < W && !halo_row(r0 + nr)) nx = (uint16_t)(nr * AT + nc);
        }
        P[i] = nx;
    }
    __syncthreads();
    // the walkers: the entry cells the graph pass left a share at, and the halo cells of this tile that flow into one of its cells
    if (tid < PERIM) {
        const unsigned long long d = nd.inflow[(int64_t)tile * NODE_STRIDE + tid];
        int r, c;
        perim_cell(tid, r, c);
        if (d != 0ull && (r0 + r) < H && (c0 + c) < W && !halo_row(r0 + r)) {
            const uint32_t kk = atomicAdd(&wl_n, 1u);
            wl_cell[kk] = (uint32_t)(r * AT + c);
            wl_add[kk] = d;
        }
    }
    for (int side = 0; side < 2; ++side) {
        const int64_t hr = side == 0 ? 0 : H - 1;
        if ((side == 0 ? !fixed_top : !fixed_bot) || hr < r0 || hr >= r0 + AT) continue;
        if (tid < AT && c0 + tid < W) {
            const int r = (int)(hr - r0), c = tid;
            const unsigned code = fd[hr * W + c0 + c];
            const double X = out[hr * W + c0 + c];
            if (code <= 7u && dir_dr((int)code) != 0 && X > 0.0) {
                const int nr = r + dir_dr((int)code), nc = c + dir_dc((int)code);
                if (nr >= 0 && nr < AT && nc >= 0 && nc < AT && (r0 + nr) < H && (c0 + nc) < W && !halo_row(r0 + nr)) {
                    const uint32_t kk = atomicAdd(&wl_n, 1u);
                    wl_cell[kk] = (uint32_t)(nr * AT + nc);
                    wl_add[kk] = (unsigned long long)X;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nw = wl_n;
    for (uint32_t w = tid; w < nw; w += ATN) {
        uint32_t a = wl_cell[w];
        const unsigned long long v = wl_add[w];
        do {
            if (S[a] == 0ull) break;                  // unresolved in the first pass: stays 0, and so does everything below it
            atomicAdd(&S[a], v);
            a = P[a];
        } while (a != SENT13);
    }
    __syncthreads();
    for (int i = tid; i < AT * AT; i += ATN) {
        const int r = i / AT, c = i - r * AT;
        const int64_t rr = r0 + r, cc = c0 + c;
        if (rr < H && cc < W && !halo_row(rr)) out[rr * W + cc] = (double)S[i];
    }
}
}  // namespace

// the second pass of a row band as a delta over the kept perimeter graph of the boundary pass; *done == false: not applicable
// (no kept graph, an unknown halo value, a flow cycle across tiles) -- the caller runs the full pass
int accum_band_delta_dev(const uint8_t *d_fd, double *d_out, int64_t H, int64_t W, hipStream_t s, int fixed_top, int fixed_bot, AccumKeep *keep, bool *done)
{
    *done = false;
    if (!keep || !keep->valid || keep->H != H || keep->W != W || keep->fixed_top != fixed_top || keep->fixed_bot != fixed_bot || !keep->nodes.p) return MHIP_OK;
    static const bool off = [] { const char *e = dev_env("MHIP_BAND_ACC"); return e && std::string(e) == "full"; }();      // (development: A/B)
    if (off) return MHIP_OK;
    const int ntr = (int)cdiv(H, AT), ntc = (int)cdiv(W, AT);
    const int64_t ntiles = (int64_t)ntr * ntc, nnodes = ntiles * NODE_STRIDE;
    char *b = keep->nodes.as<char>();
    Nodes nd;
    nd.gstate = reinterpret_cast<uint64_t *>(b + keep->o[0]);
    nd.inflow = reinterpret_cast<uint64_t *>(b + keep->o[1]);
    nd.arrived = reinterpret_cast<uint32_t *>(b + keep->o[2]);
    nd.next = reinterpret_cast<int32_t *>(b + keep->o[3]);
    nd.dst = reinterpret_cast<int32_t *>(b + keep->o[4]);
    nd.exit_of = reinterpret_cast<uint16_t *>(b + keep->o[5]);
    nd.flags = reinterpret_cast<uint8_t *>(b + keep->o[6]);
    nd.halo_first = reinterpret_cast<int32_t *>(b + keep->o[7]);
    nd.bexit = reinterpret_cast<int32_t *>(b + keep->o[8]);
    DevBuf aux;
    MH_TRY(aux.alloc((size_t)ntiles + 64));
    uint8_t *touched = aux.as<uint8_t>() + 64;
    unsigned int *flag = aux.as<unsigned int>();
    MH_HIP(hipMemsetAsync(aux.p, 0, (size_t)ntiles + 64, s));
    MH_HIP(hipMemsetAsync(nd.inflow, 0, 8 * (size_t)nnodes, s));
    hipLaunchKernelGGL(accum_delta_graph_kernel, dim3((unsigned)cdiv(2 * W, 256)), dim3(256), 0, s, nd, d_fd, (const double *)d_out, H, W, ntc, ntiles, fixed_top,
                       fixed_bot, touched, flag);
    MH_HIP(hipGetLastError());
    unsigned int h_flag = 0;
    MH_HIP(hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    keep->valid = false;                  // (one use: the halo values it was fed are this exchange's)
    if (h_flag) return MHIP_OK;           // nothing has been written yet: the full pass takes over
    hipLaunchKernelGGL(accum_delta_tile_kernel, dim3((unsigned)ntiles), dim3(ATN), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot, (const uint8_t *)touched);
    MH_HIP(hipGetLastError());
    MH_HIP(stream_sync(s));               // (aux goes back to the pool)
    *done = true;
    return MHIP_OK;
}

int accum_dev(const uint8_t *d_fd, double *d_out, int64_t H, int64_t W, hipStream_t s, int fixed_top, int fixed_bot, int halo_zero,
              int32_t *d_exit_map, PourLink *pour, AccumKeep *keep)
{
    const int ntr = (int)cdiv(H, AT), ntc = (int)cdiv(W, AT);
    const int64_t ntiles = (int64_t)ntr * ntc, nnodes = ntiles * NODE_STRIDE;
    if (nnodes >= (int64_t)INT32_MAX || H * W >= (1ll << 37)) {   // + the 38-bit sum field of the tile words
        set_error("accumulated_flow: raster too large for the int32 perimeter-node domain");
        return MHIP_ELIMIT;
    }
    DevBuf buf;
    auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t o_gstate = 0, o_inflow = align(o_gstate + 8 * (size_t)nnodes), o_arrived = align(o_inflow + 8 * (size_t)nnodes);
    const size_t o_next = align(o_arrived + 4 * (size_t)nnodes), o_dst = align(o_next + 4 * (size_t)nnodes);
    const size_t o_exit = align(o_dst + 4 * (size_t)nnodes), o_flags = align(o_exit + 2 * (size_t)nnodes);
    const size_t o_halo = align(o_flags + (size_t)nnodes);
    const size_t o_bexit = align(o_halo + (d_exit_map ? 8 * (size_t)W : 0));
    MH_TRY(buf.alloc(o_bexit + (d_exit_map ? 4 * (size_t)nnodes : 0)));
    char *b = buf.as<char>();
    Nodes nd;
    nd.gstate = reinterpret_cast<uint64_t *>(b + o_gstate);
    nd.inflow = reinterpret_cast<uint64_t *>(b + o_inflow);
    nd.arrived = reinterpret_cast<uint32_t *>(b + o_arrived);
    nd.next = reinterpret_cast<int32_t *>(b + o_next);
    nd.dst = reinterpret_cast<int32_t *>(b + o_dst);
    nd.exit_of = reinterpret_cast<uint16_t *>(b + o_exit);
    nd.flags = reinterpret_cast<uint8_t *>(b + o_flags);
    nd.halo_first = d_exit_map ? reinterpret_cast<int32_t *>(b + o_halo) : nullptr;
    nd.bexit = d_exit_map ? reinterpret_cast<int32_t *>(b + o_bexit) : nullptr;
    if (d_exit_map) MH_HIP(hipMemsetAsync(nd.halo_first, 0xff, 8 * (size_t)W, s));
    MH_HIP(hipMemsetAsync(nd.inflow, 0, o_next - o_inflow, s));      // inflow | arrived: phase 1 pushes the external in-degrees into `arrived`
    const unsigned gn = (unsigned)cdiv(nnodes, 256);
    // the local sums of phase 1 fit 32 bits unless halo cells bring the neighbouring band's flux in (the final pass of a row band)
    // the final pass as walks from the entry cells (accum_final_walk_kernel) wherever the sums fit 32 bits and no halo row is a source
    // (a row band: its boundary pass -- halo cells are sources of no flux, the sums are the band's own cells)
    const bool walk_final = (!(fixed_top || fixed_bot) || halo_zero) && (!d_exit_map || halo_zero) && H * W < (int64_t)0x7fffffff;
    if (walk_final)
        hipLaunchKernelGGL((accum_tile_kernel<false, false, true>), dim3((unsigned)ntiles), dim3(ATN), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot, halo_zero);
    else if ((fixed_top || fixed_bot) && !halo_zero)
        hipLaunchKernelGGL((accum_tile_kernel<false, true>), dim3((unsigned)ntiles), dim3(ATN), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot, halo_zero);
    else
        hipLaunchKernelGGL((accum_tile_kernel<false, false>), dim3((unsigned)ntiles), dim3(ATN), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot, halo_zero);
    hipLaunchKernelGGL(accum_link_kernel, dim3(gn), dim3(256), 0, s, nd, nnodes);
    if (d_exit_map)
        hipLaunchKernelGGL(accum_band_exit_kernel, dim3((unsigned)cdiv(2 * W, 256)), dim3(256), 0, s, nd, H, W, ntc, ntiles, fixed_top, fixed_bot, d_exit_map);
    hipLaunchKernelGGL(accum_mark_kernel, dim3(gn), dim3(256), 0, s, nd, nnodes);
    hipLaunchKernelGGL(accum_graph_walk_kernel, dim3(gn), dim3(256), 0, s, nd, nnodes);
    // final values are at most H * W without halo sources: 32-bit sums (half the LDS, cheaper atomics) below 2**31 cells
    if (walk_final) {
        PourCandDev pc;      // (no keys: the pour points stay a pass of their own)
        if (pour && pour->wait && pour->wait(pour->arg) == 1 && pour->dev.key) {
            MH_HIP(hipStreamWaitEvent(s, pour->ev, 0));
            pc = pour->dev;
            pour->consumed = true;
        }
        hipLaunchKernelGGL(accum_final_walk_kernel, dim3((unsigned)ntiles), dim3(ATN), 0, s, d_fd, d_out, H, W, ntc, nd, pc, fixed_top, fixed_bot);
    }
    else if ((!(fixed_top || fixed_bot) || halo_zero) && H * W < (int64_t)0x7fffffff)      // (halo_zero: a band's OWN cells only -- no sum passes their number)
        hipLaunchKernelGGL((accum_tile_kernel<true, false>), dim3((unsigned)ntiles), dim3(ATN), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot, halo_zero);
    else
        hipLaunchKernelGGL((accum_tile_kernel<true, true>), dim3((unsigned)ntiles), dim3(ATN), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot, halo_zero);
    MH_HIP(hipGetLastError());
    MH_HIP(stream_sync(s));  // the node buffer goes back to the pool -- or, after the boundary pass of a row band, stays for the delta pass
    if (keep) {
        keep->valid = false;
        if (d_exit_map) {
            keep->nodes.release();
            keep->nodes.p = buf.p; keep->nodes.bytes = buf.bytes;
            buf.p = nullptr; buf.bytes = 0;
            const size_t offs[9] = {o_gstate, o_inflow, o_arrived, o_next, o_dst, o_exit, o_flags, o_halo, o_bexit};
            for (int k = 0; k < 9; ++k) keep->o[k] = offs[k];
            keep->H = H; keep->W = W; keep->fixed_top = fixed_top; keep->fixed_bot = fixed_bot;
            keep->valid = true;
        }
    }
#ifdef MH_PROFILE_ACCUM
    {
        unsigned long long pr[2][8];
        MH_HIP(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_accum_prof), sizeof(pr)));
        for (int f = 0; f < 2; ++f)
            fprintf(stderr, "[accum profile phase %d] ticks/tile (thread 0): load=%.0f init=%.0f solve=%.0f publish/write=%.0f\n", f ? 3 : 1,
                    pr[f][0] / (double)ntiles, pr[f][1] / (double)ntiles, pr[f][2] / (double)ntiles, pr[f][3] / (double)ntiles);
        unsigned long long z[2][8] = {};
        MH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_accum_prof), z, sizeof(z)));
    }
#endif
    return MHIP_OK;
}

}  // namespace mh
