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
//   phase 1  per 64x64 tile, in LDS: Kahn walk restricted to the tile (external inflow ignored) gives every cell's
//            tile-local sum; for each perimeter cell we publish {local sum, resolved?, leaves-the-tile?} and for each
//            ENTRY cell (has an upstream neighbour outside the tile) the perimeter cell where its local path exits.
//   phase 2  global, on perimeter cells only (~1/16 of the raster): exit cell x forwards F(x) = local(x) + all flux
//            routed through x to the entry cell it flows into, and on to that entry's exit: the same packed
//            64-bit "pending count | sum" walk as before, but over ~4 % of the cells.
//   phase 3  per tile, in LDS again: the Kahn walk with every entry cell pre-loaded with its external inflow
//            produces the final values, written once as float64.
// One LDS state word per cell: bit 63 = source (pending 0 at start), bits 59..62 = the cell's own flow code (so a walker
// learns where to go next from the value its atomic returns: one LDS round trip per step), bits 55..58 = pending
// arrivals, bits 0..54 = running sum.
#include "common.hpp"

namespace mh {
namespace {

constexpr int AT = 64;                // tile edge
constexpr int PERIM = 4 * AT - 4;     // perimeter cells of a tile
constexpr int NODE_STRIDE = 256;      // perimeter slots per tile in the global node arrays
constexpr int FS = AT + 4;            // LDS row stride of the flow-direction window (66 used)
constexpr uint64_t SRC = 1ull << 63;
constexpr int DEG_SHIFT = 55;
constexpr int CODE_SHIFT = 59;
constexpr uint64_t ONE_PENDING = 1ull << DEG_SHIFT;
constexpr uint64_t SUM_MASK = ONE_PENDING - 1;
constexpr uint16_t NO_EXIT = 0xffffu;
// phase-2 node word: an exit cell can be fed by every entry of its tile (hundreds), so the pending field is wider
constexpr int G_SHIFT = 44;
constexpr uint64_t G_ONE = 1ull << G_SHIFT, G_SUM = G_ONE - 1, G_PEND = 0x7ffffull;

enum : uint8_t { F_EXIT = 1, F_RESOLVED = 2, F_ENTRY = 4 };

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
    uint32_t *arrived;   // per ENTRY cell: number of deliveries
    int32_t *next;       // per EXIT cell: node index of the exit its flux continues to (-1: none)
    int32_t *dst;        // per EXIT cell: node index of the entry cell it flows into
    uint16_t *exit_of;   // per ENTRY cell: slot of the exit of its tile-local path (NO_EXIT: ends inside / leaves raster)
    uint8_t *flags;
};

// ---- the tile kernel (phase 1 when FINAL == false, phase 3 when FINAL == true) ------------------------------
template <bool FINAL>
// Row-band mode: local row 0 / H-1 may be a HALO row owned by the neighbouring band.  Its cells carry the neighbour's
// final value in `out` (> 0: known, acts as a source of that much flux; <= 0: not known yet, blocks everything below it);
// they never receive and are never written here.
__global__ __launch_bounds__(256) void accum_tile_kernel(const uint8_t *__restrict__ fd, double *__restrict__ out, int64_t H,
                                                        int64_t W, int ntc, Nodes nd, int fixed_top, int fixed_bot)
{
    auto halo_row = [&](int64_t rr) { return (fixed_top && rr == 0) || (fixed_bot && rr == H - 1); };
    __shared__ uint64_t st[AT * AT];
    __shared__ uint8_t win[(AT + 2) * FS];
    const int tile = blockIdx.x;
    const int ti = tile / ntc, tj = tile - ti * ntc;
    const int64_t r0 = (int64_t)ti * AT, c0 = (int64_t)tj * AT;
    const int tid = threadIdx.x;

    // flow-direction window incl. the 1-cell ring; outside the raster = NODIR (never flows, never receives)
    for (int i = tid; i < (AT + 2) * (AT + 2); i += 256) {
        const int wr = i / (AT + 2), wc = i - wr * (AT + 2);
        const int64_t rr = r0 + wr - 1, cc = c0 + wc - 1;
        win[wr * FS + wc] = (rr >= 0 && rr < H && cc >= 0 && cc < W) ? fd[rr * W + cc] : (uint8_t)8;
    }
    __syncthreads();

    // initial state of my 16 cells
    for (int i = tid; i < AT * AT; i += 256) {
        const int r = i / AT, c = i - r * AT;
        const bool inside = (r0 + r) < H && (c0 + c) < W;
        unsigned deg_in = 0, deg_ext = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int nr = r + dir_dr(k), nc = c + dir_dc(k);
            const bool f = flows_into(win[(nr + 1) * FS + nc + 1], k);
            const bool in_tile = nr >= 0 && nr < AT && nc >= 0 && nc < AT;
            deg_in += (f && in_tile) ? 1u : 0u;
            deg_ext += (f && !in_tile) ? 1u : 0u;
        }
        uint64_t sum = 1, pending = deg_in;
        if (!inside) pending = 15;  // not a raster cell: never fires
        const bool halo = inside && halo_row(r0 + r);
        if (halo) {
            const double ext = out[(r0 + r) * W + c0 + c];
            pending = ext > 0.0 ? 0 : 15;
            sum = ext > 0.0 ? (uint64_t)ext : 0;
            deg_ext = 0;
        }
        if (FINAL && deg_ext && inside) {
            const int64_t node = (int64_t)tile * NODE_STRIDE + perim_slot(r, c);
            if (nd.arrived[node] == deg_ext) sum += nd.inflow[node];
            else pending += 1;  // some upstream flux never arrives (flow cycle upstream): stays unresolved => 0
        }
        st[i] = ((uint64_t)win[(r + 1) * FS + c + 1] << CODE_SHIFT) | (pending << DEG_SHIFT) | sum | (pending == 0 ? SRC : 0ull);
    }
    __syncthreads();

    // Kahn walk inside the tile: the last arriver at a cell owns its complete sum and carries on.  Every lane is a small
    // state machine (walking / looking for its next source cell), so a lane that finishes a short path picks up new work
    // immediately.  (Measured: all four wavefronts walking beats a single walking wavefront 2:1 -- the phase is bound by
    // the number of steps, not by the longest chain.)
    {
        int next_i = tid;
        bool walking = false;
        int r = 0, c = 0;
        unsigned code = 8;
        uint64_t total = 0;
        while (__any(walking || next_i < AT * AT)) {
            if (walking) {
                // branch-free step: (dr+1, dc+1) of the 8 codes packed 2 bits each (NODIR decodes to garbage, masked by `go`)
                constexpr unsigned DRP = (0u << 0) | (0u << 2) | (1u << 4) | (2u << 6) | (2u << 8) | (2u << 10) | (1u << 12) | (0u << 14);
                constexpr unsigned DCP = (1u << 0) | (2u << 2) | (2u << 4) | (2u << 6) | (1u << 8) | (0u << 10) | (0u << 12) | (0u << 14);
                const unsigned sh2 = (code & 7u) * 2u;
                r += (int)((DRP >> sh2) & 3u) - 1;
                c += (int)((DCP >> sh2) & 3u) - 1;
                // stays inside the tile, the raster and the band?
                const bool go = (code < 8u) & ((unsigned)r < (unsigned)AT) & ((unsigned)c < (unsigned)AT) & !halo_row(r0 + r);
                bool cont = false;
                if (go) {
                    const uint64_t delta = total - ONE_PENDING;
                    const uint64_t now = atomicAdd(reinterpret_cast<unsigned long long *>(&st[r * AT + c]), (unsigned long long)delta) + delta;
                    cont = ((now >> DEG_SHIFT) & 0xf) == 0;  // else somebody else still has to arrive
                    total = cont ? (now & SUM_MASK) : total;
                    code = cont ? ((unsigned)(now >> CODE_SHIFT) & 0xfu) : code;
                }
                walking = cont;
            } else if (next_i < AT * AT) {
                const uint64_t s0 = st[next_i];
                if (s0 & SRC) {  // SRC is only written by the initialisation above
                    r = next_i / AT;
                    c = next_i - r * AT;
                    total = s0 & SUM_MASK;
                    code = (unsigned)(s0 >> CODE_SHIFT) & 0xfu;
                    walking = true;
                }
                next_i += 256;
            }
        }
    }
    __syncthreads();

    if (FINAL) {
        for (int i = tid; i < AT * AT; i += 256) {
            const int r = i / AT, c = i - r * AT;
            if ((r0 + r) < H && (c0 + c) < W && !halo_row(r0 + r)) {
                const uint64_t s = st[i];
                out[(r0 + r) * W + c0 + c] = ((s >> DEG_SHIFT) & 0xf) ? 0.0 : (double)(s & SUM_MASK);
            }
        }
        return;
    }

    // phase 1: publish the perimeter
    if (tid < PERIM) {
        int r, c;
        perim_cell(tid, r, c);
        const int64_t node = (int64_t)tile * NODE_STRIDE + tid;
        const bool inside = (r0 + r) < H && (c0 + c) < W;
        const uint64_t s = st[r * AT + c];
        const bool resolved = inside && ((s >> DEG_SHIFT) & 0xf) == 0;
        const unsigned code = win[(r + 1) * FS + c + 1];
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
        // entry cell?  follow its tile-local path to the cell where it leaves the tile
        bool entry = false;
        if (inside && !halo_row(r0 + r)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int nr = r + dir_dr(k), nc = c + dir_dc(k);
                if ((nr < 0 || nr >= AT || nc < 0 || nc >= AT) && flows_into(win[(nr + 1) * FS + nc + 1], k)) entry = true;
            }
        }
        uint16_t ex = NO_EXIT;
        if (entry) {
            fl |= F_ENTRY;
            int pr = r, pc = c;
            for (int step = 0; step < AT * AT; ++step) {
                const unsigned cd = win[(pr + 1) * FS + pc + 1];
                if (cd > 7u) break;  // sink inside the tile
                const int nr = pr + dir_dr((int)cd), nc = pc + dir_dc((int)cd);
                if (nr < 0 || nr >= AT || nc < 0 || nc >= AT) {
                    const int64_t gr = r0 + nr, gc = c0 + nc;
                    if (gr >= 0 && gr < H && gc >= 0 && gc < W) ex = (uint16_t)perim_slot(pr, pc);  // else: leaves the raster
                    break;
                }
                if (halo_row(r0 + nr)) break;  // continues in the neighbouring band
                pr = nr;
                pc = nc;
            }
        }
        nd.flags[node] = fl;
        nd.dst[node] = dst;
        nd.exit_of[node] = ex;
        nd.gstate[node] = ((resolved ? 0ull : 1ull) << G_SHIFT) | (s & SUM_MASK);  // unresolved: blocks itself forever
        nd.inflow[node] = 0;
        nd.arrived[node] = 0;
        nd.next[node] = -1;
    }
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
    for (;;) {
        const int32_t e = nd.dst[x];
        atomicAdd(reinterpret_cast<unsigned long long *>(&nd.inflow[e]), (unsigned long long)total);
        atomicAdd(&nd.arrived[e], 1u);
        const int32_t nx = nd.next[x];
        if (nx < 0) break;
        const uint64_t delta = total - G_ONE;
        const uint64_t now = atomicAdd(reinterpret_cast<unsigned long long *>(&nd.gstate[nx]), (unsigned long long)delta) + delta;
        if ((now >> G_SHIFT) & G_PEND) break;
        total = now & G_SUM;
        x = nx;
    }
}

}  // namespace

int accum_dev(const uint8_t *d_fd, double *d_out, int64_t H, int64_t W, hipStream_t s, int fixed_top, int fixed_bot)
{
    const int ntr = (int)cdiv(H, AT), ntc = (int)cdiv(W, AT);
    const int64_t ntiles = (int64_t)ntr * ntc, nnodes = ntiles * NODE_STRIDE;
    if (nnodes >= (int64_t)INT32_MAX) {
        set_error("accumulated_flow: raster too large for the int32 perimeter-node domain");
        return MHIP_ELIMIT;
    }
    DevBuf buf;
    auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t o_gstate = 0, o_inflow = align(o_gstate + 8 * (size_t)nnodes), o_arrived = align(o_inflow + 8 * (size_t)nnodes);
    const size_t o_next = align(o_arrived + 4 * (size_t)nnodes), o_dst = align(o_next + 4 * (size_t)nnodes);
    const size_t o_exit = align(o_dst + 4 * (size_t)nnodes), o_flags = align(o_exit + 2 * (size_t)nnodes);
    MH_TRY(buf.alloc(o_flags + (size_t)nnodes));
    char *b = buf.as<char>();
    Nodes nd;
    nd.gstate = reinterpret_cast<uint64_t *>(b + o_gstate);
    nd.inflow = reinterpret_cast<uint64_t *>(b + o_inflow);
    nd.arrived = reinterpret_cast<uint32_t *>(b + o_arrived);
    nd.next = reinterpret_cast<int32_t *>(b + o_next);
    nd.dst = reinterpret_cast<int32_t *>(b + o_dst);
    nd.exit_of = reinterpret_cast<uint16_t *>(b + o_exit);
    nd.flags = reinterpret_cast<uint8_t *>(b + o_flags);
    const unsigned gn = (unsigned)cdiv(nnodes, 256);
    hipLaunchKernelGGL(accum_tile_kernel<false>, dim3((unsigned)ntiles), dim3(256), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot);
    hipLaunchKernelGGL(accum_link_kernel, dim3(gn), dim3(256), 0, s, nd, nnodes);
    hipLaunchKernelGGL(accum_mark_kernel, dim3(gn), dim3(256), 0, s, nd, nnodes);
    hipLaunchKernelGGL(accum_graph_walk_kernel, dim3(gn), dim3(256), 0, s, nd, nnodes);
    hipLaunchKernelGGL(accum_tile_kernel<true>, dim3((unsigned)ntiles), dim3(256), 0, s, d_fd, d_out, H, W, ntc, nd, fixed_top, fixed_bot);
    MH_HIP(hipGetLastError());
    MH_HIP(hipStreamSynchronize(s));  // the node buffer goes back to the pool
    return MHIP_OK;
}

}  // namespace mh
