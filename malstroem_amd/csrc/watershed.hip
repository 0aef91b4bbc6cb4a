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

__global__ __launch_bounds__(256) void ws_count_interior_nodir(const uint8_t *__restrict__ fd, int64_t H, int64_t W,
                                                              unsigned int *count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (i < H * W) {
        const int64_t r = i / W, c = i - r * W;
        hit = fd[i] > 7u && r > 0 && r < H - 1 && c > 0 && c < W - 1;
    }
    if (__any(hit)) {
        const unsigned n = (unsigned)__popcll(__ballot(hit));
        if ((threadIdx.x & 63) == 0) atomicAdd(count, n);
    }
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
__global__ void ws_neg_lut_kernel(int32_t *lab, int64_t n, const int32_t *__restrict__ lut, int64_t nlut)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = lab[i];
    if (v < 0 && (int64_t)(-(int64_t)v - 1) < nlut) lab[i] = lut[-(int64_t)v - 1];
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
    hipLaunchKernelGGL(ws_neg_lut_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_lab, n, d_lut, nlut);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

int watersheds_dev(const uint8_t *d_fd, int32_t *d_labels, int64_t H, int64_t W, int32_t unassigned, hipStream_t s, bool band_mode)
{
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
    hipLaunchKernelGGL(ws_count_interior_nodir, dim3(grid), dim3(256), 0, s, d_fd, H, W, d_cnt);
    unsigned int interior_nodir = 0;
    MH_HIP(hipMemcpyAsync(&interior_nodir, d_cnt, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    uint32_t *q = nullptr;
    if (interior_nodir && band_mode) {
        set_error("watersheds on a row band need every flow path to leave the raster (interior NODIR cell found)");
        return MHIP_EINVAL;
    }
    if (interior_nodir) {
        MH_TRY(Q.alloc(4 * (size_t)n));
        q = Q.as<uint32_t>();
    }
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
        MH_HIP(hipStreamSynchronize(s));
        round += k;
        if (!h[k - 1]) break;
    }
    hipLaunchKernelGGL(ws_assign_kernel, dim3(grid), dim3(256), 0, s, P.as<int32_t>(), q, d_labels, n, unassigned);
    MH_HIP(hipGetLastError());
    MH_HIP(hipStreamSynchronize(s));
    return MHIP_OK;
}

}  // namespace mh
