// ccl.hip -- 8-connected component labelling with scipy.ndimage.label numbering (gfx950).
//
// Reference: label.connected_components (label.py:19-40) -> scipy.ndimage.label(data, structure=ones((3,3))):
// foreground = data != 0 (NaN and negatives are foreground), background 0, int32 labels 1..n numbered in the
// order of each component's FIRST pixel in raster scan.
//
// Device schedule: union-find over linear cell indices where the representative of a set is always its
// minimum index (hooking larger root under smaller with atomicMin), so after flattening the roots are exactly
// the components' first raster pixels; an exclusive prefix sum over the root flags gives the scipy rank.
//   1. init   : parent[i] = first cell of i's horizontal run of foreground cells inside its 64-cell wavefront chunk
//               (ballot + count-leading-zeros; runs never cross a raster row), -1 for background: every run is a
//               finished set before the first atomic
//   2. merge  : ONE union per pair of vertically adjacent runs: a cell unites with N unless its left neighbour has
//               already done so (W and NW foreground), with NW only when N and W are background, with NE when N is
//               background; the first lane of a chunk also unites with W (runs cut by the chunk border)
//   3. flatten: parent[i] = root(i)
//   4. rank   : per-block root counts -> single-block scan -> roots get -(rank+1)
//   5. emit   : labels[i] = rank of root(i), 0 for background
#include <cstdlib>
#include <string>
#include <type_traits>
#include <utility>

#include "common.hpp"

namespace mh {
namespace {

constexpr int SCAN_BLOCK = 1024;        // threads
constexpr int SCAN_ITEMS = 4;           // cells per thread in the rank pass
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

template <typename T> __device__ __forceinline__ bool is_fg(T v) { return v != (T)0; }

template <typename T>
__global__ __launch_bounds__(256) void ccl_init_kernel(const T *__restrict__ data, int32_t *__restrict__ parent, int64_t n, int64_t W)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool fg = i < n && is_fg(data[i]);
    const bool fg_left = __shfl_up((int)fg, 1) != 0;
    const bool start = fg && (lane == 0 || !fg_left || (i % W) == 0);
    const uint64_t brk = __ballot(start);
    if (i >= n) return;
    int32_t p = -1;
    if (fg) {
        const uint64_t below = brk & ((2ull << lane) - 1ull);   // run starts at or before my lane (mine included)
        p = (int32_t)(i - (lane - (63 - __builtin_clzll(below))));
    }
    parent[i] = p;
}

__device__ __forceinline__ int32_t find_root(const int32_t *parent, int32_t x)
{
    int32_t p = parent[x];
    while (p != x) {
        x = p;
        p = parent[x];
    }
    return x;
}

__device__ __forceinline__ void unite(int32_t *parent, int32_t a, int32_t b)
{
    for (;;) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) return;
        if (a < b) {
            const int32_t t = a;
            a = b;
            b = t;
        }
        const int32_t old = atomicMin(&parent[a], b);  // hook the larger root under the smaller one
        if (old == a) return;
        a = old;  // somebody re-parented a meanwhile: continue from its new (smaller) parent
    }
}

__global__ __launch_bounds__(256) void ccl_merge_kernel(int32_t *parent, int64_t H, int64_t W)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    if (parent[i] < 0) return;
    const int64_t r = i / W, c = i - r * W;
    const bool hasW = c > 0 && parent[i - 1] >= 0;
    if (hasW && (threadIdx.x & 63) == 0) unite(parent, (int32_t)i, (int32_t)(i - 1));   // my run continues the previous chunk's
    if (r == 0) return;
    const bool hasN = parent[i - W] >= 0;
    const bool hasNW = c > 0 && parent[i - W - 1] >= 0;
    if (hasN) {
        // N's run touches mine; the left neighbour has made this union already when W and NW are foreground too
        if (!(hasW && hasNW)) unite(parent, (int32_t)i, (int32_t)(i - W));
        return;
    }
    const bool hasNE = c + 1 < W && parent[i - W + 1] >= 0;
    if (hasNW && !hasW) unite(parent, (int32_t)i, (int32_t)(i - W - 1));   // with W foreground, W has N == my NW
    if (hasNE) unite(parent, (int32_t)i, (int32_t)(i - W + 1));            // a run that starts above my right shoulder
}

// ---- tile-local labelling in LDS ------------------------------------------------------------------------------------------
// One workgroup = one 64 x 64 tile; wave w owns the tile rows w, w + 4, ...  The same three steps as above (runs, one union per
// pair of vertically adjacent runs, flatten) on a 16 KB table of tile-local indices, then parent[cell] = GLOBAL index of the
// cell's tile-local root (the first cell of its piece in raster order, tile-local = global order inside a tile).  What is left
// for global memory are the unions across the tile seams (ccl_seam_kernel): ~6 % of the cells.
constexpr int CT = 64;
constexpr int MAXROOTS = CT * CT / 4;   // pieces of a tile: at most every other cell in both directions (8-connectivity)
constexpr uint32_t LBG = 0xffffffffu;

__device__ __forceinline__ uint32_t find_root_l(const uint32_t *par, uint32_t x)
{
    uint32_t p = par[x];
    while (p != x) {
        x = p;
        p = par[x];
    }
    return x;
}
__device__ __forceinline__ void unite_l(uint32_t *par, uint32_t a, uint32_t b)
{
    for (;;) {
        a = find_root_l(par, a);
        b = find_root_l(par, b);
        if (a == b) return;
        if (a < b) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        const uint32_t old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}

// (the union and the flatten phase four rows at a time, not sixteen: 19 instead of 94 VGPRs -- eight workgroups per CU instead of five;
// all of the kernel's phases wait on LDS round trips, residency is what pays)
#ifndef CCL_U2
#define CCL_U2 4
#endif
#ifndef CCL_U3
#define CCL_U3 4
#endif
template <typename T>
__global__ __launch_bounds__(256) void ccl_tile_kernel(const T *__restrict__ data, int32_t *__restrict__ parent, int64_t H, int64_t W, int ntc,
                                                       int32_t *__restrict__ rootlist, int32_t *__restrict__ rootcount, int32_t *__restrict__ colpar)
{
    __shared__ uint32_t par[CT * CT];
    __shared__ int s_nroots;
    if (threadIdx.x == 0) s_nroots = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ti = blockIdx.x / ntc, tj = blockIdx.x - ti * ntc;
    const int64_t r0 = (int64_t)ti * CT, c0 = (int64_t)tj * CT;
    const int64_t cc = c0 + lane;
    // (the foreground masks of the rows are not kept for the union phase -- sixteen 64-bit values per wavefront that held the kernel
    // at 94 VGPRs: a row's mask is a ballot over its parents, which are LBG exactly on the background)
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) {
        const int r = wave + 4 * k;
        const bool fg = (r0 + r) < H && cc < W && is_fg(data[(r0 + r) * W + cc]);
        const uint64_t m = __ballot(fg);
        // first lane of my run: the highest run start at or below my lane
        const uint64_t starts = m & ~(m << 1);
        const uint64_t below = starts & ((2ull << lane) - 1ull);
        par[r * CT + lane] = fg ? (uint32_t)(r * CT + (63 - __builtin_clzll(below))) : LBG;
    }
    __syncthreads();
#pragma unroll CCL_U2
    for (int k = 0; k < CT / 4; ++k) {
        const int r = wave + 4 * k;
        if (r == 0) continue;
        const uint64_t m = __ballot(par[r * CT + lane] != LBG);
        const uint32_t pn = par[(r - 1) * CT + lane];
        const uint64_t up = __ballot(pn != LBG);
        if (!((m >> lane) & 1ull)) continue;
        const bool hasW = lane > 0 && ((m >> (lane - 1)) & 1ull), hasN = (up >> lane) & 1ull;
        const bool hasNW = lane > 0 && ((up >> (lane - 1)) & 1ull), hasNE = lane < 63 && ((up >> (lane + 1)) & 1ull);
        const uint32_t i = (uint32_t)(r * CT + lane);
        if (hasN) {
            if (!(hasW && hasNW)) unite_l(par, i, i - CT);
        } else {
            if (hasNW && !hasW) unite_l(par, i, i - CT - 1);
            if (hasNE) unite_l(par, i, i - CT + 1);
        }
    }
    __syncthreads();
#pragma unroll CCL_U3
    for (int k = 0; k < CT / 4; ++k) {
        const int r = wave + 4 * k;
        const bool inside = (r0 + r) < H && cc < W;
        const uint32_t p = inside ? par[r * CT + lane] : LBG;
        int32_t out = -1;
        // the tile's first and last column also go to a compact array (512 B per tile): the unions across the VERTICAL seams
        // read their operands there, 64 rows per 256-byte line, instead of a sector per cell in the raster
        if (lane == 0 || lane == CT - 1) {
            int32_t e = -1;
            if (p != LBG) {
                const uint32_t rt = find_root_l(par, p);
                e = (int32_t)((r0 + (rt >> 6)) * W + c0 + (rt & 63u));
            }
            colpar[((int64_t)blockIdx.x * 2 + (lane ? 1 : 0)) * CT + r] = e;
        }
        if (!inside) continue;
        if (p != LBG) {
            const uint32_t root = find_root_l(par, p);
            out = (int32_t)((r0 + (root >> 6)) * W + c0 + (root & 63u));
            // the roots of the tile's pieces: all that the global steps (flatten, rank) still have to look at
            if (root == (uint32_t)(r * CT + lane)) rootlist[(size_t)blockIdx.x * MAXROOTS + atomicAdd(&s_nroots, 1)] = out;
        }
        parent[(r0 + r) * W + cc] = out;
    }
    __syncthreads();
    if (threadIdx.x == 0) rootcount[blockIdx.x] = s_nroots;
}

// ---- the global steps on the tile roots only ---------------------------------------------------------------------------------
// flatten: parent[r] = root(r) for every tile root r; a root of the whole component (the component's first raster pixel: unions
// always hook the larger index under the smaller) sets its bit in a 1-bit-per-cell mask
__global__ __launch_bounds__(256) void ccl_flatten_roots_kernel(int32_t *parent, const int32_t *__restrict__ rootlist, const int32_t *__restrict__ rootcount,
                                                                int64_t ntiles, unsigned long long *rootbits)
{
    // four tiles per workgroup, one wavefront each
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int cnt = rootcount[tile];
    for (int j = threadIdx.x & 63; j < cnt; j += 64) {
        const int32_t r = rootlist[(size_t)tile * MAXROOTS + j];
        const int32_t root = find_root(parent, r);
        if (root != r) parent[r] = root;
        else atomicOr(&rootbits[r >> 6], 1ull << (r & 63));
    }
}

// scipy's numbering = rank of the component's first pixel in raster order = number of root bits before it: per 64-cell word
// the count of the bits before the word (block sums -> scan_blocks_kernel -> exclusive prefix inside the block)
constexpr int WB = 1024;      // words per block
__global__ __launch_bounds__(WB) void ccl_bits_count_kernel(const unsigned long long *__restrict__ rootbits, int64_t nwords, uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t wsum[WB / 64];
    const int64_t w = (int64_t)blockIdx.x * WB + threadIdx.x;
    uint32_t cnt = w < nwords ? (uint32_t)__popcll(rootbits[w]) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < WB / 64; ++k) t += wsum[k];
        block_counts[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(WB) void ccl_bits_prefix_kernel(const unsigned long long *__restrict__ rootbits, int64_t nwords,
                                                              const uint32_t *__restrict__ block_offsets, uint32_t *__restrict__ wordprefix)
{
    __shared__ uint32_t wsum[WB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * WB + threadIdx.x;
    const uint32_t cnt = w < nwords ? (uint32_t)__popcll(rootbits[w]) : 0u;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wave; ++k) woff += wsum[k];
    if (w < nwords) wordprefix[w] = block_offsets[blockIdx.x] + woff + incl - cnt;
}

// labels[i] = 1 + rank of root(i):  i -> its tile root -> (flattened) the component's root -> bits before it
// (four cells per thread, 16-byte loads / stores: the tile-root -> component-root -> rank look-ups of the four are independent)
__global__ __launch_bounds__(256) void ccl_emit_ranked_kernel(const int32_t *__restrict__ parent, const unsigned long long *__restrict__ rootbits,
                                                              const uint32_t *__restrict__ wordprefix, int32_t *__restrict__ labels, int64_t n)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    auto rank_of = [&](int32_t p) -> int32_t {
        if (p < 0) return 0;
        const int32_t g = parent[p];      // parent[tile root] = component root; a component root is its own parent
        return (int32_t)(wordprefix[g >> 6] + (uint32_t)__popcll(rootbits[g >> 6] & ((1ull << (g & 63)) - 1ull))) + 1;
    };
    if (i0 + 4 <= n) {
        const int4 p = *reinterpret_cast<const int4 *>(parent + i0);
        int4 l;
        l.x = rank_of(p.x);
        l.y = rank_of(p.y);
        l.z = rank_of(p.z);
        l.w = rank_of(p.w);
        *reinterpret_cast<int4 *>(labels + i0) = l;
    } else {
        for (int64_t i = i0; i < n; ++i) labels[i] = rank_of(parent[i]);
    }
}

// unions across the tile seams.  Horizontal seams (rows that start a tile): the rule of ccl_merge_kernel for N / NW / NE.
// Vertical seams (columns that start a tile): W; NW and SW only when W is background (else W is united with them -- inside
// its tile, or by the horizontal-seam rule of its own row).
__global__ __launch_bounds__(256) void ccl_seam_kernel(int32_t *parent, int64_t H, int64_t W, int64_t nh, int64_t total, const int32_t *__restrict__ colpar,
                                                       int64_t ntc)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    if (k < nh) {                                   // cell (r, c) of a seam row r = 64, 128, ...
        const int64_t r = (k / W + 1) * CT, c = k % W;
        const int64_t i = r * W + c;
        if (parent[i] < 0) return;
        const bool hasW = c > 0 && parent[i - 1] >= 0, hasN = parent[i - W] >= 0, hasNW = c > 0 && parent[i - W - 1] >= 0;
        if (hasN) {
            if (!(hasW && hasNW)) unite(parent, (int32_t)i, (int32_t)(i - W));
            return;
        }
        const bool hasNE = c + 1 < W && parent[i - W + 1] >= 0;
        if (hasNW && !hasW) unite(parent, (int32_t)i, (int32_t)(i - W - 1));
        if (hasNE) unite(parent, (int32_t)i, (int32_t)(i - W + 1));
    } else {                                        // row r of the seam between tile columns sj and sj + 1: the cell right of it
        const int64_t q = k - nh;
        const int64_t r = q % H, sj = q / H;
        const int64_t ti = r / CT, lr = r - ti * CT;
        const int32_t *left = colpar + ((ti * ntc + sj) * 2 + 1) * CT, *right = colpar + ((ti * ntc + sj + 1) * 2 + 0) * CT;
        const int32_t me = right[lr];               // (the tile-local root of the cell: where a find would start anyway)
        if (me < 0) return;
        const int32_t w = left[lr];
        if (w >= 0) {
            unite(parent, me, w);
            return;
        }
        if (r > 0) {
            const int32_t nw = lr > 0 ? left[lr - 1] : colpar[(((ti - 1) * ntc + sj) * 2 + 1) * CT + CT - 1];
            if (nw >= 0) unite(parent, me, nw);
        }
        if (r + 1 < H) {
            const int32_t sw = lr < CT - 1 ? left[lr + 1] : colpar[(((ti + 1) * ntc + sj) * 2 + 1) * CT];
            if (sw >= 0) unite(parent, me, sw);
        }
    }
}

__global__ __launch_bounds__(256) void ccl_flatten_kernel(int32_t *parent, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = parent[i];
    if (p < 0 || p == (int32_t)i) return;
    parent[i] = find_root(parent, p);
}

__global__ __launch_bounds__(SCAN_BLOCK) void ccl_count_roots_kernel(const int32_t *__restrict__ parent, int64_t n,
                                                                    uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t wsum[SCAN_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = base + k;
        if (i < n) cnt += parent[i] == (int32_t)i;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += wsum[w];
        block_counts[blockIdx.x] = t;
    }
}

// exclusive scan of block_counts (nb entries) by ONE block; total written to *total
__global__ __launch_bounds__(1024) void scan_blocks_kernel(uint32_t *block_counts, int64_t nb, unsigned long long *total)
{
    __shared__ unsigned long long part[1024];
    const int64_t per = (nb + 1023) / 1024;
    const int64_t b0 = (int64_t)threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
    unsigned long long s = 0;
    for (int64_t b = b0; b < b1; ++b) s += block_counts[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) {
            const unsigned long long v = part[t];
            part[t] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int64_t b = b0; b < b1; ++b) {
        const uint32_t v = block_counts[b];
        block_counts[b] = (uint32_t)run;
        run += v;
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void ccl_rank_roots_kernel(int32_t *parent, int64_t n,
                                                                   const uint32_t *__restrict__ block_offsets)
{
    __shared__ uint32_t wsum[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    bool root[SCAN_ITEMS];
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = base + k;
        root[k] = i < n && parent[i] == (int32_t)i;
        cnt += root[k];
    }
    uint32_t incl = cnt;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint32_t rank = block_offsets[blockIdx.x] + woff + incl - cnt;  // roots before this thread
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (root[k]) {
            ++rank;
            parent[base + k] = -(int32_t)rank - 1;  // root -> -(label+1); background stays -1
        }
}

__global__ __launch_bounds__(256) void ccl_emit_kernel(const int32_t *__restrict__ parent, int32_t *__restrict__ labels,
                                                      int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t p = parent[i];
    if (p >= 0) p = parent[p];  // non-root: its (flattened) parent is a ranked root
    labels[i] = p == -1 ? 0 : -p - 1;
}

template <typename T>
int ccl8_dev(const T *d_data, int32_t *d_labels, int32_t *d_tmp, int64_t H, int64_t W, int64_t *nlabels, hipStream_t s, DevBuf *stats_out = nullptr,
             CclKeep *keep = nullptr)
{
    if (keep) keep->valid = false;
    const int64_t n = H * W;
    if (n >= (int64_t)INT32_MAX - 1) {
        set_error("connected components: %lld cells exceed the int32 index domain", (long long)n);
        return MHIP_ELIMIT;
    }
    const unsigned g256 = (unsigned)cdiv(n, 256);
    const int64_t nb = cdiv(n, SCAN_TILE);
    DevBuf counts, total;
    MH_TRY(counts.alloc(sizeof(uint32_t) * (size_t)nb));
    MH_TRY(total.alloc(sizeof(unsigned long long)));
    int32_t *parent = d_tmp;
    static const bool global_uf = [] { const char *e = dev_env("MHIP_CCL"); return e && std::string(e) == "global"; }();
    if (global_uf) {   // the round-1 schedule: every union and every flatten / rank step over all cells through global memory
        hipLaunchKernelGGL((ccl_init_kernel<T>), dim3(g256), dim3(256), 0, s, d_data, parent, n, W);
        hipLaunchKernelGGL(ccl_merge_kernel, dim3(g256), dim3(256), 0, s, parent, H, W);
        hipLaunchKernelGGL(ccl_flatten_kernel, dim3(g256), dim3(256), 0, s, parent, n);
        hipLaunchKernelGGL(ccl_count_roots_kernel, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, s, parent, n, counts.as<uint32_t>());
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, counts.as<uint32_t>(), nb, total.as<unsigned long long>());
        hipLaunchKernelGGL(ccl_rank_roots_kernel, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, s, parent, n, counts.as<uint32_t>());
        hipLaunchKernelGGL(ccl_emit_kernel, dim3(g256), dim3(256), 0, s, parent, d_labels, n);
    } else {
        // three passes over the raster: tiles (read data, write parent), seams (6 % of the cells), emit (read parent, write labels);
        // everything in between works on the tile roots and on one bit per cell
        const int64_t ntr = cdiv(H, CT), ntc = cdiv(W, CT), ntiles = ntr * ntc;
        const int64_t nwords = cdiv(n, 64), nwb = cdiv(nwords, WB);
        DevBuf roots, rcount, bits, wprefix, bcounts, colpar;
        MH_TRY(colpar.alloc(4 * 2 * CT * (size_t)ntiles));
        MH_TRY(roots.alloc(4 * (size_t)ntiles * MAXROOTS));
        MH_TRY(rcount.alloc(4 * (size_t)ntiles));
        MH_TRY(bits.alloc(8 * (size_t)nwords));
        MH_TRY(wprefix.alloc(4 * (size_t)nwords));
        MH_TRY(bcounts.alloc(4 * (size_t)nwb));
        MH_HIP(hipMemsetAsync(bits.p, 0, 8 * (size_t)nwords, s));
        hipLaunchKernelGGL((ccl_tile_kernel<T>), dim3((unsigned)ntiles), dim3(256), 0, s, d_data, parent, H, W, (int)ntc, roots.as<int32_t>(),
                           rcount.as<int32_t>(), colpar.as<int32_t>());
        const int64_t nh = (ntr - 1) * W, nv = (ntc - 1) * H;
        if (nh + nv > 0)
            hipLaunchKernelGGL(ccl_seam_kernel, dim3((unsigned)cdiv(nh + nv, 256)), dim3(256), 0, s, parent, H, W, nh, nh + nv, colpar.as<int32_t>(), ntc);
        hipLaunchKernelGGL(ccl_flatten_roots_kernel, dim3((unsigned)cdiv(ntiles, 4)), dim3(256), 0, s, parent, roots.as<int32_t>(),
                           rcount.as<int32_t>(), ntiles, bits.as<unsigned long long>());
        hipLaunchKernelGGL(ccl_bits_count_kernel, dim3((unsigned)nwb), dim3(WB), 0, s, bits.as<unsigned long long>(), nwords, bcounts.as<uint32_t>());
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, bcounts.as<uint32_t>(), nwb, total.as<unsigned long long>());
        hipLaunchKernelGGL(ccl_bits_prefix_kernel, dim3((unsigned)nwb), dim3(WB), 0, s, bits.as<unsigned long long>(), nwords, bcounts.as<uint32_t>(),
                           wprefix.as<uint32_t>());
        unsigned long long h_total = 0;
        if constexpr (std::is_same<T, float>::value) {
            if (stats_out) {
                // the statistics ride on the emit pass (label_ops.hip: EmitArgs); their buffers want the number of labels first
                MH_HIP(hipGetLastError());
                MH_HIP(hipMemcpyAsync(&h_total, total.p, sizeof(h_total), hipMemcpyDeviceToHost, s));
                MH_HIP(stream_sync(s));
                *nlabels = (int64_t)h_total;
                MH_TRY(stats_out->alloc(sizeof(mhip_stat_record) * (size_t)(h_total + 1)));
                return label_emit_stats_dev(parent, bits.as<unsigned long long>(), wprefix.as<uint32_t>(), d_data, d_labels, H, W, (int64_t)h_total,
                                            stats_out->as<mhip_stat_record>(), s);     // (synchronises: the scratch buffers may go)
            }
        }
        if (keep) {
            // a row band: the emit pass waits for the seam merge (label_ops.hip: label_emit_sparse_dev); the merge wants the two top and
            // the two bottom rows of band-local labels now
            const int64_t top = (H < 4 ? H : 2) * W;
            MH_TRY(ccl_emit_rows_dev(parent, bits.as<unsigned long long>(), wprefix.as<uint32_t>(), d_labels, 0, top, s));
            if (H >= 4) MH_TRY(ccl_emit_rows_dev(parent, bits.as<unsigned long long>(), wprefix.as<uint32_t>(), d_labels, (H - 2) * W, 2 * W, s));
        } else {
            hipLaunchKernelGGL(ccl_emit_ranked_kernel, dim3((unsigned)cdiv(cdiv(n, 4), 256)), dim3(256), 0, s, parent, bits.as<unsigned long long>(), wprefix.as<uint32_t>(), d_labels, n);
        }
        MH_HIP(hipGetLastError());
        MH_HIP(hipMemcpyAsync(&h_total, total.p, sizeof(h_total), hipMemcpyDeviceToHost, s));
        MH_HIP(stream_sync(s));     // (the scratch buffers of this branch go back to the pool after the sync)
        *nlabels = (int64_t)h_total;
        if (keep) {
            keep->bits.release();
            keep->wprefix.release();
            std::swap(keep->bits.p, bits.p);
            std::swap(keep->bits.bytes, bits.bytes);
            std::swap(keep->wprefix.p, wprefix.p);
            std::swap(keep->wprefix.bytes, wprefix.bytes);
            keep->H = H;
            keep->W = W;
            keep->nlocal = (int64_t)h_total;
            keep->valid = true;
        }
        return MHIP_OK;
    }
    MH_HIP(hipGetLastError());
    unsigned long long h_total = 0;
    MH_HIP(hipMemcpyAsync(&h_total, total.p, sizeof(h_total), hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    *nlabels = (int64_t)h_total;
    if constexpr (std::is_same<T, float>::value) {
        if (stats_out) {
            MH_TRY(stats_out->alloc(sizeof(mhip_stat_record) * (size_t)(h_total + 1)));
            MH_TRY(label_stats_dev(d_data, d_labels, n, (int64_t)h_total, stats_out->as<mhip_stat_record>(), s, W, true));
        }
    }
    return MHIP_OK;
}

}  // namespace

int ccl8_f32_dev(const float *d_data, int32_t *d_labels, int32_t *d_tmp, int64_t H, int64_t W, int64_t *nlabels,
                 hipStream_t s, DevBuf *stats_out)
{
    return ccl8_dev<float>(d_data, d_labels, d_tmp, H, W, nlabels, s, stats_out);
}
int ccl8_f32_begin_dev(const float *d_data, int32_t *d_labels, int32_t *d_tmp, int64_t H, int64_t W, int64_t *nlabels, hipStream_t s, CclKeep *keep)
{
    return ccl8_dev<float>(d_data, d_labels, d_tmp, H, W, nlabels, s, nullptr, keep);
}
int ccl8_u8_dev(const uint8_t *d_data, int32_t *d_labels, int32_t *d_tmp, int64_t H, int64_t W, int64_t *nlabels,
                hipStream_t s)
{
    return ccl8_dev<uint8_t>(d_data, d_labels, d_tmp, H, W, nlabels, s);
}

}  // namespace mh
