// label_ops.hip -- per-label reductions and label re-mapping (gfx950).
//
// Reference:
//   label.label_stats      _label.pyx:68-97 (python label.py:43-75): min/max (f64), sequential f64 sum, count (i64)
//   label.label_min_index  _label.pyx:99-128: strict '<', first raster-order occurrence, init (inf,-1,-1)
//   label.label_max_index  label.py:135-166 : strict '>', first raster-order occurrence, init (-inf,-1,-1)
//   label.label_count      label.py:169-180 : np.bincount
//   label.keep_labels      label.py:78-98   : keep[background] = False; mask = keep[labelled]
//
// All of them are scatter reductions keyed by label.  Label 0 (background) typically owns ~90 % of the
// cells, so every kernel first combines equal-label runs inside a wavefront (segmented shuffle reduction).
// label_stats / label_count then combine the runs of a 32-row x 256-column tile in an LDS hash table keyed by
// label (labels are spatially compact: a bluespot or watershed crosses many rows of few tiles), and only one
// record per (label, tile) goes to the global atomics; the other kernels send their run heads there directly.  min/max go through monotone integer keys; counts are
// integers; the f64 sum uses global float atomics (order independent whenever every partial sum is exactly
// representable, which holds for bluespot depths -- SURVEY.md 8a row S; otherwise within 1 ulp-scale
// rounding of the sequential sum, see tests).
#include "common.hpp"

namespace mh {
namespace {

// ---- wave-level segmented combine over runs of equal labels --------------------------------------
// Lanes are consecutive raster cells; a run is a maximal stretch of lanes with equal label.
// distance (in lanes) from this lane to the end of its equal-label run (exclusive), computed once per wave
__device__ __forceinline__ int run_length_from(int32_t lab, int lane, bool valid)
{
    // bit i of `brk` set <=> lane i starts a new run
    const int32_t prev = __shfl_up(lab, 1);
    const bool prev_valid = __shfl_up((int)valid, 1) != 0;
    const bool start = (lane == 0) || (lab != prev) || (valid != prev_valid);
    const uint64_t brk = __ballot(start);
    // next run start strictly after this lane
    const uint64_t above = lane == 63 ? 0ull : (brk >> (lane + 1)) << (lane + 1);
    const int next = above ? __builtin_ctzll(above) : 64;
    return next - lane;  // >= 1
}

__device__ __forceinline__ bool is_run_head(int32_t lab, int lane, bool valid)
{
    const int32_t prev = __shfl_up(lab, 1);
    const bool prev_valid = __shfl_up((int)valid, 1) != 0;
    return (lane == 0) || (lab != prev) || (valid != prev_valid);
}

// generic segmented suffix-combine: after log2(64) steps lane i holds op over lanes [i, i+len_i)
template <typename T, typename Op> __device__ __forceinline__ T seg_reduce(T v, int len, Op op)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T ov = __shfl_down(v, o);
        if (o < len) {
            // combine only the part that lies inside my run: lanes [i+o, i+min(2o,len))
            v = op(v, ov);
        }
        // after this step lane i covers min(2o, len) lanes provided lane i+o covered min(o, len-o) lanes,
        // which holds because lane i+o has len' = len - o for lanes of the same run.
    }
    return v;
}

// ---- label_stats ------------------------------------------------------------------------------
struct StatAcc {
    uint32_t *minkey;           // f32 ordered keys
    uint32_t *maxkey;
    double *sum;
    unsigned long long *count;
};

__global__ __launch_bounds__(256) void stats_init_kernel(StatAcc a, int64_t nrec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    a.minkey[i] = f32_key(__builtin_inff());
    a.maxkey[i] = f32_key(-__builtin_inff());
    a.sum[i] = 0.0;
    a.count[i] = 0ull;
}

// ---- tile geometry shared by the LDS-table kernels ----------------------------------------------------------------
// The raster (width W, n cells; a flat array is treated as 256 columns wide) is cut into tiles of TR rows x 256 columns;
// a block owns a tile, its four wavefronts the four 64-column strips of every row.
constexpr int TR = 32;
struct TileGeom {
    int64_t n, W;
    int64_t ntr, ntc;
};
__host__ __device__ inline TileGeom tile_geom(int64_t n, int64_t W)
{
    TileGeom g;
    g.n = n;
    g.W = (W > 0 && n % W == 0) ? W : 256;
    const int64_t H = cdiv(n, g.W);
    g.ntr = cdiv(H, TR);
    g.ntc = cdiv(g.W, 256);
    return g;
}

// open-addressing slot of `key` in an LDS table of TS (power of two) slots, -1 when the probe limit is hit (the caller
// then falls back to the global atomics); keys[] holds -1 for an empty slot
template <int TS> __device__ __forceinline__ int table_slot(int *keys, int key)
{
    unsigned h = ((unsigned)key * 2654435761u) >> 7;
#pragma unroll 1
    for (int probe = 0; probe < 16; ++probe) {
        h &= (unsigned)(TS - 1);
        const int prev = atomicCAS(&keys[h], -1, key);
        if (prev == -1 || prev == key) return (int)h;
        ++h;
    }
    return -1;
}

// A thread follows its column down the tile (32 rows x 256 columns per workgroup, four rows of loads in flight) and keeps the
// VERTICAL run of equal labels it is in -- label, min, max, sum, count -- in registers; when the label changes the run goes to the
// label's slot of the tile's LDS table, and the table leaves as one record per (label, tile).  Background cells go to registers
// that leave once per wavefront.  (Until round 3 the equal-label runs ALONG the rows were combined by segmented shuffles first:
// 35 ds_bpermute per row of 64 cells, 0.8 of the kernel's 1.6 ms; half the cells of the benchmark DEM carry a label, in runs
// of 7.6 cells, and a vertical run costs nothing until it ends.)
// COMPONENTS: the caller vouches that the labels are the 8-connected components of a raster this one is a row range of.  A label
// none of whose cells in the tile lies on the tile's outline (or on the raster's last row) then lives in this tile alone and
// its record is WRITTEN instead of merged by a load and up to four atomics.
#ifndef MH_STATS_TS
#define MH_STATS_TS 512
#endif
constexpr int STATS_TS = MH_STATS_TS;      // (512 slots = 12 KB: the table does not limit the resident workgroups; a tile that holds more labels goes to the global atomics)
// EMIT: the labels do not exist yet -- the kernel is the last pass of the connected-component labelling (ccl.hip) as well: it
// turns parent[] (cell -> root of its tile piece -> root of its component) into ranks, WRITES the label raster and reduces on
// the way, which saves the labelling's own emit pass the statistics would read back (4 + 4 B per cell and a launch).
struct EmitArgs {
    const int32_t *parent;                  // [n] tile-local root of every cell (-1: background); parent[root] = component root
    const unsigned long long *rootbits;     // one bit per cell: component roots
    const uint32_t *wordprefix;             // root bits before each 64-bit word
    int32_t *labels;                        // out
    // a ROW BAND (EMIT == 2): `data` starts at cell `cell0` of the labelled raster (the band's first owned row; parent / rootbits /
    // wordprefix / labels cover the whole local raster, halo rows included), and the rank -- the band-local label -- goes through the
    // sparse map of the band protocol on the way out (see sparse_map): what leaves is the GLOBAL label, and the records are the
    // global label's
    int64_t cell0;
    const uint32_t *cnt256;                 // number of dropped labels below k * 256
    const int32_t *dropped, *target;
    int32_t offset;
};
// band-local label l (>= 1) -> global label: offset + l - (number of dropped labels below l), a dropped label (sorted list: the
// labels another band numbers, or that own no cell here) -> its explicit target.  The list is short and a table of its counts per
// 256 labels takes most of the search away: two loads give the dropped labels that share l's 256-range [a, b) -- none for nearly every
// label; the labels of the band's first rows (components that touch the top seam: all dropped, and some of them lakes of 100 000
// cells) find 256 of them there and search those in 8 steps.
__device__ __forceinline__ int32_t sparse_map(int32_t l, uint32_t a, uint32_t b, const int32_t *__restrict__ dropped, const int32_t *__restrict__ target,
                                              int32_t offset)
{
    const uint32_t end = b;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (dropped[mid] < l) a = mid + 1;
        else b = mid;
    }
    if (a < end && dropped[a] == l) return target[a];
    return offset + l - (int32_t)a;
}
template <bool COMPONENTS, int EMIT>
__attribute__((amdgpu_waves_per_eu(8, 8)))      // eight workgroups per CU (tests/test_build_lint.py); the band variant asked for 68 registers without it
__global__ __launch_bounds__(256) void stats_kernel(const float *__restrict__ data, const int32_t *__restrict__ lab,
                                                   TileGeom g, int64_t nlab, StatAcc a, unsigned int *bad, EmitArgs em)
{
    __shared__ int keys[STATS_TS];
    __shared__ unsigned int tcnt[STATS_TS], tmin[STATS_TS], tmax[STATS_TS];
    __shared__ double tsum[STATS_TS];
    const int lane = threadIdx.x & 63;
    float bmin = __builtin_inff(), bmax = -__builtin_inff();
    double bsum = 0.0;
    unsigned long long bcnt = 0;
    auto to_global = [&](int32_t l, uint32_t kmin, uint32_t kmax, double sum, unsigned long long cnt) {
        if (kmin < a.minkey[l]) atomicMin(&a.minkey[l], kmin);     // (a stale read is conservative)
        if (kmax > a.maxkey[l]) atomicMax(&a.maxkey[l], kmax);
        atomicAdd(&a.sum[l], sum);
        atomicAdd(&a.count[l], cnt);
    };
    const int64_t ntiles = g.ntr * g.ntc;
    const int64_t last_row = cdiv(g.n, g.W) - 1;
    const bool edge_col = threadIdx.x == 0 || threadIdx.x == 255;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int k = threadIdx.x; k < STATS_TS; k += 256) {
            keys[k] = -1;
            tcnt[k] = 0u;
            tmin[k] = 0xffffffffu;
            tmax[k] = 0u;
            tsum[k] = 0.0;
        }
        __syncthreads();
        const int64_t tr = tile / g.ntc, tc = tile - tr * g.ntc;
        const int64_t col = tc * 256 + threadIdx.x;
        // the run this thread is in: label (<= 0: none), first tile row, min, max, sum, count
        int32_t cl = -1;
        int cstart = 0;
        float cmin = 0.0f, cmax = 0.0f;
        double csum = 0.0;
        unsigned int ccnt = 0;
        auto end_run = [&](int end_row) {     // the run cl ended in tile row end_row
            const uint32_t kmin = f32_key(cmin), kmax = f32_key(cmax);
            const int h = table_slot<STATS_TS>(keys, cl);
            if (h >= 0) {
                atomicMin(&tmin[h], kmin);
                atomicMax(&tmax[h], kmax);
                atomicAdd(&tsum[h], csum);
                atomicAdd(&tcnt[h], ccnt);
                // bit 31 of the count (a tile holds 8192 cells): the label reaches the tile's outline
                if (COMPONENTS && (edge_col || cstart == 0 || end_row == TR - 1 || tr * TR + end_row >= last_row)) atomicOr(&tcnt[h], 0x80000000u);
            } else {
                to_global(cl, kmin, kmax, csum, (unsigned long long)ccnt);
            }
        };
        static_assert(TR % 4 == 0, "rows in batches of four");
        for (int r4 = 0; r4 < TR; r4 += 4) {
            int32_t lq[4];           // four rows' loads in flight
            float dq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (tr * TR + r4 + u) * g.W + col;
                const bool valid = col < g.W && i < g.n;
                lq[u] = valid ? (EMIT ? em.parent[(EMIT == 2 ? em.cell0 : 0) + i] : lab[i]) : -1;
                dq[u] = valid ? data[i] : 0.0f;
                if (!EMIT && valid && (lq[u] < 0 || lq[u] > nlab)) {
                    atomicOr(bad, 1u);
                    lq[u] = -1;
                }
            }
            if (EMIT) {
                // rank of the component root = the label (the four rows' look-ups are independent: in flight together)
                int32_t gq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) gq[u] = lq[u] >= 0 ? em.parent[lq[u]] : -1;
                // (the four rows' prefix words and root bits in flight together as well: looked up row by row inside the `>= 0` branch,
                // each pair was waited for before the next row's went out)
                uint32_t wp[4];
                unsigned long long rb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t gg = gq[u] >= 0 ? gq[u] : 0;
                    wp[u] = em.wordprefix[gg >> 6];
                    rb[u] = em.rootbits[gg >> 6];
                }
                int32_t lr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t gg = gq[u];
                    lr[u] = gg >= 0 ? (int32_t)(wp[u] + (uint32_t)__popcll(rb[u] & ((1ull << (gg & 63)) - 1ull))) + 1 : 0;
                }
                if (EMIT == 2) {
                    // (two rows' table entries in flight together: all four cost the kernel its eighth workgroup per CU in registers)
#pragma unroll
                    for (int h = 0; h < 4; h += 2) {
                        const uint32_t a0 = em.cnt256[(uint32_t)lr[h] >> 8], b0 = em.cnt256[((uint32_t)lr[h] >> 8) + 1];
                        const uint32_t a1 = em.cnt256[(uint32_t)lr[h + 1] >> 8], b1 = em.cnt256[((uint32_t)lr[h + 1] >> 8) + 1];
                        if (lr[h] > 0) lr[h] = sparse_map(lr[h], a0, b0, em.dropped, em.target, em.offset);
                        if (lr[h + 1] > 0) lr[h + 1] = sparse_map(lr[h + 1], a1, b1, em.dropped, em.target, em.offset);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = (tr * TR + r4 + u) * g.W + col;
                    const bool valid = col < g.W && i < g.n;
                    if (valid) em.labels[(EMIT == 2 ? em.cell0 : 0) + i] = lr[u];
                    lq[u] = valid ? lr[u] : -1;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t l = lq[u];
                const float v = dq[u];
                const bool isnan = v != v;
                // NaN never wins `val < min` / `val > max` in the reference; it does poison the sum.
                const float vmin = isnan ? __builtin_inff() : v, vmax = isnan ? -__builtin_inff() : v;
                if (l != cl) {
                    if (cl > 0) end_run(r4 + u - 1);
                    cl = l;
                    cstart = r4 + u;
                    cmin = __builtin_inff();
                    cmax = -__builtin_inff();
                    csum = 0.0;
                    ccnt = 0;
                }
                if (l == 0) {
                    bmin = fminf(bmin, vmin);
                    bmax = fmaxf(bmax, vmax);
                    bsum += (double)v;
                    ++bcnt;
                } else if (l > 0) {
                    cmin = fminf(cmin, vmin);
                    cmax = fmaxf(cmax, vmax);
                    csum += (double)v;
                    ++ccnt;
                }
            }
        }
        if (cl > 0) end_run(TR - 1);
        __syncthreads();
        for (int k = threadIdx.x; k < STATS_TS; k += 256) {
            const int l = keys[k];
            if (l < 0) continue;
            const unsigned int cnt = tcnt[k];
            if (!COMPONENTS || (cnt >> 31)) {
                to_global(l, tmin[k], tmax[k], tsum[k], (unsigned long long)(cnt & 0x7fffffffu));
            } else {
                a.minkey[l] = tmin[k];
                a.maxkey[l] = tmax[k];
                a.sum[l] = tsum[k];
                a.count[l] = (unsigned long long)cnt;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bmin = fminf(bmin, __shfl_xor(bmin, o));
        bmax = fmaxf(bmax, __shfl_xor(bmax, o));
        bsum += __shfl_xor(bsum, o);
        bcnt += __shfl_xor(bcnt, o);
    }
    // (the background's record: one set of atomics per WORKGROUP -- every wavefront of a 2048-workgroup launch at the same four words
    // was 8192 same-address atomics each, ~0.1 ms of serialised tail at the end of a kernel on the chain's critical path)
    __shared__ float wmin[4], wmax[4];
    __shared__ double wsum[4];
    __shared__ unsigned long long wcnt[4];
    __syncthreads();
    if (lane == 0) {
        wmin[threadIdx.x >> 6] = bmin; wmax[threadIdx.x >> 6] = bmax; wsum[threadIdx.x >> 6] = bsum; wcnt[threadIdx.x >> 6] = bcnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            bmin = fminf(bmin, wmin[k]); bmax = fmaxf(bmax, wmax[k]); bsum += wsum[k]; bcnt += wcnt[k];
        }
        if (bcnt) {
            atomicMin(&a.minkey[0], f32_key(bmin));
            atomicMax(&a.maxkey[0], f32_key(bmax));
            atomicAdd(&a.sum[0], bsum);
            atomicAdd(&a.count[0], bcnt);
        }
    }
}

__global__ __launch_bounds__(256) void stats_finish_kernel(StatAcc a, int64_t nrec, mhip_stat_record *rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    mhip_stat_record r;
    r.min = (double)key_f32(a.minkey[i]);
    r.max = (double)key_f32(a.maxkey[i]);
    r.sum = a.sum[i];
    r.count = (int64_t)a.count[i];
    rec[i] = r;
}

// ---- label_stats on float64 data (the reference's generic path, label.py:43-75; not on the hot path) -------------
// One thread per cell; equal-label runs are combined inside a wavefront and the run heads go to the global atomics.
struct StatAcc64 {
    uint64_t *minkey, *maxkey;  // f64 ordered keys
    double *sum;
    unsigned long long *count;
};
__global__ __launch_bounds__(256) void stats64_init_kernel(StatAcc64 a, int64_t nrec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    a.minkey[i] = f64_key(__builtin_inf());
    a.maxkey[i] = f64_key(-__builtin_inf());
    a.sum[i] = 0.0;
    a.count[i] = 0ull;
}
__global__ __launch_bounds__(256) void stats64_kernel(const double *__restrict__ data, const int32_t *__restrict__ lab, int64_t n,
                                                      int64_t nlab, StatAcc64 a, unsigned int *bad)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    int32_t l = valid ? lab[i] : -1;
    const double v = valid ? data[i] : 0.0;
    if (valid && (l < 0 || l > nlab)) {
        atomicOr(bad, 1u);
        l = -1;
    }
    const bool ok = l >= 0;
    const bool isnan = v != v;   // never wins `val < min` / `val > max`; it does poison the sum (label.py:68-73)
    double vmin = isnan ? __builtin_inf() : v, vmax = isnan ? -__builtin_inf() : v;
    const int len = run_length_from(l, lane, ok);
    const bool head = is_run_head(l, lane, ok);
    vmin = seg_reduce(vmin, len, [](double x, double y) { return fmin(x, y); });
    vmax = seg_reduce(vmax, len, [](double x, double y) { return fmax(x, y); });
    const double sm = seg_reduce(v, len, [](double x, double y) { return x + y; });
    if (head && ok) {
        atomicMin((unsigned long long *)&a.minkey[l], (unsigned long long)f64_key(vmin));
        atomicMax((unsigned long long *)&a.maxkey[l], (unsigned long long)f64_key(vmax));
        atomicAdd(&a.sum[l], sm);
        atomicAdd(&a.count[l], (unsigned long long)len);
    }
}
__global__ __launch_bounds__(256) void stats64_finish_kernel(StatAcc64 a, int64_t nrec, mhip_stat_record *rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    mhip_stat_record r;
    r.min = key_f64(a.minkey[i]);
    r.max = key_f64(a.maxkey[i]);
    r.sum = a.sum[i];
    r.count = (int64_t)a.count[i];
    rec[i] = r;
}

// ---- label_min_index / label_max_index ----------------------------------------------------------
// pass 1: per label extreme of the monotone f64 key (-0.0 folded onto +0.0 so that float equality == key equality)
// pass 2: smallest linear index whose value equals the extreme
__device__ __forceinline__ uint64_t arg_key(double v) { return f64_key(v == 0.0 ? 0.0 : v); }

__global__ __launch_bounds__(256) void arg_init_kernel(uint64_t *key, unsigned long long *idx, int64_t nrec, bool is_max)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    key[i] = f64_key(is_max ? -__builtin_inf() : __builtin_inf());
    idx[i] = ~0ull;
}

__global__ __launch_bounds__(256) void arg_pass1_kernel(const double *__restrict__ data, const int32_t *__restrict__ lab,
                                                       int64_t n, int64_t nlab, uint64_t *key, bool is_max,
                                                       unsigned int *bad)
{
    const int lane = threadIdx.x & 63;
    const double ident = is_max ? -__builtin_inf() : __builtin_inf();
    uint64_t bkey = f64_key(ident);  // background (label 0) extreme, kept in registers
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nloop = cdiv(n, stride) * stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nloop; i += stride) {
        const bool valid = i < n;
        int32_t l = valid ? lab[i] : -1;
        if (valid && (l < 0 || l > nlab)) {
            atomicOr(bad, 1u);
            l = -1;
        }
        const double v = valid ? data[i] : 0.0;
        // NaN never wins a strict compare: map it to the identity
        uint64_t k = arg_key(v != v ? ident : v);
        if (__all(l <= 0)) {
            if (l == 0) bkey = is_max ? (k > bkey ? k : bkey) : (k < bkey ? k : bkey);
            continue;
        }
        const bool ok = l >= 0;
        const int len = run_length_from(l, lane, ok);
        const bool head = is_run_head(l, lane, ok);
        k = is_max ? seg_reduce(k, len, [](uint64_t x, uint64_t y) { return x > y ? x : y; })
                   : seg_reduce(k, len, [](uint64_t x, uint64_t y) { return x < y ? x : y; });
        if (head && ok) {
            if (l == 0) bkey = is_max ? (k > bkey ? k : bkey) : (k < bkey ? k : bkey);
            else if (is_max) { if (k > key[l]) atomicMax(reinterpret_cast<unsigned long long *>(&key[l]), (unsigned long long)k); }
            else { if (k < key[l]) atomicMin(reinterpret_cast<unsigned long long *>(&key[l]), (unsigned long long)k); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t ob = __shfl_xor(bkey, o);
        bkey = is_max ? (ob > bkey ? ob : bkey) : (ob < bkey ? ob : bkey);
    }
    if (lane == 0) {
        if (is_max) atomicMax(reinterpret_cast<unsigned long long *>(&key[0]), (unsigned long long)bkey);
        else atomicMin(reinterpret_cast<unsigned long long *>(&key[0]), (unsigned long long)bkey);
    }
}

__global__ __launch_bounds__(256) void arg_pass2_kernel(const double *__restrict__ data, const int32_t *__restrict__ lab,
                                                       int64_t n, int64_t nlab, const uint64_t *__restrict__ key,
                                                       unsigned long long *idx, bool is_max)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = lab[i];
    if (l < 0 || l > nlab) return;
    const double v = data[i];
    if (v != v) return;
    const uint64_t k = arg_key(v);
    // the init value (+-inf) is never a strict improvement over the record's own +-inf: skip it like the reference
    if (k == f64_key(is_max ? -__builtin_inf() : __builtin_inf())) return;
    // stale reads of idx[] are conservative (the true value is never larger), so skipping on them is safe
    if (k == key[l] && (unsigned long long)i < idx[l]) atomicMin(&idx[l], (unsigned long long)i);
}

// ---- single-pass arg-max for integer-valued data (accumulated flow: the pour points, bluespots.py:195-206) -----------
// value and position share one 64-bit key, (value << 32) | (0xffffffff - index): the largest key is the largest value at
// its FIRST raster position, so one atomicMax per run replaces the two passes above.  Needs 0 <= value < 2**32 integral
// and fewer than 2**32 - 1 cells; a cell that does not qualify raises `notint` and the caller falls back to the two passes.
// Tiled like stats_kernel: a thread follows its column down the tile and keeps the largest
// key of the vertical run it is in; runs go to the tile's LDS table when they end, the table leaves as one atomicMax per
// (label, tile) -- or as a plain store for a component that lies inside the tile (COMPONENTS, see stats_kernel).
constexpr int ARG_TS = 1024;
template <bool COMPONENTS>
__global__ __launch_bounds__(256) void arg_packed_kernel(const double *__restrict__ data, const int32_t *__restrict__ lab, TileGeom g, int64_t nlab,
                                                        uint64_t *key, unsigned int *bad, unsigned int *notint)
{
    __shared__ int keys[ARG_TS];
    __shared__ unsigned long long tkey[ARG_TS];
    __shared__ unsigned char tout[ARG_TS];
    const int lane = threadIdx.x & 63;
    uint64_t bkey = 0;  // background (label 0), kept in registers
    const int64_t ntiles = g.ntr * g.ntc;
    const int64_t last_row = cdiv(g.n, g.W) - 1;
    const bool edge_col = threadIdx.x == 0 || threadIdx.x == 255;
    unsigned int any_bad = 0, any_notint = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int k = threadIdx.x; k < ARG_TS; k += 256) {
            keys[k] = -1;
            tkey[k] = 0ull;
            tout[k] = 0;
        }
        __syncthreads();
        const int64_t tr = tile / g.ntc, tc = tile - tr * g.ntc;
        const int64_t col = tc * 256 + threadIdx.x;
        int32_t cl = -1;
        int cstart = 0;
        uint64_t ckey = 0;
        auto end_run = [&](int end_row) {
            const int h = table_slot<ARG_TS>(keys, cl);
            if (h >= 0) {
                atomicMax(&tkey[h], (unsigned long long)ckey);
                if (COMPONENTS && (edge_col || cstart == 0 || end_row == TR - 1 || tr * TR + end_row >= last_row)) tout[h] = 1;
            } else if (ckey > key[cl]) {
                atomicMax(reinterpret_cast<unsigned long long *>(&key[cl]), (unsigned long long)ckey);
            }
        };
        for (int r4 = 0; r4 < TR; r4 += 4) {
            int32_t lq[4];       // four rows' loads in flight
            double dq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (tr * TR + r4 + u) * g.W + col;
                const bool valid = col < g.W && i < g.n;
                lq[u] = valid ? lab[i] : -1;
                dq[u] = valid ? data[i] : 0.0;
                if (valid && (lq[u] < 0 || lq[u] > nlab)) {
                    any_bad = 1;
                    lq[u] = -1;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (tr * TR + r4 + u) * g.W + col;
                const int32_t l = lq[u];
                const double v = dq[u];
                const bool isint = v >= 0.0 && v < 4294967296.0 && (double)(uint32_t)v == v;   // false for NaN
                if (l >= 0 && !isint) any_notint = 1;
                const uint64_t k = ((uint64_t)(uint32_t)v << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
                if (l != cl) {
                    if (cl > 0) end_run(r4 + u - 1);
                    cl = l;
                    cstart = r4 + u;
                    ckey = 0;
                }
                if (l == 0) bkey = k > bkey ? k : bkey;
                else if (l > 0) ckey = k > ckey ? k : ckey;
            }
        }
        if (cl > 0) end_run(TR - 1);
        __syncthreads();
        for (int k = threadIdx.x; k < ARG_TS; k += 256) {
            const int l = keys[k];
            if (l < 0) continue;
            const unsigned long long kk = tkey[k];
            if (!COMPONENTS || tout[k]) {
                if (kk > key[l]) atomicMax(reinterpret_cast<unsigned long long *>(&key[l]), kk);
            } else {
                key[l] = kk;
            }
        }
        __syncthreads();
    }
    if (any_bad) atomicOr(bad, 1u);
    if (any_notint) atomicOr(notint, 1u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t ob = __shfl_xor(bkey, o);
        bkey = ob > bkey ? ob : bkey;
    }
    if (lane == 0 && bkey) atomicMax(reinterpret_cast<unsigned long long *>(&key[0]), (unsigned long long)bkey);
}

__global__ __launch_bounds__(256) void arg_packed_finish_kernel(const double *__restrict__ data, const uint64_t *__restrict__ key,
                                                               int64_t nrec, int64_t W, mhip_index_record *rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    mhip_index_record r;
    const uint64_t k = key[i];
    if (k == 0) {  // no cell: a real key never is 0 (its low word is 0xffffffff - index > 0)
        r.value = -__builtin_inf();
        r.row = -1;
        r.col = -1;
    } else {
        const uint64_t p = 0xffffffffu - (uint32_t)k;
        r.value = data[p];
        r.row = (int64_t)(p / (uint64_t)W);
        r.col = (int64_t)(p % (uint64_t)W);
    }
    rec[i] = r;
}

__global__ __launch_bounds__(256) void arg_finish_kernel(const double *__restrict__ data, const unsigned long long *idx,
                                                        int64_t nrec, int64_t W, bool is_max, mhip_index_record *rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    mhip_index_record r;
    const unsigned long long p = idx[i];
    if (p == ~0ull) {
        r.value = is_max ? -__builtin_inf() : __builtin_inf();
        r.row = -1;
        r.col = -1;
    } else {
        r.value = data[p];
        r.row = (int64_t)(p / (unsigned long long)W);
        r.col = (int64_t)(p % (unsigned long long)W);
    }
    rec[i] = r;
}

// ---- bincount / max / lut / mask -------------------------------------------------------------------
constexpr int COUNT_TS = 2048;
__global__ __launch_bounds__(256) void count_kernel(const int32_t *__restrict__ lab, TileGeom g, int64_t nlab,
                                                   unsigned long long *counts, unsigned int *bad)
{
    __shared__ int keys[COUNT_TS];
    __shared__ unsigned int tcnt[COUNT_TS];
    const int lane = threadIdx.x & 63;
    unsigned long long bcnt = 0;
    const int64_t ntiles = g.ntr * g.ntc;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int k = threadIdx.x; k < COUNT_TS; k += 256) {
            keys[k] = -1;
            tcnt[k] = 0u;
        }
        __syncthreads();
        const int64_t tr = tile / g.ntc, tc = tile - tr * g.ntc;
        const int64_t col = tc * 256 + threadIdx.x;
        static_assert(TR % 4 == 0, "rows in batches of four");
        for (int r4 = 0; r4 < TR; r4 += 4) {
            int32_t lq[4];       // four rows' loads in flight
            bool vq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (tr * TR + r4 + u) * g.W + col;
                vq[u] = col < g.W && i < g.n;
                lq[u] = vq[u] ? lab[i] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int32_t l = lq[u];
                if (vq[u] && (l < 0 || l > nlab)) {
                    atomicOr(bad, 1u);
                    l = -1;
                }
                if (__all(l <= 0)) {
                    bcnt += l == 0;
                    continue;
                }
                const bool ok = l >= 0;
                const int len = run_length_from(l, lane, ok);
                if (is_run_head(l, lane, ok) && ok) {
                    if (l == 0) bcnt += (unsigned long long)len;
                    else {
                        const int h = table_slot<COUNT_TS>(keys, l);
                        if (h >= 0) atomicAdd(&tcnt[h], (unsigned int)len);
                        else atomicAdd(&counts[l], (unsigned long long)len);
                    }
                }
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < COUNT_TS; k += 256)
            if (keys[k] >= 0) atomicAdd(&counts[keys[k]], (unsigned long long)tcnt[k]);
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bcnt += __shfl_xor(bcnt, o);
    __shared__ unsigned long long wcnt0[4];      // (one atomic per workgroup: see stats_kernel)
    __syncthreads();
    if (lane == 0) wcnt0[threadIdx.x >> 6] = bcnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        bcnt = wcnt0[0] + wcnt0[1] + wcnt0[2] + wcnt0[3];
        if (bcnt) atomicAdd(&counts[0], bcnt);
    }
}

__global__ __launch_bounds__(256) void max_kernel(const int32_t *__restrict__ lab, int64_t n, int *out)
{
    int m = INT32_MIN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = lab[i] > m ? lab[i] : m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int v = __shfl_xor(m, o);
        m = v > m ? v : m;
    }
    // (one atomic per workgroup, few workgroups: same-address atomics are served one after the other -- see reduce.hip: minmax_kernel)
    __shared__ int sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) m = sm[k] > m ? sm[k] : m;
        atomicMax(out, m);
    }
}

__global__ __launch_bounds__(256) void lut_kernel(int32_t *lab, const int32_t *__restrict__ lut, int64_t nlab, int64_t n,
                                                 unsigned int *bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = lab[i];
    if (l < 0 || l > nlab) {
        atomicOr(bad, 1u);
        return;
    }
    if (l != 0) lab[i] = lut[l];
}

// band-local -> global labels without a dense LUT: local label l becomes offset + l - (number of dropped labels below l);
// a dropped label (sorted `dropped`) becomes its explicit target.  The dropped list is short (labels that touch a band
// boundary), so the binary search stays in cache.
__global__ __launch_bounds__(256) void relabel_sparse_kernel(int32_t *lab, int64_t n, int64_t nlocal, int32_t offset,
                                                            const int32_t *__restrict__ dropped, const int32_t *__restrict__ target,
                                                            int32_t ndropped, unsigned int *bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = lab[i];
    if (l < 0 || l > nlocal) {
        atomicOr(bad, 1u);
        return;
    }
    if (l == 0) return;
    int32_t lo = 0, hi = ndropped;   // first index with dropped[idx] >= l
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (dropped[mid] < l) lo = mid + 1;
        else hi = mid;
    }
    lab[i] = (lo < ndropped && dropped[lo] == l) ? target[lo] : offset + l - lo;
}

// labels in [lo, hi] (the labels this band numbered) go through a dense table, every other positive label (a bluespot numbered by
// another band that reaches into this one: a short sorted list) through a binary search; a label found in neither becomes 0
__global__ __launch_bounds__(256) void relabel_range_kernel(int32_t *lab, int64_t n, int32_t lo, int32_t hi, const int32_t *__restrict__ lut,
                                                           const int32_t *__restrict__ fid, const int32_t *__restrict__ fnew, int32_t nf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = lab[i];
    if (l <= 0) return;
    if (l >= lo && l <= hi) {
        lab[i] = lut[l - lo];
        return;
    }
    int32_t a = 0, b = nf;
    while (a < b) {
        const int32_t mid = (a + b) >> 1;
        if (fid[mid] < l) a = mid + 1;
        else b = mid;
    }
    lab[i] = (a < nf && fid[a] == l) ? fnew[a] : 0;
}

__global__ __launch_bounds__(256) void mask_kernel(const int32_t *__restrict__ lab, const uint8_t *__restrict__ keep,
                                                  int64_t nlab, int64_t n, uint8_t *mask, unsigned int *bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = lab[i];
    if (l < 0 || l > nlab) {
        atomicOr(bad, 1u);
        return;
    }
    mask[i] = keep[l] ? 1 : 0;  // the caller has already cleared keep[background]
}

// grid for the grid-stride reductions: 256 CUs x 8 blocks
unsigned stride_grid(int64_t n) { return (unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048); }

int check_bad(DevBuf &bad, hipStream_t s, const char *what)
{
    unsigned int h = 0;
    MH_HIP(hipMemcpyAsync(&h, bad.p, sizeof(h), hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    if (h) {
        set_error("%s: label outside [0, nlabels]", what);
        return MHIP_EINVAL;
    }
    return MHIP_OK;
}

}  // namespace

// one block per tile up to 256 CUs x 8 blocks, then tile-stride
static unsigned tile_grid(const TileGeom &g)
{
    const int64_t nt = g.ntr * g.ntc;
    return (unsigned)(nt < 2048 ? (nt > 0 ? nt : 1) : 2048);
}

int label_stats_dev(const float *d_data, const int32_t *d_labels, int64_t n, int64_t nlab, mhip_stat_record *d_rec,
                    hipStream_t s, int64_t W, bool components)
{
    const int64_t nrec = nlab + 1;
    DevBuf mn, mx, sm, ct, bad;
    MH_TRY(mn.alloc(4 * (size_t)nrec));
    MH_TRY(mx.alloc(4 * (size_t)nrec));
    MH_TRY(sm.alloc(8 * (size_t)nrec));
    MH_TRY(ct.alloc(8 * (size_t)nrec));
    MH_TRY(bad.alloc(4));
    MH_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    StatAcc a{mn.as<uint32_t>(), mx.as<uint32_t>(), sm.as<double>(), ct.as<unsigned long long>()};
    hipLaunchKernelGGL(stats_init_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec);
    const TileGeom g = tile_geom(n, W);
    // (the shortcut for components needs the raster's real geometry: tile_geom treats a flat array as 256 columns wide)
    // (the shortcut for components needs the raster's real geometry: tile_geom treats a flat array as 256 columns wide)
    if (components && W > 0 && n % W == 0)
        hipLaunchKernelGGL((stats_kernel<true, 0>), dim3(tile_grid(g)), dim3(256), 0, s, d_data, d_labels, g, nlab, a, bad.as<unsigned int>(), EmitArgs{});
    else
        hipLaunchKernelGGL((stats_kernel<false, 0>), dim3(tile_grid(g)), dim3(256), 0, s, d_data, d_labels, g, nlab, a, bad.as<unsigned int>(), EmitArgs{});
    hipLaunchKernelGGL(stats_finish_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec, d_rec);
    MH_HIP(hipGetLastError());
    return check_bad(bad, s, "label_stats");
}

// the last pass of ccl8_dev fused with label_stats (see EmitArgs): labels written, records of the labels in d_rec
int label_emit_stats_dev(const int32_t *d_parent, const unsigned long long *d_rootbits, const uint32_t *d_wordprefix, const float *d_data,
                         int32_t *d_labels, int64_t H, int64_t W, int64_t nlab, mhip_stat_record *d_rec, hipStream_t s)
{
    const int64_t nrec = nlab + 1, n = H * W;
    DevBuf mn, mx, sm, ct;
    MH_TRY(mn.alloc(4 * (size_t)nrec));
    MH_TRY(mx.alloc(4 * (size_t)nrec));
    MH_TRY(sm.alloc(8 * (size_t)nrec));
    MH_TRY(ct.alloc(8 * (size_t)nrec));
    StatAcc a{mn.as<uint32_t>(), mx.as<uint32_t>(), sm.as<double>(), ct.as<unsigned long long>()};
    hipLaunchKernelGGL(stats_init_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec);
    TileGeom g;
    g.n = n; g.W = W; g.ntr = cdiv(H, TR); g.ntc = cdiv(W, 256);
    hipLaunchKernelGGL((stats_kernel<true, 1>), dim3(tile_grid(g)), dim3(256), 0, s, d_data, (const int32_t *)nullptr, g, nlab, a, (unsigned int *)nullptr,
                       EmitArgs{d_parent, d_rootbits, d_wordprefix, d_labels, 0, nullptr, nullptr, nullptr, 0});
    hipLaunchKernelGGL(stats_finish_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec, d_rec);
    MH_HIP(hipGetLastError());
    MH_HIP(stream_sync(s));      // the accumulators go back to the pool
    return MHIP_OK;
}

// ---- a row band's labelling in two halves (ccl.hip: ccl8_f32_begin_dev keeps parent / rootbits / wordprefix) ----------------------
// cells [base, base + count) of the labelled raster: rank of the cell's component (the band-local label), SPARSE: through the
// band protocol's map (sparse_map) to the global label
template <bool SPARSE>
__global__ __launch_bounds__(256) void emit_range_kernel(EmitArgs em, int64_t base, int64_t count)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t i = base + k;
    const int32_t p = em.parent[i];
    int32_t l = 0;
    if (p >= 0) {
        const int32_t g = em.parent[p];
        l = (int32_t)(em.wordprefix[g >> 6] + (uint32_t)__popcll(em.rootbits[g >> 6] & ((1ull << (g & 63)) - 1ull))) + 1;
        if (SPARSE) l = sparse_map(l, em.cnt256[(uint32_t)l >> 8], em.cnt256[((uint32_t)l >> 8) + 1], em.dropped, em.target, em.offset);
    }
    em.labels[i] = l;
}
// cnt256[k] = number of dropped labels below k * 256 (lower bound in the sorted list)
__global__ __launch_bounds__(256) void sparse_index_kernel(const int32_t *__restrict__ dropped, int32_t ndropped, int64_t nk, uint32_t *__restrict__ cnt256)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    const int64_t key = k << 8;
    int32_t lo = 0, hi = ndropped;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if ((int64_t)dropped[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    cnt256[k] = (uint32_t)lo;
}

int ccl_emit_rows_dev(const int32_t *d_parent, const unsigned long long *d_rootbits, const uint32_t *d_wordprefix, int32_t *d_labels, int64_t base,
                      int64_t count, hipStream_t s)
{
    if (count <= 0) return MHIP_OK;
    hipLaunchKernelGGL((emit_range_kernel<false>), dim3((unsigned)cdiv(count, 256)), dim3(256), 0, s,
                       EmitArgs{d_parent, d_rootbits, d_wordprefix, d_labels, 0, nullptr, nullptr, nullptr, 0}, base, count);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

// The second half: every cell of the local raster (H rows, the owned ones are [ht, ht + H_owned)) gets its GLOBAL label in ONE pass
// from parent[] -- rank (= band-local label) -> sparse map -- instead of an emit pass and a relabelling pass over the labels; with
// d_rec the owned rows' label_stats ride on that pass as in the single context (records by GLOBAL label, nlab_global + 1 of them).
int label_emit_sparse_dev(const int32_t *d_parent, const unsigned long long *d_rootbits, const uint32_t *d_wordprefix, const float *d_data,
                          int32_t *d_labels, int64_t H, int64_t W, int ht, int64_t H_owned, int64_t nlocal, int32_t offset,
                          const int32_t *d_dropped, const int32_t *d_target, int32_t ndropped, int64_t nlab_global, mhip_stat_record *d_rec,
                          hipStream_t s)
{
    const int64_t nk = (nlocal >> 8) + 2;
    DevBuf cnt;
    MH_TRY(cnt.alloc(4 * (size_t)nk));
    hipLaunchKernelGGL(sparse_index_kernel, dim3((unsigned)cdiv(nk, 256)), dim3(256), 0, s, d_dropped, ndropped, nk, cnt.as<uint32_t>());
    EmitArgs em{d_parent, d_rootbits, d_wordprefix, d_labels, (int64_t)ht * W, cnt.as<uint32_t>(), d_dropped, d_target, offset};
    const int64_t n_owned = H_owned * W;
    DevBuf mn, mx, sm, ct;
    if (d_rec) {
        const int64_t nrec = nlab_global + 1;
        MH_TRY(mn.alloc(4 * (size_t)nrec));
        MH_TRY(mx.alloc(4 * (size_t)nrec));
        MH_TRY(sm.alloc(8 * (size_t)nrec));
        MH_TRY(ct.alloc(8 * (size_t)nrec));
        StatAcc a{mn.as<uint32_t>(), mx.as<uint32_t>(), sm.as<double>(), ct.as<unsigned long long>()};
        hipLaunchKernelGGL(stats_init_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec);
        TileGeom g;
        g.n = n_owned; g.W = W; g.ntr = cdiv(H_owned, TR); g.ntc = cdiv(W, 256);
        hipLaunchKernelGGL((stats_kernel<true, 2>), dim3(tile_grid(g)), dim3(256), 0, s, d_data + em.cell0, (const int32_t *)nullptr, g, nlab_global, a,
                           (unsigned int *)nullptr, em);
        hipLaunchKernelGGL(stats_finish_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec, d_rec);
    } else if (n_owned > 0) {
        hipLaunchKernelGGL((emit_range_kernel<true>), dim3((unsigned)cdiv(n_owned, 256)), dim3(256), 0, s, em, em.cell0, n_owned);
    }
    if (ht > 0) hipLaunchKernelGGL((emit_range_kernel<true>), dim3((unsigned)cdiv(ht * W, 256)), dim3(256), 0, s, em, (int64_t)0, (int64_t)ht * W);
    const int64_t below = (H - ht - H_owned) * W;
    if (below > 0) hipLaunchKernelGGL((emit_range_kernel<true>), dim3((unsigned)cdiv(below, 256)), dim3(256), 0, s, em, (ht + H_owned) * W, below);
    MH_HIP(hipGetLastError());
    MH_HIP(stream_sync(s));      // the table and the accumulators go back to the pool
    return MHIP_OK;
}

int label_stats64_dev(const double *d_data, const int32_t *d_labels, int64_t n, int64_t nlab, mhip_stat_record *d_rec, hipStream_t s)
{
    const int64_t nrec = nlab + 1;
    DevBuf mn, mx, sm, ct, bad;
    MH_TRY(mn.alloc(8 * (size_t)nrec));
    MH_TRY(mx.alloc(8 * (size_t)nrec));
    MH_TRY(sm.alloc(8 * (size_t)nrec));
    MH_TRY(ct.alloc(8 * (size_t)nrec));
    MH_TRY(bad.alloc(4));
    MH_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    StatAcc64 a{mn.as<uint64_t>(), mx.as<uint64_t>(), sm.as<double>(), ct.as<unsigned long long>()};
    hipLaunchKernelGGL(stats64_init_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec);
    hipLaunchKernelGGL(stats64_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_data, d_labels, n, nlab, a, bad.as<unsigned int>());
    hipLaunchKernelGGL(stats64_finish_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, a, nrec, d_rec);
    MH_HIP(hipGetLastError());
    return check_bad(bad, s, "label_stats");
}

int label_arg_dev(const double *d_data, const int32_t *d_labels, int64_t H, int64_t W, int64_t nlab, bool is_max,
                  mhip_index_record *d_rec, hipStream_t s, bool components)
{
    const int64_t nrec = nlab + 1, n = H * W;
    DevBuf key, idx, bad;
    MH_TRY(key.alloc(8 * (size_t)nrec));
    MH_TRY(idx.alloc(8 * (size_t)nrec));
    MH_TRY(bad.alloc(4));
    MH_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    const unsigned gr = (unsigned)cdiv(nrec, 256), gn = (unsigned)cdiv(n, 256);
    if (is_max && n < 0xffffffffll) {
        // integer-valued data (accumulated flow): value and position in one key, one pass; anything else falls through
        DevBuf ni;
        MH_TRY(ni.alloc(4));
        MH_HIP(hipMemsetAsync(ni.p, 0, 4, s));
        MH_HIP(hipMemsetAsync(key.p, 0, 8 * (size_t)nrec, s));
        const TileGeom g = tile_geom(n, W);
        if (components)
            hipLaunchKernelGGL(arg_packed_kernel<true>, dim3(tile_grid(g)), dim3(256), 0, s, d_data, d_labels, g, nlab, key.as<uint64_t>(),
                               bad.as<unsigned int>(), ni.as<unsigned int>());
        else
            hipLaunchKernelGGL(arg_packed_kernel<false>, dim3(tile_grid(g)), dim3(256), 0, s, d_data, d_labels, g, nlab, key.as<uint64_t>(),
                               bad.as<unsigned int>(), ni.as<unsigned int>());
        hipLaunchKernelGGL(arg_packed_finish_kernel, dim3(gr), dim3(256), 0, s, d_data, key.as<uint64_t>(), nrec, W, d_rec);
        MH_HIP(hipGetLastError());
        unsigned int h_ni = 0;
        MH_HIP(hipMemcpyAsync(&h_ni, ni.p, 4, hipMemcpyDeviceToHost, s));
        MH_TRY(check_bad(bad, s, "label_max_index"));   // synchronises the stream
        if (!h_ni) return MHIP_OK;
    }
    hipLaunchKernelGGL(arg_init_kernel, dim3(gr), dim3(256), 0, s, key.as<uint64_t>(), idx.as<unsigned long long>(), nrec, is_max);
    hipLaunchKernelGGL(arg_pass1_kernel, dim3(stride_grid(n)), dim3(256), 0, s, d_data, d_labels, n, nlab, key.as<uint64_t>(), is_max,
                       bad.as<unsigned int>());
    hipLaunchKernelGGL(arg_pass2_kernel, dim3(gn), dim3(256), 0, s, d_data, d_labels, n, nlab, key.as<uint64_t>(),
                       idx.as<unsigned long long>(), is_max);
    hipLaunchKernelGGL(arg_finish_kernel, dim3(gr), dim3(256), 0, s, d_data, idx.as<unsigned long long>(), nrec, W, is_max,
                       d_rec);
    MH_HIP(hipGetLastError());
    return check_bad(bad, s, "label_min/max_index");
}

int label_count_dev(const int32_t *d_labels, int64_t n, int64_t nlab, int64_t *d_counts, hipStream_t s, int64_t W)
{
    DevBuf bad;
    MH_TRY(bad.alloc(4));
    MH_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    MH_HIP(hipMemsetAsync(d_counts, 0, 8 * (size_t)(nlab + 1), s));
    const TileGeom g = tile_geom(n, W);
    hipLaunchKernelGGL(count_kernel, dim3(tile_grid(g)), dim3(256), 0, s, d_labels, g, nlab,
                       reinterpret_cast<unsigned long long *>(d_counts), bad.as<unsigned int>());
    MH_HIP(hipGetLastError());
    return check_bad(bad, s, "label_count");
}

int label_max_dev(const int32_t *d_labels, int64_t n, int32_t *out_max, hipStream_t s)
{
    DevBuf m;
    MH_TRY(m.alloc(4));
    const int init = INT32_MIN;
    MH_HIP(hipMemcpyAsync(m.p, &init, 4, hipMemcpyHostToDevice, s));
    const unsigned grid = (unsigned)(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024);
    hipLaunchKernelGGL(max_kernel, dim3(grid), dim3(256), 0, s, d_labels, n, m.as<int>());
    MH_HIP(hipGetLastError());
    MH_HIP(hipMemcpyAsync(out_max, m.p, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(stream_sync(s));
    return MHIP_OK;
}

int relabel_lut_dev(int32_t *d_labels, const int32_t *d_lut, int64_t nlab, int64_t n, hipStream_t s)
{
    DevBuf bad;
    MH_TRY(bad.alloc(4));
    MH_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    hipLaunchKernelGGL(lut_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_labels, d_lut, nlab, n,
                       bad.as<unsigned int>());
    MH_HIP(hipGetLastError());
    return check_bad(bad, s, "relabel_keep");
}

int relabel_sparse_dev(int32_t *d_labels, int64_t n, int64_t nlocal, int32_t offset, const int32_t *d_dropped, const int32_t *d_target,
                       int32_t ndropped, hipStream_t s)
{
    DevBuf bad;
    MH_TRY(bad.alloc(4));
    MH_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    hipLaunchKernelGGL(relabel_sparse_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_labels, n, nlocal, offset, d_dropped, d_target,
                       ndropped, bad.as<unsigned int>());
    MH_HIP(hipGetLastError());
    return check_bad(bad, s, "band_relabel_sparse");
}

int relabel_range_dev(int32_t *d_labels, int64_t n, int32_t lo, int32_t hi, const int32_t *d_lut, const int32_t *d_fid, const int32_t *d_fnew, int32_t nf,
                      hipStream_t s)
{
    hipLaunchKernelGGL(relabel_range_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_labels, n, lo, hi, d_lut, d_fid, d_fnew, nf);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

int keep_mask_dev(const int32_t *d_labels, const uint8_t *d_keep, int64_t nlab, int64_t n, uint8_t *d_mask, hipStream_t s)
{
    DevBuf bad;
    MH_TRY(bad.alloc(4));
    MH_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_labels, d_keep, nlab, n, d_mask,
                       bad.as<unsigned int>());
    MH_HIP(hipGetLastError());
    return check_bad(bad, s, "keep_labels");
}

}  // namespace mh
