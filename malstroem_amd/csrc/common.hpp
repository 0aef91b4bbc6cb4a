// common.hpp -- shared host/device helpers of libmalstroem_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/malstroem_hip.h"

namespace mh {

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char *fmt, ...);
const char *get_error();
// Development knobs and test hooks (engine selection for A/B runs, debug prints, fault injection for the liveness tests of the
// run-time checks) are read ONLY when MHIP_DEVELOPER=1 is set: a stray variable in a user's environment never steers the product.
const char *dev_env(const char *name);
// hipStreamSynchronize with a short busy wait first: the round loops of the fills read a few words back every 16-32 launches and
// the device idles while a sleeping host thread is woken (30-50 us per read-back; MALSTROEM_HIP_SPIN_US=0 turns the busy wait off)
hipError_t stream_sync(hipStream_t s);

#define MH_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            mh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MHIP_EHIP;                                                                    \
        }                                                                                        \
    } while (0)

#define MH_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != MHIP_OK) return _rc; \
    } while (0)

#define MH_ARG(cond, msg)                   \
    do {                                    \
        if (!(cond)) {                      \
            mh::set_error("invalid argument: %s", msg); \
            return MHIP_EINVAL;             \
        }                                   \
    } while (0)

// ---- device scratch buffer (RAII, stream-ordered free is not needed: all work is on one stream and
//      we synchronise before releasing) ------------------------------------------------------------
// Blocks come from a small caching pool (api.hip) so that the iterative stages and repeated pipeline runs do
// not pay hipMalloc/hipFree (the latter synchronises the device) inside the hot path.
int pool_alloc(void **p, size_t bytes);
void pool_free(void *p, size_t bytes);

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n)
    {
        if (n == 0) n = 16;
        if (p && bytes >= n && bytes <= 2 * n + 4096) return MHIP_OK;  // reuse what we already hold
        release();
        MH_TRY(pool_alloc(&p, n));
        bytes = n;
        return MHIP_OK;
    }
    void release()
    {
        if (p) pool_free(p, bytes);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- AGNPS direction codes (reference flow.py:30-38, deltas _flow.pyx:36-46) -------------------
// code k: U 0, UR 1, R 2, DR 3, D 4, DL 5, L 6, UL 7, NODIR 8
__host__ __device__ inline int dir_dr(int k) { return (k == 0 || k == 1 || k == 7) ? -1 : (k >= 3 && k <= 5) ? 1 : 0; }
__host__ __device__ inline int dir_dc(int k) { return (k >= 1 && k <= 3) ? 1 : (k >= 5 && k <= 7) ? -1 : 0; }

// ---- D8 of one cell (d8.hip; noflat_geo.hip: ng_finish_kernel) ------------------------------------------------------------
__device__ __forceinline__ unsigned d8_code(double z, double u, double ur, double r, double dr, double d, double dl,
                                            double l, double ul)
{
    const double INV_SQRT2 = 0.7071067811865475;  // 1 / 2**0.5, _flow.pyx:93-94
    // `if dz > dzmax: dzmax = dz; i = k` in the reference's order.  The running maximum is a v_max_f64 (same value as the
    // conditional move: it only changes when dz > dzmax; NaN drops are ignored by both), the index a 32-bit select.
    unsigned i = 8;
    double dzmax = 0.0, dz;
#define MH_D8_STEP(expr, k)          \
    dz = (expr);                     \
    i = dz > dzmax ? (k) : i;        \
    dzmax = fmax(dzmax, dz);
    MH_D8_STEP(__dsub_rn(z, u), 0u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, ur), INV_SQRT2), 1u)
    MH_D8_STEP(__dsub_rn(z, r), 2u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, dr), INV_SQRT2), 3u)
    MH_D8_STEP(__dsub_rn(z, d), 4u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, dl), INV_SQRT2), 5u)
    MH_D8_STEP(__dsub_rn(z, l), 6u)
    MH_D8_STEP(__dmul_rn(__dsub_rn(z, ul), INV_SQRT2), 7u)
#undef MH_D8_STEP
    return i;
}

// flow.py:130-139: rows first, then columns overwrite, corners last (same order => same result on 1-wide rasters)
__device__ __forceinline__ unsigned edge_code(int64_t r, int64_t c, int64_t maxr, int64_t maxc)
{
    unsigned code = 8;
    if (r == 0) code = 0;
    if (r == maxr) code = 4;
    if (c == 0) code = 6;
    if (c == maxc) code = 2;
    if (r == 0 && c == 0) code = 7;
    if (r == 0 && c == maxc) code = 1;
    if (r == maxr && c == 0) code = 5;
    if (r == maxr && c == maxc) code = 3;
    return code;
}

// monotone float/double <-> unsigned keys (for atomicMin/Max); NaN must be filtered by the caller
__host__ __device__ inline uint32_t f32_key(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float key_f32(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline uint64_t f64_key(double f)
{
    uint64_t u;
    memcpy(&u, &f, 8);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double key_f64(uint64_t k)
{
    uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double f;
    memcpy(&f, &u, 8);
    return f;
}

// ---- device-pointer stage implementations (one translation unit per stage) --------------------
struct FillStats {
    int32_t rounds = 0;   // kernel launches (tile rounds)
    int64_t visits = 0;   // tile visits over all rounds
    int64_t cycles = 0;   // local (down, up, right, left) cycles over all visits
    int64_t tiles = 0;    // tiles in the raster
    float hot_ms = 0;     // device time of the stage's dominant kernel, HIP events around its launches: pf_tile_kernel (one launch) /
    int32_t hot_launches = 0;   // the ng_round_kernel launches (the span of the round loop: compaction launches and gaps included)
    int32_t algorithm = 0;  // 0: iterative tile schedule (fill.hip), 1: tiled priority-flood (pflood.hip; rounds = kernel launches),
                            // 2: integer geodesic transform (noflat_geo.hip)
    // the priority-flood reads every DEM cell anyway: smallest / largest elevation of the local raster and "holds a NaN" ride along
    bool have_minmax = false, dem_nan = false;
    float dem_min = 0.0f, dem_max = 0.0f;
    // why the integer geodesic transform handed a raster back to the float64 relaxation (diagnostics: ctx_get_int "noflat_reject*"):
    // 0 it did not, 1 not applicable (irregular levels / NaN cells / epsilons without weights), 2 its final check failed
    int32_t geo_reject = 0;
    int64_t geo_irregular = 0, geo_unreached = 0, geo_mismatch = 0;
};

// fill.hip
// Resumable fill run (row-band mode interleaves halo refreshes between batches of rounds).
struct FillRun {
    bool noflat = false;
    const float *dem = nullptr;
    void *out = nullptr;            // float* (plain) or double* (no-flats)
    int64_t H = 0, W = 0;           // local raster (band + halo rows)
    double sh = 0, dg = 0;
    const float *seed = nullptr;    // plain-filled surface: no-flats upper-bound start (see fill_noflat_dev)
    double seed_add = 0;
    int fixed_top = 0, fixed_bot = 0;  // local row 0 / H-1 is a halo row owned by the neighbouring band
    int rounds_per_batch = 8;          // rounds launched between two host checks (a band exchanges its halo rows after each batch)
    bool attach_only = false;          // (internal: attach() is begin() without the initialising round)
    struct Impl;
    Impl *impl;
    FillRun();
    ~FillRun();
    FillRun(const FillRun &) = delete;
    FillRun &operator=(const FillRun &) = delete;
    int begin(hipStream_t s, bool *active);      // initialising round over every tile
    int batch(hipStream_t s, bool *active);      // a batch of rounds; *active == false: locally converged
    int activate_row(int side, hipStream_t s);   // halo row `side` (0 top, 1 bottom) changed: revisit its tile row
    int attach(hipStream_t s);                   // instead of begin(): `out` already holds an upper bound of the fixed point
    int certify(hipStream_t s, bool *changed);   // one sweep over EVERY tile, iterated to local convergence; *changed: a tile moved
    int finish(hipStream_t s, FillStats *st);
};
void noflat_seed(FillRun &f, const float *d_filled, double sh, double dg, int64_t ncells_global);
int fill_plain_dev(const float *d_dem, float *d_out, int64_t H, int64_t W, hipStream_t s, FillStats *st, float *d_depths = nullptr,
                   bool *depths_done = nullptr);
// check.hip: *d_flag (zeroed by the caller) = 1 when `filled` is not a fixed point of the plain fill with border == dem
int fill_check_f32_dev(const float *d_dem, const float *d_filled, int64_t H, int64_t W, int fixed_top, int fixed_bot, hipStream_t s,
                       unsigned int *d_flag);
// pflood.hip
int fill_plain_pflood_dev(const float *d_dem, float *d_out, float *d_depths, int64_t H, int64_t W, hipStream_t s, FillStats *st, bool *violated = nullptr);
// pflood.hip: the exact tiled priority-flood, resumable for row bands (whose halo rows of `out` carry the neighbours' current
// estimates of their filled edge rows; this band's own edge rows in `out` are kept current for them)
struct PfRun {
    const float *dem = nullptr;
    float *out = nullptr;            // the filled surface
    int64_t H = 0, W = 0;
    int fixed_top = 0, fixed_bot = 0;
    struct Impl;
    Impl *impl;
    PfRun();
    ~PfRun();
    PfRun(const PfRun &) = delete;
    PfRun &operator=(const PfRun &) = delete;
    int begin(hipStream_t s);                     // MHIP_ELIMIT: not applicable (capacity, band alignment): run the iterative schedule
    int halo_changed(int side, hipStream_t s);
    int batch(hipStream_t s);                     // MHIP_ELIMIT as above
    // writes the raster, then proves it (check.hip); *violated: the surface is not the fixed point (it still is an upper bound of
    // it): the caller lets the iterative schedule finish the job (FillRun::attach + certify)
    int finish(hipStream_t s, float *d_depths, FillStats *st, bool *violated = nullptr);
    int solve(hipStream_t s);
    int pack(hipStream_t s, int row0, int nrows);
    int publish_edges(hipStream_t s);
};
// a call-back fired once, on the stage's stream, at a chosen point of a stage (mhip_ctx_run starts the label branch when the
// no-flats fill has left its throughput-bound first rounds: see api.hip)
struct StageHook {
    void (*fn)(void *, hipStream_t) = nullptr;
    void *arg = nullptr;
    bool fired = false;
    void fire(hipStream_t s)
    {
        if (fn && !fired) {
            fired = true;
            fn(arg, s);
        }
    }
};

// noflat_geo.hip: the no-flats fill as an integer geodesic distance transform; resumable for row bands
struct GeoRun {
    const float *dem = nullptr, *filled = nullptr;   // local raster incl. halo rows; `filled` = the converged plain fill
    double *out = nullptr;                            // the no-flats surface (written by end())
    uint32_t *dist = nullptr;                         // distances, H * W (nullptr: kept in the run's own workspace)
    int64_t H = 0, W = 0;
    double sh = 0, dg = 0;
    int fixed_top = 0, fixed_bot = 0;
    bool allow_partial = false;                       // flats of irregular levels: leave them to the caller (else: not applicable)
    double seed_add = 0;                              // ... with the upper bound F + seed_add in `out`
    bool partial = false;                             // set by end(): `out` still needs the float64 relaxation on those flats
    StageHook *tail_hook = nullptr;                   // fired when the first batch of rounds has been launched
    uint8_t *d8_out = nullptr;                        // optional: end() also writes the flow directions of the surface (one context, no halo rows)
    unsigned int *d8_nodir = nullptr;                 // ... and sets this word when an interior cell has none
    bool d8_done = false;                             // set by end(): d8_out holds the directions of the verified surface
    struct Impl;
    Impl *impl;
    GeoRun();
    ~GeoRun();
    GeoRun(const GeoRun &) = delete;
    GeoRun &operator=(const GeoRun &) = delete;
    int begin(hipStream_t s, bool *applicable, bool *active);
    int batch(hipStream_t s, bool *active);
    int halo_changed(int side, hipStream_t s);
    int end(hipStream_t s, bool *ok, FillStats *st);
    int launch_rounds(hipStream_t s, int nb);
};
// noflat_geo.hip: MHIP_ELIMIT = not applicable, run the float64 relaxation
// D8Sink: the caller wants the flow directions of the surface too; when the geodesic transform's finishing pass could write them
// (`done`) the D8 pass over the surface is not needed
struct D8Sink {
    uint8_t *flowdir = nullptr;
    unsigned int *nodir = nullptr;     // zeroed by the caller; set when an interior cell has no downslope neighbour
    bool done = false;
};
int fill_noflat_geodesic_dev(const float *d_dem, const float *d_filled, double *d_out, int64_t H, int64_t W, double sh, double dg, double seed_add,
                             hipStream_t s, FillStats *st, bool *partial, StageHook *tail_hook = nullptr, D8Sink *d8 = nullptr);
int noflat_verify_dev(const float *d_dem, const double *d_out, int64_t H, int64_t W, double sh, double dg, hipStream_t s, bool *ok, int fixed_top = 0,
                      int fixed_bot = 0);
int fill_noflat_dev(const float *d_dem, double *d_out, int64_t H, int64_t W, double sh, double dg, hipStream_t s,
                    FillStats *st = nullptr, const float *d_filled = nullptr, StageHook *tail_hook = nullptr, D8Sink *d8 = nullptr);
int short_diag_dev(const float *d_dem, int64_t n, double *sh, double *dg, hipStream_t s);
int depths_dev(const float *d_filled, const float *d_dem, float *d_out, int64_t n, hipStream_t s);
// d8.hip
int d8_dev(const double *d_z, uint8_t *d_out, int64_t H, int64_t W, int edges_outward, hipStream_t s, int64_t row_off = 0,
           int64_t Hg = 0, unsigned int *d_interior_nodir = nullptr);   // d_interior_nodir (edges_outward only): left != 0 when an interior cell got NODIR (pre-zeroed by the caller)
void short_diag_from_minmax(float mn, float mx, bool has_nan, double *sh, double *dg);   // fill.py:235-250
int minmax_dev(const float *d_x, int64_t n, float *mn, float *mx, int *has_nan, hipStream_t s);
int row_update_dev(void *d_dst, const void *d_src, int64_t nbytes, int *changed, hipStream_t s);
int copy_bandwidth_dev(size_t bytes, int reps, double *gbs, hipStream_t s);
int read_bandwidth_dev(size_t bytes, int reps, double *gbs, hipStream_t s);
int row_update_async(void *d_dst, const void *d_src, int64_t nbytes, int *d_changed, hipStream_t s);
// trace.hip
int trace_downstream_dev(const uint8_t *d_fd, const int32_t *d_lab, int64_t H, int64_t W, const int64_t *d_cells, int64_t n, int use_bg,
                         int32_t bg, int32_t *d_label, int32_t *d_found, int64_t *d_len, const int64_t *d_offsets, int64_t *d_out_cells,
                         hipStream_t s);
// comm.hip (RCCL, opened at run time)
int comm_unique_id(void *id128);
int comm_available();
int comm_create(void **comm, const void *id128, int rank, int nranks);
void comm_destroy(void *comm);
int comm_exchange_rows(void *comm, int rank, int nranks, const void *first_row, const void *last_row, void *stage, size_t rowbytes,
                       hipStream_t s);
int comm_allreduce_max(void *comm, double *d_value, hipStream_t s);
// Pour points out of the accumulation's final pass (bluespots.py:195-206: the cell of the largest accumulated flow of every label,
// the first in raster order among equals).  Accumulated flow grows strictly along a flow path, so the largest value of a label
// sits on a cell whose downstream cell is NOT of that label -- a CANDIDATE: 3 % of the labelled cells, a fifth of the unlabelled
// ones.  The watersheds' tile pass (labels + flow directions of a 64 x 64 tile in one place) lists the candidates; the
// accumulation's final pass (the final sums of the same tile in LDS) turns them into one packed key per label,
// (value << 32) | (0xffffffff - cell), by atomicMax -- and the pass over accumulation + labels that the pour points were
// (12.6 B/cell at the end of a step) goes.  Exact unless a cell stays unresolved (a flow cycle: the value 0 does not grow along
// the path) or the list overflows: both raise a flag and the caller runs the general pass.
struct PourCandDev {             // device side, tile = 64 x 64 cells as in accum.hip / watershed.hip, thread t <-> row t >> 2, columns (t & 3) * 16 ..
    uint16_t *mask0 = nullptr;   // [ntiles * 256] unlabelled candidates: bit k of word t = column (t & 3) * 16 + k of row t >> 2
    uint2 *list = nullptr;       // [ntiles * POUR_TILE_CAP] labelled candidates: (cell, label); a tile's entries start at tile * POUR_TILE_CAP
    uint32_t *tile_cnt = nullptr;          // [ntiles]   (no global counters: 65 536 atomics on one address cost a millisecond)
    unsigned long long *tile_key0 = nullptr;   // [ntiles] the best key among the tile's unlabelled candidates (label 0: reduced by pour_finish_dev)
    uint32_t *flags = nullptr;   // [0] a tile's list overflowed, [1] an unresolved cell
    int components = 0;          // the labels are 8-connected components: two neighbouring labelled cells share their label
    unsigned long long *key = nullptr;   // [nlabels + 1], zeroed
    uint32_t nlab = 0;           // a candidate with a label outside [0, nlab] (uploaded labels) raises flags[1]: the general pass reports it
};
constexpr uint32_t POUR_TILE_CAP = 512;
struct PourLink {                // host side: hand-over between the thread of the watersheds and the thread of the accumulation
    PourCandDev dev;
    hipEvent_t ev = nullptr;                       // recorded behind the tile pass that wrote the candidates
    void (*notify)(void *, int) = nullptr;         // watersheds_dev: candidates recorded (1) / there will be none (0)
    int (*wait)(void *) = nullptr;                 // accum_dev, before its final pass: blocks; 1: the candidates exist (wait for `ev`)
    void *arg = nullptr;
    bool consumed = false;                         // the final pass has written the keys
};
// accum.hip
struct AccumKeep {               // the perimeter graph of a row band's boundary pass, kept for the delta pass that follows it
    DevBuf nodes;
    size_t o[9] = {};
    int64_t H = 0, W = 0;
    int fixed_top = 0, fixed_bot = 0;
    bool valid = false;
};
int accum_dev(const uint8_t *d_fd, double *d_out, int64_t H, int64_t W, hipStream_t s, int fixed_top = 0, int fixed_bot = 0, int halo_zero = 0,
              int32_t *d_exit_map = nullptr, PourLink *pour = nullptr, AccumKeep *keep = nullptr);
int accum_band_delta_dev(const uint8_t *d_fd, double *d_out, int64_t H, int64_t W, hipStream_t s, int fixed_top, int fixed_bot, AccumKeep *keep, bool *done);
// ccl.hip   (d_tmp: H*W int32 scratch)
// stats_out: the labelling's last pass also reduces label_stats(d_data, labels) into *stats_out (allocated here: nlabels + 1 records)
int ccl8_f32_dev(const float *d_data, int32_t *d_labels, int32_t *d_tmp, int64_t H, int64_t W, int64_t *nlabels,
                 hipStream_t s, DevBuf *stats_out = nullptr);
int label_emit_stats_dev(const int32_t *d_parent, const unsigned long long *d_rootbits, const uint32_t *d_wordprefix, const float *d_data,
                         int32_t *d_labels, int64_t H, int64_t W, int64_t nlab, mhip_stat_record *d_rec, hipStream_t s);
int ccl8_u8_dev(const uint8_t *d_data, int32_t *d_labels, int32_t *d_tmp, int64_t H, int64_t W, int64_t *nlabels,
                hipStream_t s);
// a row band labels in two halves: ccl8_f32_begin_dev stops before the emit pass -- the labels raster gets its two top and two bottom
// rows only (band-local ranks: what the seam merge on the host compares), parent (d_tmp) / root bits / word prefixes stay in `keep`
// -- and label_emit_sparse_dev writes the GLOBAL labels once the merge has numbered them (label_ops.hip)
struct CclKeep {
    DevBuf bits, wprefix;
    int64_t H = 0, W = 0, nlocal = 0;
    bool valid = false;       // false: the labels raster holds band-local labels everywhere (a schedule without the kept tables)
};
int ccl8_f32_begin_dev(const float *d_data, int32_t *d_labels, int32_t *d_tmp, int64_t H, int64_t W, int64_t *nlabels, hipStream_t s, CclKeep *keep);
int ccl_emit_rows_dev(const int32_t *d_parent, const unsigned long long *d_rootbits, const uint32_t *d_wordprefix, int32_t *d_labels, int64_t base,
                      int64_t count, hipStream_t s);
int label_emit_sparse_dev(const int32_t *d_parent, const unsigned long long *d_rootbits, const uint32_t *d_wordprefix, const float *d_data,
                          int32_t *d_labels, int64_t H, int64_t W, int ht, int64_t H_owned, int64_t nlocal, int32_t offset,
                          const int32_t *d_dropped, const int32_t *d_target, int32_t ndropped, int64_t nlab_global, mhip_stat_record *d_rec,
                          hipStream_t s);
// label_ops.hip
int relabel_lut_dev(int32_t *d_labels, const int32_t *d_lut, int64_t nlab, int64_t n, hipStream_t s);
int relabel_range_dev(int32_t *d_labels, int64_t n, int32_t lo, int32_t hi, const int32_t *d_lut, const int32_t *d_fid, const int32_t *d_fnew, int32_t nf,
                      hipStream_t s);
int band_trace_dev(const uint8_t *d_fd, const int32_t *d_lab, int64_t Hl, int64_t W, int64_t row_lo, int64_t own0, int64_t own1, int64_t Hg,
                   const int64_t *d_cells, const int32_t *d_src, int64_t n, int use_bg, int32_t bg, int32_t *d_label, int32_t *d_status,
                   int32_t *d_src_out, int64_t *d_exit, int64_t *d_len, const int64_t *d_offsets, int64_t *d_out_cells, hipStream_t s);
int relabel_sparse_dev(int32_t *d_labels, int64_t n, int64_t nlocal, int32_t offset, const int32_t *d_dropped, const int32_t *d_target,
                       int32_t ndropped, hipStream_t s);
int keep_mask_dev(const int32_t *d_labels, const uint8_t *d_keep, int64_t nlab, int64_t n, uint8_t *d_mask,
                  hipStream_t s);
// W: raster width if the caller knows it (2-D tiles combine a label's rows before the global atomics), 0 = flat array
// components: the labels are the 8-connected components of a raster of which these n cells (rows of W) are a row range
int label_stats_dev(const float *d_data, const int32_t *d_labels, int64_t n, int64_t nlab, mhip_stat_record *d_rec,
                    hipStream_t s, int64_t W = 0, bool components = false);
int label_stats64_dev(const double *d_data, const int32_t *d_labels, int64_t n, int64_t nlab, mhip_stat_record *d_rec, hipStream_t s);
int label_arg_dev(const double *d_data, const int32_t *d_labels, int64_t H, int64_t W, int64_t nlab, bool is_max,
                  mhip_index_record *d_rec, hipStream_t s, bool components = false);
int label_count_dev(const int32_t *d_labels, int64_t n, int64_t nlab, int64_t *d_counts, hipStream_t s, int64_t W = 0);
int label_max_dev(const int32_t *d_labels, int64_t n, int32_t *out_max, hipStream_t s);
// watershed.hip
int watersheds_dev(const uint8_t *d_fd, int32_t *d_labels, int64_t H, int64_t W, int32_t unassigned, hipStream_t s,
                   bool band_mode = false, const unsigned int *d_known_interior_nodir = nullptr, const int32_t *d_src = nullptr, PourLink *pour = nullptr);
int pour_finish_dev(unsigned long long *d_key, const unsigned long long *d_tile_key0, int64_t ntiles, int64_t nlab, int64_t W, mhip_index_record *d_rec,
                    hipStream_t s);
int band_pseudo_labels_dev(int32_t *d_ws, int64_t H, int64_t W, int top, int bottom, hipStream_t s);
int negative_lut_dev(int32_t *d_lab, int64_t n, const int32_t *d_lut, int64_t nlut, hipStream_t s);

__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// "Does any thread of the workgroup say yes?" with ONE barrier per call.  The library's __syncthreads_or is a reduction through 256
// bytes of LDS of its own with two barriers; the loops that end on such a vote (pointer doubling, label-correcting sweeps) are
// bound by exactly these barriers.  Three LDS words take turns: call n raises w[n % 3] in front of its barrier and reads it behind;
// thread 0 clears the word of call n + 2 there (its last readers left before call n's barrier, its next writers come behind call
// n + 1's).  The caller zeroes the three words in front of a barrier; every call must be reached by all threads of the workgroup.
struct WgVote {
    int *w;
    int turn;
    __device__ __forceinline__ explicit WgVote(int *lds3) : w(lds3), turn(0) {}
    __device__ __forceinline__ bool any(bool p)
    {
        if (__any(p) && (threadIdx.x & 63u) == 0u) w[turn] = 1;
        __syncthreads();
        const bool r = w[turn] != 0;
        if (threadIdx.x == 0) w[turn == 0 ? 2 : turn - 1] = 0;
        turn = turn == 2 ? 0 : turn + 1;
        return r;
    }
};
#endif

}  // namespace mh
