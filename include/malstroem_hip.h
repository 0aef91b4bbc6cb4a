/*
 * malstroem_hip.h -- C-ABI of libmalstroem_hip.so, the MI355X (gfx950) raster-hydrology core.
 *
 * Drop-in boundary: these entry points are what a malstroem maintainer binds with ctypes in
 * place of the Cython `speedups` modules (reference malstroem/algorithms/speedups/__init__.py:37-77).
 * The reference patches 8 per-sweep / per-cell functions; a GPU needs whole-stage granularity
 * (SURVEY.md 8b), so each entry point below replaces one whole stage of malstroem.algorithms.
 *
 * Conventions
 *   - return 0 on success, <0 on error (MHIP_E*); mhip_last_error() gives the message (thread local).
 *   - rasters are C-contiguous row-major, H rows x W cols; host pointers are borrowed for the call
 *     and never retained; the caller allocates every output.
 *   - dtype contract of reference algorithms/dtypes.py:20-31: DEM/filled float32, no-flats surface
 *     float64, flow direction uint8 (AGNPS codes 0..7, 8 = no direction), accumulation float64,
 *     labels int32 (scipy.ndimage.label output dtype).
 *   - record layouts are the packed NumPy dtypes of reference _label.pyx:22-28.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with MHIP_ENODEV.
 */
#ifndef MALSTROEM_HIP_H
#define MALSTROEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHIP_OK 0
#define MHIP_EINVAL (-1)   /* bad argument (NULL pointer, H/W < 1, label out of range ...) */
#define MHIP_EHIP (-2)     /* HIP runtime error */
#define MHIP_ENODEV (-3)   /* no HIP device visible */
#define MHIP_ELIMIT (-4)   /* raster too large for the int32 label / local-index domain */
#define MHIP_ENOTCONV (-5) /* iteration cap hit (flow cycle through an edge cell, see watersheds) */
#define MHIP_ECOMM (-6)    /* RCCL error (multi-GPU context) */

/* packed record of reference _label.pyx:22-24 == numpy [('min','<f8'),('max','<f8'),('sum','<f8'),('count','<i8')] */
typedef struct { double min, max, sum; int64_t count; } mhip_stat_record;
/* packed record of reference _label.pyx:26-28 == numpy [('value','<f8'),('row','<i8'),('col','<i8')] */
typedef struct { double value; int64_t row, col; } mhip_index_record;

/* ---- library / device ------------------------------------------------------------------------- */
const char *mhip_last_error(void);
const char *mhip_version(void);
int mhip_device_count(void);                 /* number of HIP devices, 0 if none / runtime missing */
int mhip_set_device(int device);             /* device used by the calling thread's later calls */

/* measurement aid: rate of a 16-byte-per-lane device-to-device copy of `bytes` (read + written bytes per second, GB/s),
 * i.e. the achievable share of the HBM spec peak on this device; reported next to the roofline fractions by bench.py */
int mhip_copy_bandwidth(int64_t bytes, int32_t reps, double *gbs);
/* ... and of a read-only stream of 16-byte loads (the ceiling of a kernel that reads much more than it writes) */
int mhip_read_bandwidth(int64_t bytes, int32_t reps, double *gbs);

/* ---- whole-stage entry points on HOST rasters (upload -> kernels -> download) ----------------- */

/* fill.fill_terrain(dtm)  reference fill.py:112-171 (+ sweep _fill.pyx:28-70).
 * Greatest fixed point of W = max(dtm, min(W, 8 nbrs)), border cells fixed to dtm.
 * out_rounds (optional): number of tile rounds the device schedule needed. */
int mhip_fill_f32(const float *dem, float *out_filled, int64_t H, int64_t W, int32_t *out_rounds);

/* fill.fill_terrain_no_flats(dtm, short, diag)  reference fill.py:174-232 (+ _fill.pyx:72-124). */
int mhip_fill_noflat_f64(const float *dem, double *out, int64_t H, int64_t W, double short_, double diag,
                         int32_t *out_rounds);

/* fill.minimum_safe_short_and_diag(dem)  reference fill.py:235-250. */
int mhip_short_diag(const float *dem, int64_t n, double *short_, double *diag);

/* depths = filled - dem  reference dem.py:71. */
int mhip_depths_f32(const float *filled, const float *dem, float *out, int64_t n);

/* flow.terrain_flowdirection(terrain, edges_flow_outward)  reference flow.py:142-167 ->
 * _flow.pyx:98-176 (diagonal drop MULTIPLIED by 1/2**0.5) + flow.py:118-139. float64 input only. */
int mhip_d8_f64(const double *z, uint8_t *out, int64_t H, int64_t W, int edges_outward);

/* flow.accumulated_flow(flowdir)  reference _flow.pyx:225-273 (python flow.py:304-364). */
int mhip_accum(const uint8_t *flowdir, double *out, int64_t H, int64_t W);

/* label.connected_components(data)  reference label.py:19-40 -> scipy.ndimage.label(data, ones((3,3))):
 * 8-connected, foreground = data != 0, labels 1..n ordered by first raster-scan pixel, int32. */
int mhip_ccl8_f32(const float *data, int32_t *labels, int64_t H, int64_t W, int64_t *nlabels);
int mhip_ccl8_u8(const uint8_t *data, int32_t *labels, int64_t H, int64_t W, int64_t *nlabels);

/* label.keep_labels + second connected_components (reference label.py:78-98, bluespots.py:167-170)
 * fused: rank = cumsum(keep)*keep with keep[0] forced false; labels[i] = rank[labels[i]] in place.
 * keep has nlab+1 entries. nkept receives the number of surviving labels. */
int mhip_relabel_keep(int32_t *labels, const uint8_t *keep, int64_t nlab, int64_t n, int64_t *nkept);
/* label.keep_labels alone: mask[i] = keep[labels[i]]; the caller clears keep[background] first (label.py:91). */
int mhip_keep_mask(const int32_t *labels, const uint8_t *keep, int64_t nlab, int64_t n, uint8_t *mask);

/* label.label_stats(data, labelled, nlabels)  reference _label.pyx:68-97 (python label.py:43-75).
 * records has nlab+1 entries (label 0 included). */
int mhip_label_stats_f32(const float *data, const int32_t *labels, int64_t n, int64_t nlab,
                         mhip_stat_record *records);

/* the same on float64 data: the reference's generic path (label.py:43-75) for rasters that are not float32
 * (integer rasters are converted to float64 by the caller, as the reference's float64 record fields do). */
int mhip_label_stats_f64(const double *data, const int32_t *labels, int64_t n, int64_t nlab,
                         mhip_stat_record *records);

/* label.label_min_index / label.label_max_index  reference _label.pyx:99-128, label.py:135-166:
 * per label extreme value and its FIRST raster-order position (strict compare). nlab+1 records. */
int mhip_label_argmin_f64(const double *data, const int32_t *labels, int64_t H, int64_t W, int64_t nlab,
                          mhip_index_record *records);
int mhip_label_argmax_f64(const double *data, const int32_t *labels, int64_t H, int64_t W, int64_t nlab,
                          mhip_index_record *records);

/* label.label_count(labelled) == np.bincount  reference label.py:169-180. counts has nlab+1 entries. */
int mhip_label_count(const int32_t *labels, int64_t n, int64_t nlab, int64_t *counts);
/* max(labelled) helper (the reference computes nlabels = np.max(labelled) when not given). */
int mhip_label_max(const int32_t *labels, int64_t n, int32_t *out_max);

/* flow.watersheds_from_labels(flowdir, labelled, unassigned)  reference flow.py:398-412 ->
 * _flow.pyx:276-403. In place on labels. Terminates on flow cycles (reference does not). */
int mhip_watersheds_i32(const uint8_t *flowdir, int32_t *labels_inout, int64_t H, int64_t W, int32_t unassigned);

/* net.next_downstream_label(flowdir, labeled, cell, background_label, geometry) for a BATCH of cells -- what
 * net.pourpoint_network / geometric_pourpoint_network loop over (reference net.py:142-192, 195-224).  cells_rc: n (row, col)
 * pairs.  Per cell: out_found = 1 and out_label = the first label on the downstream walk that differs from the start cell's
 * (and from `background` when use_background), else out_found = 0 (the reference's None); out_len = number of cells walked,
 * start and end included (the reference's geometry).  Geometry in a second call: offsets = n + 1 prefix sums of out_len,
 * out_cells receives the walked cells as linear indices row * W + col.  Any output pointer may be NULL.  A walk caught in
 * a flow cycle (the reference never returns from it) is cut after H * W steps and reports "none". */
int mhip_trace_downstream_i32(const uint8_t *flowdir, const int32_t *labels, int64_t H, int64_t W, const int64_t *cells_rc, int64_t n,
                              int use_background, int32_t background, int32_t *out_label, int32_t *out_found, int64_t *out_len,
                              const int64_t *offsets, int64_t *out_cells);

/* network.Network.rain_event for several rain events at once (reference network.py:75-129, rain.py:48-87): leaf-to-root fill
 * and spill over the node forest.  Host code (O(nodes) per event), no device needed.  down_index[i]: position of node i's
 * downstream node, -1 = root (downstream id None), -2 = downstream id unknown (the node is never evaluated, like in the
 * reference).  Outputs are [nevents][n]; NaN marks "not evaluated" and, in pctv, the reference's None (bspot_vol == 0).
 * order[0 .. *ncomputed): the nodes in the reference's evaluation (= result list) order. */
int mhip_rain_events(int64_t n, const int64_t *down_index, const double *wshed_area, const double *bspot_vol, int32_t nevents,
                     const double *mmrain, double *rainv, double *spillv, double *v, double *pctv, int64_t *order, int64_t *ncomputed);

/* The two boundary systems of the row-band protocol (malstroem_amd/distributed.py; no reference counterpart: the reference works
 * on one undivided raster).  Host code, no device needed; every rank solves them on the gathered seam rows of all bands.
 * band_forest_solve: flow accumulation over the forest of seam crossings.  val[i] > 0 = node i's own (known) contribution,
 *   parent[i] = the node its flux continues in or -1.  On return val[i] = own + everything upstream for every node whose upstream
 *   is completely known, 0 for every other node (_flow.pyx:212-247 leaves cells behind a flow cycle 0 as well).
 * band_ws_resolve: vals[i] >= 0 is a watershed label (0 = none), vals[i] < 0 means "whatever node -vals[i] - 1 resolves to";
 *   chains are followed to their end, a cycle resolves to 0 (_flow.pyx:276-314 leaves such cells unassigned). */
int mhip_band_forest_solve(int64_t n, const int64_t *parent, double *val);
int mhip_band_ws_resolve(int64_t n, int64_t *vals);
/* classes of an undirected graph (n nodes, m edges a[k] -- b[k]): cls[i] = smallest node of i's class (the label merge across
 * band seams works on the seam pairs the bands publish) */
int mhip_band_union_find(int64_t n, int64_t m, const int64_t *a, const int64_t *b, int64_t *cls);
/* The host sections of the band protocol between its exchanges, one or two passes each over the seam rows (2 * W cells per band)
 * and the gathered pairs (malstroem_amd/distributed.py: BandPipeline.accum / label / watershed describe the protocol; node of
 * (band, side, column) = (2 * band + side) * W + column, label keys = (band << 32) | local label).
 * band_accum_pairs: one pair {int32 child, int32 parent, double own_child, double own_parent} (24 bytes) per cell of a neighbour's
 *   edge row whose flux crosses this band and leaves it again.
 * band_accum_solve: the forest of the gathered pairs solved as in band_forest_solve; the neighbours' edge rows top / bot updated.
 * band_label_pairs: one (mine, theirs) pair per run along a halo row, and the row's phantoms (components without an owned cell).
 * band_label_merge: classes of the gathered pairs -> offsets[R + 1] of every band's numbering, band `me`'s dropped labels and
 *   their global labels, and the global labels that live in more than one band.
 * band_ws_publish:  the edge cells of this band on watershed paths that are still open after looking at the neighbours' rows.
 * band_ws_lut:      gathered chains followed to their ends; lut[2 * W] = label behind each halo cell of this band. */
int mhip_band_accum_pairs(int64_t W, const int32_t *exit_half, const double *nbr, const double *own_edge, int64_t child_base,
                          int64_t parent_base, void *pairs, int64_t *n);
int mhip_band_accum_solve(int64_t nspace, int64_t m, const void *pairs, int64_t W, int64_t base_top, double *top, int64_t base_bot,
                          double *bot);
int mhip_band_label_pairs(int64_t W, const int32_t *halo, const int32_t *edge, const int32_t *nbr, int64_t key_mine, int64_t key_nbr,
                          int64_t *ea, int64_t *eb, int64_t *npairs, int64_t *ph, int64_t *nph);
int mhip_band_label_merge(int32_t R, int32_t me, const int64_t *nloc, int64_t m, const int64_t *EA, const int64_t *EB, int64_t nph,
                          const int64_t *PH, int64_t *offsets, int32_t *dropped, int32_t *target, int64_t *ndrop, int64_t *shared,
                          int64_t *nshared);
int mhip_band_ws_publish(int64_t W, int32_t me, const int32_t *mine, const int32_t *up, const int32_t *dn, int64_t *N, int64_t *V,
                         int64_t *n);
int mhip_band_ws_lut(int64_t W, int32_t me, int64_t n, const int64_t *N, const int64_t *V, const int32_t *up, const int32_t *dn,
                     int32_t *lut);
/* Rendezvous of the band threads of ONE process (distributed.ThreadComm / HybridComm: k bands per process where a raster is beyond
 * 2**31 cells; the reference has no counterpart -- one process, one raster).  A thread blocks inside the call (ctypes: no GIL),
 * spins a few microseconds and then sleeps; a wait longer than timeout_ms returns MHIP_ECOMM and breaks the group for good.
 * allreduce_max: the maximum of every thread's value.  offer + take: the neighbour exchange in two calls -- rank r offers to_up (to
 * r - 1) / to_down (to r + 1) with eight words of description each (meta[0] = bytes; NULL = nothing) and learns what its neighbours
 * offer (meta_from_*[0] = -1: nothing); take copies those bytes into buffers the caller sized from that. */
int mhip_tg_create(int32_t n, void **group);
int mhip_tg_destroy(void *group);
int mhip_tg_barrier(void *group, int32_t timeout_ms);
int mhip_tg_allreduce_max(void *group, int32_t rank, double value, double *out, int32_t timeout_ms);
int mhip_tg_offer(void *group, int32_t rank, const void *to_up, const int64_t *meta_up /* [8] */, const void *to_down, const int64_t *meta_down,
                  int64_t *meta_from_up /* [8] */, int64_t *meta_from_down, int32_t timeout_ms);
int mhip_tg_take(void *group, int32_t rank, void *from_up, void *from_down, int32_t timeout_ms);
/* the barrier of ranks that are PROCESSES of one host (distributed.ShmComm keeps the payloads of its collectives in a POSIX shared-memory
 * segment every rank maps): two 64-bit words at `base` (8-byte aligned, zero at the start), n ranks; MHIP_ECOMM after timeout_ms */
int mhip_shm_barrier(void *base, int32_t n, int32_t timeout_ms);

/* per-label records of labels that live in several bands: parts[r] = band r's n partial records, merged in band order into out.
 * kind 0: {min, max, sum, count} (label_stats, _label.pyx:22-24); kind 2 / 3: {value, row, col} of label_max_index / label_min_index
 * (_label.pyx:26-28; row < 0: the band holds no cell of the label; the earlier band wins ties = first raster position) */
int mhip_band_merge_records(int32_t kind, int32_t R, int64_t n, const void *const *parts, void *out);

/* ---- device-resident pipeline (DemTool / BluespotTool sequences, reference dem.py:53-93,
 *      bluespots.py:138-216): one upload, all stages in HBM, downloads only for the writers. ------- */
typedef struct mhip_ctx mhip_ctx;

enum mhip_stage {
    MHIP_STAGE_FILL = 1 << 0,        /* filled f32 (+ depths f32 fused epilogue) */
    MHIP_STAGE_NOFLAT = 1 << 1,      /* short/diag + no-flats surface f64 */
    MHIP_STAGE_FLOWDIR = 1 << 2,     /* D8 on the no-flats surface, edges outward */
    MHIP_STAGE_ACCUM = 1 << 3,       /* accumulated flow f64 */
    MHIP_STAGE_LABEL = 1 << 4,       /* raw bluespot labels i32 + raw label_stats */
    MHIP_STAGE_WATERSHED = 1 << 5,   /* watersheds i32 from (filtered) labels + label_count */
    MHIP_STAGE_POURPOINTS = 1 << 6,  /* argmax(accum) or argmin(no-flats) per label */
    MHIP_STAGE_ALL = 0x7f
};

enum mhip_raster {
    MHIP_R_DEM = 0, MHIP_R_FILLED, MHIP_R_DEPTHS, MHIP_R_NOFLAT, MHIP_R_FLOWDIR, MHIP_R_ACCUM,
    MHIP_R_LABELS, MHIP_R_WATERSHEDS,
    MHIP_R_NGDIST,   /* uint32: distances of the geodesic no-flats fill (row bands exchange its edge rows), see mhip_ctx_geo_begin */
    MHIP_R_COUNT_
};

/* Single-GPU context for an H x W raster on `device`. */
int mhip_ctx_create(mhip_ctx **out, int64_t H, int64_t W, int device);
/* Multi-GPU context: this rank owns rows [row0, row0+H_local) of a H_global x W raster; bands are contiguous in rank
 * order, one rank (process) per GPU.
 * RCCL transport: rank 0 calls mhip_comm_unique_id (ncclGetUniqueId, 128 bytes), the launcher hands the bytes to every
 * rank (any channel: a file, a socket, torch.distributed ...), and every rank passes them to mhip_ctx_create_band, which
 * joins the communicator (ncclCommInitRank: a COLLECTIVE call over all nranks bands).  With nccl_unique_id == NULL the
 * context has no communicator and the launcher moves the edge rows itself (mhip_ctx_get_edge_row / _set_halo_row). */
int mhip_comm_unique_id(void *id128);
/* 1 when librccl and every symbol the band transport needs can be loaded (no device call, no collective): every rank votes on
 * this BEFORE any rank enters mhip_ctx_create_band with an id -- ncclCommInitRank would wait for a rank that cannot join */
int mhip_comm_available(void);
int mhip_ctx_create_band(mhip_ctx **out, int64_t H_global, int64_t W, int64_t row0, int64_t H_local,
                         int device, int rank, int nranks, const void *nccl_unique_id);
int mhip_ctx_has_comm(mhip_ctx *ctx);        /* 0: no RCCL communicator, 1: one, 2: also the side communicator */
/* A second communicator over the same bands (ncclCommInitRank with another id from mhip_comm_unique_id: a COLLECTIVE call) for the
 * thread between mhip_ctx_side_begin / _end: the labelling branch trades its seam rows over RCCL next to the main thread's halo
 * exchanges, and two threads must not interleave operations on one communicator. */
int mhip_ctx_comm_add_side(mhip_ctx *ctx, const void *nccl_unique_id);
/* Neighbour exchange over RCCL (ncclGroupStart / ncclSend + ncclRecv per neighbour / ncclGroupEnd on the context's stream,
 * straight out of the raster's edge rows; band neighbours are xGMI peers): the rows that arrive are compared with and
 * stored into the halo rows on the device; changed[0] / changed[1] = the top / bottom halo row changed. */
int mhip_ctx_exchange_halo(mhip_ctx *ctx, int which, int32_t *changed);
/* the same exchange into host buffers (W elements each, NULL where there is no neighbour) without touching the halo rows: what the
 * boundary systems of labelling / accumulation / watersheds need of a neighbour -- its edge row next to this band's own halo row */
int mhip_ctx_exchange_edge_rows(mhip_ctx *ctx, int which, void *host_from_up, void *host_from_down);
/* max over all bands of one value (ncclAllReduce): ends the fill / accumulation loops ("is anybody still active") */
int mhip_ctx_allreduce_max(mhip_ctx *ctx, double value, double *out);
int mhip_ctx_destroy(mhip_ctx *ctx);
/* ---- row-band protocol.  A band context holds its owned rows plus one halo row per neighbour (a copy of
 * the neighbouring band's edge row).  Edge rows move between neighbours over RCCL (mhip_ctx_exchange_halo above) or,
 * for a context without a communicator, through the launcher (host buffers, or device buffers: the *_dev variants
 * below); either way the launcher drives the fills to a GLOBAL fixed point: begin -> { batch; exchange edge rows; halo_changed } until no band is active. */
int mhip_ctx_band_info(mhip_ctx *ctx, int64_t *row_off, int64_t *rows_local, int32_t *halo_top, int32_t *halo_bottom);
int mhip_ctx_get_edge_row(mhip_ctx *ctx, int which, int side, void *host);   /* side 0: first owned row, 1: last owned row,
                                                                                2: top halo row, 3: bottom halo row */
int mhip_ctx_set_halo_row(mhip_ctx *ctx, int which, int side, const void *host, int32_t *changed); /* 0: top halo, 1: bottom */
/* both rows of a halo exchange over the host transport with ONE synchronisation per call (the reference trades nothing: one raster,
 * fill.py:112-232; the bands' fill loops exchange 7-24 times per step).  get: first / last OWNED row (a NULL buffer: not wanted);
 * set: the neighbours' rows into the halo rows (NULL: no neighbour there), changed[0 / 1] = the top / bottom halo row changed */
int mhip_ctx_get_edge_rows(mhip_ctx *ctx, int which, void *host_first, void *host_last);
int mhip_ctx_set_halo_rows(mhip_ctx *ctx, int which, const void *host_top, const void *host_bottom, int32_t *changed /* [2] */);
/* the same with DEVICE buffers of the caller (W * element size bytes on the context's GPU), for a launcher that brings
 * its own device-to-device transport */
int mhip_ctx_get_edge_row_dev(mhip_ctx *ctx, int which, int side, void *dev_dst);
int mhip_ctx_set_halo_row_dev(mhip_ctx *ctx, int which, int side, const void *dev_src, int32_t *changed);
/* extremes of the band's DEM rows, for the GLOBAL extremes of fill.py:235-250: over the owned rows -- or, when the band's flood has
 * just folded them (over its local rows, halo rows included: cells of the same raster), those */
int mhip_ctx_dem_minmax(mhip_ctx *ctx, float *mn, float *mx, int32_t *has_nan);
/* kind 0: fill.fill_terrain, kind 1: fill.fill_terrain_no_flats (short/diag from the GLOBAL dem extremes) */
int mhip_ctx_fill_begin(mhip_ctx *ctx, int kind, double short_, double diag, int32_t *active);
int mhip_ctx_fill_batch(mhip_ctx *ctx, int kind, int32_t *active);
int mhip_ctx_fill_halo_changed(mhip_ctx *ctx, int kind, int side);
/* certification: one sweep over EVERY tile of the band, iterated to local convergence; *changed = some tile moved.  The
 * launcher calls it once all bands are quiescent and resumes the exchange loop if any band reports a change: a sweep that
 * changes nothing anywhere proves the global state is the fixed point (the worklist schedule itself only revisits tiles
 * whose halo a neighbour's probe saw drop) */
int mhip_ctx_fill_certify(mhip_ctx *ctx, int kind, int32_t *changed);
int mhip_ctx_fill_end(mhip_ctx *ctx, int kind);   /* kind 0 also computes the bluespot depths */
/* the no-flats fill of a band as an integer geodesic distance transform (csrc/noflat_geo.hip; needs the converged plain fill
 * incl. its halo rows).  geo_begin classifies and relaxes every tile once; *applicable == 0: a level without integer weights
 * (a flat at elevation 0, NaN, ...) -- then EVERY band runs mhip_ctx_fill_begin(kind 1) instead.  Loop: swap the edge rows of
 * MHIP_R_NGDIST, geo_halo_changed(side) for a halo row that changed, geo_batch (to local convergence) while anything moved
 * anywhere.  geo_end writes MHIP_R_NOFLAT = filled + ulp * distance and checks the reference's equation (_fill.pyx:107-117) at
 * every owned cell; *ok == 0 on any band: run the float64 relaxation (kind 1) on all of them. */
int mhip_ctx_geo_begin(mhip_ctx *ctx, double short_, double diag, int32_t *applicable, int32_t *active);
int mhip_ctx_geo_batch(mhip_ctx *ctx, int32_t *active);
int mhip_ctx_geo_halo_changed(mhip_ctx *ctx, int side);
int mhip_ctx_geo_end(mhip_ctx *ctx, int32_t *ok, int32_t *partial);
/* *partial (with *ok): MHIP_R_NOFLAT is exact except on flats the transform does not cover (a level without integer weights, a
 * distance beyond the uint32 headroom), which hold an upper bound.  If any band reports it, ALL bands call
 * mhip_ctx_fill_attach(kind 1) -- mhip_ctx_fill_begin without the initialising round: the surface in MHIP_R_NOFLAT is the start --
 * and run the usual loop (edge rows of MHIP_R_NOFLAT, _halo_changed, _batch, _certify, _end); mhip_ctx_noflat_verify then checks
 * the reference's equation at every owned cell (any band failing: relaxation from scratch, mhip_ctx_fill_begin kind 1). */
int mhip_ctx_fill_attach(mhip_ctx *ctx, int kind, double short_, double diag);
int mhip_ctx_noflat_verify(mhip_ctx *ctx, int32_t *ok);
/* accumulation on a band: mhip_ctx_zero_raster(ACCUM) once, then { mhip_ctx_run(ACCUM); swap ACCUM edge rows } until no
 * halo row changes (a halo value <= 0 means "not known yet" and blocks the cells below it). */
int mhip_ctx_zero_raster(mhip_ctx *ctx, int which);
/* ... or without iterating (two local passes + one all-gather of edge rows, however often a river crosses the seams):
 * the boundary pass accumulates with the halo rows as sources of NO flux (ACCUM = the band's own contribution A0) and
 * reports, for each halo cell k (top halo row: k = column, bottom halo row: k = W + column), exit_map[k] = side * W + column
 * of the cell of the first (side 0) / last (side 1) owned row through which the flux of k leaves the band again, or -1 when
 * it stays inside.  Every EXIT cell e then satisfies  x[e] = A0[e] + sum of x[neighbour's edge cell under k] over the halo
 * cells k with exit_map[k] = e  -- a forest over all bands' edge cells that the launcher solves on the host; it writes the
 * solved values into the ACCUM halo rows and runs mhip_ctx_run(ACCUM) once (reference: the same sums as _flow.pyx:212-247). */
int mhip_ctx_band_accum_boundary(mhip_ctx *ctx, int32_t *exit_map /* 2 * W, host */);
/* labelling on a band: local components -> host merges the boundary equivalences of all bands -> global LUT.
 * The relabel calls below may only JOIN components that are connected across bands (or drop components): the record passes take a
 * label without a cell on a tile outline for a component that lies inside that tile and write its record without atomics.  (Labels
 * that were uploaded with mhip_ctx_upload / _upload_rows carry no such promise and take the general path.) */
int mhip_ctx_band_ccl_local(mhip_ctx *ctx, int64_t *nlocal);
int mhip_ctx_band_relabel(mhip_ctx *ctx, const int32_t *lut, int64_t nlocal, int64_t nlabels_global);
/* the same without a dense table: local label l -> offset + l - #(dropped labels < l); dropped[k] (sorted: the local labels
 * numbered by another band, or without an owned cell) -> target[k] */
int mhip_ctx_band_relabel_sparse(mhip_ctx *ctx, int64_t nlocal, int64_t offset, const int32_t *dropped, const int32_t *target,
                                 int64_t ndropped, int64_t nlabels_global);
/* mhip_ctx_band_ccl_local + mhip_ctx_band_relabel_sparse as two HALVES of one labelling, without the two passes over the label raster
 * between them (reference: label.py:8-23 numbers one raster in one go; a band has to wait for its neighbours).  begin: everything
 * up to the emit pass -- *nlocal band-local labels, and of the LABELS raster only the two top and the two bottom local rows are
 * written (band-local labels: what mhip_ctx_get_edge_row / mhip_ctx_exchange_edge_rows hand to the seam merge).  finish: the
 * GLOBAL label of every cell in one pass (the map of mhip_ctx_band_relabel_sparse); with_stats != 0: label_stats of the depths over
 * the owned rows by global label ride on it -- the records of mhip_ctx_band_records(ctx, 0), which need not run then.  Between the
 * two calls LABELS is not a raster of labels (mhip_ctx_download etc. refuse it). */
int mhip_ctx_band_ccl_begin(mhip_ctx *ctx, int64_t *nlocal);
int mhip_ctx_band_ccl_finish(mhip_ctx *ctx, int64_t offset, const int32_t *dropped, const int32_t *target, int64_t ndropped,
                             int64_t nlabels_global, int with_stats);
/* watersheds on a band: local pointer jumping with pseudo labels on the halo rows, then a boundary LUT */
/* the bluespot filter on a band (bluespots.py:165-172 as a rank relabel): labels in [lo, hi] (numbered by this band) -> lut[l - lo]
 * (0 = dropped); a label another band numbered -> fnew[k] where fid[k] == l (fid sorted, unique); nlabels_new = the global count */
int mhip_ctx_band_relabel_range(mhip_ctx *ctx, int64_t lo, int64_t hi, const int32_t *lut, const int32_t *fid, const int32_t *fnew, int64_t nf,
                                int64_t nlabels_new);
/* One leg of net.next_downstream_label (net.py:142-169) on a band.  cells_rc: GLOBAL (row, col) of n walkers standing on owned rows;
 * src_label[i] >= 0: the walker's source label (it came from another band), -1 or src_label == NULL: its start cell's label.
 * out_status: 0 ended without a label, 1 found out_label, 2 stepped onto a neighbour's row at out_exit_rc (GLOBAL row, col) -- the
 * launcher hands it to that band.  Geometry (GLOBAL linear indices) in two passes: lengths, then offsets + out_cells. */
int mhip_ctx_band_trace(mhip_ctx *ctx, const int64_t *cells_rc, const int32_t *src_label, int64_t n, int use_background, int32_t background,
                        int32_t *out_label, int32_t *out_status, int32_t *out_src, int64_t *out_exit_rc, int64_t *out_len, const int64_t *offsets,
                        int64_t *out_cells);
int mhip_ctx_band_watershed_local(mhip_ctx *ctx);
int mhip_ctx_band_apply_neg_lut(mhip_ctx *ctx, int which, const int32_t *lut, int64_t n);
/* Two host threads per band (the labelling branch next to no-flats fill -> D8 -> accumulation, like the stage DAG of
 * mhip_ctx_run): the thread that drives the labelling branch calls _side_begin after the plain fill has been issued and
 * _side_end when it is done; in between, the band / data-movement entry points IT calls use the context's side stream.
 * _side_end returns when the side stream has drained, so the caller only has to join the thread. */
int mhip_ctx_side_begin(mhip_ctx *ctx);
int mhip_ctx_side_end(mhip_ctx *ctx);
/* per-label records over the OWNED rows of the band, indexed by GLOBAL label (reference bluespots.py:159-206 on one
 * raster).  which: 0 = label_stats of the depths (mhip_stat_record), 1 = np.bincount of the watersheds (int64), 2 = first
 * arg-max of the accumulated flow (mhip_index_record, rows are global raster rows), 3 = first arg-min of the no-flats surface (the
 * pour points when no accumulated flow was asked for, bluespots.py:203-205; shares the buffer of 2).  _records computes nlabels_global + 1
 * entries and keeps them on the device; the launcher fetches the slice of the labels this band numbered (_fetch), the few
 * labels that cross a band boundary (_gather) and the non-zero watershed counts of labels outside [lo, hi] (_foreign_counts:
 * up to cap pairs, *nfound = how many exist) and merges them across bands */
int mhip_ctx_band_records(mhip_ctx *ctx, int which);
int mhip_ctx_band_fetch(mhip_ctx *ctx, int which, int64_t first, int64_t count, void *out);
int mhip_ctx_band_gather(mhip_ctx *ctx, int which, const int64_t *ids, int64_t nids, void *out);
int mhip_ctx_band_foreign_counts(mhip_ctx *ctx, int64_t lo, int64_t hi, int64_t cap, int64_t *ids, int64_t *counts, int64_t *nfound);
int mhip_ctx_upload_dem(mhip_ctx *ctx, const float *dem_band);       /* H_local x W host raster */
int mhip_ctx_upload(mhip_ctx *ctx, int which, const void *host);     /* any raster (for sub-commands) */
int mhip_ctx_download(mhip_ctx *ctx, int which, void *host);         /* H_local x W */
/* the same in row windows [row0, row0 + nrows) of the owned raster (the reference's io.py:21-159 moves whole rasters): a
 * streaming reader / writer keeps one window on the host whatever the raster's size; a raster counts as present once its
 * last row has arrived */
int mhip_ctx_upload_rows(mhip_ctx *ctx, int which, int64_t row0, int64_t nrows, const void *host);
int mhip_ctx_download_rows(mhip_ctx *ctx, int which, int64_t row0, int64_t nrows, void *host);
int mhip_ctx_run(mhip_ctx *ctx, int stage_mask);                     /* asynchronous on the ctx stream */
int mhip_ctx_sync(mhip_ctx *ctx);
/* mhip_trace_downstream_i32 on the context's resident flow directions and bluespot labels (StreamTool after BluespotTool) */
int mhip_ctx_trace_downstream(mhip_ctx *ctx, const int64_t *cells_rc, int64_t n, int use_background, int32_t background,
                              int32_t *out_label, int32_t *out_found, int64_t *out_len, const int64_t *offsets, int64_t *out_cells);
/* after mhip_ctx_sync: milliseconds (HIP events on the ctx stream) of `stage` (single bit) in the last run */
int mhip_ctx_stage_ms(mhip_ctx *ctx, int stage, float *ms);
/* milliseconds / launch count of one named kernel family in the last run ("d8", "fill_round", "noflat_round");
 * "d8_steady" is a measurement of its own: it LAUNCHES the D8 stencil on the resident no-flats surface 16 times back to back
 * between one pair of events (steady-state throughput; an undivided context only) and returns their total */
int mhip_ctx_kernel_ms(mhip_ctx *ctx, const char *kernel, float *ms_total, int32_t *launches);
/* "nlabels_raw", "nlabels", "fill_rounds", "fill_launches", "fill_visits", "noflat_rounds", "noflat_visits", ... and which engine
 * the last run of a stage took: "fill_algorithm" (1 tiled priority-flood, 0 iterative schedule, 4 flood + iterative repair),
 * "noflat_algorithm" (2 integer geodesic transform, 3 + float64 relaxation of irregular flats, 0 float64 relaxation; why a
 * transform was handed back: "noflat_reject*"), "pour_algorithm" (1 keys out of the accumulation's final pass, 0 a pass over
 * values + labels), "accum_algorithm" (row bands: 1 the second pass as a delta over the boundary pass's graph, 0 a full pass) */
int mhip_ctx_get_i64(mhip_ctx *ctx, const char *key, int64_t *value);
int mhip_ctx_get_f64(mhip_ctx *ctx, const char *key, double *value);  /* "short", "diag" */
/* label filter between MHIP_STAGE_LABEL and MHIP_STAGE_WATERSHED (reference bluespots.py:165-172):
 * download raw stats (nlabels_raw+1 records), decide on host, upload keep flags. */
int mhip_ctx_raw_stats(mhip_ctx *ctx, mhip_stat_record *records);
int mhip_ctx_apply_keep(mhip_ctx *ctx, const uint8_t *keep);          /* NULL = keep all */
int mhip_ctx_stats(mhip_ctx *ctx, mhip_stat_record *records);         /* nlabels+1, after apply_keep */
int mhip_ctx_watershed_counts(mhip_ctx *ctx, int64_t *counts);        /* nlabels+1 */
int mhip_ctx_pourpoints(mhip_ctx *ctx, mhip_index_record *records);   /* nlabels+1 */

#ifdef __cplusplus
}
#endif
#endif /* MALSTROEM_HIP_H */
