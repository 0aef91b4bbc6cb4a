// trace.hip -- batched stream tracing: net.next_downstream_label for many pour points at once
// (reference malstroem/algorithms/net.py:142-169; the walk is flow.trace_downstream, flow.py:286-301).
//
// The reference walks one pour point at a time in Python (pourpoint_network, net.py:172-192: a loop over all bluespots);
// here one GPU thread walks one pour point over the resident flow direction + label rasters.  Semantics per point:
//   src = labelled[cell];  for c in cell, downstream(cell), ... while c is inside the raster and has a direction:
//       (geometry: append c);  lbl = labelled[c];  if lbl != src and (no background given or lbl != background): found lbl
//   nothing found: None.  The start cell itself is part of the walk (its label is src by definition).
// A walk that would not terminate in the reference (a flow cycle) is cut after H*W steps and reports "none".
// Two passes for the geometry: lengths first (the caller sizes one flat buffer by a prefix sum), then the cells.
#include "common.hpp"

namespace mh {

namespace {

__global__ __launch_bounds__(256) void trace_kernel(const uint8_t *__restrict__ fd, const int32_t *__restrict__ lab, int64_t H, int64_t W,
                                                    const int64_t *__restrict__ cells, int64_t n, int use_bg, int32_t bg,
                                                    int32_t *__restrict__ out_label, int32_t *__restrict__ out_found,
                                                    int64_t *__restrict__ out_len, const int64_t *__restrict__ offsets,
                                                    int64_t *__restrict__ out_cells)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = cells[2 * i], c = cells[2 * i + 1];
    int64_t len = 0;
    int32_t found = 0, flabel = 0;
    int64_t *dst = (out_cells && offsets) ? out_cells + offsets[i] : nullptr;
    if (r >= 0 && r < H && c >= 0 && c < W) {
        const int32_t src = lab[r * W + c];
        const int64_t cap = H * W;
        while (len < cap) {
            const int64_t idx = r * W + c;
            if (dst) dst[len] = idx;
            ++len;
            const int32_t l = lab[idx];
            if (l != src && (!use_bg || l != bg)) {
                found = 1;
                flabel = l;
                break;
            }
            const int k = fd[idx];
            if (k > 7) break;                                   // NODIR: the walk ends here (flow.py:296-301)
            r += dir_dr(k);
            c += dir_dc(k);
            if (r < 0 || r >= H || c < 0 || c >= W) break;      // left the raster
        }
    }
    if (out_label) out_label[i] = flabel;
    if (out_found) out_found[i] = found;
    if (out_len) out_len[i] = len;
}

// The same walk on one ROW BAND of a larger raster: `fd` / `lab` are the band's local rasters (owned rows + halo rows), `row_lo` the
// global row of local row 0, owned rows are local [own0, own1).  A walker carries its source label (-1: take it from its start cell)
// and walks while it stays on owned rows; stepping onto a neighbour's row ends this leg with status 2 and the GLOBAL cell it
// stepped onto (the next band continues there: BandPipeline.trace_downstream).  status 1: label found, 0: the walk ended (NODIR,
// left the raster, or cut after `cap` steps: a flow cycle).  Cells (geometry) are GLOBAL linear indices row * W + col.
__global__ __launch_bounds__(256) void band_trace_kernel(const uint8_t *__restrict__ fd, const int32_t *__restrict__ lab, int64_t Hl, int64_t W,
                                                         int64_t row_lo, int64_t own0, int64_t own1, int64_t Hg, const int64_t *__restrict__ cells,
                                                         const int32_t *__restrict__ src_in, int64_t n, int use_bg, int32_t bg, int64_t cap,
                                                         int32_t *__restrict__ out_label, int32_t *__restrict__ out_status, int32_t *__restrict__ out_src,
                                                         int64_t *__restrict__ out_exit, int64_t *__restrict__ out_len,
                                                         const int64_t *__restrict__ offsets, int64_t *__restrict__ out_cells)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t gr = cells[2 * i], c = cells[2 * i + 1];
    int64_t r = gr - row_lo, len = 0;
    int32_t status = 0, flabel = 0, src = src_in ? src_in[i] : -1;
    int64_t ex_r = -1, ex_c = -1;
    int64_t *dst = (out_cells && offsets) ? out_cells + offsets[i] : nullptr;
    if (gr >= 0 && gr < Hg && c >= 0 && c < W && r >= own0 && r < own1) {
        if (src < 0) src = lab[r * W + c];
        while (len < cap) {
            const int64_t idx = r * W + c;
            if (dst) dst[len] = (r + row_lo) * W + c;
            ++len;
            const int32_t l = lab[idx];
            if (l != src && (!use_bg || l != bg)) {
                status = 1;
                flabel = l;
                break;
            }
            const int k = fd[idx];
            if (k > 7) break;
            r += dir_dr(k);
            c += dir_dc(k);
            if (r + row_lo < 0 || r + row_lo >= Hg || c < 0 || c >= W) break;      // left the raster
            if (r < own0 || r >= own1) {                                         // a neighbour's row: hand the walker over
                status = 2;
                ex_r = r + row_lo;
                ex_c = c;
                break;
            }
        }
    }
    if (out_label) out_label[i] = flabel;
    if (out_status) out_status[i] = status;
    if (out_src) out_src[i] = src;
    if (out_exit) { out_exit[2 * i] = ex_r; out_exit[2 * i + 1] = ex_c; }
    if (out_len) out_len[i] = len;
}

}  // namespace

int band_trace_dev(const uint8_t *d_fd, const int32_t *d_lab, int64_t Hl, int64_t W, int64_t row_lo, int64_t own0, int64_t own1, int64_t Hg,
                   const int64_t *d_cells, const int32_t *d_src, int64_t n, int use_bg, int32_t bg, int32_t *d_label, int32_t *d_status,
                   int32_t *d_src_out, int64_t *d_exit, int64_t *d_len, const int64_t *d_offsets, int64_t *d_out_cells, hipStream_t s)
{
    if (n <= 0) return MHIP_OK;
    hipLaunchKernelGGL(band_trace_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_fd, d_lab, Hl, W, row_lo, own0, own1, Hg, d_cells, d_src, n,
                       use_bg, bg, Hg * W, d_label, d_status, d_src_out, d_exit, d_len, d_offsets, d_out_cells);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

// d_cells: n (row, col) pairs on the device.  d_offsets / d_out_cells optional (second pass).
int trace_downstream_dev(const uint8_t *d_fd, const int32_t *d_lab, int64_t H, int64_t W, const int64_t *d_cells, int64_t n, int use_bg,
                         int32_t bg, int32_t *d_label, int32_t *d_found, int64_t *d_len, const int64_t *d_offsets, int64_t *d_out_cells,
                         hipStream_t s)
{
    if (n <= 0) return MHIP_OK;
    hipLaunchKernelGGL(trace_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_fd, d_lab, H, W, d_cells, n, use_bg, bg, d_label, d_found,
                       d_len, d_offsets, d_out_cells);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

}  // namespace mh
