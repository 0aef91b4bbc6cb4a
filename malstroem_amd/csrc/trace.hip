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

}  // namespace

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
