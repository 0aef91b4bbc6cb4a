// taillab.hip -- development bench for the per-label reductions of the tail (not part of the library).
//   python tools/lab/dump_rasters.py && hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/taillab.hip -o /tmp/taillab && /tmp/taillab
// Reads /tmp/mlab/{depths,labels,watersheds,accum}.bin (a real pipeline's rasters), times the library kernels and ablations of them.
#include <cstdarg>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../malstroem_amd/csrc/label_ops.hip"

namespace mh {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const char *get_error() { return ""; }
int pool_alloc(void **p, size_t n) { return hipMalloc(p, n) == hipSuccess ? 0 : -1; }
void pool_free(void *p, size_t) { (void)hipFree(p); }
hipError_t stream_sync(hipStream_t s) { return hipStreamSynchronize(s); }
const char *dev_env(const char *name) { return getenv(name); }
}  // namespace mh
using namespace mh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <typename F> static float time_ms(F f, int reps = 9)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <typename T> static T *load(const char *name, size_t n)
{
    std::vector<T> h(n);
    char path[256]; snprintf(path, sizeof path, "/tmp/mlab/%s.bin", name);
    FILE *f = fopen(path, "rb");
    if (!f || fread(h.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    T *d; CK(hipMalloc(&d, n * sizeof(T))); CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

// ---- ablations of stats_kernel: what does its structure cost without the work? ------------------------------------------------
// MODE 0: the tile loop and its loads only (every cell counted as background in registers)
// MODE 1: + LDS table init / flush scan per tile (no inserts)
template <int MODE, int ROWS>
__global__ __launch_bounds__(256) void stats_floor_kernel(const float *__restrict__ data, const int32_t *__restrict__ lab, TileGeom g, StatAcc a)
{
    __shared__ int keys[STATS_TS];
    __shared__ unsigned int tcnt[STATS_TS], tmin[STATS_TS], tmax[STATS_TS];
    __shared__ double tsum[STATS_TS];
    float bmin = __builtin_inff(), bmax = -__builtin_inff();
    double bsum = 0.0;
    unsigned long long bcnt = 0;
    const int64_t ntiles = g.ntr * g.ntc;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (MODE >= 1) {
            for (int k = threadIdx.x; k < STATS_TS; k += 256) { keys[k] = -1; tcnt[k] = 0u; tmin[k] = 0xffffffffu; tmax[k] = 0u; tsum[k] = 0.0; }
            __syncthreads();
        }
        const int64_t tr = tile / g.ntc, tc = tile - tr * g.ntc;
        const int64_t col = tc * 256 + threadIdx.x;
        for (int r4 = 0; r4 < TR; r4 += ROWS) {
            int32_t lq[ROWS]; float dq[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const int64_t i = (tr * TR + r4 + u) * g.W + col;
                const bool v = col < g.W && i < g.n;
                lq[u] = v ? lab[i] : -1;
                dq[u] = v ? data[i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                if (lq[u] >= 0) { bmin = fminf(bmin, dq[u]); bmax = fmaxf(bmax, dq[u]); bsum += (double)dq[u]; bcnt += (unsigned)lq[u]; }
            }
        }
        if (MODE >= 1) {
            __syncthreads();
            for (int k = threadIdx.x; k < STATS_TS; k += 256)
                if (keys[k] >= 0) atomicAdd(&a.count[keys[k]], (unsigned long long)tcnt[k]);
            __syncthreads();
        }
    }
    if (bcnt == 12345ull) { atomicMin(&a.minkey[0], f32_key(bmin)); atomicMax(&a.maxkey[0], f32_key(bmax)); atomicAdd(&a.sum[0], bsum); }
    if ((threadIdx.x & 63) == 0) atomicAdd(&a.count[0], bcnt);
}

// ---- ablations of the work inside stats_kernel --------------------------------------------------------------------------------
// WHAT bit 0: the segmented reductions; bit 1: the run heads' LDS table updates; bit 2: the flush of the table to global memory;
// 8: VERTICAL runs instead (a thread follows its column down the tile and flushes when the label changes)
template <int WHAT>
__global__ __launch_bounds__(256) void stats_abl_kernel(const float *__restrict__ data, const int32_t *__restrict__ lab, TileGeom g, int64_t nlab, StatAcc a)
{
    __shared__ int keys[STATS_TS];
    __shared__ unsigned int tcnt[STATS_TS], tmin[STATS_TS], tmax[STATS_TS];
    __shared__ double tsum[STATS_TS];
    const int lane = threadIdx.x & 63;
    float bmin = __builtin_inff(), bmax = -__builtin_inff();
    double bsum = 0.0;
    unsigned long long bcnt = 0;
    auto to_global = [&](int32_t l, uint32_t kmin, uint32_t kmax, double sum, unsigned long long cnt) {
        if (kmin < a.minkey[l]) atomicMin(&a.minkey[l], kmin);
        if (kmax > a.maxkey[l]) atomicMax(&a.maxkey[l], kmax);
        atomicAdd(&a.sum[l], sum);
        atomicAdd(&a.count[l], cnt);
    };
    const int64_t ntiles = g.ntr * g.ntc;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int k = threadIdx.x; k < STATS_TS; k += 256) { keys[k] = -1; tcnt[k] = 0u; tmin[k] = 0xffffffffu; tmax[k] = 0u; tsum[k] = 0.0; }
        __syncthreads();
        const int64_t tr = tile / g.ntc, tc = tile - tr * g.ntc;
        const int64_t col = tc * 256 + threadIdx.x;
        // vertical-run state
        int32_t cl = 0; float cmin = 0, cmax = 0; double csum = 0; unsigned ccnt = 0;
        auto flush_lane = [&]() {
            if (cl > 0) {
                const int h = table_slot<STATS_TS>(keys, cl);
                if (h >= 0) {
                    atomicMin(&tmin[h], f32_key(cmin)); atomicMax(&tmax[h], f32_key(cmax)); atomicAdd(&tsum[h], csum); atomicAdd(&tcnt[h], ccnt);
                } else to_global(cl, f32_key(cmin), f32_key(cmax), csum, ccnt);
            } else if (cl == 0 && ccnt) { bmin = fminf(bmin, cmin); bmax = fmaxf(bmax, cmax); bsum += csum; bcnt += ccnt; }
        };
        for (int r4 = 0; r4 < TR; r4 += 4) {
            int32_t lq[4]; float dq[4]; bool vq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (tr * TR + r4 + u) * g.W + col;
                vq[u] = col < g.W && i < g.n;
                lq[u] = vq[u] ? lab[i] : -1;
                dq[u] = vq[u] ? data[i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int32_t l = lq[u];
                const float v = dq[u];
                float vmin = v, vmax = v;
                if (WHAT & 8) {
                    if (l != cl || (r4 + u) == 0) {
                        if ((r4 + u) != 0) flush_lane();
                        cl = l; cmin = v; cmax = v; csum = (double)v; ccnt = 1;
                    } else { cmin = fminf(cmin, v); cmax = fmaxf(cmax, v); csum += (double)v; ++ccnt; }
                    continue;
                }
                if (__all(l <= 0)) {
                    if (l == 0) { bmin = fminf(bmin, vmin); bmax = fmaxf(bmax, vmax); bsum += (double)v; ++bcnt; }
                    continue;
                }
                const bool ok = l >= 0;
                const int len = run_length_from(l, lane, ok);
                const bool head = is_run_head(l, lane, ok);
                double sm = (double)v;
                if (WHAT & 1) {
                    vmin = seg_reduce(vmin, len, [](float x, float y) { return fminf(x, y); });
                    vmax = seg_reduce(vmax, len, [](float x, float y) { return fmaxf(x, y); });
                    sm = seg_reduce((double)v, len, [](double x, double y) { return x + y; });
                }
                if (head && ok) {
                    if (l == 0) { bmin = fminf(bmin, vmin); bmax = fmaxf(bmax, vmax); bsum += sm; bcnt += (unsigned long long)len; }
                    else if (WHAT & 2) {
                        const uint32_t kmin = f32_key(vmin), kmax = f32_key(vmax);
                        const int h = table_slot<STATS_TS>(keys, l);
                        if (h >= 0) { atomicMin(&tmin[h], kmin); atomicMax(&tmax[h], kmax); atomicAdd(&tsum[h], sm); atomicAdd(&tcnt[h], (unsigned int)len); }
                        else to_global(l, kmin, kmax, sm, (unsigned long long)len);
                    } else { bsum += sm; bcnt += len; }
                }
            }
        }
        if (WHAT & 8) flush_lane();
        __syncthreads();
        if (WHAT & 4)
            for (int k = threadIdx.x; k < STATS_TS; k += 256)
                if (keys[k] >= 0) to_global(keys[k], tmin[k], tmax[k], tsum[k], (unsigned long long)tcnt[k]);
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bmin = fminf(bmin, __shfl_xor(bmin, o)); bmax = fmaxf(bmax, __shfl_xor(bmax, o)); bsum += __shfl_xor(bsum, o); bcnt += __shfl_xor(bcnt, o);
    }
    if (lane == 0 && bcnt) { atomicMin(&a.minkey[0], f32_key(bmin)); atomicMax(&a.maxkey[0], f32_key(bmax)); atomicAdd(&a.sum[0], bsum); atomicAdd(&a.count[0], bcnt); }
}

// how are the labelled cells spread?  per wave-row (64 cells): any labelled cell / run heads
__global__ __launch_bounds__(256) void census_kernel(const int32_t *__restrict__ lab, int64_t n, unsigned long long *out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t l = i < n ? lab[i] : 0;
    const bool any = __any(l > 0);
    const bool ok = true;
    const bool head = is_run_head(l, lane, ok) && l > 0;
    const unsigned long long heads = __popcll(__ballot(head)), cells = __popcll(__ballot(l > 0));
    if (lane == 0) {
        atomicAdd(&out[0], 1ull);
        if (any) atomicAdd(&out[1], 1ull);
        atomicAdd(&out[2], heads);
        atomicAdd(&out[3], cells);
    }
}

int main(int argc, char **argv)
{
    long H, W, nlab;
    { FILE *f = fopen("/tmp/mlab/meta.txt", "r"); if (!f || fscanf(f, "%ld %ld %ld", &H, &W, &nlab) != 3) { fprintf(stderr, "no meta\n"); return 2; } fclose(f); }
    const size_t n = (size_t)H * W;
    printf("raster %ld x %ld, %ld labels\n", H, W, nlab);
    float *depths = load<float>("depths", n);
    int32_t *labels = load<int32_t>("labels", n);
    int32_t *ws = load<int32_t>("watersheds", n);
    double *accum = load<double>("accum", n);
    hipStream_t s = 0;
    {
        unsigned long long *c; CK(hipMalloc(&c, 64)); CK(hipMemset(c, 0, 64));
        hipLaunchKernelGGL(census_kernel, dim3((unsigned)cdiv((int64_t)n, 256)), dim3(256), 0, s, labels, (int64_t)n, c);
        unsigned long long h[4]; CK(hipMemcpy(h, c, 32, hipMemcpyDeviceToHost));
        printf("labels: wave-rows %llu, with a labelled cell %llu (%.1f %%), run heads %llu, labelled cells %llu (%.2f %%)\n", h[0], h[1], 100.0 * h[1] / h[0], h[2], h[3],
               100.0 * h[3] / (double)n);
        CK(hipMemset(c, 0, 64));
        hipLaunchKernelGGL(census_kernel, dim3((unsigned)cdiv((int64_t)n, 256)), dim3(256), 0, s, ws, (int64_t)n, c);
        CK(hipMemcpy(h, c, 32, hipMemcpyDeviceToHost));
        printf("watersheds: wave-rows %llu, with a labelled cell %llu (%.1f %%), run heads %llu, labelled cells %llu (%.2f %%)\n", h[0], h[1], 100.0 * h[1] / h[0], h[2], h[3],
               100.0 * h[3] / (double)n);
    }
    mhip_stat_record *rec; CK(hipMalloc(&rec, sizeof(mhip_stat_record) * (nlab + 1)));
    mhip_index_record *irec; CK(hipMalloc(&irec, sizeof(mhip_index_record) * (nlab + 1)));
    int64_t *counts; CK(hipMalloc(&counts, 8 * (nlab + 1)));
    auto report = [&](const char *name, float ms, double bytes_per_cell) {
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes_per_cell * (double)n / 1e9 / (ms * 1e-3)); fflush(stdout);
    };
    report("label_stats_dev (components)", time_ms([&] { label_stats_dev(depths, labels, (int64_t)n, nlab, rec, s, W, true); }), 8);
    report("label_stats_dev (generic)", time_ms([&] { label_stats_dev(depths, labels, (int64_t)n, nlab, rec, s, W, false); }), 8);
    report("label_count_dev (watersheds)", time_ms([&] { label_count_dev(ws, (int64_t)n, nlab, counts, s, W); }), 4);
    report("label_arg_dev (accum, labels)", time_ms([&] { label_arg_dev(accum, labels, H, W, nlab, true, irec, s); }), 12);
    report("label_arg_dev (accum, labels, components)", time_ms([&] { label_arg_dev(accum, labels, H, W, nlab, true, irec, s, true); }), 12);
    {
        DevBuf mn, mx, sm, ct;
        mn.alloc(4 * (nlab + 1)); mx.alloc(4 * (nlab + 1)); sm.alloc(8 * (nlab + 1)); ct.alloc(8 * (nlab + 1));
        StatAcc a{mn.as<uint32_t>(), mx.as<uint32_t>(), sm.as<double>(), ct.as<unsigned long long>()};
        const TileGeom g = tile_geom((int64_t)n, W);
        auto run_abl = [&](auto kern, const char *name) {
            report(name, time_ms([&] {
                hipLaunchKernelGGL(stats_init_kernel, dim3((unsigned)cdiv((int64_t)nlab + 1, 256)), dim3(256), 0, s, a, (int64_t)nlab + 1);
                hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, s, depths, labels, g, (int64_t)nlab, a); }), 8);
        };
        run_abl(stats_abl_kernel<7>, "abl: seg + heads + flush (= library)");
        run_abl(stats_abl_kernel<6>, "abl: heads + flush (no seg reduce)");
        run_abl(stats_abl_kernel<5>, "abl: seg + flush (no head updates)");
        run_abl(stats_abl_kernel<3>, "abl: seg + heads (no flush)");
        run_abl(stats_abl_kernel<1>, "abl: seg only");
        run_abl(stats_abl_kernel<0>, "abl: run logic only");
        run_abl(stats_abl_kernel<12>, "abl: VERTICAL runs + flush");
        run_abl(stats_abl_kernel<8>, "abl: VERTICAL runs, no global flush");
        {   // is the vertical variant right?  compare its records with the library's
            label_stats_dev(depths, labels, (int64_t)n, nlab, rec, s, W, false);
            std::vector<mhip_stat_record> r1(nlab + 1), r2(nlab + 1);
            CK(hipMemcpy(r1.data(), rec, sizeof(mhip_stat_record) * (nlab + 1), hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(stats_init_kernel, dim3((unsigned)cdiv((int64_t)nlab + 1, 256)), dim3(256), 0, s, a, (int64_t)nlab + 1);
            hipLaunchKernelGGL(stats_abl_kernel<12>, dim3(2048), dim3(256), 0, s, depths, labels, g, (int64_t)nlab, a);
            hipLaunchKernelGGL(stats_finish_kernel, dim3((unsigned)cdiv((int64_t)nlab + 1, 256)), dim3(256), 0, s, a, (int64_t)nlab + 1, rec);
            CK(hipMemcpy(r2.data(), rec, sizeof(mhip_stat_record) * (nlab + 1), hipMemcpyDeviceToHost));
            size_t bad = 0, badsum = 0;
            for (long i = 0; i <= nlab; ++i) {
                if (r1[i].min != r2[i].min || r1[i].max != r2[i].max || r1[i].count != r2[i].count) ++bad;
                if (r1[i].sum != r2[i].sum) ++badsum;
            }
            printf("vertical vs library: %zu records differ in min/max/count, %zu in the sum bits\n", bad, badsum);
        }
        report("stats floor: loads only, 4 rows", time_ms([&] { hipLaunchKernelGGL((stats_floor_kernel<0, 4>), dim3(2048), dim3(256), 0, s, depths, labels, g, a); }), 8);
        report("stats floor: loads only, 8 rows", time_ms([&] { hipLaunchKernelGGL((stats_floor_kernel<0, 8>), dim3(2048), dim3(256), 0, s, depths, labels, g, a); }), 8);
        report("stats floor: + table init/flush, 4 rows", time_ms([&] { hipLaunchKernelGGL((stats_floor_kernel<1, 4>), dim3(2048), dim3(256), 0, s, depths, labels, g, a); }), 8);
        report("stats floor: + table init/flush, 8 rows", time_ms([&] { hipLaunchKernelGGL((stats_floor_kernel<1, 8>), dim3(2048), dim3(256), 0, s, depths, labels, g, a); }), 8);
    }
    return 0;
}
