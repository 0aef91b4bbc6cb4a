// ccllab.hip -- development bench for the tile kernel of the connected-component labelling (not part of the library).
//   python tools/lab/dump_rasters.py && hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/ccllab.hip -o tools/bin/ccllab && tools/bin/ccllab
#include <cstdarg>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../malstroem_amd/csrc/ccl.hip"

namespace mh {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const char *get_error() { return ""; }
int pool_alloc(void **p, size_t n) { return hipMalloc(p, n) == hipSuccess ? 0 : -1; }
void pool_free(void *p, size_t) { (void)hipFree(p); }
const char *dev_env(const char *name) { return getenv(name); }
hipError_t stream_sync(hipStream_t s) { return hipStreamSynchronize(s); }
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

// the library's tile kernel cut short after phase STOP (1: runs, 2: + unions, 3: + flatten and stores = all of it)
template <int STOP>
__global__ __launch_bounds__(256) void ccl_tile_cut(const float *__restrict__ data, int32_t *__restrict__ parent, int64_t H, int64_t W, int ntc,
                                                    int32_t *__restrict__ rootlist, int32_t *__restrict__ rootcount)
{
    __shared__ uint32_t par[CT * CT];
    __shared__ int s_nroots;
    if (threadIdx.x == 0) s_nroots = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ti = blockIdx.x / ntc, tj = blockIdx.x - ti * ntc;
    const int64_t r0 = (int64_t)ti * CT, c0 = (int64_t)tj * CT;
    const int64_t cc = c0 + lane;
    uint64_t fgrow[CT / 4];
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) {
        const int r = wave + 4 * k;
        const bool fg = (r0 + r) < H && cc < W && is_fg(data[(r0 + r) * W + cc]);
        const uint64_t m = __ballot(fg);
        fgrow[k] = m;
        const uint64_t starts = m & ~(m << 1);
        const uint64_t below = starts & ((2ull << lane) - 1ull);
        par[r * CT + lane] = fg ? (uint32_t)(r * CT + (63 - __builtin_clzll(below))) : LBG;
    }
    __syncthreads();
    if (STOP >= 2) {
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) {
            const int r = wave + 4 * k;
            if (r == 0) continue;
            const uint64_t m = fgrow[k];
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
    }
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) {
        const int r = wave + 4 * k;
        if ((r0 + r) >= H || cc >= W) continue;
        const uint32_t p = par[r * CT + lane];
        int32_t out = -1;
        if (p != LBG) {
            const uint32_t root = STOP >= 3 ? find_root_l(par, p) : p;
            out = (int32_t)((r0 + (root >> 6)) * W + c0 + (root & 63u));
            if (root == (uint32_t)(r * CT + lane)) rootlist[(size_t)blockIdx.x * MAXROOTS + atomicAdd(&s_nroots, 1)] = out;
        }
        parent[(r0 + r) * W + cc] = out;
    }
    __syncthreads();
    if (threadIdx.x == 0) rootcount[blockIdx.x] = s_nroots;
}

int main()
{
    long H, W, nlab;
    { FILE *f = fopen("/tmp/mlab/meta.txt", "r"); if (!f || fscanf(f, "%ld %ld %ld", &H, &W, &nlab) != 3) { fprintf(stderr, "no meta\n"); return 2; } fclose(f); }
    const size_t n = (size_t)H * W;
    float *depths = load<float>("depths", n);
    int32_t *ref = load<int32_t>("labels", n);
    int32_t *labels, *tmp; CK(hipMalloc(&labels, n * 4)); CK(hipMalloc(&tmp, n * 4));
    const int64_t ntr = cdiv(H, CT), ntc = cdiv(W, CT), ntiles = ntr * ntc;
    int32_t *roots, *rcount; CK(hipMalloc(&roots, 4 * (size_t)ntiles * MAXROOTS)); CK(hipMalloc(&rcount, 4 * (size_t)ntiles));
    hipStream_t s = 0;
    auto report = [&](const char *name, float ms) { printf("%-52s %8.3f ms  %7.1f GB/s\n", name, ms, 8.0 * (double)n / 1e9 / (ms * 1e-3)); fflush(stdout); };
    int64_t nl = 0;
    report("ccl8_f32_dev (whole labelling)", time_ms([&] { ccl8_f32_dev(depths, labels, tmp, H, W, &nl, s); }));
    {
        std::vector<int32_t> a(n), b(n);
        CK(hipMemcpy(a.data(), labels, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
        printf("labels vs the pipeline's: %zu differ (%lld labels)\n", bad, (long long)nl);
    }
    report("ccl_tile_kernel (library)", time_ms([&] { hipLaunchKernelGGL((ccl_tile_kernel<float>), dim3((unsigned)ntiles), dim3(256), 0, s, depths, tmp, H, W, (int)ntc, roots, rcount); }));
    report("  cut after the runs (+ stores)", time_ms([&] { hipLaunchKernelGGL((ccl_tile_cut<1>), dim3((unsigned)ntiles), dim3(256), 0, s, depths, tmp, H, W, (int)ntc, roots, rcount); }));
    report("  cut after the unions (+ stores)", time_ms([&] { hipLaunchKernelGGL((ccl_tile_cut<2>), dim3((unsigned)ntiles), dim3(256), 0, s, depths, tmp, H, W, (int)ntc, roots, rcount); }));
    report("  all phases", time_ms([&] { hipLaunchKernelGGL((ccl_tile_cut<3>), dim3((unsigned)ntiles), dim3(256), 0, s, depths, tmp, H, W, (int)ntc, roots, rcount); }));
    return 0;
}
