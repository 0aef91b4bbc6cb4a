// d8lab.hip -- development bench for D8 kernel variants (not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/lab/d8lab.hip -o tools/bin/d8lab && tools/bin/d8lab [N]
// Every variant is checked against the library's kernel (included below) on a random surface and on a surface full of ties.
#include <cstdarg>
#include <vector>
#include <random>
#include "../../malstroem_amd/csrc/d8.hip"

namespace mh {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const char *get_error() { return ""; }
int pool_alloc(void **p, size_t n) { return hipMalloc(p, n) == hipSuccess ? 0 : -1; }
void pool_free(void *p, size_t) { (void)hipFree(p); }
hipError_t stream_sync(hipStream_t s) { return hipStreamSynchronize(s); }
}  // namespace mh

using namespace mh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <typename F> static float time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 16384;
    const int64_t H = N, W = N;
    const size_t n = (size_t)H * W;
    double *z; uint8_t *ref, *out; unsigned *nodir;
    CK(hipMalloc(&z, n * 8)); CK(hipMalloc(&ref, n)); CK(hipMalloc(&out, n)); CK(hipMalloc(&nodir, 4));
    std::vector<double> hz(n);
    std::vector<uint8_t> h1(n), h2(n);
    for (int pass = 0; pass < 2; ++pass) {
        std::mt19937_64 rng(7 + pass);
        if (pass == 1) {   // smooth-ish surface + noise: few ties, no interior NODIR (what the pipeline's no-flats surfaces look like)
            for (size_t i = 0; i < n; ++i) hz[i] = 50.0 + 30.0 * sin(1e-3 * (double)(i % W)) * cos(1.3e-3 * (double)(i / W)) + 1e-3 * (double)(rng() >> 40) / 16777216.0 - 1e-2 * (double)(i / W) - 1.1e-2 * (double)(i % W);
        } else {           // ties everywhere, some NaN / inf
            for (size_t i = 0; i < n; ++i) { uint64_t r = rng(); hz[i] = (double)(r & 3); if ((r >> 20) % 5000 == 0) hz[i] = NAN; if ((r >> 20) % 7001 == 0) hz[i] = INFINITY; }
        }
        CK(hipMemcpy(z, hz.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemset(nodir, 0, 4));
        hipLaunchKernelGGL((d8_kernel<128, false, 1>), dim3((unsigned)cdiv(cdiv(W, STRIP), 4), (unsigned)cdiv(H, 128)), dim3(256), 0, 0, z, ref, H, W, 1, (int64_t)0, H, nodir, (int64_t)0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h1.data(), ref, n, hipMemcpyDeviceToHost));
        unsigned nd_ref; CK(hipMemcpy(&nd_ref, nodir, 4, hipMemcpyDeviceToHost));
        auto check = [&](const char *name) {
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h2.data(), out, n, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < n; ++i) if (h1[i] != h2[i]) { if (!bad) first = i; ++bad; }
            unsigned nd; CK(hipMemcpy(&nd, nodir, 4, hipMemcpyDeviceToHost));
            printf("  check pass %d %-28s %s (%zu differ, first at %zu: %d vs %d) nodir %s\n", pass, name, bad ? "FAIL" : "ok", bad, first, h1[first], h2[first], (nd != 0) == (nd_ref != 0) ? "ok" : "FAIL");
            CK(hipMemset(out, 0xee, n));
        };
#define VARIANT(CPL_, PF_, RPW_, WPS_) do { CK(hipMemset(nodir, 0, 4)); d8_launch<CPL_, PF_, RPW_, WPS_>(z, out, H, W, 1, 0, 0, H, nodir); check("cpl" #CPL_ " pf" #PF_ " rpw" #RPW_ " wps" #WPS_); } while (0)
        VARIANT(4, 1, 128, 4); VARIANT(4, 2, 64, 4); VARIANT(2, 1, 128, 6); VARIANT(2, 2, 64, 6); VARIANT(4, 1, 16, 4); VARIANT(4, 1, 7, 4); VARIANT(2, 2, 8, 6);
#undef VARIANT
    }
    // timing on the last surface reloaded as the smooth one (values do not steer control flow)
    const double gb = 9.0 * (double)n / 1e9;
    auto report = [&](const char *name, float ms) { printf("%-34s %8.4f ms  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80.0); fflush(stdout); };
    report("library d8_kernel<128,nt0,pf1>", time_ms([&] { hipLaunchKernelGGL((d8_kernel<128, false, 1>), dim3((unsigned)cdiv(cdiv(W, STRIP), 4), (unsigned)cdiv(H, 128)), dim3(256), 0, 0, z, ref, H, W, 1, (int64_t)0, H, nodir, (int64_t)0); }, 20));
    // a producer writes the surface top to bottom right before D8 reads it, like the no-flats stage does in the pipeline
    double *zsrc; CK(hipMalloc(&zsrc, n * 8)); CK(hipMemcpy(zsrc, z, n * 8, hipMemcpyDeviceToDevice));
    auto timed = [&](const char *name, auto f) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        std::vector<float> t;
        for (int i = 0; i < 12; ++i) {
            CK(hipMemcpyAsync(z, zsrc, n * 8, hipMemcpyDeviceToDevice, 0));
            CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (i >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        report(name, t[t.size() / 2]);
    };
#define TIME(CPL_, PF_, RPW_, WPS_, W_, MODE_) timed("cpl" #CPL_ " nb" #PF_ " rpw" #RPW_ " wps" #WPS_ " wide " #W_ " mode" #MODE_, [&] { d8_launch<CPL_, PF_, RPW_, WPS_, W_, MODE_>(z, out, H, W, 1, 0, 0, H, nodir); })
    TIME(4, 1, 16, 4, true, 0); TIME(4, 1, 16, 4, false, 0); TIME(4, 1, 16, 4, true, 1); TIME(4, 1, 16, 4, true, 2);
    TIME(4, 1, 32, 4, true, 0); TIME(4, 1, 128, 4, true, 0); TIME(4, 2, 16, 4, true, 0); TIME(2, 1, 16, 6, true, 0); TIME(2, 2, 16, 6, true, 0);
    return 0;
}
