// fill.hip -- depression filling on gfx950: fill.fill_terrain (f32) and fill.fill_terrain_no_flats (f64).
//
// Reference semantics (fill.py:112-232, sweeps _fill.pyx:28-124): the greatest fixed point of
//     plain   : W = max(dtm, min(W, 8 neighbours))
//     no-flats: W = max(dtm, min(W, min4(diagonal nbrs) + diag, min4(edge nbrs) + short))
// on interior cells, border cells fixed to dtm, start from +inf.  Both operators are monotone and every
// intermediate state stays >= the fixed point, so ANY chaotic schedule of single-cell updates converges
// to the same bits as the reference's four-direction Gauss-Seidel raster sweeps (single IEEE adds, no
// FMA, no reassociation in the no-flats case).
//
// Device schedule ("tiled iterative sweep"):
//   * the raster is cut into 62x62-cell tiles; ONE WAVEFRONT owns a tile and holds its 64x64 window
//     (tile + 1-cell halo ring) entirely in VGPRs: register r = window row r, lane = window column.
//   * a tile visit runs row-sequential passes (down, up) where the vertical dependency is carried in
//     registers and the horizontal neighbours come from wave_shr/wave_shl DPP shifts; then the window is
//     transposed through a wave-private LDS scratch and the same two passes run along the other axis.
//     Cycles repeat until the tile is locally converged (or a cap is hit).
//   * changed tiles write back and raise "active" flags of the neighbours whose halo they touched; the
//     host launches rounds until no flag is raised (flags are double buffered; a round is one launch).
// Halo lanes/rows stay fixed because their "dem" register is set to their own value (plain) or because
// the update is lane-masked (no-flats).  Raster border cells have W == dem from the start and therefore
// never move; cells outside the raster are +inf and never win a min.
#include "common.hpp"

namespace mh {

namespace {

constexpr int WN = 64;  // window edge (tile + halo)
constexpr int TI = 62;  // tile interior edge
constexpr int MAXCYC = 6;
constexpr int DPP_WF_SL1 = 0x130;  // lane i <- lane i+1
constexpr int DPP_WF_SR1 = 0x138;  // lane i <- lane i-1

__device__ __forceinline__ float from_left(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WF_SR1, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WF_SL1, 0xf, 0xf, true));
}
__device__ __forceinline__ double from_left(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, DPP_WF_SR1, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, DPP_WF_SR1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, DPP_WF_SL1, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, DPP_WF_SL1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ---- one row-sequential pass over the register window -------------------------------------------
// plain f32: every lane computes; halo lanes hold d == w so they cannot move.
template <bool DOWN>
__device__ __forceinline__ void pass_plain(float (&w)[WN], const float (&d)[WN], uint64_t &any, uint64_t &first, uint64_t &last)
{
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int r = DOWN ? 1 + i : TI - i;
        const float up = w[r - 1], cu = w[r], dn = w[r + 1];
        float m = fminf(fminf(from_left(up), from_right(up)), up);
        float n = fminf(fminf(from_left(dn), from_right(dn)), dn);
        float o = fminf(fminf(from_left(cu), from_right(cu)), cu);
        m = fminf(fminf(m, n), o);
        const float nv = fmaxf(m, d[r]);
        const uint64_t ch = __ballot(nv != cu);
        w[r] = nv;
        any |= ch;
        if (r == 1) first |= ch;
        if (r == TI) last |= ch;
        __builtin_amdgcn_sched_barrier(0);  // keep rows in program order: hoisted DPP shifts blow the VGPR budget
    }
}

// no-flats f64: min4(diagonals)+diag, min4(edges)+short, self; single IEEE adds (_fill.pyx:107-117).
template <bool DOWN>
__device__ __forceinline__ void pass_noflat(double (&w)[WN], const float (&d)[WN], bool upd, double sh, double dg,
                                            uint64_t &any, uint64_t &first, uint64_t &last)
{
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int r = DOWN ? 1 + i : TI - i;
        const double up = w[r - 1], cu = w[r], dn = w[r + 1];
        double md = fmin(fmin(from_left(up), from_right(up)), fmin(from_left(dn), from_right(dn)));
        double me = fmin(fmin(up, dn), fmin(from_left(cu), from_right(cu)));
        md = __dadd_rn(md, dg);
        me = __dadd_rn(me, sh);
        double m = fmin(fmin(md, me), cu);
        double nv = fmax(m, (double)d[r]);
        nv = upd ? nv : cu;
        const uint64_t ch = __ballot(nv != cu);
        w[r] = nv;
        any |= ch;
        if (r == 1) first |= ch;
        if (r == TI) last |= ch;
        __builtin_amdgcn_sched_barrier(0);  // keep rows in program order: hoisted DPP shifts blow the VGPR budget
    }
}

// ---- 64x64 transpose of 32-bit words through a wave-private LDS scratch [64][65] ------------------
__device__ __forceinline__ void transpose32(uint32_t (&x)[WN], uint32_t *scr, int lane)
{
#pragma unroll
    for (int r = 0; r < WN; ++r) scr[r * (WN + 1) + lane] = x[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int c = 0; c < WN; ++c) x[c] = scr[lane * (WN + 1) + c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void transpose(float (&x)[WN], uint32_t *scr, int lane)
{
    uint32_t t[WN];
#pragma unroll
    for (int r = 0; r < WN; ++r) t[r] = __float_as_uint(x[r]);
    transpose32(t, scr, lane);
#pragma unroll
    for (int r = 0; r < WN; ++r) x[r] = __uint_as_float(t[r]);
}
__device__ __forceinline__ void transpose(double (&x)[WN], uint32_t *scr, int lane)
{
    uint32_t lo[WN], hi[WN];
#pragma unroll
    for (int r = 0; r < WN; ++r) {
        lo[r] = (uint32_t)__double2loint(x[r]);
        hi[r] = (uint32_t)__double2hiint(x[r]);
    }
    transpose32(lo, scr, lane);
    transpose32(hi, scr, lane);
#pragma unroll
    for (int r = 0; r < WN; ++r) x[r] = __hiloint2double((int)hi[r], (int)lo[r]);
}

template <typename WT> struct Inf;
template <> struct Inf<float> { static __device__ __forceinline__ float v() { return __builtin_inff(); } };
template <> struct Inf<double> { static __device__ __forceinline__ double v() { return __builtin_inf(); } };

// ---- one round: every active tile is visited by one wavefront -------------------------------------
// flags: cur[t] != 0 => tile t must be visited this round; visited tiles clear cur[t] and raise nxt[] of the
// neighbours whose halo they changed.  counter += number of raised flags (0 => converged).
template <typename WT, bool NOFLAT>
__global__ __launch_bounds__(256, NOFLAT ? 1 : 2) void fill_round_kernel(const float *__restrict__ dem, WT *__restrict__ W, int64_t H,
                                                        int64_t Wd, int ntr, int ntc, uint8_t *cur, uint8_t *nxt,
                                                        unsigned int *counter, int first_round, double sh, double dg,
                                                        unsigned long long *stats)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform => scalar addressing below
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= (int64_t)ntr * ntc) return;
    if (!first_round) {
        if (cur[tile] == 0) return;  // wave-uniform
        if (lane == 0) cur[tile] = 0;
    }
    uint32_t *scr = lds + wave * (WN * (WN + 1));
    const int ti = (int)(tile / ntc), tj = (int)(tile % ntc);
    const int64_t r0 = (int64_t)ti * TI, c0 = (int64_t)tj * TI;  // raster coords of window (0,0)
    const int64_t cc = c0 + lane;
    const bool col_in = cc < Wd;
    const bool col_border = (cc == 0) || (cc == Wd - 1);

    WT w[WN];
    float d[WN];
    const WT INF = Inf<WT>::v();
    // scalar row base + one shared per-lane offset: keeps the 128 row loads off the VGPR address budget
    const float *dem0 = dem + (r0 * Wd + c0);
    const WT *W0 = W + (r0 * Wd + c0);
#pragma unroll
    for (int r = 0; r < WN; ++r) {
        const int64_t rr = r0 + r;
        const bool in = col_in && rr < H;
        float dv = in ? dem0[(int64_t)r * Wd + lane] : __builtin_inff();
        WT wv;
        if (first_round) {
            const bool border = (rr == 0) || (rr == H - 1) || col_border;
            wv = (in && border) ? (WT)dv : INF;  // fill.py:102-109 _initialize_filled
        } else {
            wv = in ? W0[(int64_t)r * Wd + lane] : INF;
        }
        // NaN handling: a NaN neighbour never wins `a <= b` in the reference (_fill.pyx:22) and a NaN dem
        // cell is never updated (`fv > NaN` is false): both behave like +inf inside the window.
        if (dv != dv) dv = __builtin_inff();
        if (wv != wv) wv = INF;
        w[r] = wv;
        d[r] = dv;
    }
    const bool upd = lane >= 1 && lane <= TI;
    if constexpr (!NOFLAT) {
        // freeze the halo ring: d == w makes max(min(..), d) a no-op there (both layouts: rows 0/63, lanes 0/63)
        d[0] = w[0];
        d[WN - 1] = w[WN - 1];
        if (!upd) {
#pragma unroll
            for (int r = 0; r < WN; ++r) d[r] = w[r];
        }
    }

    uint64_t anyN = 0, topN = 0, botN = 0, anyT = 0, leftT = 0, rightT = 0;
    bool capped = true;
    int ncyc = 0;
    for (int cyc = 0; cyc < MAXCYC; ++cyc) {
        ++ncyc;
        uint64_t a1 = 0, a2 = 0;
        if constexpr (NOFLAT) {
            pass_noflat<true>(w, d, upd, sh, dg, a1, topN, botN);
            pass_noflat<false>(w, d, upd, sh, dg, a1, topN, botN);
        } else {
            pass_plain<true>(w, d, a1, topN, botN);
            pass_plain<false>(w, d, a1, topN, botN);
        }
        transpose(w, scr, lane);
        transpose(d, scr, lane);
        if constexpr (NOFLAT) {
            pass_noflat<true>(w, d, upd, sh, dg, a2, leftT, rightT);
            pass_noflat<false>(w, d, upd, sh, dg, a2, leftT, rightT);
        } else {
            pass_plain<true>(w, d, a2, leftT, rightT);
            pass_plain<false>(w, d, a2, leftT, rightT);
        }
        transpose(w, scr, lane);
        transpose(d, scr, lane);
        anyN |= a1;
        anyT |= a2;
        if ((a1 | a2) == 0) {
            capped = false;
            break;
        }
    }

    const bool changed = (anyN | anyT) != 0;
    if (lane == 0) {  // schedule statistics: tile visits, local cycles
        atomicAdd(&stats[0], 1ull);
        atomicAdd(&stats[1], (unsigned long long)ncyc);
    }
    if (changed || first_round) {
        // interior write-back.  Raster border cells never move: they are written once (first round, straight
        // from dem, including row 0 / column 0 which only ever sit in a halo ring) and skipped afterwards.
        WT *Wst = W + (r0 * Wd + c0);
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            const int64_t rr = r0 + r;
            const bool interior_pos = (r >= 1 && r <= TI) && upd;
            if (rr < H && col_in) {
                const bool border = (rr == 0) || (rr == H - 1) || col_border;
                if (!border) {
                    if (interior_pos) Wst[(int64_t)r * Wd + lane] = w[r];
                } else if (first_round) {  // wherever the border cell sits in the window (halo ring included)
                    Wst[(int64_t)r * Wd + lane] = (WT)dem0[(int64_t)r * Wd + lane];
                }
            }
        }
    }
    if (changed && lane == 0) {
        // which neighbours saw their halo change?  N layout masks are indexed by column, T layout by row.
        const uint64_t B1 = 1ull << 1, BT = 1ull << TI;
        const bool top = topN != 0 || (anyT & B1), bot = botN != 0 || (anyT & BT);
        const bool left = leftT != 0 || (anyN & B1), right = rightT != 0 || (anyN & BT);
        const bool tl = (topN & B1) || (leftT & B1), tr = (topN & BT) || (rightT & B1);
        const bool bl = (botN & B1) || (leftT & BT), br = (botN & BT) || (rightT & BT);
        unsigned int raised = 0;
        auto raise = [&](int di, int dj, bool cond) {
            const int a = ti + di, b = tj + dj;
            if (cond && a >= 0 && a < ntr && b >= 0 && b < ntc) {
                nxt[(int64_t)a * ntc + b] = 1;
                ++raised;
            }
        };
        raise(-1, 0, top);
        raise(1, 0, bot);
        raise(0, -1, left);
        raise(0, 1, right);
        raise(-1, -1, tl);
        raise(-1, 1, tr);
        raise(1, -1, bl);
        raise(1, 1, br);
        raise(0, 0, capped);
        if (raised) atomicAdd(counter, raised);
    }
}

template <typename WT> __global__ void copy_dem_kernel(const float *dem, WT *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (WT)dem[i];
}

template <typename WT, bool NOFLAT>
int fill_dev(const float *d_dem, WT *d_out, int64_t H, int64_t W, double sh, double dg, hipStream_t s, FillStats *st)
{
    if (st) st->rounds = 0;
    if (H < 3 || W < 3) {  // no interior cell: filled == dem (fill.py:102-109 with an empty sweep area)
        int64_t n = H * W;
        hipLaunchKernelGGL((copy_dem_kernel<WT>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d_dem, d_out, n);
        MH_HIP(hipGetLastError());
        return MHIP_OK;
    }
    const int ntr = (int)cdiv(H - 2, TI), ntc = (int)cdiv(W - 2, TI);
    const int64_t nt = (int64_t)ntr * ntc;
    constexpr int BATCH = 8;
    DevBuf flags, counters, statbuf;
    MH_TRY(statbuf.alloc(16));
    MH_HIP(hipMemsetAsync(statbuf.p, 0, 16, s));
    unsigned long long *d_stats = statbuf.as<unsigned long long>();
    MH_TRY(flags.alloc((size_t)nt * 2));
    MH_TRY(counters.alloc(sizeof(unsigned int) * BATCH));
    MH_HIP(hipMemsetAsync(flags.p, 0, (size_t)nt * 2, s));
    uint8_t *fl[2] = {flags.as<uint8_t>(), flags.as<uint8_t>() + nt};
    unsigned int *cnt = counters.as<unsigned int>();
    const size_t lds = 4 * WN * (WN + 1) * sizeof(uint32_t);
    const dim3 grid((unsigned)cdiv(nt, 4)), block(256);
    auto kern = fill_round_kernel<WT, NOFLAT>;
    MH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    int round = 0;
    unsigned int h_cnt[BATCH];
    // round 0 initialises every tile; then batches of BATCH rounds between host checks
    MH_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int) * BATCH, s));
    hipLaunchKernelGGL(kern, grid, block, lds, s, d_dem, d_out, H, W, ntr, ntc, fl[0], fl[1], cnt, 1, sh, dg, d_stats);
    MH_HIP(hipGetLastError());
    round = 1;
    MH_HIP(hipMemcpyAsync(h_cnt, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    bool active = h_cnt[0] != 0;
    const int64_t max_rounds = 64 + 8 * (cdiv(H, TI) + cdiv(W, TI)) * 64;  // generous safety cap
    while (active) {
        MH_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int) * BATCH, s));
        for (int b = 0; b < BATCH; ++b) {
            uint8_t *cur = fl[(round + b) & 1], *nxt = fl[(round + b + 1) & 1];
            hipLaunchKernelGGL(kern, grid, block, lds, s, d_dem, d_out, H, W, ntr, ntc, cur, nxt, cnt + b, 0, sh, dg, d_stats);
        }
        MH_HIP(hipGetLastError());
        MH_HIP(hipMemcpyAsync(h_cnt, cnt, sizeof(unsigned int) * BATCH, hipMemcpyDeviceToHost, s));
        MH_HIP(hipStreamSynchronize(s));
        int used = BATCH;
        for (int b = 0; b < BATCH; ++b)
            if (h_cnt[b] == 0) {
                used = b + 1;
                active = false;
                break;
            }
        round += used;
        if (round > max_rounds) {
            set_error("fill did not converge within %lld rounds", (long long)max_rounds);
            return MHIP_ENOTCONV;
        }
    }
    if (st) {
        unsigned long long h_stats[2] = {0, 0};
        MH_HIP(hipMemcpyAsync(h_stats, d_stats, 16, hipMemcpyDeviceToHost, s));
        MH_HIP(hipStreamSynchronize(s));
        st->rounds = round;
        st->visits = (int64_t)h_stats[0];
        st->cycles = (int64_t)h_stats[1];
        st->tiles = nt;
    }
    return MHIP_OK;
}

// ---- min/max reduction for minimum_safe_short_and_diag -------------------------------------------
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ x, int64_t n, unsigned int *out)
{
    float mx = -__builtin_inff(), mn = __builtin_inff();
    bool has_nan = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        has_nan |= (v != v);
        mx = fmaxf(mx, v);
        mn = fminf(mn, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        mn = fminf(mn, __shfl_xor(mn, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out[0], f32_key(mx));
        atomicMin(&out[1], f32_key(mn));
    }
    if (has_nan) atomicOr(&out[2], 1u);
}

__global__ void depths_kernel(const float *__restrict__ f, const float *__restrict__ d, float *__restrict__ o, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = n >> 2;
    if (i < n4) {
        const float4 a = reinterpret_cast<const float4 *>(f)[i], b = reinterpret_cast<const float4 *>(d)[i];
        float4 r;
        r.x = __fsub_rn(a.x, b.x);
        r.y = __fsub_rn(a.y, b.y);
        r.z = __fsub_rn(a.z, b.z);
        r.w = __fsub_rn(a.w, b.w);
        reinterpret_cast<float4 *>(o)[i] = r;
    }
    if (i == 0)
        for (int64_t k = n4 << 2; k < n; ++k) o[k] = __fsub_rn(f[k], d[k]);
}

}  // namespace

int fill_plain_dev(const float *d_dem, float *d_out, int64_t H, int64_t W, hipStream_t s, FillStats *st)
{
    return fill_dev<float, false>(d_dem, d_out, H, W, 0.0, 0.0, s, st);
}

int fill_noflat_dev(const float *d_dem, double *d_out, int64_t H, int64_t W, double sh, double dg, hipStream_t s,
                    FillStats *st)
{
    return fill_dev<double, true>(d_dem, d_out, H, W, sh, dg, s, st);
}

// fill.py:235-250: maxval = f64(max(|amax|,|amin|)); short = (nextafter(maxval, inf) - maxval) * 1024; diag = short * 2**0.5
int short_diag_dev(const float *d_dem, int64_t n, double *sh, double *dg, hipStream_t s)
{
    DevBuf acc;
    MH_TRY(acc.alloc(sizeof(unsigned int) * 4));
    unsigned int init[4] = {0u, 0xffffffffu, 0u, 0u};
    MH_HIP(hipMemcpyAsync(acc.p, init, sizeof(init), hipMemcpyHostToDevice, s));
    const unsigned grid = (unsigned)(cdiv(n, 256) < 4096 ? cdiv(n, 256) : 4096);
    hipLaunchKernelGGL(minmax_kernel, dim3(grid), dim3(256), 0, s, d_dem, n, acc.as<unsigned int>());
    MH_HIP(hipGetLastError());
    unsigned int h[4];
    MH_HIP(hipMemcpyAsync(h, acc.p, sizeof(h), hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    double amax = (double)key_f32(h[0]), amin = (double)key_f32(h[1]);
    if (h[2]) amax = amin = __builtin_nan("");  // np.amax/np.amin propagate NaN
    double a = __builtin_fabs(amax), b = __builtin_fabs(amin);
    double maxval = a > b ? a : b;  // python max(): first wins on ties/NaN ordering is irrelevant here
    double nextval = __builtin_nextafter(maxval, __builtin_inf());
    *sh = (nextval - maxval) * 1024.0;
    *dg = *sh * __builtin_pow(2.0, 0.5);
    return MHIP_OK;
}

int depths_dev(const float *d_filled, const float *d_dem, float *d_out, int64_t n, hipStream_t s)
{
    const int64_t n4 = n >> 2;
    const unsigned grid = (unsigned)(cdiv(n4 > 0 ? n4 : 1, 256));
    hipLaunchKernelGGL(depths_kernel, dim3(grid), dim3(256), 0, s, d_filled, d_dem, d_out, n);
    MH_HIP(hipGetLastError());
    return MHIP_OK;
}

}  // namespace mh
