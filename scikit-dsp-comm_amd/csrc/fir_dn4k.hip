// fir_dn4k.hip -- multirate_FIR.dn (multirate_helper.py:121-127, downsample(lfilter(b, [1], x), M)) in the frequency domain,
// one workgroup per OUTPUT tile: the transpose of the interpolator of fir_up4k.hip.  With the input cut into aligned blocks
// u_r[i] = x[i M + r], r = 0 .. M-1,
//   y[k] = sum_n b[n] x[k M - n] = sum_r sum_j g_r[j] u_r[k - j],   g_r[j] = b[j M - r]   (b[negative] = 0),
// i.e. M filters of ~Ntaps / M taps over the M phase signals, summed: per tile M forward transforms, their products with G_r
// ACCUMULATED in the frequency domain, and ONE inverse transform -- (M + 1) transforms per V kept outputs where the
// decimating store of the overlap-save tile (fir_ols.hip, DEC) spends 2 M and throws (M - 1) / M of its outputs away.
// A thread loads the 8 M contiguous bytes x[i M .. i M + M - 1] of each of its 16 samples (M <= 4: every byte of a line is used by
// the same wave at the same time) and transforms the M phase signals one after the other in place.
//
// Tile: 4096 complex64 points, 256 threads x 16 points (ols4k_core.hpp); V = 4096 - OV outputs per tile, OV = taps per phase - 1
// rounded up to 256.  float32 signals with real taps ride TWO output tiles per complex tile (re = tile A, im = tile B: real taps
// commute with taking real and imaginary parts), their phase signals loaded as 16 contiguous bytes per tile and sample.
// Algorithmic bytes: 8 B x (n + n / M) complex64, 4 B x (n + n / M) float32.
#include "skdsp_internal.hpp"
#include "ols4k_tables.hpp"

namespace skdsp {

using namespace ols4k;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));

struct Dn4kPlan {
    int M = 0, T = 0, ov = 0, V = 0;
    float2 *tw = nullptr, *T2 = nullptr;
    float4 *Hp = nullptr;
};

struct Dn4kArgs {
    const void *x;
    void *y;
    int64_t n, n_hist;       // input samples (at the high rate); valid history in front of x[0]
    int64_t n_out;           // floor(n / M) outputs
    const float2 *tw, *T2;
    const float4 *Hp;        // M x 2048 float4
    int ov, V, a0;           // a0 = ov / 256: first stored 256-block of a tile
    int M;
    int aligned;             // x and y element-aligned
    int64_t ntiles;          // tiles (float32: pairs of tiles) of V outputs
    CarefulFir cf;           // the filter as the exact path of a poisoned tile reads it (careful.hpp)
};

// first output of `tile` (float32: of the pair's tile A; tile B follows V outputs later), minus the overlap
template <bool REAL> __device__ __forceinline__ int64_t dn4k_k0(const Dn4kArgs &A, int64_t tile) { return tile * (REAL ? 2 : 1) * (int64_t)A.V - A.ov; }
// every input sample the tile touches exists and the accesses are element-aligned
template <bool REAL> __device__ __forceinline__ bool dn4k_interior(const Dn4kArgs &A, int64_t tile)
{
    const int64_t k0 = dn4k_k0<REAL>(A, tile);
    const int64_t k_last = k0 + kN - 1 + (REAL ? A.V : 0);
    return A.aligned && k0 * A.M >= -A.n_hist && (k_last + 1) * A.M <= A.n;
}

// The phases r0 .. r0 + CNT - 1 of the tile: in[16 j + a] = u_(r0 + j)[k0 + 256 a + t] (float32: (tile A, tile B) as (re, im)).
template <bool REAL, int CNT> __device__ __forceinline__ void dn4k_load_interior(const Dn4kArgs &A, int64_t tile, int r0, int t, cf *in)
{
    const int64_t k0 = dn4k_k0<REAL>(A, tile);
    int tt = t;   // (opaque copy: the addresses are rebuilt per tile instead of living in registers across the tile loop)
    asm volatile("" : "+v"(tt));
    if constexpr (REAL) {
        const char *xa = reinterpret_cast<const char *>(A.x) + (k0 * A.M + r0) * 4;   // uniform
        const char *xb = xa + (int64_t)A.V * A.M * 4;
        const unsigned lane_off = (unsigned)tt * (unsigned)A.M * 4u;
        const size_t step = (size_t)256 * A.M * 4;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const char *pa = xa + a * step + lane_off, *pb = xb + a * step + lane_off;
            float va[4], vb[4];
            if constexpr (CNT == 4) {
                const v4f_t wa = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(pa)), wb = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(pb));
                va[0] = wa.x; va[1] = wa.y; va[2] = wa.z; va[3] = wa.w;
                vb[0] = wb.x; vb[1] = wb.y; vb[2] = wb.z; vb[3] = wb.w;
            } else {
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    va[j] = __builtin_nontemporal_load(reinterpret_cast<const float *>(pa) + j);
                    vb[j] = __builtin_nontemporal_load(reinterpret_cast<const float *>(pb) + j);
                }
            }
#pragma unroll
            for (int j = 0; j < CNT; ++j) in[16 * j + a] = make_float2(va[j], vb[j]);
        }
    } else {
        const char *xa = reinterpret_cast<const char *>(A.x) + (k0 * A.M + r0) * 8;   // uniform
        const unsigned lane_off = (unsigned)tt * (unsigned)A.M * 8u;
        const size_t step = (size_t)256 * A.M * 8;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const char *p = xa + a * step + lane_off;
            if constexpr (CNT >= 2) {
                const v4f_t w = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
                in[a] = make_float2(w.x, w.y);
                in[16 + a] = make_float2(w.z, w.w);
            }
            if constexpr (CNT == 4) {
                const v4f_t w = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p + 16));
                in[32 + a] = make_float2(w.x, w.y);
                in[48 + a] = make_float2(w.z, w.w);
            }
            if constexpr (CNT == 1 || CNT == 3) {
                const v2f_t w = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p + 8 * (CNT - 1)));
                in[16 * (CNT - 1) + a] = make_float2(w.x, w.y);
            }
        }
    }
}
// (out of line, through a small array in scratch: the tiles at the ends of a signal only; arguments by value, so that the kernel's
// argument block stays in scalar registers)
template <bool REAL> __device__ __noinline__ void dn4k_load_edge(const void *x, int64_t k0, int64_t kB, int M, int r, int64_t n_hist, int64_t n, int t, cf *v)
{
    for (int a = 0; a < 16; ++a) {
        const int64_t g = (k0 + 256 * a + t) * M + r;
        cf val = make_float2(0.f, 0.f);
        if (REAL) {
            const int64_t gb = g + kB * M;
            if (g >= -n_hist && g < n) val.x = reinterpret_cast<const float *>(x)[g];
            if (gb >= -n_hist && gb < n) val.y = reinterpret_cast<const float *>(x)[gb];
        } else if (g >= -n_hist && g < n) {
            val = reinterpret_cast<const cf *>(x)[g];
        }
        v[a] = val;
    }
}

__device__ __forceinline__ float4 dn4k_vld(const volatile float4 *p)
{
    float4 r;
    r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
    return r;
}
__device__ __forceinline__ void dn4k_load_H(const Dn4kArgs &A, int r, int t, float4 *hh)
{
    int tt = t;
    asm volatile("" : "+v"(tt));
    const volatile float4 *hp = reinterpret_cast<const volatile float4 *>(A.Hp) + (size_t)r * 2048;
#pragma unroll
    for (int k = 0; k < 8; ++k) hh[k] = dn4k_vld(hp + (unsigned)(k * 256 + tt));
}
__device__ __forceinline__ void dn4k_settle(const float4 *hh)
{
#pragma unroll
    for (int k = 0; k < 8; k += 4)
        asm volatile("" ::"v"(hh[k].x), "v"(hh[k].y), "v"(hh[k].z), "v"(hh[k].w), "v"(hh[k + 1].x), "v"(hh[k + 1].y), "v"(hh[k + 1].z), "v"(hh[k + 1].w),
                     "v"(hh[k + 2].x), "v"(hh[k + 2].y), "v"(hh[k + 2].z), "v"(hh[k + 2].w), "v"(hh[k + 3].x), "v"(hh[k + 3].y), "v"(hh[k + 3].z), "v"(hh[k + 3].w)
                     : "memory");
}
// "these values, in these registers, now" (see fir_up2k.hip: hipcc otherwise carries a finished transform in a form of its own)
__device__ __forceinline__ void dn4k_pin(cf *v)
{
#pragma unroll
    for (int i = 0; i < 16; i += 8)
        asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y), "+v"(v[i + 2].x), "+v"(v[i + 2].y), "+v"(v[i + 3].x), "+v"(v[i + 3].y),
                     "+v"(v[i + 4].x), "+v"(v[i + 4].y), "+v"(v[i + 5].x), "+v"(v[i + 5].y), "+v"(v[i + 6].x), "+v"(v[i + 6].y), "+v"(v[i + 7].x), "+v"(v[i + 7].y));
}

// y[out0 + 256 (a - a0) + t] = v[a] for a >= a0 (float32: tile A from the real parts, tile B from the imaginary parts)
template <bool REAL> __device__ __forceinline__ void dn4k_store(const Dn4kArgs &A, int64_t tile, int t, const cf *v)
{
    int a0 = A.a0;   // (opaque copies: nothing of the store addressing is hoisted out of the tile loop)
    asm volatile("" : "+s"(a0));
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int64_t out0 = tile * (REAL ? 2 : 1) * (int64_t)A.V;
    const int64_t left = A.n_out - out0;
    const bool whole = left >= (REAL ? 2 : 1) * (int64_t)A.V;
    const int lim = (int)(left > (1 << 20) ? (1 << 20) : left) - tt;   // this lane's outputs 256 (a - a0) < lim exist
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        if (a < a0) continue;
        const int s = 256 * (a - a0);
        if constexpr (REAL) {
            float *ya = reinterpret_cast<float *>(A.y) + out0 + s, *yb = ya + A.V;   // uniform
            if (whole || s < lim) __builtin_nontemporal_store(v[a].x, ya + tt);
            if (whole || s + A.V < lim) __builtin_nontemporal_store(v[a].y, yb + tt);
        } else {
            v2f_t *yp = reinterpret_cast<v2f_t *>(A.y) + out0 + s;   // uniform
            if (whole || s < lim) __builtin_nontemporal_store(v2f_t{v[a].x, v[a].y}, yp + tt);
        }
    }
}

// pass 2 of the forward transform with the caller's registers as its working array (ols4k_core.hpp's fwd_pass2 keeps an array of its own)
__device__ __forceinline__ void dn4k_fwd_pass2(int t, const cf *T2, cf *img, cf *w)
{
    const int k1 = t >> 4, c = t & 15;
#pragma unroll
    for (int b = 0; b < 16; ++b) w[b] = img[unit(k1, b, c)];
    dft16_f(w);
    img[unit(k1, 0, c)] = w[P16(0)];
    static_for<1, 16>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        img[unit(k1, k2, c)] = cmul(w[P16(k2)], T2[k2 * 16 + c]);
    });
}

// A poisoned tile (one inf / nan among the M x 4096 inputs of a tile makes all of its outputs non-finite, where the reference confines the
// sample to the kept ones among the Ntaps outputs that multiply it): the thread recomputes the outputs it stored by the reference's own sum
// (careful.hpp) -- its own stores, in program order: no barrier.
template <bool REAL> __device__ __forceinline__ void dn4k_careful_outputs(const void *x, void *y, int64_t n_out, int64_t n_hist, const CarefulFir cf, int M, int64_t out0, int V, int a0, int t)
{
#pragma unroll 1
    for (int a = a0; a < 16; ++a) {
        const int64_t o = out0 + 256 * (a - a0) + t;
        if (REAL) {
            if (o < n_out) careful_fir_store<float, false>(reinterpret_cast<const float *>(x), n_hist, cf, 1, M, o, reinterpret_cast<float *>(y) + o);
            if (o + V < n_out) careful_fir_store<float, false>(reinterpret_cast<const float *>(x), n_hist, cf, 1, M, o + V, reinterpret_cast<float *>(y) + o + V);
        } else if (o < n_out) {
            careful_fir_store<float, true>(reinterpret_cast<const float *>(x), n_hist, cf, 1, M, o, reinterpret_cast<float *>(y) + 2 * o);
        }
    }
}

// Persistent: 2 workgroups per CU walk the output tiles (XCD-contiguous runs per round: neighbouring tiles share their overlap
// through that XCD's L2).  MS = M, a compile-time 2 ... 4: ONE load group holds all phases of a sample (32 contiguous bytes per
// lane at M = 4), and with the phase count static every phase's registers are known dead where the next one starts -- the same
// kernel with a run-time phase count and load groups of four compiled to 230 - 790 spilled registers, with it to 208 - 213
// registers and none.  Every phase is transformed in place in the registers it was loaded into; its spectrum times G_r lands
// in the accumulator; one inverse transform per tile.
// Vector-memory order (vmcnt retires in order: a wait behind a store burst waits for its acknowledgements): the next tile's
// phase signals are requested in front of the inverse transform -- their registers are dead by then -- have that transform to
// arrive and are waited for in front of the stores.  The tables are requested at the top of their phase.
template <bool REAL, int MS> __global__ __launch_bounds__(256, 2) void dn4k_kernel(Dn4kArgs A)
{
    __shared__ cf img[kImgUnits];
    __shared__ cf T2f[kT2Units], T2t[kT2Units];
    __shared__ cf twl[kTwUnits];
    __shared__ unsigned long long dn_noted;   // poisoned tiles, by walk step (careful.hpp)
    const int t = threadIdx.x;
    if (t == 0) dn_noted = 0;
    {
        const cf w = A.T2[t];
        T2f[t] = w;
        T2t[(t & 15) * 16 + (t >> 4)] = w;
#pragma unroll
        for (int k = 0; k < 15; ++k) twl[k * 256 + t] = A.tw[k * 256 + t];
    }
    __syncthreads();
    int64_t tile = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
    auto tile_first = [&]() -> int64_t { return (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x; };
    cf in[MS * 16];         // the phase signals of the tile; each is transformed in place
    bool have_in = false;   // `in` holds the phase signals of `tile` (requested ahead: interior tiles only)
    for (; tile < A.ntiles; tile += gridDim.x) {
        const bool pre_next = tile + gridDim.x < A.ntiles && dn4k_interior<REAL>(A, tile + gridDim.x);
        if (!have_in) {
            if (dn4k_interior<REAL>(A, tile)) {
                dn4k_load_interior<REAL, MS>(A, tile, 0, t, in);
            } else {   // (tiles at the ends of the signal: guarded accesses, out of line)
                static_for<0, MS>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    cf e[16];
                    dn4k_load_edge<REAL>(A.x, dn4k_k0<REAL>(A, tile), A.V, A.M, j, A.n_hist, A.n, t, e);
#pragma unroll
                    for (int a = 0; a < 16; ++a) in[16 * j + a] = e[a];
                });
            }
        }
        have_in = pre_next;
        cf acc[16];
        static_for<0, MS>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            float4 hh[8];
            dn4k_load_H(A, j, t, hh);
            cf *Z = in + 16 * j;
            fwd_pass1(t, Z, twl, img);
            __syncthreads();
            dn4k_fwd_pass2(t, T2f, img, Z);   // (wave-local from here to the product)
            fwd_pass3(t, img, Z);
            dn4k_pin(Z);
            if (j == 0) mul_H(hh, Z, acc); else mac_H(hh, Z, acc);
            dn4k_pin(acc);
            __syncthreads();   // every wave has read the image of this transform before the next one (or the inverse) overwrites it
        });
        if (pre_next) dn4k_load_interior<REAL, MS>(A, tile + gridDim.x, 0, t, in);   // the phase signals are dead: the next tile's, with the inverse transform to arrive
        inv_pass3(t, T2t, img, acc);
        inv_pass2(t, img);
        __syncthreads();
        inv_pass1(t, twl, img, acc);
        __syncthreads();
        dn4k_pin(acc);
        if (pre_next) static_for<0, MS>([&](auto jc) __attribute__((always_inline)) { dn4k_pin(in + 16 * decltype(jc)::value); });   // (waited for in front of the stores)
        dn4k_store<REAL>(A, tile, t, acc);
        if (__builtin_expect(__any(not_finite(acc[15].x) | not_finite(acc[15].y)), 0)) careful_note(&dn_noted, (tile - tile_first()) / gridDim.x);
    }
    const unsigned long long noted = careful_noted(&dn_noted);
    if (__builtin_expect(noted != 0, 0)) {
        int64_t k = 0;
        for (int64_t tl = tile_first(); tl < A.ntiles; tl += gridDim.x, ++k)
            if (careful_step_noted(noted, k))
                dn4k_careful_outputs<REAL>(A.x, A.y, A.n_out, A.n_hist, A.cf, A.M, tl * (REAL ? 2 : 1) * (int64_t)A.V, A.V, A.a0, t);
    }
}

struct Dn4kPlanList { std::vector<Dn4kPlan *> plans; };

static void dn4k_free_plan(Dn4kPlan *p)
{
    if (!p) return;
    if (p->tw) (void)hipFree(p->tw);
    if (p->T2) (void)hipFree(p->T2);
    if (p->Hp) (void)hipFree(p->Hp);
    delete p;
}

void fir_dn4k_free(void *list)
{
    Dn4kPlanList *l = static_cast<Dn4kPlanList *>(list);
    if (!l) return;
    for (Dn4kPlan *p : l->plans) dn4k_free_plan(p);
    delete l;
}

// complex64 (any taps) or float32 with real taps; per phase at most 2049 taps (half a tile of overlap)
bool fir_dn4k_supported(const FirHandle *h, int M)
{
    if (M < 2 || M > 4) return false;   // (one load group: all phases of a sample in one thread)
    const int T = dn_taps_per_phase(h->ntaps, M);
    if (T - 1 > 2048) return false;
    return h->dtype == SKDSP_C64 || (h->dtype == SKDSP_F32 && !h->taps_complex);
}

static int dn4k_plan(FirHandle *h, int M, Dn4kPlan **out)
{
    if (!h->dn4k) h->dn4k = new Dn4kPlanList();
    Dn4kPlanList *l = static_cast<Dn4kPlanList *>(h->dn4k);
    for (Dn4kPlan *p : l->plans)
        if (p->M == M) { *out = p; return SKDSP_OK; }
    Dn4kPlan *p = new Dn4kPlan();
    p->M = M;
    p->T = dn_taps_per_phase(h->ntaps, M);
    p->ov = ((p->T - 1 + 255) / 256) * 256;
    if (p->ov == 0) p->ov = 256;
    p->V = kN - p->ov;
    std::vector<float2> tw, T2;
    std::vector<float4> Hp;
    make_tw(tw);
    make_T2(T2);
    make_dn_tables(h->taps_host.data(), h->ntaps, h->taps_complex ? 2 : 1, M, Hp);
    hipError_t e;
    if ((e = hipMalloc((void **)&p->tw, tw.size() * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc((void **)&p->T2, T2.size() * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc((void **)&p->Hp, Hp.size() * sizeof(float4))) != hipSuccess ||
        (e = hipMemcpy(p->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->T2, T2.data(), T2.size() * sizeof(float2), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->Hp, Hp.data(), Hp.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess) {
        dn4k_free_plan(p);
        return hip_fail(e, "dn4k tables", __FILE__, __LINE__);
    }
    l->plans.push_back(p);
    *out = p;
    return SKDSP_OK;
}

int fir_dn4k_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int M, void *y, hipStream_t s)
{
    note_path("fir_dn4k");
    const int64_t n_out = n / M;
    if (n_out <= 0) return SKDSP_OK;
    SK_CHECK(fir_dn4k_supported(h, M), SKDSP_ERR_UNSUPPORTED, "fir_dn4k: needs complex64 (or float32 with real taps), 2 <= M <= 4, at most 2049 taps per phase");
    Dn4kPlan *p = nullptr;
    int rc = dn4k_plan(h, M, &p);
    if (rc) return rc;
    const bool real = h->dtype == SKDSP_F32;
    const int esz = real ? 4 : 8;
    Dn4kArgs A;
    A.x = x; A.y = y; A.n = n_out * M; A.n_hist = n_hist; A.n_out = n_out;
    A.tw = p->tw; A.T2 = p->T2; A.Hp = p->Hp;
    A.ov = p->ov; A.V = p->V; A.a0 = p->ov / 256;
    A.M = M;
    A.aligned = ((((uintptr_t)x) | ((uintptr_t)y)) & (esz - 1)) == 0;
    const int64_t per = (int64_t)p->V * (real ? 2 : 1);
    A.ntiles = (n_out + per - 1) / per;
    if ((rc = fir_careful(h, &A.cf))) return rc;
    SK_CHECK(A.ntiles < (int64_t)1 << 31, SKDSP_ERR_BADARG, "fir_dn4k: too many tiles");
    int64_t grid = 2 * (int64_t)ctx().num_cus;
    const int reserve_wgs = opt().ols_reserve;
    if (reserve_wgs > 0 && grid >= 4 * (int64_t)reserve_wgs) grid -= reserve_wgs;
    if (grid > A.ntiles) grid = A.ntiles;
    const dim3 g((unsigned)grid), b(256);
    auto launch = [&](auto rl) {
        constexpr bool R = decltype(rl)::value;
        if (M == 2) hipLaunchKernelGGL((dn4k_kernel<R, 2>), g, b, 0, s, A);
        else if (M == 3) hipLaunchKernelGGL((dn4k_kernel<R, 3>), g, b, 0, s, A);
        else hipLaunchKernelGGL((dn4k_kernel<R, 4>), g, b, 0, s, A);
    };
    if (real) launch(std::true_type{}); else launch(std::false_type{});
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

}  // namespace skdsp
