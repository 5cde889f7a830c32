// fir_up4k.hip -- multirate_FIR.up (multirate_helper.py:112-118, lfilter(b, [1], L * upsample(x, L))) in the frequency
// domain, one workgroup per INPUT tile: the tile is loaded and transformed ONCE, its spectrum stays in registers, and the L
// phase filters  y[i L + p] = L sum_t b[p + L t] x[i - t]  are L pointwise products + inverse transforms of that one spectrum.
// A thread keeps the results of up to four consecutive phases and stores them together: 32 contiguous bytes of y per lane
// (L = 4: a wave's store instruction pair writes 2 KiB of consecutive output) instead of one 8-byte element between the
// elements of other phases -- the stride-L stores of the walk over (tile, phase) pairs this replaces (fir_ols.hip, UP) put
// 2 - 2.5 x the output bytes on the fabric and re-loaded + re-transformed the tile L times.
//
// Tile: 4096 complex64 points, 256 threads x 16 points (ols4k_core.hpp); V = 4096 - OV inputs -> V L outputs per tile,
// OV = taps per phase - 1 rounded up to 256.  float32 signals with real taps run their phases in PAIRS: x * (h_2q + i h_2q+1)
// = y_2q + i y_2q+1 is one complex pass over the real tile, and its output IS the interleaved pair (y[i L + 2q], y[i L + 2q + 1])
// as one 8-byte element -- four passes are 32 contiguous bytes again.
// HBM traffic = the input once (+ OV / V overlap) and the output once; the L phase tables (32 KiB each) stream from L2.
// Algorithmic bytes: 8 B x (n + n L) complex64, 4 B x (n + n L) float32.
#include "skdsp_internal.hpp"
#include "ols4k_tables.hpp"

namespace skdsp {

using namespace ols4k;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));

struct Up4kPlan {
    int L = 0, T = 0, ov = 0, V = 0, passes = 0;
    bool pairs = false;
    float2 *tw = nullptr, *T2 = nullptr;
    float4 *Hp = nullptr;
};

struct Up4kArgs {
    const void *x;
    void *y;
    int64_t n, n_hist;       // input samples; valid history in front of x[0]
    const float2 *tw, *T2;
    const float4 *Hp;        // passes x 2048 float4
    int ov, V, a0;           // a0 = ov / 256: first stored 256-block of a tile
    int passes;              // complex passes per tile: L (complex64) or ceil(L / 2) (float32: two phases per pass)
    int row_bytes;           // bytes of one input sample's L outputs
    int odd_tail;            // float32, odd L: the last pass carries ONE phase (4 bytes)
    int aligned;             // x and y element-aligned
    int64_t ntiles;
    int staged;              // four-pass groups leave through the wave-private staging image (option fir_up4k_staged; 0 = as they lie, for A/B)
    CarefulFir cf;           // the filter as the exact path of a poisoned tile reads it (careful.hpp)
};

// x[in0 + 256 a + t] -> v[a]; zero outside [-n_hist, n).  XR: a float32 signal into the real parts (the imaginary parts are
// set where the tile is transformed).  A tile is "interior" when all 4096 samples exist and the accesses are element-aligned.
__device__ __forceinline__ bool up4k_interior(const Up4kArgs &A, int64_t tile)
{
    const int64_t in0 = tile * A.V - A.ov;
    return A.aligned && in0 >= -A.n_hist && in0 + kN <= A.n;
}
template <bool XR> __device__ __forceinline__ void up4k_load_interior(const Up4kArgs &A, int64_t tile, int t, cf *v)
{
    const int64_t in0 = tile * A.V - A.ov;
    int tt = t;   // (opaque copy: the 16 addresses are rebuilt per tile instead of living in registers across the tile loop)
    asm volatile("" : "+v"(tt));
    if (XR) {
        const float *xp = reinterpret_cast<const float *>(A.x) + in0;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a].x = __builtin_nontemporal_load(xp + (unsigned)(a * 256 + tt));
    } else {
        const v2f_t *xp = reinterpret_cast<const v2f_t *>(A.x) + in0;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const v2f_t r = __builtin_nontemporal_load(xp + (unsigned)(a * 256 + tt));
            v[a] = make_float2(r.x, r.y);
        }
    }
}
// (out of line, through a small array in scratch: the first and last tiles of a signal only; arguments by value, so that the
// kernel's argument block stays in scalar registers)
template <bool XR> __device__ __noinline__ void up4k_load_edge(const void *x, int64_t in0, int64_t n_hist, int64_t n, int t, cf *v)
{
    for (int a = 0; a < 16; ++a) {
        const int64_t g = in0 + 256 * a + t;
        cf val = make_float2(0.f, 0.f);
        if (g >= -n_hist && g < n) {
            if (XR) val.x = reinterpret_cast<const float *>(x)[g];
            else val = reinterpret_cast<const cf *>(x)[g];
        }
        v[a] = val;
    }
}

// The passes g0 .. g0 + CNT - 1 of this tile leave together: sample i = out0 + 256 (a - a0) + t owns the 8 CNT bytes at
// y + i row_bytes + 8 g0 (the last 4 of them missing when TAIL: the odd L's single last phase of a float32 signal).
template <int CNT, bool TAIL> __device__ __forceinline__ void up4k_store(const Up4kArgs &A, int64_t tile, int g0, int t, const cf *out)
{
    int a0 = A.a0;   // (opaque copies: nothing of the store addressing is hoisted out of the tile loop)
    asm volatile("" : "+s"(a0));
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int64_t out0 = tile * A.V;
    const int64_t left = A.n - out0;
    const bool whole = left >= A.V;
    const int lim = (int)(left > kN ? kN : left) - tt;               // this lane's samples 256 (a - a0) < lim exist
    char *ub = reinterpret_cast<char *>(A.y) + out0 * A.row_bytes + 8 * g0;   // uniform
    const unsigned lane_off = (unsigned)tt * (unsigned)A.row_bytes;
    const size_t step = (size_t)256 * A.row_bytes;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        if (a < a0) continue;
        if (!whole && 256 * (a - a0) >= lim) continue;
        char *p = ub + (size_t)(a - a0) * step + lane_off;
        if constexpr (CNT >= 2) {
            v4f_t w;
            w.x = out[0 + a].x; w.y = out[0 + a].y; w.z = out[16 + a].x; w.w = out[16 + a].y;
            if (TAIL && CNT == 2) {
                *reinterpret_cast<v2f_t *>(p) = v2f_t{w.x, w.y};
                *reinterpret_cast<float *>(p + 8) = w.z;
            } else {
                *reinterpret_cast<v4f_t *>(p) = w;
            }
        }
        if constexpr (CNT == 4) {
            v4f_t w;
            w.x = out[32 + a].x; w.y = out[32 + a].y; w.z = out[48 + a].x; w.w = out[48 + a].y;
            if (TAIL) {
                *reinterpret_cast<v2f_t *>(p + 16) = v2f_t{w.x, w.y};
                *reinterpret_cast<float *>(p + 24) = w.z;
            } else {
                *reinterpret_cast<v4f_t *>(p + 16) = w;
            }
        }
        if constexpr (CNT == 1 || CNT == 3) {
            constexpr int j = CNT - 1;
            if (TAIL) *reinterpret_cast<float *>(p + 8 * j) = out[16 * j + a].x;
            else *reinterpret_cast<v2f_t *>(p + 8 * j) = v2f_t{out[16 * j + a].x, out[16 * j + a].y};
        }
    }
}

// Four passes, no tail: the same bytes through a wave-private corner of the LDS, so that a lane PAIR writes a sample's 32
// contiguous bytes.  Stored as they lie (each lane its own sample's first 16 bytes, then the second 16) a store instruction
// carries 16 bytes into each of 64 different 64-byte blocks once the rows are 64 bytes or longer, and the vector L1 forwards
// one write request per block: measured 278 clocks per store instruction at L = 8 and 379 at L = 12 against 68 at L = 4 (two lanes
// per block) -- 0.27 and 0.37 ms per 2^26 outputs for the stores alone, whether or not a row's pieces leave at the same time
//.  Through the staging image lane l of the i-th instruction writes half l & 1 of sample 32 i + (l >> 1):
// half the requests at any L, and at L = 4 whole 64-byte blocks (1 KiB of consecutive output per instruction).  Wave-private:
// the 64 lanes of a wave own the 64 samples they stage, LDS operations of a wave execute in order -- no barrier.
__device__ __forceinline__ void up4k_store4_staged(const Up4kArgs &A, int64_t tile, int g0, int t, const cf *out, float4 *stage /* this wave's 128 float4 */)
{
    int a0 = A.a0;   // (opaque copies: nothing of the store addressing is hoisted out of the tile loop)
    asm volatile("" : "+s"(a0));
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int lane = tt & 63, wv = tt >> 6;
    const int64_t out0 = tile * A.V;
    const int64_t left = A.n - out0;
    const bool whole = left >= A.V;
    const int lim = (int)(left > kN ? kN : left);                    // tile-local samples s < lim exist
    char *ub = reinterpret_cast<char *>(A.y) + out0 * A.row_bytes + 8 * g0 + (size_t)(64 * wv) * A.row_bytes;   // uniform per wave
    const unsigned off0 = (unsigned)(lane >> 1) * (unsigned)A.row_bytes + 16u * (unsigned)(lane & 1);            // sample lane >> 1 (instruction 0)
    const unsigned off1 = off0 + 32u * (unsigned)A.row_bytes;                                                    // sample 32 + (lane >> 1)
    const size_t step = (size_t)256 * A.row_bytes;
    const int s0 = 64 * wv + (lane >> 1);   // tile-local sample of instruction 0 inside its 256-block (instruction 1: + 32)
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        if (a < a0) continue;
        if (!whole && 256 * (a - a0) + 64 * wv >= lim) continue;    // (uniform per wave)
        stage[2 * lane] = make_float4(out[0 + a].x, out[0 + a].y, out[16 + a].x, out[16 + a].y);
        stage[2 * lane + 1] = make_float4(out[32 + a].x, out[32 + a].y, out[48 + a].x, out[48 + a].y);
        const float4 w0 = stage[lane], w1 = stage[64 + lane];
        char *p = ub + (size_t)(a - a0) * step;
        if (whole || 256 * (a - a0) + s0 < lim) *reinterpret_cast<v4f_t *>(p + off0) = v4f_t{w0.x, w0.y, w0.z, w0.w};
        if (whole || 256 * (a - a0) + s0 + 32 < lim) *reinterpret_cast<v4f_t *>(p + off1) = v4f_t{w1.x, w1.y, w1.z, w1.w};
    }
}

// "these are the results, in these registers, now": without it hipcc carries a finished pass in a form of its own (more live registers per
// pass than its 32 results: fir_up2k.hip measured 24 instead of 16 for its eight)
__device__ __forceinline__ void up4k_pin(cf *v)
{
#pragma unroll
    for (int i = 0; i < 16; i += 8)
        asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y), "+v"(v[i + 2].x), "+v"(v[i + 2].y), "+v"(v[i + 3].x), "+v"(v[i + 3].y),
                     "+v"(v[i + 4].x), "+v"(v[i + 4].y), "+v"(v[i + 5].x), "+v"(v[i + 5].y), "+v"(v[i + 6].x), "+v"(v[i + 6].y), "+v"(v[i + 7].x), "+v"(v[i + 7].y));
}
// volatile 16-byte load: keeps the request at its program position (the scheduler would otherwise sink a prefetch to its first use)
__device__ __forceinline__ float4 up4k_vld(const volatile float4 *p)
{
    float4 r;
    r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
    return r;
}
// this thread's 8 float4 of pass q's transfer function
__device__ __forceinline__ void up4k_load_H(const Up4kArgs &A, int q, int t, float4 *hh)
{
    int tt = t;   // (opaque copy: the addresses are rebuilt where they are used)
    asm volatile("" : "+v"(tt));
    const volatile float4 *hp = reinterpret_cast<const volatile float4 *>(A.Hp) + (size_t)q * 2048;
#pragma unroll
    for (int k = 0; k < 8; ++k) hh[k] = up4k_vld(hp + (unsigned)(k * 256 + tt));
}
// (a table that is NOT requested: a defined value, so that the previous one does not stay alive across the whole loop body)
__device__ __forceinline__ void up4k_no_H(float4 *hh)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) hh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// "the values must be in their registers HERE": makes hipcc place its wait for a prefetch at this point
__device__ __forceinline__ void up4k_settle(const float4 *hh)
{
#pragma unroll
    for (int k = 0; k < 8; k += 4)
        asm volatile("" ::"v"(hh[k].x), "v"(hh[k].y), "v"(hh[k].z), "v"(hh[k].w), "v"(hh[k + 1].x), "v"(hh[k + 1].y), "v"(hh[k + 1].z), "v"(hh[k + 1].w),
                     "v"(hh[k + 2].x), "v"(hh[k + 2].y), "v"(hh[k + 2].z), "v"(hh[k + 2].w), "v"(hh[k + 3].x), "v"(hh[k + 3].y), "v"(hh[k + 3].z), "v"(hh[k + 3].w)
                     : "memory");
}
template <bool XR> __device__ __forceinline__ void up4k_settle_x(const cf *v)
{
    if constexpr (XR) {
#pragma unroll
        for (int i = 0; i < 16; i += 8)
            asm volatile("" ::"v"(v[i].x), "v"(v[i + 1].x), "v"(v[i + 2].x), "v"(v[i + 3].x), "v"(v[i + 4].x), "v"(v[i + 5].x), "v"(v[i + 6].x), "v"(v[i + 7].x) : "memory");
    } else {
#pragma unroll
        for (int i = 0; i < 16; i += 8)
            asm volatile("" ::"v"(v[i].x), "v"(v[i].y), "v"(v[i + 1].x), "v"(v[i + 1].y), "v"(v[i + 2].x), "v"(v[i + 2].y), "v"(v[i + 3].x), "v"(v[i + 3].y),
                         "v"(v[i + 4].x), "v"(v[i + 4].y), "v"(v[i + 5].x), "v"(v[i + 5].y), "v"(v[i + 6].x), "v"(v[i + 6].y), "v"(v[i + 7].x), "v"(v[i + 7].y)
                         : "memory");
    }
}

// A poisoned tile (one inf / nan among its 4096 inputs makes every result of every pass non-finite, where the reference confines the sample
// to the Ntaps outputs that multiply it): the thread recomputes the rows it stored -- samples out0 + 256 (a - a0) + t, all L phases each -- by
// the reference's own sum (careful.hpp).  Rows are written by their own thread or, through the wave-private staging image, by its own wave,
// so the second store follows the first in program order: no barrier.
template <bool XR> __device__ __forceinline__ void up4k_careful_rows(const void *x, void *y, int64_t n, int64_t n_hist, const CarefulFir cf, int L, int64_t out0, int a0, int t)
{
#pragma unroll 1
    for (int a = a0; a < 16; ++a) {
        const int64_t i = out0 + 256 * (a - a0) + t;
        if (i >= n) break;
        careful_up_row<XR>(x, y, n_hist, cf, L, i);
    }
}

// Persistent: 2 workgroups per CU walk the input tiles (XCD-contiguous runs per round, like ols_tile_kernel: neighbouring
// tiles share their overlap through that XCD's L2).  G = passes whose results a thread holds before it stores.
//
// Vector-memory order.  vmcnt retires in order, so a wait for a load issued BEHIND a store burst is a wait for the stores'
// acknowledgements -- the first version of this kernel (pass tables requested at the top of every pass, the tile at the top of
// every tile) paid each group's whole store drain that way: + 0.09 ms per 2^26 outputs at L = 4, + 0.30 ms at L = 12.  Now
//   * the next tile's samples are requested behind the H product of the tile's LAST pass -- they land in the registers of the
//     spectrum, which is dead by then, have that pass's inverse transform to arrive, and are waited for in front of the stores;
//   * the table of the next group's first pass is requested AND waited for in front of the stores (an exposed L2 round trip);
//   * the tables of a group's other passes are requested at the top of their pass: an L2 round trip that is exposed, but never
//     behind a store.  (Requested a pass ahead -- behind the previous H product, or in front of the previous pass's last
//     exchange -- they are 32 more live registers at the peak: 153 - 210 spilled registers with four results per thread.)
template <bool XR, int G> __global__ __launch_bounds__(256, 2) void up4k_kernel(Up4kArgs A)
{
    __shared__ cf img[kImgUnits];
    __shared__ cf T2f[kT2Units], T2t[kT2Units];
    __shared__ cf twl[kTwUnits];
    __shared__ float4 stage[G == 4 ? 4 * 128 : 1];   // 2 KiB per wave: up4k_store4_staged
    __shared__ unsigned long long up_noted;          // poisoned tiles, by walk step (careful.hpp)
    const int t = threadIdx.x;
    if (t == 0) up_noted = 0;
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
    cf Z[16];          // the tile's samples, then its spectrum, then (behind the last H product) the next tile's samples
    float4 hh[G][8];   // the tables of the current group's passes
    bool have_x = false;   // Z holds the samples of `tile` (requested a pass ahead: interior tiles only)
    if (tile < A.ntiles) {
        up4k_load_H(A, 0, t, hh[0]);
    } else {
        up4k_no_H(hh[0]);
    }
    for (; tile < A.ntiles; tile += gridDim.x) {
        const bool has_next = tile + gridDim.x < A.ntiles;
        const bool pre_next = has_next && up4k_interior(A, tile + gridDim.x);
        if (!have_x) {   // the first tile of this workgroup, and tiles at the ends of the signal (guarded accesses)
            if (up4k_interior(A, tile)) {
                up4k_load_interior<XR>(A, tile, t, Z);
            } else {
                cf e[16];
                up4k_load_edge<XR>(A.x, tile * A.V - A.ov, A.n_hist, A.n, t, e);
#pragma unroll
                for (int a = 0; a < 16; ++a) Z[a] = e[a];
            }
        }
        have_x = pre_next;
        if constexpr (XR) {
#pragma unroll
            for (int a = 0; a < 16; ++a) Z[a].y = 0.f;
        }
        fwd_pass1(t, Z, twl, img);
        __syncthreads();
        fwd_pass2(t, T2f, img);
        fwd_pass3(t, img, Z);
        bool poisoned = false;
        // (the first inverse pass writes the rows this thread's 16-lane group just read: wave-local, no barrier)
        for (int g0 = 0; g0 < A.passes; g0 += G) {
            const int cnt = A.passes - g0 < G ? A.passes - g0 : G;
            const bool last_group = g0 + cnt == A.passes;
            cf out[G * 16];
            static_for<0, G>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if (j < cnt) {
                    cf P[16];
                    if (j > 0) up4k_load_H(A, g0 + j, t, hh[j]);
                    mul_H(hh[j], Z, P);
                    if (j == cnt - 1 && last_group && pre_next)   // the tile's last pass: the spectrum is dead, the next tile's samples land in its registers
                        up4k_load_interior<XR>(A, tile + gridDim.x, t, Z);
                    inv_pass3(t, T2t, img, P);
                    inv_pass2(t, img);
                    __syncthreads();
                    inv_pass1(t, twl, img, out + 16 * j);
                    __syncthreads();   // every wave has read the image before the next pass (or tile) overwrites it
                    up4k_pin(out + 16 * j);
                }
            });
            // the table of the next group's first pass: requested and waited for in front of the stores (an L2 round trip, exposed; held
            // across the last inverse transform it costs 32 registers at the peak and the kernel spills)
            if (!last_group) up4k_load_H(A, g0 + G, t, hh[0]);
            else if (has_next) up4k_load_H(A, 0, t, hh[0]);
            else up4k_no_H(hh[0]);
            up4k_settle(hh[0]);
            if (last_group && pre_next) up4k_settle_x<XR>(Z);
            poisoned |= not_finite(out[15].x) | not_finite(out[15].y);
            const bool tail = A.odd_tail && last_group;
            auto store = [&](auto tc) __attribute__((always_inline)) {
                constexpr bool TAIL = decltype(tc)::value;
                if constexpr (G == 4) {
                    if (cnt == 4) {
                        if (!TAIL && A.staged) up4k_store4_staged(A, tile, g0, t, out, stage + 128 * (t >> 6));
                        else up4k_store<4, TAIL>(A, tile, g0, t, out);
                        return;
                    }
                    if (cnt == 3) { up4k_store<3, TAIL>(A, tile, g0, t, out); return; }
                }
                if (cnt == 2) up4k_store<2, TAIL>(A, tile, g0, t, out);
                else up4k_store<1, TAIL>(A, tile, g0, t, out);
            };
            if constexpr (XR) {
                if (tail) store(std::true_type{}); else store(std::false_type{});
            } else {
                store(std::false_type{});
            }
        }
        if (__builtin_expect(__any(poisoned), 0)) careful_note(&up_noted, (tile - tile_first()) / gridDim.x);
    }
    const unsigned long long noted = careful_noted(&up_noted);
    if (__builtin_expect(noted != 0, 0)) {
        int64_t k = 0;
        for (int64_t tl = tile_first(); tl < A.ntiles; tl += gridDim.x, ++k)
            if (careful_step_noted(noted, k)) up4k_careful_rows<XR>(A.x, A.y, A.n, A.n_hist, A.cf, A.row_bytes / (XR ? 4 : 8), tl * A.V, A.a0, t);
    }
}

struct Up4kPlanList { std::vector<Up4kPlan *> plans; };

static void up4k_free_plan(Up4kPlan *p)
{
    if (!p) return;
    if (p->tw) (void)hipFree(p->tw);
    if (p->T2) (void)hipFree(p->T2);
    if (p->Hp) (void)hipFree(p->Hp);
    delete p;
}

void fir_up4k_free(void *list)
{
    Up4kPlanList *l = static_cast<Up4kPlanList *>(list);
    if (!l) return;
    for (Up4kPlan *p : l->plans) up4k_free_plan(p);
    delete l;
}

// complex64 (any taps) or float32 with real taps; per phase at most 2049 taps (half a tile of overlap)
bool fir_up4k_supported(const FirHandle *h, int L)
{
    // (a plan holds one 32 / 16 KiB table per pass, built on first use under the handle's lock: 256 passes are 8 / 4 MiB and a few ms of host transforms;
    // beyond that the walk over (tile, phase) pairs and the polyphase kernels serve the call)
    if (L < 2 || L > 256) return false;
    const int T = up_taps_per_phase(h->ntaps, L);
    if (T - 1 > 2048) return false;
    return h->dtype == SKDSP_C64 || (h->dtype == SKDSP_F32 && !h->taps_complex);
}

static int up4k_plan(FirHandle *h, int L, Up4kPlan **out)
{
    if (!h->up4k) h->up4k = new Up4kPlanList();
    Up4kPlanList *l = static_cast<Up4kPlanList *>(h->up4k);
    for (Up4kPlan *p : l->plans)
        if (p->L == L) { *out = p; return SKDSP_OK; }
    Up4kPlan *p = new Up4kPlan();
    p->L = L;
    p->pairs = h->dtype == SKDSP_F32;
    p->T = up_taps_per_phase(h->ntaps, L);
    p->ov = ((p->T - 1 + 255) / 256) * 256;
    if (p->ov == 0) p->ov = 256;
    p->V = kN - p->ov;
    p->passes = up_passes(L, p->pairs);
    std::vector<float2> tw, T2;
    std::vector<float4> Hp;
    make_tw(tw);
    make_T2(T2);
    make_up_tables(h->taps_host.data(), h->ntaps, h->taps_complex ? 2 : 1, L, p->pairs, Hp);
    hipError_t e;
    if ((e = hipMalloc((void **)&p->tw, tw.size() * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc((void **)&p->T2, T2.size() * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc((void **)&p->Hp, Hp.size() * sizeof(float4))) != hipSuccess ||
        (e = hipMemcpy(p->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->T2, T2.data(), T2.size() * sizeof(float2), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->Hp, Hp.data(), Hp.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess) {
        up4k_free_plan(p);
        return hip_fail(e, "up4k tables", __FILE__, __LINE__);
    }
    l->plans.push_back(p);
    *out = p;
    return SKDSP_OK;
}

int fir_up4k_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, void *y, hipStream_t s)
{
    note_path("fir_up4k");
    if (n <= 0) return SKDSP_OK;
    SK_CHECK(fir_up4k_supported(h, L), SKDSP_ERR_UNSUPPORTED, "fir_up4k: needs complex64 (or float32 with real taps), 2 <= L <= 4096, at most 2049 taps per phase");
    Up4kPlan *p = nullptr;
    int rc = up4k_plan(h, L, &p);
    if (rc) return rc;
    const int esz = h->dtype == SKDSP_F32 ? 4 : 8;
    Up4kArgs A;
    A.x = x; A.y = y; A.n = n; A.n_hist = n_hist;
    A.tw = p->tw; A.T2 = p->T2; A.Hp = p->Hp;
    A.ov = p->ov; A.V = p->V; A.a0 = p->ov / 256;
    A.passes = p->passes;
    A.row_bytes = L * esz;
    A.odd_tail = p->pairs && (L & 1);
    A.aligned = ((((uintptr_t)x) | ((uintptr_t)y)) & (esz - 1)) == 0;
    A.ntiles = (n + p->V - 1) / p->V;
    A.staged = opt().fir_up4k_staged;
    if ((rc = fir_careful(h, &A.cf))) return rc;
    SK_CHECK(A.ntiles < (int64_t)1 << 31, SKDSP_ERR_BADARG, "fir_up4k: too many tiles");
    int64_t grid = 2 * (int64_t)ctx().num_cus;
    const int reserve_wgs = opt().ols_reserve;
    if (reserve_wgs > 0 && grid >= 4 * (int64_t)reserve_wgs) grid -= reserve_wgs;
    if (grid > A.ntiles) grid = A.ntiles;
    const int G = opt().fir_up4k_group;
    if (p->pairs) {
        if (G == 2) hipLaunchKernelGGL((up4k_kernel<true, 2>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((up4k_kernel<true, 4>), dim3((unsigned)grid), dim3(256), 0, s, A);
    } else {
        if (G == 2) hipLaunchKernelGGL((up4k_kernel<false, 2>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((up4k_kernel<false, 4>), dim3((unsigned)grid), dim3(256), 0, s, A);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

}  // namespace skdsp
