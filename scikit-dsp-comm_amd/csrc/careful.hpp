// careful.hpp -- the reference's arithmetic for ONE output of multirate_FIR.filter / .up / .dn, on the device.
//
// scipy.signal.lfilter(b, [1], x) (multirate_helper.py:108, 117, 125) confines a non-finite sample x[k] to the P outputs
// y[k .. k+P-1] that multiply it; the fast engines here do not: a frequency-domain tile spreads it over all of its ~8000 outputs,
// a Toeplitz product on the matrix pipe over the zero-padded part of its lag blocks, and the fp16 split of that kernel scales a
// whole window by its largest magnitude.  Every engine therefore notices a tile / window it cannot have computed correctly
// (non-finite results, a non-finite or out-of-range window maximum) and recomputes THAT tile's outputs here: the plain
// direct-form sum in float64 -- y[m] = L sum_t b[phi + L t] x[i - t], j = m M, phi = j mod L, i = j div L -- straight from
// global memory, with IEEE propagation (so the P outputs that do see the sample come out non-finite, like the reference's).
// Slow (one thread per output, Ntaps / L loads each) and rare.
//
// Where it runs.  NOT inside the engines' loops, and not as a call: a call anywhere in a kernel reserves scalar registers for its
// frame, and the loops of these kernels have none to spare (measured with the call: the headline + 3.8 %, the 127-tap kernel + 10 %,
// 8 - 45 spilled SGPRs).  A kernel that notices a poisoned tile / window only sets a bit in a workgroup-shared word (careful_note); BEHIND
// its loop, inline, it walks the noted steps again and recomputes them (careful_noted).  The hot loop keeps no state for this.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace skdsp {

struct CarefulFir {
    const double *taps;   // device: ntaps doubles (real taps) or 2 ntaps (complex, interleaved), natural order
    int ntaps;
    int taps_complex;
};

// T: float / double; CX: the signal is complex (interleaved).  Returns the output as (re, im).
// gain = 0: without the factor L (kernels that apply it themselves behind this point)
template <typename T, bool CX>
__device__ __forceinline__ void careful_fir_point(const T *__restrict__ x, int64_t n_hist, const CarefulFir c, int L, int M, int64_t m, double *re_out, double *im_out, int gain = 1)
{
    const int64_t j = m * M;
    const int64_t i = j / L;
    const int phi = (int)(j - i * L);
    double re = 0.0, im = 0.0;
#pragma unroll 1
    for (int k = phi; k < c.ntaps; k += L) {
        const int64_t g = i - (k - phi) / L;
        if (g < -n_hist) break;
        double br = c.taps_complex ? c.taps[2 * k] : c.taps[k];
        double bi = c.taps_complex ? c.taps[2 * k + 1] : 0.0;
        if (CX) {
            const double xr = (double)x[2 * g], xi = (double)x[2 * g + 1];
            if (c.taps_complex) {
                re += br * xr - bi * xi;
                im += br * xi + bi * xr;
            } else {   // (NumPy widens real taps to complex: the products with the zero imaginary part are formed, 0 * inf = nan and all)
                re += br * xr - 0.0 * xi;
                im += br * xi + 0.0 * xr;
            }
        } else {
            re += br * (double)x[g];
        }
    }
    *re_out = gain ? re * (double)L : re;
    *im_out = gain ? im * (double)L : im;
}

// the same, stored: y_elem points at the output element (T or T[2])
template <typename T, bool CX>
__device__ __forceinline__ void careful_fir_store(const T *x, int64_t n_hist, const CarefulFir &c, int L, int M, int64_t m, T *y_elem)
{
    double re, im;
    careful_fir_point<T, CX>(x, n_hist, c, L, M, m, &re, &im);
    y_elem[0] = (T)re;
    if (CX) y_elem[1] = (T)im;
}

// a value of the kind the engines check: true for inf / nan
__device__ __forceinline__ bool not_finite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ bool not_finite(double v) { return ((unsigned)(__double_as_longlong(v) >> 32) & 0x7ff00000u) == 0x7ff00000u; }
__device__ __forceinline__ bool not_finite(float2 v) { return not_finite(v.x) || not_finite(v.y); }
__device__ __forceinline__ bool not_finite(double2 v) { return not_finite(v.x) || not_finite(v.y); }

// Direct-form engines: an output is non-finite exactly when one of its products is -- including the products of the ZERO taps their
// tables are padded with (0 x inf = nan), which reach a few samples past the reference's Ntaps outputs.  They re-evaluate every
// non-finite result by the reference's sum: finite where only padding met the sample, non-finite where a real tap did.
// X: float / float2 / double / double2 (interleaved complex); returns the corrected value.
template <typename X> struct CarefulOf;
template <> struct CarefulOf<float> { using T = float; static constexpr bool CX = false; static __device__ __forceinline__ float make(double re, double) { return (float)re; } };
template <> struct CarefulOf<double> { using T = double; static constexpr bool CX = false; static __device__ __forceinline__ double make(double re, double) { return re; } };
template <> struct CarefulOf<float2> { using T = float; static constexpr bool CX = true; static __device__ __forceinline__ float2 make(double re, double im) { return make_float2((float)re, (float)im); } };
template <> struct CarefulOf<double2> { using T = double; static constexpr bool CX = true; static __device__ __forceinline__ double2 make(double re, double im) { return make_double2(re, im); } };
template <typename X>
__device__ __forceinline__ X careful_fir_value(const X *x, int64_t n_hist, const CarefulFir &c, int L, int M, int64_t m, int gain = 1)
{
    double re, im;
    careful_fir_point<typename CarefulOf<X>::T, CarefulOf<X>::CX>(reinterpret_cast<const typename CarefulOf<X>::T *>(x), n_hist, c, L, M, m, &re, &im, gain);
    return CarefulOf<X>::make(re, im);
}

// all L outputs y[i L .. i L + L - 1] of input sample i of an interpolator (the tile interpolators fir_up4k.hip / fir_up2k.hip: a thread
// -- or its own wave, through a wave-private staging image -- stores whole rows, so it may overwrite them without a barrier)
template <bool XR>
__device__ __forceinline__ void careful_up_row(const void *x, void *y, int64_t n_hist, const CarefulFir &c, int L, int64_t i)
{
    constexpr int W = XR ? 1 : 2;
#pragma unroll 1
    for (int p = 0; p < L; ++p) {
        const int64_t m = i * L + p;
        careful_fir_store<float, !XR>(reinterpret_cast<const float *>(x), n_hist, c, L, 1, m, reinterpret_cast<float *>(y) + W * m);
    }
}

// The poisoned steps of a persistent workgroup's walk: bit min(step, 63) of a workgroup-shared word (bit 63: "some step from 63 on",
// every such step is then recomputed -- exact, only slow).  careful_note: by one lane of a wave that found one; careful_noted: after the
// loop, by every thread (a barrier on both sides: the flags of all waves are in, and the loop's stores are ordered in front of the
// recomputed ones whichever thread made them).
__device__ __forceinline__ void careful_note(unsigned long long *word, int64_t step)
{
    if ((threadIdx.x & 63) == 0) atomicOr(word, 1ull << (step < 63 ? (int)step : 63));
}
__device__ __forceinline__ unsigned long long careful_noted(const unsigned long long *word)
{
    __syncthreads();
    const unsigned long long w = *reinterpret_cast<const volatile unsigned long long *>(word);
    return w;
}
__device__ __forceinline__ bool careful_step_noted(unsigned long long noted, int64_t step) { return (noted >> (step < 63 ? (int)step : 63)) & 1; }

// a contiguous run of outputs [m0, m0 + count) of y (L / M as the call has them), every thread every 256th
template <typename T, bool CX>
__device__ __forceinline__ void careful_fir_range(const void *x, void *y, int64_t n_hist, int64_t n_out, int64_t m0, int64_t count, int L, int M, const CarefulFir &cf, int tid)
{
#pragma unroll 1
    for (int64_t i = tid; i < count; i += 256) {
        const int64_t m = m0 + i;
        if (m >= n_out) break;
        careful_fir_store<T, CX>(reinterpret_cast<const T *>(x), n_hist, cf, L, M, m, reinterpret_cast<T *>(y) + (CX ? 2 : 1) * m);
    }
}
// the same run, but only the outputs that came out non-finite (direct-form engines: a non-finite result may be the sample meeting the ZERO
// padding of a tap table -- 0 x inf = nan -- a few outputs past the reference's Ntaps; read back, re-evaluated, rewritten where so).
// Behind a barrier and an agent-scope fence; the loads bypass the vector L1.
template <typename T, bool CX>
__device__ __forceinline__ void careful_fir_recheck(const void *x, void *y, int64_t n_hist, int64_t n_out, int64_t m0, int64_t count, int L, int M, const CarefulFir &cf, int tid)
{
    constexpr int W = CX ? 2 : 1;
    T *yy = reinterpret_cast<T *>(y);
#pragma unroll 1
    for (int64_t i = tid; i < count; i += 256) {
        const int64_t m = m0 + i;
        if (m >= n_out) break;
        bool bad = false;
#pragma unroll
        for (int c = 0; c < W; ++c) {
            if constexpr (sizeof(T) == 4) {
                const unsigned u = __hip_atomic_load(reinterpret_cast<const unsigned *>(yy + W * m + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad |= (u & 0x7f800000u) == 0x7f800000u;
            } else {
                const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(yy + W * m + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad |= ((unsigned)(u >> 32) & 0x7ff00000u) == 0x7ff00000u;
            }
        }
        if (bad) careful_fir_store<T, CX>(reinterpret_cast<const T *>(x), n_hist, cf, L, M, m, yy + W * m);
    }
}

// ---- frequency-domain tiles (fir_ols.hip, fir_ols64.hip) -------------------------------------------------------------------------------
// One non-finite input makes EVERY result of an overlap-save tile non-finite; the tile's outputs are then recomputed here, every thread
// taking every 256th output position, and overwrite what the tile stored (the caller places this behind the barrier that ends the tile).
struct OlsCareful {   // (scalars by value: taking the address of a kernel's argument block would move it to scratch)
    const void *x;
    void *y;
    int64_t n, n_hist, n_keep, up_pitch;
    int V, dec, up;   // outputs per tile; decimation; phases this launch walks (UP)
    int L, p0;        // UP without decimation: the interpolation factor and the first phase of this pass (XR: phases p0, p0 + 1)
    CarefulFir cf;
};
// tile: the tile, or the pair of real tiles (REAL); ph: the pass's index among the launch's phases (UP)
template <typename T, bool REAL, bool DEC, bool UP, bool XR>
__device__ __forceinline__ void careful_ols_tile(const OlsCareful c, int64_t tile, int ph, int t)
{
    constexpr bool CX = !(REAL || XR);
    constexpr int W = CX ? 2 : 1;
    const T *x = reinterpret_cast<const T *>(c.x);
    T *y = reinterpret_cast<T *>(c.y);
    const int64_t out0 = (REAL ? 2 * tile : tile) * c.V;
    const int span = REAL ? 2 * c.V : c.V;
#pragma unroll 1
    for (int g = t; g < span; g += 256) {
        const int64_t gi = out0 + g;
        if (gi >= c.n) break;
        if (!UP) {
            if (!DEC) {
                careful_fir_store<T, CX>(x, c.n_hist, c.cf, 1, 1, gi, y + W * gi);
            } else if (gi % c.dec == 0 && gi / c.dec < c.n_keep / c.dec) {
                careful_fir_store<T, CX>(x, c.n_hist, c.cf, 1, c.dec, gi / c.dec, y + W * (gi / c.dec));
            }
            continue;
        }
        if (DEC) {   // L / M: up-rate index j = gi up + ph is kept iff M divides it (such a launch walks all phases: up IS L)
            const int64_t j = gi * c.up + ph;
            if (j % c.dec == 0 && j / c.dec < c.n_keep) careful_fir_store<T, CX>(x, c.n_hist, c.cf, c.up, c.dec, j / c.dec, y + W * (j / c.dec));
            continue;
        }
#pragma unroll 1
        for (int e = 0; e < (XR ? 2 : 1); ++e) {
            const int64_t m = gi * c.L + c.p0 + e;
            T *dst = c.up_pitch ? (XR ? y + 2 * ((int64_t)ph * c.up_pitch + gi) + e : y + W * ((int64_t)ph * c.up_pitch + gi)) : y + W * m;
            careful_fir_store<T, CX>(x, c.n_hist, c.cf, c.L, 1, m, dst);
        }
    }
}

}  // namespace skdsp
