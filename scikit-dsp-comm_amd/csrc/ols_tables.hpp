// ols_tables.hpp -- host-side (float64) construction of the constant tables used by
// the overlap-save tile: inter-pass twiddles T1/T2 and the pre-permuted, pre-scaled
// transfer function Hp.  Everything is computed in double and rounded ONCE to float
// ("twiddles from a float64-computed table, H computed in float64").
// Host only; shared by fir_ols.hip and tests/host/ols_emul.cpp.
#pragma once
#include <complex>
#include <vector>
#include <cmath>
#include "ols_core.hpp"

namespace skdsp {
namespace ols {

typedef std::complex<double> cd;

inline cd wexp(long long num, long long den)  // exp(-2 pi i num/den), exact quadrant handling
{
    num %= den;
    if (num < 0) num += den;
    const double ang = -2.0 * M_PI * (double)num / (double)den;
    // exact values on the axes
    if (num == 0) return cd(1, 0);
    if (4 * num == den) return cd(0, -1);
    if (2 * num == den) return cd(-1, 0);
    if (4 * num == 3 * den) return cd(0, 1);
    return cd(std::cos(ang), std::sin(ang));
}

// in-place iterative radix-2 FFT (float64), forward
inline void fft_host(std::vector<cd> &a)
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1)
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const cd w = wexp((long long)k, (long long)len);
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
}

// T1[k1*256 + t] = (W_8192^((2t)k1), W_8192^((2t+1)k1))
inline void make_T1(std::vector<float4> &T1)
{
    T1.resize(16 * 256);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int t = 0; t < 256; ++t) {
            const cd w0 = wexp((long long)(2 * t) * k1, kN), w1 = wexp((long long)(2 * t + 1) * k1, kN);
            T1[k1 * 256 + t] = make_float4((float)w0.real(), (float)w0.imag(), (float)w1.real(), (float)w1.imag());
        }
}

// T2[k2*16 + q] = (W_512^((2q)k2), W_512^((2q+1)k2))
inline void make_T2(std::vector<float4> &T2)
{
    T2.resize(16 * 16);
    for (int k2 = 0; k2 < 16; ++k2)
        for (int q = 0; q < 16; ++q) {
            const cd w0 = wexp((long long)(2 * q) * k2, 512), w1 = wexp((long long)(2 * q + 1) * k2, 512);
            T2[k2 * 16 + q] = make_float4((float)w0.real(), (float)w0.imag(), (float)w1.real(), (float)w1.imag());
        }
}

// Hp[j*256 + t] = (H[k(2j)], H[k(2j+1)]) / N  with k(k3) = k1 + 16 k2 + 256 k3, t = 16 k1 + k2.
// taps: ntaps real (comp=1) or interleaved complex (comp=2) doubles.
inline void make_Hp(const double *taps, int ntaps, int comp, std::vector<float4> &Hp)
{
    std::vector<cd> h(kN, cd(0, 0));
    for (int k = 0; k < ntaps; ++k) h[k] = comp == 2 ? cd(taps[2 * k], taps[2 * k + 1]) : cd(taps[k], 0.0);
    fft_host(h);
    Hp.resize(16 * 256);
    const double sc = 1.0 / (double)kN;
    for (int j = 0; j < 16; ++j)
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, k2 = t & 15;
            const cd a = h[k1 + 16 * k2 + 256 * (2 * j)] * sc, b = h[k1 + 16 * k2 + 256 * (2 * j + 1)] * sc;
            Hp[j * 256 + t] = make_float4((float)a.real(), (float)a.imag(), (float)b.real(), (float)b.imag());
        }
}

}  // namespace ols
}  // namespace skdsp
