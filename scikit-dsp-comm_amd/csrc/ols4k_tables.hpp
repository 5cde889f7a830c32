// ols4k_tables.hpp -- host-side (float64) tables of the 4096-point tile (ols4k_core.hpp): inter-pass twiddles and the
// pre-permuted, pre-scaled transfer functions of the interpolator's / decimator's phase filters.  Computed in double,
// rounded ONCE to float.  Host only; shared by fir_up4k.hip / fir_dn4k.hip and tests/host/ols4k_emul.cpp.
#pragma once
#include "ols_tables.hpp"
#include "ols4k_core.hpp"

namespace skdsp {
namespace ols4k {

using ols::cd;
using ols::wexp;

// tw[(k1 - 1) * 256 + t] = W_4096^(t k1), k1 = 1..15
inline void make_tw(std::vector<float2> &tw)
{
    tw.resize(kTwUnits);
    for (int k1 = 1; k1 < 16; ++k1)
        for (int t = 0; t < 256; ++t) {
            const cd w = wexp((long long)t * k1, kN);
            tw[(k1 - 1) * 256 + t] = make_float2((float)w.real(), (float)w.imag());
        }
}

// T2[k2 * 16 + c] = W_256^(c k2)
inline void make_T2(std::vector<float2> &T2)
{
    T2.resize(kT2Units);
    for (int k2 = 0; k2 < 16; ++k2)
        for (int c = 0; c < 16; ++c) {
            const cd w = wexp((long long)c * k2, 256);
            T2[k2 * 16 + c] = make_float2((float)w.real(), (float)w.imag());
        }
}

// BY SLOT of the in-place transform: Hp[j * 256 + t] = (H[k(P16(2j))], H[k(P16(2j + 1))]) / N with k(k3) = k1 + 16 k2 + 256 k3,
// t = 16 k1 + k2; appended to `Hp` (2048 float4 = 32 KiB).
// h: `len` complex taps (len <= 4096).
inline void append_Hp(const cd *h, int len, std::vector<float4> &Hp)
{
    std::vector<cd> H(kN, cd(0, 0));
    for (int k = 0; k < len; ++k) H[k] = h[k];
    ols::fft_host(H);
    const double sc = 1.0 / (double)kN;
    const size_t base = Hp.size();
    Hp.resize(base + 8 * 256);
    for (int j = 0; j < 8; ++j)
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, k2 = t & 15;
            const cd a = H[k1 + 16 * k2 + 256 * P16(2 * j)] * sc, b = H[k1 + 16 * k2 + 256 * P16(2 * j + 1)] * sc;
            Hp[base + j * 256 + t] = make_float4((float)a.real(), (float)a.imag(), (float)b.real(), (float)b.imag());
        }
}

// ---- multirate_FIR.up (multirate_helper.py:112-118): y[i L + p] = L sum_t b[p + L t] x[i - t] -----------------------------------
// complex64 signals: pass q IS phase q, h_q[t] = L b[q + L t].  float32 signals with real taps: pass q carries phases 2q and 2q + 1 as
// the real and imaginary part of ONE complex filter over the real signal (x * (h_2q + i h_2q+1) = y_2q + i y_2q+1: the pass's output is
// the interleaved pair (y[i L + 2q], y[i L + 2q + 1]) as one 8-byte element); an odd L leaves the last pass's imaginary part empty.
inline int up_taps_per_phase(int ntaps, int L) { return (ntaps + L - 1) / L; }
inline int up_passes(int L, bool real_pairs) { return real_pairs ? (L + 1) / 2 : L; }
// taps: ntaps real (comp = 1) or interleaved complex (comp = 2) doubles
inline void make_up_tables(const double *taps, int ntaps, int comp, int L, bool real_pairs, std::vector<float4> &Hp)
{
    const int T = up_taps_per_phase(ntaps, L);
    std::vector<cd> h(T);
    Hp.clear();
    auto tap = [&](int k) -> cd {
        if (k >= ntaps) return cd(0, 0);
        return comp == 2 ? cd(taps[2 * k], taps[2 * k + 1]) : cd(taps[k], 0.0);
    };
    for (int q = 0; q < up_passes(L, real_pairs); ++q) {
        for (int t = 0; t < T; ++t) {
            if (real_pairs) {
                const double re = tap(2 * q + L * t).real();
                const double im = 2 * q + 1 < L ? tap(2 * q + 1 + L * t).real() : 0.0;
                h[t] = cd((double)L * re, (double)L * im);
            } else {
                h[t] = (double)L * tap(q + L * t);
            }
        }
        append_Hp(h.data(), T, Hp);
    }
}

// ---- multirate_FIR.dn (multirate_helper.py:121-127): y[k] = sum_n b[n] x[k M - n] -------------------------------------------------
// With the input cut into ALIGNED blocks u_r[i] = x[i M + r], r = 0..M-1 (what a lane reads as contiguous bytes):
//   y[k] = sum_r sum_j g_r[j] u_r[k - j],   g_r[j] = b[j M - r]  (b[negative] = 0: for r > 0 the phase filter starts at j = 1)
inline int dn_taps_per_phase(int ntaps, int M) { return (ntaps - 1 + M - 1) / M + 1; }
inline void make_dn_tables(const double *taps, int ntaps, int comp, int M, std::vector<float4> &Hp)
{
    const int T = dn_taps_per_phase(ntaps, M);
    std::vector<cd> g(T);
    Hp.clear();
    for (int r = 0; r < M; ++r) {
        for (int j = 0; j < T; ++j) {
            const long long k = (long long)j * M - r;
            g[j] = (k < 0 || k >= ntaps) ? cd(0, 0) : (comp == 2 ? cd(taps[2 * k], taps[2 * k + 1]) : cd(taps[k], 0.0));
        }
        append_Hp(g.data(), T, Hp);
    }
}

}  // namespace ols4k
}  // namespace skdsp
