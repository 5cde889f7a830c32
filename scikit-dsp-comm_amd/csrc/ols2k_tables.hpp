// ols2k_tables.hpp -- host-side (float64) tables of the 2048-point, 8-points-per-thread tile (ols2k_core.hpp): inter-pass
// twiddles and the pre-permuted, pre-scaled transfer functions of the phase filters.  Computed in double, rounded ONCE to
// float.  Host only; shared by fir_up2k.hip / fir_dn2k.hip and tests/host/ols2k_emul.cpp.
#pragma once
#include "ols_tables.hpp"
#include "ols2k_core.hpp"

namespace skdsp {
namespace ols2k {

using ols::cd;
using ols::wexp;

inline void make_tw1(std::vector<float2> &tw)   // [(k1 - 1) * 256 + t] = W_2048^(t k1), k1 = 1..3
{
    tw.resize(kTw1Units);
    for (int k1 = 1; k1 < 4; ++k1)
        for (int t = 0; t < 256; ++t) {
            const cd w = wexp((long long)t * k1, k2N);
            tw[(k1 - 1) * 256 + t] = make_float2((float)w.real(), (float)w.imag());
        }
}
inline void make_tw2(std::vector<float2> &tw)   // [k2 * 64 + r] = W_512^(r k2)
{
    tw.resize(kTw2Units);
    for (int k2 = 0; k2 < 8; ++k2)
        for (int r = 0; r < 64; ++r) {
            const cd w = wexp((long long)r * k2, 512);
            tw[k2 * 64 + r] = make_float2((float)w.real(), (float)w.imag());
        }
}
inline void make_tw3(std::vector<float2> &tw)   // [k3 * 8 + d] = W_64^(d k3)
{
    tw.resize(kTw3Units);
    for (int k3 = 0; k3 < 8; ++k3)
        for (int d = 0; d < 8; ++d) {
            const cd w = wexp((long long)d * k3, 64);
            tw[k3 * 8 + d] = make_float2((float)w.real(), (float)w.imag());
        }
}

// BY SLOT of the in-place transform: Hp[j * 256 + t] = (H[k(Q8(2j))], H[k(Q8(2j + 1))]) / N with k(k4) = k1 + 4 k2 + 32 k3 + 256 k4,
// t = 64 k1 + 8 k2 + k3; appended to `Hp` (1024 float4 = 16 KiB).  h: `len` complex taps (len <= 2048).
inline void append_Hp(const cd *h, int len, std::vector<float4> &Hp)
{
    std::vector<cd> H(k2N, cd(0, 0));
    for (int k = 0; k < len; ++k) H[k] = h[k];
    ols::fft_host(H);
    const double sc = 1.0 / (double)k2N;
    const size_t base = Hp.size();
    Hp.resize(base + 4 * 256);
    for (int j = 0; j < 4; ++j)
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 6, k2 = (t >> 3) & 7, k3 = t & 7;
            const int kb = k1 + 4 * k2 + 32 * k3;
            const cd a = H[kb + 256 * Q8(2 * j)] * sc, b = H[kb + 256 * Q8(2 * j + 1)] * sc;
            Hp[base + j * 256 + t] = make_float4((float)a.real(), (float)a.imag(), (float)b.real(), (float)b.imag());
        }
}

// multirate_FIR.up (multirate_helper.py:112-118): pass q IS phase q (complex64), h_q[t] = L b[q + L t]; float32 signals with real
// taps: pass q carries phases 2q and 2q + 1 as real and imaginary part of one complex filter over the real signal
inline int up_taps_per_phase(int ntaps, int L) { return (ntaps + L - 1) / L; }
inline int up_passes(int L, bool real_pairs) { return real_pairs ? (L + 1) / 2 : L; }
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

// multirate_FIR.dn (multirate_helper.py:121-127) over ALIGNED input blocks u_r[i] = x[i M + r]:
//   y[k] = sum_r sum_j g_r[j] u_r[k - j],   g_r[j] = b[j M - r]  (b[negative] = 0)
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

}  // namespace ols2k
}  // namespace skdsp
