// Host emulation of the 4096-point tile of the frequency-domain interpolator / decimator (no GPU): runs the exact per-thread
// phase functions of csrc/ols4k_core.hpp for all 256 "threads" with an array standing in for LDS (every exchange split where
// the hardware's in-order wave execution splits it), with the tables of csrc/ols4k_tables.hpp, and checks
//   (1) the placement of the forward spectrum,
//   (2) multirate_FIR.up: one tile, every pass, complex64 signal (complex taps, odd L) and float32 signal (phases in pairs, odd
//       and even L) against the float64 polyphase sum  y[i L + p] = L sum_t b[p + L t] x[i - t],
//   (3) multirate_FIR.dn: one tile, M forward transforms accumulated in the frequency domain and ONE inverse, against the
//       float64 sum  y[k] = sum_n b[n] x[k M - n].
// Build: g++ -O1 -std=c++17 -I scikit-dsp-comm_amd/csrc tests/host/ols4k_emul.cpp -o /tmp/ols4k_emul
#include <cstdio>
#include <cstdlib>
#include <random>
#include "ols4k_tables.hpp"

using namespace skdsp::ols4k;
using skdsp::ols::cd;

struct Tables {
    std::vector<float2> tw, T2, T2t;
    Tables()
    {
        make_tw(tw);
        make_T2(T2);
        T2t.resize(256);
        for (int i = 0; i < 256; ++i) T2t[(i & 15) * 16 + (i >> 4)] = T2[i];
    }
};

// x[4096] (natural order) -> Z[t * 16 + k3]
static void fwd_tile(const Tables &tb, const std::vector<cf> &x, std::vector<cf> &Z)
{
    std::vector<cf> img(kImgUnits), regs(256 * 16);
    for (int t = 0; t < 256; ++t)
        for (int a = 0; a < 16; ++a) regs[t * 16 + a] = x[256 * a + t];
    for (int t = 0; t < 256; ++t) fwd_pass1(t, &regs[t * 16], tb.tw.data(), img.data());
    for (int t = 0; t < 256; ++t) fwd_pass2(t, tb.T2.data(), img.data());
    Z.resize(256 * 16);
    for (int t = 0; t < 256; ++t) fwd_pass3(t, img.data(), &Z[t * 16]);
}
// P[t * 16 + k3] -> y[4096] (natural order)
static void inv_tile(const Tables &tb, const std::vector<cf> &P, std::vector<cf> &y)
{
    std::vector<cf> img(kImgUnits), regs(256 * 16);
    std::vector<cf> W(P);   // (the inverse works in place)
    for (int t = 0; t < 256; ++t) inv_pass3(t, tb.T2t.data(), img.data(), &W[t * 16]);
    for (int t = 0; t < 256; ++t) inv_pass2(t, img.data());
    for (int t = 0; t < 256; ++t) inv_pass1(t, tb.tw.data(), img.data(), &regs[t * 16]);
    y.resize(kN);
    for (int t = 0; t < 256; ++t)
        for (int a = 0; a < 16; ++a) y[256 * a + t] = regs[t * 16 + a];
}

static int check_up(const Tables &tb, int ntaps, int comp, int L, bool real_sig, unsigned seed)
{
    std::mt19937 g(seed);
    std::normal_distribution<double> nd;
    std::vector<double> taps((size_t)ntaps * comp);
    for (auto &v : taps) v = nd(g) / 16;
    const int T = up_taps_per_phase(ntaps, L), ov = ((T - 1 + 255) / 256) * 256 ? ((T - 1 + 255) / 256) * 256 : 256, V = kN - ov;
    const int n = 3 * V, tile = 1, in0 = tile * V - ov;
    std::vector<cd> x(n);
    for (auto &v : x) v = real_sig ? cd((float)nd(g), 0) : cd((float)nd(g), (float)nd(g));
    std::vector<float4> Hp;
    make_up_tables(taps.data(), ntaps, comp, L, real_sig, Hp);
    std::vector<cf> xt(kN), Z, P(256 * 16), yt;
    for (int i = 0; i < kN; ++i) xt[i] = make_float2((float)x[in0 + i].real(), (float)x[in0 + i].imag());
    fwd_tile(tb, xt, Z);
    auto tap = [&](int k) -> cd { return k >= ntaps ? cd(0, 0) : (comp == 2 ? cd(taps[2 * k], taps[2 * k + 1]) : cd(taps[k], 0)); };
    double worst = 0, peak = 0;
    for (int q = 0; q < up_passes(L, real_sig); ++q) {
        // (mul_H takes the thread's OWN 8 float4: Hp[q][j * 256 + t])
        for (int t = 0; t < 256; ++t) {
            float4 hh[8];
            for (int j = 0; j < 8; ++j) hh[j] = Hp[(size_t)q * 2048 + j * 256 + t];
            mul_H(hh, &Z[t * 16], &P[t * 16]);
        }
        inv_tile(tb, P, yt);
        for (int il = ov; il < kN; il += 3) {
            const int i = in0 + il;
            if (real_sig) {
                for (int c = 0; c < 2; ++c) {
                    const int p = 2 * q + c;
                    if (p >= L) continue;
                    double acc = 0;
                    for (int t = 0; t < T && i - t >= 0; ++t) acc += (double)L * tap(p + L * t).real() * x[i - t].real();
                    const double got = c ? yt[il].y : yt[il].x;
                    worst = std::max(worst, std::abs(acc - got));
                    peak = std::max(peak, std::abs(acc));
                }
            } else {
                cd acc(0, 0);
                for (int t = 0; t < T && i - t >= 0; ++t) acc += (double)L * tap(q + L * t) * x[i - t];
                worst = std::max(worst, std::abs(acc - cd(yt[il].x, yt[il].y)));
                peak = std::max(peak, std::abs(acc));
            }
        }
    }
    printf("up   %s L=%2d %4d taps (%s, %d per phase, overlap %d): rel err %.3g\n", real_sig ? "float32  " : "complex64", L, ntaps,
           comp == 2 ? "complex" : "real", T, ov, worst / peak);
    return worst / peak > 2e-6;
}

static int check_dn(const Tables &tb, int ntaps, int comp, int M, unsigned seed)
{
    std::mt19937 g(seed);
    std::normal_distribution<double> nd;
    std::vector<double> taps((size_t)ntaps * comp);
    for (auto &v : taps) v = nd(g) / 16;
    const int T = dn_taps_per_phase(ntaps, M), ov = ((T - 1 + 255) / 256) * 256 ? ((T - 1 + 255) / 256) * 256 : 256, V = kN - ov;
    const int nk = 3 * V, n = nk * M, tile = 1, k0 = tile * V - ov;   // the tile holds outputs k0 .. k0 + 4095, valid from k0 + ov on
    std::vector<cd> x(n);
    for (auto &v : x) v = cd((float)nd(g), (float)nd(g));
    std::vector<float4> Hp;
    make_dn_tables(taps.data(), ntaps, comp, M, Hp);
    std::vector<cf> xt(kN), Z, A(256 * 16, make_float2(0.f, 0.f)), yt;
    for (int r = 0; r < M; ++r) {
        for (int i = 0; i < kN; ++i) {
            const cd v = x[(size_t)(k0 + i) * M + r];   // u_r[k0 + i]
            xt[i] = make_float2((float)v.real(), (float)v.imag());
        }
        fwd_tile(tb, xt, Z);
        for (int t = 0; t < 256; ++t) {
            float4 hh[8];
            for (int j = 0; j < 8; ++j) hh[j] = Hp[(size_t)r * 2048 + j * 256 + t];
            mac_H(hh, &Z[t * 16], &A[t * 16]);
        }
    }
    inv_tile(tb, A, yt);
    auto tap = [&](int k) -> cd { return comp == 2 ? cd(taps[2 * k], taps[2 * k + 1]) : cd(taps[k], 0); };
    double worst = 0, peak = 0;
    for (int il = ov; il < kN; il += 5) {
        const long long k = k0 + il;
        cd acc(0, 0);
        for (int j = 0; j < ntaps && k * M - j >= 0; ++j) acc += tap(j) * x[(size_t)(k * M - j)];
        worst = std::max(worst, std::abs(acc - cd(yt[il].x, yt[il].y)));
        peak = std::max(peak, std::abs(acc));
    }
    printf("dn   complex64 M=%2d %4d taps (%s, %d per phase, overlap %d): rel err %.3g\n", M, ntaps, comp == 2 ? "complex" : "real", T, ov, worst / peak);
    return worst / peak > 2e-6;
}

int main()
{
    Tables tb;
    int fails = 0;
    {   // forward spectrum placement: Z[t][P16(k3)] == X[k1 + 16 k2 + 256 k3], t = 16 k1 + k2
        std::mt19937 g(5);
        std::normal_distribution<float> nd;
        std::vector<cf> x(kN), Z;
        for (auto &v : x) v = make_float2(nd(g), nd(g));
        fwd_tile(tb, x, Z);
        std::vector<cd> X(kN);
        for (int i = 0; i < kN; ++i) X[i] = cd(x[i].x, x[i].y);
        skdsp::ols::fft_host(X);
        double worst = 0, peak = 0;
        for (int t = 0; t < 256; ++t)
            for (int k3 = 0; k3 < 16; ++k3) {
                const int k = (t >> 4) + 16 * (t & 15) + 256 * k3;   // bin k3 of thread t sits at slot P16(k3)
                worst = std::max(worst, std::abs(X[k] - cd(Z[t * 16 + P16(k3)].x, Z[t * 16 + P16(k3)].y)));
                peak = std::max(peak, std::abs(X[k]));
            }
        printf("forward 4096 spectrum rel err %.3g\n", worst / peak);
        fails += worst / peak > 5e-6;
        std::vector<cf> y;
        inv_tile(tb, Z, y);   // unnormalised inverse: 4096 x
        worst = 0;
        for (int i = 0; i < kN; ++i) worst = std::max(worst, (double)std::hypot(y[i].x / 4096.f - x[i].x, y[i].y / 4096.f - x[i].y));
        printf("forward + inverse round trip max abs err %.3g\n", worst);
        fails += worst > 5e-6;
    }
    fails += check_up(tb, 700, 2, 3, false, 11);     // complex taps, odd L
    fails += check_up(tb, 1024, 1, 4, false, 12);    // the bench shape: 256 per phase
    fails += check_up(tb, 512, 1, 12, false, 13);    // the reference's default L_change on its 512-tap prototype
    fails += check_up(tb, 3000, 1, 2, false, 14);    // 1500 per phase: overlap 1536
    fails += check_up(tb, 1024, 1, 4, true, 15);     // float32: two passes of two phases
    fails += check_up(tb, 777, 1, 5, true, 16);      // float32, odd L: the last pass carries one phase
    fails += check_dn(tb, 1024, 1, 4, 21);
    fails += check_dn(tb, 700, 2, 3, 22);
    fails += check_dn(tb, 512, 1, 12, 23);
    fails += check_dn(tb, 4000, 1, 2, 24);
    printf(fails ? "FAIL\n" : "OK\n");
    return fails ? 1 : 0;
}
