// Host emulation of the 2048-point, 8-points-per-thread tile of the frequency-domain interpolator / decimator (no GPU): runs the exact per-thread
// phase functions of csrc/ols2k_core.hpp for all 256 "threads" with an array standing in for LDS (every exchange split where
// the hardware's in-order wave execution splits it), with the tables of csrc/ols2k_tables.hpp, and checks
//   (1) the placement of the forward spectrum,
//   (2) multirate_FIR.up: one tile, every pass, complex64 signal (complex taps, odd L) and float32 signal (phases in pairs, odd
//       and even L) against the float64 polyphase sum  y[i L + p] = L sum_t b[p + L t] x[i - t],
//   (3) multirate_FIR.dn: one tile, M forward transforms accumulated in the frequency domain and ONE inverse, against the
//       float64 sum  y[k] = sum_n b[n] x[k M - n].
// Build: g++ -O1 -std=c++17 -I scikit-dsp-comm_amd/csrc tests/host/ols2k_emul.cpp -o /tmp/ols2k_emul
#include <cstdio>
#include <cstdlib>
#include <random>
#include "ols2k_tables.hpp"

using namespace skdsp::ols2k;
using skdsp::ols::cd;

struct Tables {
    std::vector<float2> tw1, tw2, tw3;
    Tables()
    {
        make_tw1(tw1);
        make_tw2(tw2);
        make_tw3(tw3);
    }
};

constexpr int kN = k2N;
// x[2048] (natural order) -> Z[t * 8 + slot]
static void fwd_tile(const Tables &tb, const std::vector<cf> &x, std::vector<cf> &Z)
{
    std::vector<cf> img(kImgUnits), regs(256 * 8);
    for (int t = 0; t < 256; ++t)
        for (int m = 0; m < 8; ++m) regs[t * 8 + m] = x[256 * m + t];   // v[2 a + e] = x[512 a + 256 e + t]
    for (int t = 0; t < 256; ++t) fwd_pass1(t, &regs[t * 8], tb.tw1.data(), img.data());
    for (int t = 0; t < 256; ++t) fwd_pass2(t, tb.tw2.data(), img.data());
    for (int t = 0; t < 256; ++t) fwd_pass3(t, tb.tw3.data(), img.data());
    Z.resize(256 * 8);
    for (int t = 0; t < 256; ++t) fwd_pass4(t, img.data(), &Z[t * 8]);
}
// P[t * 8 + slot] -> y[2048] (natural order)
static void inv_tile(const Tables &tb, const std::vector<cf> &P, std::vector<cf> &y)
{
    std::vector<cf> img(kImgUnits), regs(256 * 8), W(P);
    for (int t = 0; t < 256; ++t) inv_pass4(t, img.data(), &W[t * 8]);
    for (int t = 0; t < 256; ++t) inv_pass3(t, tb.tw3.data(), img.data());
    for (int t = 0; t < 256; ++t) inv_pass2(t, tb.tw2.data(), img.data());
    for (int t = 0; t < 256; ++t) inv_pass1(t, tb.tw1.data(), img.data(), &regs[t * 8]);
    y.resize(kN);
    for (int t = 0; t < 256; ++t)
        for (int m = 0; m < 8; ++m) y[256 * m + t] = regs[t * 8 + m];
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
    std::vector<cf> xt(kN), Z, P(256 * 8), yt;
    for (int i = 0; i < kN; ++i) xt[i] = make_float2((float)x[in0 + i].real(), (float)x[in0 + i].imag());
    fwd_tile(tb, xt, Z);
    auto tap = [&](int k) -> cd { return k >= ntaps ? cd(0, 0) : (comp == 2 ? cd(taps[2 * k], taps[2 * k + 1]) : cd(taps[k], 0)); };
    double worst = 0, peak = 0;
    for (int q = 0; q < up_passes(L, real_sig); ++q) {
        // (mul_H takes the thread's OWN 8 float4: Hp[q][j * 256 + t])
        for (int t = 0; t < 256; ++t) {
            float4 hh[4];
            for (int j = 0; j < 4; ++j) hh[j] = Hp[(size_t)q * 1024 + j * 256 + t];
            mul_H(hh, &Z[t * 8], &P[t * 8]);
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
    std::vector<cf> xt(kN), Z, A(256 * 8, make_float2(0.f, 0.f)), yt;
    for (int r = 0; r < M; ++r) {
        for (int i = 0; i < kN; ++i) {
            const cd v = x[(size_t)(k0 + i) * M + r];   // u_r[k0 + i]
            xt[i] = make_float2((float)v.real(), (float)v.imag());
        }
        fwd_tile(tb, xt, Z);
        for (int t = 0; t < 256; ++t) {
            float4 hh[4];
            for (int j = 0; j < 4; ++j) hh[j] = Hp[(size_t)r * 1024 + j * 256 + t];
            mac_H(hh, &Z[t * 8], &A[t * 8]);
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
    {   // forward spectrum placement: Z[t][P8(k4)] == X[k1 + 4 k2 + 32 k3 + 256 k4], t = 64 k1 + 8 k2 + k3
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
            for (int k4 = 0; k4 < 8; ++k4) {
                const int k = (t >> 6) + 4 * ((t >> 3) & 7) + 32 * (t & 7) + 256 * k4;   // bin k4 of thread t sits at slot P8(k4)
                worst = std::max(worst, std::abs(X[k] - cd(Z[t * 8 + P8(k4)].x, Z[t * 8 + P8(k4)].y)));
                peak = std::max(peak, std::abs(X[k]));
            }
        printf("forward 2048 spectrum rel err %.3g\n", worst / peak);
        fails += worst / peak > 5e-6;
        std::vector<cf> y;
        inv_tile(tb, Z, y);   // unnormalised inverse: 4096 x
        worst = 0;
        for (int i = 0; i < kN; ++i) worst = std::max(worst, (double)std::hypot(y[i].x / 2048.f - x[i].x, y[i].y / 2048.f - x[i].y));
        printf("forward + inverse round trip max abs err %.3g\n", worst);
        fails += worst > 5e-6;
    }
    fails += check_up(tb, 700, 2, 3, false, 11);     // complex taps, odd L
    fails += check_up(tb, 1024, 1, 4, false, 12);    // the bench shape: 256 per phase
    fails += check_up(tb, 512, 1, 12, false, 13);    // the reference's default L_change on its 512-tap prototype
    fails += check_up(tb, 1500, 1, 2, false, 14);    // 750 per phase: overlap 768
    fails += check_up(tb, 1024, 1, 4, true, 15);     // float32: two passes of two phases
    fails += check_up(tb, 777, 1, 5, true, 16);      // float32, odd L: the last pass carries one phase
    fails += check_dn(tb, 1024, 1, 4, 21);
    fails += check_dn(tb, 700, 2, 3, 22);
    fails += check_dn(tb, 512, 1, 12, 23);
    fails += check_dn(tb, 2000, 1, 2, 24);
    printf(fails ? "FAIL\n" : "OK\n");
    return fails ? 1 : 0;
}
