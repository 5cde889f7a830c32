// Host emulation of the overlap-save tile (no GPU): runs the exact per-thread phase
// functions of csrc/ols_core.hpp for all 256 "threads" with an array standing in
// for LDS, and checks (1) the register DFT16/DFT32 against an O(N^2) float64 DFT,
// (2) one whole tile (forward FFT, xH, inverse FFT) against a float64 direct FIR.
// Build: g++ -O1 -std=c++17 -I scikit-dsp-comm_amd/csrc tests/host/ols_emul.cpp -o /tmp/ols_emul
#include <cstdio>
#include <cstdlib>
#include <random>
#include "ols_tables.hpp"

using namespace skdsp::ols;

template <int N> static double check_dft(bool inv)
{
    std::mt19937 g(123 + N);
    std::normal_distribution<float> nd;
    cf x[N], X[N];
    for (int i = 0; i < N; ++i) x[i] = make_float2(nd(g), nd(g));
    if (inv) Dft<N, 1, true>::run(x, X); else Dft<N, 1, false>::run(x, X);
    double worst = 0, peak = 0;
    for (int k = 0; k < N; ++k) {
        cd acc(0, 0);
        for (int n = 0; n < N; ++n) {
            cd w = wexp((long long)n * k, N);
            if (inv) w = std::conj(w);
            acc += cd(x[n].x, x[n].y) * w;
        }
        worst = std::max(worst, std::abs(acc - cd(X[k].x, X[k].y)));
        peak = std::max(peak, std::abs(acc));
    }
    return worst / peak;
}

int main()
{
    int fails = 0;
    double e;
    e = check_dft<2>(false);  printf("dft2  fwd rel err %.3g\n", e); fails += e > 2e-6;
    e = check_dft<4>(false);  printf("dft4  fwd rel err %.3g\n", e); fails += e > 2e-6;
    e = check_dft<8>(false);  printf("dft8  fwd rel err %.3g\n", e); fails += e > 2e-6;
    e = check_dft<16>(false); printf("dft16 fwd rel err %.3g\n", e); fails += e > 2e-6;
    e = check_dft<32>(false); printf("dft32 fwd rel err %.3g\n", e); fails += e > 2e-6;
    e = check_dft<8>(true);   printf("dft8  inv rel err %.3g\n", e); fails += e > 2e-6;
    e = check_dft<16>(true);  printf("dft16 inv rel err %.3g\n", e); fails += e > 2e-6;
    e = check_dft<32>(true);  printf("dft32 inv rel err %.3g\n", e); fails += e > 2e-6;

    // ---- whole tile ----
    const int P = 1024;
    std::mt19937 g(7);
    std::normal_distribution<double> nd;
    std::vector<double> taps(2 * P);
    for (int k = 0; k < P; ++k) { taps[2 * k] = nd(g) / 32; taps[2 * k + 1] = nd(g) / 32; }  // complex taps: hardest case
    std::vector<float4> T1, T2, Hp;
    make_T1(T1); make_T2(T2); make_Hp(taps.data(), P, 2, Hp);
    std::vector<cf> x(kN);
    for (auto &v : x) v = make_float2((float)nd(g), (float)nd(g));
    std::vector<float4> lds(kLdsUnits);
    std::vector<cf> regs(256 * 32), Z(256 * 32);
    for (int t = 0; t < 256; ++t)
        for (int a = 0; a < 16; ++a)
            for (int ee = 0; ee < 2; ++ee) regs[t * 32 + 2 * a + ee] = x[512 * a + 2 * t + ee];
    // register twiddles exactly as the kernel derives them: tw[k1] = lo(T1[k1][t]) = W_4096^(t k1)
    std::vector<cf> tw(256 * 16);
    for (int t = 0; t < 256; ++t) {
        tw[t * 16] = make_float2(1.f, 0.f);
        for (int k1 = 1; k1 < 16; ++k1) tw[t * 16 + k1] = lo(T1[k1 * 256 + t]);
    }
    std::vector<float4> T2t(256);
    for (int i = 0; i < 256; ++i) T2t[(i & 15) * 16 + (i >> 4)] = T2[i];
    for (int t = 0; t < 256; ++t) fwd_pass1(t, &regs[t * 32], &tw[t * 16], lds.data());
    // pass 2/3 are wave-local after the barrier: emulate 16-lane groups in lock-step by
    // running the function per thread only works if reads follow ALL writes of the group,
    // so split it here exactly like the hardware does (in-order per wave).
    {
        std::vector<float4> stage(lds);
        // exchange-1 read + pass2 + exchange-2 write for every thread into a copy, then read back
        struct Tmp { cf o0[16], o1[16]; };
        std::vector<Tmp> tmp(256);
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, q = t & 15;
            cf in0[16], in1[16];
            for (int b = 0; b < 16; ++b) { float4 f = lds[lds_unit(k1, b, q)]; in0[b] = lo(f); in1[b] = hi(f); }
            Dft<16, 1, false>::run(in0, tmp[t].o0);
            Dft<16, 1, false>::run(in1, tmp[t].o1);
        }
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, q = t & 15;
            for (int k2 = 0; k2 < 16; ++k2) {
                float4 w = T2[k2 * 16 + q];
                lds[lds_unit(k1, k2, q)] = pack(cmul(tmp[t].o0[k2], lo(w)), cmul(tmp[t].o1[k2], hi(w)));
            }
        }
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, k2 = t & 15;
            cf z[32];
            for (int qq = 0; qq < 16; ++qq) { float4 f = lds[lds_unit(k1, k2, qq)]; z[2 * qq] = lo(f); z[2 * qq + 1] = hi(f); }
            Dft<32, 1, false>::run(z, &Z[t * 32]);
        }
    }
    // check the forward spectrum placement: Z[t][k3] == X[k1 + 16 k2 + 256 k3]
    {
        std::vector<cd> X(kN);
        for (int i = 0; i < kN; ++i) X[i] = cd(x[i].x, x[i].y);
        fft_host(X);
        double worst = 0, peak = 0;
        for (int t = 0; t < 256; ++t)
            for (int k3 = 0; k3 < 32; ++k3) {
                const int k = (t >> 4) + 16 * (t & 15) + 256 * k3;
                worst = std::max(worst, std::abs(X[k] - cd(Z[t * 32 + k3].x, Z[t * 32 + k3].y)));
                peak = std::max(peak, std::abs(X[k]));
            }
        printf("forward 8192 spectrum rel err %.3g\n", worst / peak);
        fails += worst / peak > 5e-6;
    }
    for (int t = 0; t < 256; ++t) { float4 hh[16]; load_H(t, Hp.data(), hh); mul_H(hh, &Z[t * 32]); }
    // inverse: pass 3 + exchange-2' write for all, then the rest
    {
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, k2 = t & 15;
            cf z[32];
            Dft<32, 1, true>::run(&Z[t * 32], z);
            for (int qq = 0; qq < 16; ++qq) {
                float4 w = T2t[qq * 16 + k2];
                lds[lds_unit(k1, k2, qq)] = pack(cmulc(z[2 * qq], lo(w)), cmulc(z[2 * qq + 1], hi(w)));
            }
        }
        struct Tmp { cf o0[16], o1[16]; };
        std::vector<Tmp> tmp(256);
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, q = t & 15;
            cf in0[16], in1[16];
            for (int kk = 0; kk < 16; ++kk) { float4 f = lds[lds_unit(k1, kk, q)]; in0[kk] = lo(f); in1[kk] = hi(f); }
            Dft<16, 1, true>::run(in0, tmp[t].o0);
            Dft<16, 1, true>::run(in1, tmp[t].o1);
        }
        for (int t = 0; t < 256; ++t) {
            const int k1 = t >> 4, q = t & 15;
            for (int b = 0; b < 16; ++b) lds[lds_unit(k1, b, q)] = pack(tmp[t].o0[b], tmp[t].o1[b]);
        }
    }
    for (int t = 0; t < 256; ++t) inv_pass1(t, &tw[t * 16], lds.data(), &regs[t * 32]);
    // reference: circular convolution == linear FIR for n >= P-1
    double worst = 0, peak = 0;
    for (int n = P - 1; n < kN; n += 7) {
        cd acc(0, 0);
        for (int k = 0; k < P; ++k) acc += cd(taps[2 * k], taps[2 * k + 1]) * cd(x[n - k].x, x[n - k].y);
        const int a = n / 512, rho = n % 512, t = rho / 2, ee = rho & 1;
        const cf y = regs[t * 32 + 2 * a + ee];
        worst = std::max(worst, std::abs(acc - cd(y.x, y.y)));
        peak = std::max(peak, std::abs(acc));
    }
    printf("tile FIR (1024 complex taps) rel err %.3g\n", worst / peak);
    fails += worst / peak > 2e-6;
    printf(fails ? "FAIL\n" : "OK\n");
    return fails ? 1 : 0;
}
