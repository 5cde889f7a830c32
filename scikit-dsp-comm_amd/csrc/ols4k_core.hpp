// ols4k_core.hpp -- the 4096-point complex64 overlap-save tile of the frequency-domain interpolator / decimator
// (fir_up4k.hip: multirate_FIR.up, multirate_helper.py:112-118; fir_dn4k.hip: multirate_FIR.dn, :121-127).
// Same code for the device (hipcc, gfx950) and the host (g++: tests/host/ols4k_emul.cpp checks the index algebra
// without a GPU).  Butterflies, packed complex arithmetic and the register DFTs come from ols_core.hpp.
//
// Why a second tile size.  An L-fold interpolator is L filters of ceil(Ntaps / L) taps over ONE input, a decimator the
// transpose of that: per tile of input (output) ONE forward (inverse) transform and L inverse (M forward) ones.  The
// phases are short (the reference's 512-tap prototype at L = 12: 43 taps per phase), so a 4096-point tile loses little to
// its overlap, and with 16 points per thread instead of 32 a thread can hold the spectrum of the tile (32 registers) AND
// the results of up to FOUR phases (128 registers) -- which is what lets a lane store 32 contiguous bytes of the
// interleaved output y[i L + p .. p + 3] (read 32 contiguous bytes of the input x[i M + r .. r + 3]) instead of one
// 8-byte element between the elements of other phases.
//
// Tile: N = 4096 complex64 points, 256 threads x 16 points, N = 16 x 16 x 16:
//   n = 256 a + rho,  rho = 16 b + c                     (a, b, c in [0,16))
//   k = k1 + 16 k2 + 256 k3                              (k1, k2, k3 in [0,16))
//   pass 1  thread (b,c)  : DFT16 over a -> k1, times W_4096^(rho k1)
//   xchg 1  (k1; b, c): thread (b,c)  -> thread (k1,c)     [workgroup-wide: one barrier]
//   pass 2  thread (k1,c) : DFT16 over b -> k2, times W_256^(c k2)
//   xchg 2  (k1, k2; c): thread (k1,c) -> thread (k1,k2)   [inside 16-lane groups: wave-local]
//   pass 3  thread (k1,k2): DFT16 over c -> k3
// and the inverse is the mirror image (decimation in time), so the spectrum never leaves its scrambled, thread-major
// order and the transfer functions are stored pre-permuted: thread 16 k1 + k2 holds bins k1 + 16 k2 + 256 k3.
//
// LDS image: 16 rows (k1) x 16 x 17 complex64 units of 8 bytes; with the +1 pad every ds_read_b64 / ds_write_b64 pattern
// below touches 32 distinct bank pairs per 32-lane group (MI355X: 64 banks x 4 B), but for the thread (b,c) <-> row k1
// pattern of pass 1, where lanes (b, 0) and (b + 1, 15) of a group meet in one bank pair (one extra LDS cycle).
#pragma once
#include "ols_core.hpp"

namespace skdsp {
namespace ols4k {

using ols::cf;
using ols::Dft;
using ols::cmul;
using ols::cmulc;
using ols::cadd;
using ols::static_for;
using ols::lo;
using ols::hi;
using ols::csub;
using ols::madd_mi;
using ols::msub_mi;
using ols::twmul;

constexpr int kN = 4096;
constexpr int kThreads = 256;
constexpr int kRowPitch = 272;             // cf units per k1 row (16 x 17)
constexpr int kImgUnits = 16 * kRowPitch;  // 4352 cf = 34816 B
constexpr int kT2Units = 256;              // one 16 x 16 cf twiddle table = 2 KiB
constexpr int kTwUnits = 15 * 256;         // this workgroup's pass-1 twiddles W_4096^(t k1), k1 = 1..15: [k1 - 1][t], 30 KiB

SK_HD int unit(int row_k1, int mid, int c) { return row_k1 * kRowPitch + mid * 17 + c; }

// ---- 16-point DFTs IN PLACE (radix 4 x 4): a pass works on ONE 16-element register array ------------------------
//   dft16_f  natural order in, X[k] out at slot P16(k) = 4 (k & 3) + (k >> 2)      (decimation in frequency)
//   dft16_g  input Z[m] at slot P16(m), natural order out; unnormalised inverse     (decimation in time)
// P16 is its own inverse (the transpose of the 4 x 4 index grid) and every index is a compile-time constant, so the
// permutation is free: a forward pass hands its output on through P16, the inverse pass 3 takes the spectrum where the
// forward pass 3 left it (the transfer functions are stored BY SLOT).  The out-of-place DFT16 of ols_core.hpp keeps its
// input, four partial DFT4s and its output alive together; with four phases' results, the spectrum and the next table
// in registers this kernel has no room for that.
constexpr int P16(int k) { return ((k & 3) << 2) | (k >> 2); }

template <bool INV> SK_HD void dft4_ip(cf &x0, cf &x1, cf &x2, cf &x3)
{
    const cf s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = madd_mi<INV>(d02, d13);
    x3 = msub_mi<INV>(d02, d13);
}
SK_HD void dft16_f(cf *v)
{
    // stage 1: DFT4 over n2 for each n1 (slots n1, n1 + 4, n1 + 8, n1 + 12): slot n1 + 4 k2 = a[n1][k2]
    SK_UNROLL
    for (int n1 = 0; n1 < 4; ++n1) dft4_ip<false>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    // twiddle W_16^(n1 k2), stage 2: DFT4 over n1 for each k2 (slots 4 k2 .. 4 k2 + 3): slot 4 k2 + k1 = X[k2 + 4 k1]
    static_for<1, 4>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        v[4 * k2 + 1] = twmul<16, k2, false>(v[4 * k2 + 1]);
        v[4 * k2 + 2] = twmul<16, 2 * k2, false>(v[4 * k2 + 2]);
        v[4 * k2 + 3] = twmul<16, 3 * k2, false>(v[4 * k2 + 3]);
    });
    SK_UNROLL
    for (int k2 = 0; k2 < 4; ++k2) dft4_ip<false>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}
SK_HD void dft16_g(cf *v)
{
    // input Z[m1 + 4 m2] at slot 4 m1 + m2.  stage A: inverse DFT4 over m2 for each m1 (slots 4 m1 .. 4 m1 + 3)
    SK_UNROLL
    for (int m1 = 0; m1 < 4; ++m1) dft4_ip<true>(v[4 * m1], v[4 * m1 + 1], v[4 * m1 + 2], v[4 * m1 + 3]);
    // conj twiddle W_16^(m1 r) on slot 4 m1 + r; stage B: inverse DFT4 over m1 for each r (slots r, r + 4, r + 8, r + 12)
    static_for<1, 4>([&](auto mc) {
        constexpr int m1 = decltype(mc)::value;
        v[4 * m1 + 1] = twmul<16, m1, true>(v[4 * m1 + 1]);
        v[4 * m1 + 2] = twmul<16, 2 * m1, true>(v[4 * m1 + 2]);
        v[4 * m1 + 3] = twmul<16, 3 * m1, true>(v[4 * m1 + 3]);
    });
    SK_UNROLL
    for (int r = 0; r < 4; ++r) dft4_ip<true>(v[r], v[r + 4], v[r + 8], v[r + 12]);
}

// ---- forward ----------------------------------------------------------------------------------------------------
// v[a] = x[256 a + t] on entry (destroyed).  tw[(k1 - 1) * 256 + t] = W_4096^(t k1) (LDS copy, lane-consecutive: conflict-free).
SK_HD void fwd_pass1(int t, cf *v, const cf *tw, cf *img)
{
    dft16_f(v);
    const int b = t >> 4, c = t & 15;
    img[unit(0, b, c)] = v[P16(0)];
    static_for<1, 16>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        img[unit(k1, b, c)] = cmul(v[P16(k1)], tw[(k1 - 1) * 256 + t]);
    });
}

// exchange-1 read + pass 2 + twiddle + exchange-2 write (thread t = 16 k1 + c; T2[k2 * 16 + c] = W_256^(c k2)), then
// exchange-2 read + pass 3 (thread t = 16 k1 + k2; the spectrum bin k3 lands at slot P16(k3) of Z).  Wave-local behind the
// barrier: the 16 lanes of a k1 row only read what they wrote, and a wave's LDS operations execute in order.
SK_HD void fwd_pass2(int t, const cf *T2, cf *img)
{
    const int k1 = t >> 4, c = t & 15;
    cf in[16];
    SK_UNROLL
    for (int b = 0; b < 16; ++b) in[b] = img[unit(k1, b, c)];
    dft16_f(in);
    img[unit(k1, 0, c)] = in[P16(0)];
    static_for<1, 16>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        img[unit(k1, k2, c)] = cmul(in[P16(k2)], T2[k2 * 16 + c]);
    });
}
SK_HD void fwd_pass3(int t, const cf *img, cf *Z)
{
    const int k1 = t >> 4, k2 = t & 15;
    SK_UNROLL
    for (int cc = 0; cc < 16; ++cc) Z[cc] = img[unit(k1, k2, cc)];
    dft16_f(Z);
}

// ---- pointwise product with a pre-permuted, pre-scaled transfer function --------------------------------------
// BY SLOT: Hp[j * 256 + t] = (H[k(P16(2j))], H[k(P16(2j + 1))]) / N,  k(k3) = k1 + 16 k2 + 256 k3,  t = 16 k1 + k2,  j = 0..7.
SK_HD void mul_H(const float4 *hh, const cf *Z, cf *P)
{
    SK_UNROLL
    for (int j = 0; j < 8; ++j) {
        P[2 * j] = cmul(Z[2 * j], lo(hh[j]));
        P[2 * j + 1] = cmul(Z[2 * j + 1], hi(hh[j]));
    }
}
// the decimator's accumulation over its M input phases: A += Z * H
SK_HD void mac_H(const float4 *hh, const cf *Z, cf *A)
{
    SK_UNROLL
    for (int j = 0; j < 8; ++j) {
        A[2 * j] = cadd(A[2 * j], cmul(Z[2 * j], lo(hh[j])));
        A[2 * j + 1] = cadd(A[2 * j + 1], cmul(Z[2 * j + 1], hi(hh[j])));
    }
}

// ---- inverse ----------------------------------------------------------------------------------------------------
// inverse pass 3 (in place on P: spectrum by slot in, natural order out) + conj twiddle + exchange-2' write (thread 16 k1 + k2;
// T2t[c * 16 + k2] = W_256^(c k2), the transposed copy: lane-consecutive in k2), then exchange-2' read + inverse pass 2 +
// exchange-1' write (thread 16 k1 + c).
SK_HD void inv_pass3(int t, const cf *T2t, cf *img, cf *P)
{
    const int k1 = t >> 4, k2 = t & 15;
    dft16_g(P);
    img[unit(k1, k2, 0)] = P[0];
    SK_UNROLL
    for (int cc = 1; cc < 16; ++cc) img[unit(k1, k2, cc)] = cmulc(P[cc], T2t[cc * 16 + k2]);
}
SK_HD void inv_pass2(int t, cf *img)
{
    const int k1 = t >> 4, c = t & 15;
    cf in[16];
    static_for<0, 16>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        in[P16(kk)] = img[unit(k1, kk, c)];
    });
    dft16_g(in);
    SK_UNROLL
    for (int b = 0; b < 16; ++b) img[unit(k1, b, c)] = in[b];
}

// exchange-1' read + conj twiddle + inverse pass 1, in place in v.  v[a] = y[256 a + t] out.
SK_HD void inv_pass1(int t, const cf *tw, const cf *img, cf *v)
{
    const int b = t >> 4, c = t & 15;
    v[P16(0)] = img[unit(0, b, c)];
    static_for<1, 16>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        v[P16(k1)] = cmulc(img[unit(k1, b, c)], tw[(k1 - 1) * 256 + t]);
    });
    dft16_g(v);
}

}  // namespace ols4k
}  // namespace skdsp
