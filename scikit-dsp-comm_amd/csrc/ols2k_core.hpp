// ols2k_core.hpp -- the 2048-point complex64 overlap-save tile with EIGHT points per thread (fir_up2k.hip / fir_dn2k.hip):
// the frequency-domain interpolator / decimator for 5 ... 12 phases (complex64) or up to 24 (float32, phases in pairs).
// Same code for the device (hipcc, gfx950) and the host (g++: tests/host/ols2k_emul.cpp).
//
// Why eight points per thread.  A row of the interleaved output -- one input sample's L outputs -- must leave the chip in
// ONE burst: pieces of a row written at different times cost 3 - 4 x (tools/ubench_strided_store.hip: 32-byte pieces of 64 ...
// 512-byte rows written back to back 5 TB/s, written a sweep apart 1.2 - 1.6 TB/s), so a thread has to hold the results of
// ALL phases of its samples.  With 16 points per thread (ols4k_core.hpp) that is 32 registers per phase: four phases.  With
// 8 points it is 16 registers per phase: twelve phases in 192 registers, next to the spectrum (16) and the working set of
// an in-place DFT8 pass (16).
//
// Tile: N = 2048 complex64 points, 256 threads x 8 points, N = 4 x 8 x 8 x 8:
//   n = 512 a + rho (a < 4),  rho = 64 b + 8 c + d (b, c, d < 8);   k = k1 + 4 k2 + 32 k3 + 256 k4 (k1 < 4; k2, k3, k4 < 8)
//   pass 1  thread t        : two columns rho = t, t + 256: DFT4 over a -> k1, times W_2048^(rho k1)
//   xchg 1  (k1; b, c, d): thread (rho & 255) -> thread (k1, c, d) = 64 k1 + 8 c + d        [workgroup-wide: one barrier]
//   pass 2  thread (k1,c,d) : DFT8 over b -> k2, times W_512^((8 c + d) k2)
//   xchg 2  (k1, k2; c, d): -> thread (k1, k2, d)                                           [one wave per k1: wave-local]
//   pass 3  thread (k1,k2,d): DFT8 over c -> k3, times W_64^(d k3)
//   xchg 3  (k1, k2, k3; d): -> thread (k1, k2, k3)                                         [wave-local]
//   pass 4  thread (k1,k2,k3): DFT8 over d -> k4
// and the inverse is the mirror image, so the spectrum stays in its thread-major order (thread 64 k1 + 8 k2 + k3 holds the
// bins k1 + 4 k2 + 32 k3 + 256 k4, bin k4 at slot P8(k4) of the in-place DFT8) and the transfer functions are stored by slot.
//
// LDS image: per k1 a region of 8 x 72 complex64 units, element (i2, i3, d) at 72 i2 + 9 i3 + d (i2 = b or k2, i3 = c or k3):
// every pass transforms one index in place, lanes run over the two others.  ds_read_b64 / ds_write_b64 are serviced in
// 32-lane groups over 32 bank pairs: lanes (i2, d) -> 8 i2 + d and lanes (i2, i3) -> 8 i2 + 9 i3 are 32 distinct residues;
// lanes (i3, d) -> 9 i3 + d wrap three of them (a two-way conflict on 3 of 32 lanes: one extra LDS cycle).
#pragma once
#include "ols_core.hpp"

namespace skdsp {
namespace ols2k {

using ols::cf;
using ols::cmul;
using ols::cmulc;
using ols::cadd;
using ols::csub;
using ols::madd_mi;
using ols::msub_mi;
using ols::twmul;
using ols::static_for;
using ols::lo;
using ols::hi;

constexpr int k2N = 2048;
constexpr int kRegion = 576;               // cf units per k1 region (8 x 72)
constexpr int kImgUnits = 4 * kRegion;     // 2304 cf = 18432 B
constexpr int kTw1Units = 3 * 256;         // W_2048^(t k1), k1 = 1..3: [k1 - 1][t]
constexpr int kTw2Units = 8 * 64;          // W_512^(r k2): [k2][r], r = 8 c + d
constexpr int kTw3Units = 8 * 8;           // W_64^(d k3): [k3][d]

SK_HD int unit(int k1, int i2, int i3, int d) { return k1 * kRegion + i2 * 72 + i3 * 9 + d; }

// ---- 8-point DFT in place (2 x 4): natural order in, X[k] out at slot P8(k) = 2 (k & 3) + (k >> 2); the inverse takes
// Z[m] at slot P8(m) and leaves natural order (unnormalised).  P8 is NOT its own inverse: the slot of bin k is P8(k), the bin
// of slot s is Q8(s) = (s >> 1) + 4 (s & 1).
constexpr int P8(int k) { return ((k & 3) << 1) | (k >> 2); }
constexpr int Q8(int s) { return (s >> 1) | ((s & 1) << 2); }

template <bool INV> SK_HD void dft4_ip(cf &x0, cf &x1, cf &x2, cf &x3)
{
    const cf s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = madd_mi<INV>(d02, d13);
    x3 = msub_mi<INV>(d02, d13);
}
SK_HD void dft2_ip(cf &x0, cf &x1)
{
    const cf s = cadd(x0, x1), d = csub(x0, x1);
    x0 = s;
    x1 = d;
}
SK_HD void dft8_f(cf *v)
{
    // n = n1 + 2 n2: DFT4 over n2 for each n1 (slots n1, n1 + 2, n1 + 4, n1 + 6): slot n1 + 2 k2 = a[n1][k2]
    dft4_ip<false>(v[0], v[2], v[4], v[6]);
    dft4_ip<false>(v[1], v[3], v[5], v[7]);
    // twiddle W_8^(n1 k2) on slot 1 + 2 k2, then DFT2 over n1 for each k2 (slots 2 k2, 2 k2 + 1): slot 2 k2 + k1 = X[k2 + 4 k1]
    v[3] = twmul<8, 1, false>(v[3]);
    v[5] = twmul<8, 2, false>(v[5]);
    v[7] = twmul<8, 3, false>(v[7]);
    dft2_ip(v[0], v[1]);
    dft2_ip(v[2], v[3]);
    dft2_ip(v[4], v[5]);
    dft2_ip(v[6], v[7]);
}
SK_HD void dft8_g(cf *v)
{
    // Z[m2 + 4 m1] at slot 2 m2 + m1: inverse DFT2 over m1 for each m2, conj twiddle W_8^(r m2) on slot 2 m2 + r, inverse DFT4 over m2
    dft2_ip(v[0], v[1]);
    dft2_ip(v[2], v[3]);
    dft2_ip(v[4], v[5]);
    dft2_ip(v[6], v[7]);
    v[3] = twmul<8, 1, true>(v[3]);
    v[5] = twmul<8, 2, true>(v[5]);
    v[7] = twmul<8, 3, true>(v[7]);
    dft4_ip<true>(v[0], v[2], v[4], v[6]);
    dft4_ip<true>(v[1], v[3], v[5], v[7]);
}

// ---- forward ----------------------------------------------------------------------------------------------------
// v[2 a + e] = x[512 a + 256 e + t] on entry (destroyed).  tw1[(k1 - 1) * 256 + t] = W_2048^(t k1); the second column's
// twiddle is that times W_8^k1 (rho = t + 256).
SK_HD void fwd_pass1(int t, cf *v, const cf *tw1, cf *img)
{
    dft4_ip<false>(v[0], v[2], v[4], v[6]);
    dft4_ip<false>(v[1], v[3], v[5], v[7]);
    const int b = t >> 6, c = (t >> 3) & 7, d = t & 7;
    img[unit(0, b, c, d)] = v[0];
    img[unit(0, b + 4, c, d)] = v[1];
    static_for<1, 4>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        const cf w = tw1[(k1 - 1) * 256 + t];
        img[unit(k1, b, c, d)] = cmul(v[2 * k1], w);
        img[unit(k1, b + 4, c, d)] = cmul(twmul<8, k1, false>(v[2 * k1 + 1]), w);
    });
}
// thread t = 64 k1 + 8 c + d: DFT8 over b; tw2[k2 * 64 + (t & 63)] = W_512^((8 c + d) k2)
SK_HD void fwd_pass2(int t, const cf *tw2, cf *img)
{
    const int k1 = t >> 6, c = (t >> 3) & 7, d = t & 7;
    cf v[8];
    SK_UNROLL
    for (int b = 0; b < 8; ++b) v[b] = img[unit(k1, b, c, d)];
    dft8_f(v);
    img[unit(k1, 0, c, d)] = v[P8(0)];
    static_for<1, 8>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        img[unit(k1, k2, c, d)] = cmul(v[P8(k2)], tw2[k2 * 64 + (t & 63)]);
    });
}
// thread t = 64 k1 + 8 k2 + d: DFT8 over c; tw3[k3 * 8 + d] = W_64^(d k3)
SK_HD void fwd_pass3(int t, const cf *tw3, cf *img)
{
    const int k1 = t >> 6, k2 = (t >> 3) & 7, d = t & 7;
    cf v[8];
    SK_UNROLL
    for (int c = 0; c < 8; ++c) v[c] = img[unit(k1, k2, c, d)];
    dft8_f(v);
    img[unit(k1, k2, 0, d)] = v[P8(0)];
    static_for<1, 8>([&](auto kc) {
        constexpr int k3 = decltype(kc)::value;
        img[unit(k1, k2, k3, d)] = cmul(v[P8(k3)], tw3[k3 * 8 + d]);
    });
}
// thread t = 64 k1 + 8 k2 + k3: DFT8 over d; bin k4 lands at slot P8(k4) of Z
SK_HD void fwd_pass4(int t, const cf *img, cf *Z)
{
    const int k1 = t >> 6, k2 = (t >> 3) & 7, k3 = t & 7;
    SK_UNROLL
    for (int d = 0; d < 8; ++d) Z[d] = img[unit(k1, k2, k3, d)];
    dft8_f(Z);
}

// ---- pointwise product with a pre-permuted, pre-scaled transfer function, BY SLOT -----------------------------
// Hp[j * 256 + t] = (H[k(Q8(2j))], H[k(Q8(2j + 1))]) / N,  k(k4) = k1 + 4 k2 + 32 k3 + 256 k4,  t = 64 k1 + 8 k2 + k3,  j = 0..3.
SK_HD void mul_H(const float4 *hh, const cf *Z, cf *P)
{
    SK_UNROLL
    for (int j = 0; j < 4; ++j) {
        P[2 * j] = cmul(Z[2 * j], lo(hh[j]));
        P[2 * j + 1] = cmul(Z[2 * j + 1], hi(hh[j]));
    }
}
SK_HD void mac_H(const float4 *hh, const cf *Z, cf *A)
{
    SK_UNROLL
    for (int j = 0; j < 4; ++j) {
        A[2 * j] = cadd(A[2 * j], cmul(Z[2 * j], lo(hh[j])));
        A[2 * j + 1] = cadd(A[2 * j + 1], cmul(Z[2 * j + 1], hi(hh[j])));
    }
}

// ---- inverse (each pass in place on one 8-element array) ---------------------------------------------------------
SK_HD void inv_pass4(int t, cf *img, cf *P)
{
    const int k1 = t >> 6, k2 = (t >> 3) & 7, k3 = t & 7;
    dft8_g(P);
    SK_UNROLL
    for (int d = 0; d < 8; ++d) img[unit(k1, k2, k3, d)] = P[d];
}
// LEAN: the eight image reads and seven twiddle reads of a pass are issued in two batches with a fence between them (14 instead
// of 30 landing registers in flight) -- for the kernel that holds twelve results per thread and has no register to spare
#if defined(__HIP_DEVICE_COMPILE__)
#define SK_2K_FENCE() asm volatile("" ::: "memory")
#else
#define SK_2K_FENCE() do {} while (0)
#endif
template <bool LEAN = false> SK_HD void inv_pass3(int t, const cf *tw3, cf *img)
{
    const int k1 = t >> 6, k2 = (t >> 3) & 7, d = t & 7;
    cf v[8];
    v[P8(0)] = img[unit(k1, k2, 0, d)];
    static_for<1, 4>([&](auto kc) {
        constexpr int k3 = decltype(kc)::value;
        v[P8(k3)] = cmulc(img[unit(k1, k2, k3, d)], tw3[k3 * 8 + d]);
    });
    if (LEAN) SK_2K_FENCE();
    static_for<4, 8>([&](auto kc) {
        constexpr int k3 = decltype(kc)::value;
        v[P8(k3)] = cmulc(img[unit(k1, k2, k3, d)], tw3[k3 * 8 + d]);
    });
    dft8_g(v);
    SK_UNROLL
    for (int c = 0; c < 8; ++c) img[unit(k1, k2, c, d)] = v[c];
}
template <bool LEAN = false> SK_HD void inv_pass2(int t, const cf *tw2, cf *img)
{
    const int k1 = t >> 6, c = (t >> 3) & 7, d = t & 7;
    cf v[8];
    v[P8(0)] = img[unit(k1, 0, c, d)];
    static_for<1, 4>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        v[P8(k2)] = cmulc(img[unit(k1, k2, c, d)], tw2[k2 * 64 + (t & 63)]);
    });
    if (LEAN) SK_2K_FENCE();
    static_for<4, 8>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        v[P8(k2)] = cmulc(img[unit(k1, k2, c, d)], tw2[k2 * 64 + (t & 63)]);
    });
    dft8_g(v);
    SK_UNROLL
    for (int b = 0; b < 8; ++b) img[unit(k1, b, c, d)] = v[b];
}
// v[2 a + e] = y[512 a + 256 e + t] out: block m = 2 a + e of the tile holds the samples 256 m .. 256 m + 255
SK_HD void inv_pass1(int t, const cf *tw1, const cf *img, cf *v)
{
    const int b = t >> 6, c = (t >> 3) & 7, d = t & 7;
    v[0] = img[unit(0, b, c, d)];
    v[1] = img[unit(0, b + 4, c, d)];
    static_for<1, 4>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        const cf w = tw1[(k1 - 1) * 256 + t];
        v[2 * k1] = cmulc(img[unit(k1, b, c, d)], w);
        v[2 * k1 + 1] = twmul<8, k1, true>(cmulc(img[unit(k1, b + 4, c, d)], w));
    });
    dft4_ip<true>(v[0], v[2], v[4], v[6]);
    dft4_ip<true>(v[1], v[3], v[5], v[7]);
}

}  // namespace ols2k
}  // namespace skdsp
