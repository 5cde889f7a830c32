// ols_core.hpp -- register-resident FFT building blocks and the per-thread phases of
// the 8192-point overlap-save tile (fir_ols.hip).  Written so that the SAME code
// compiles for the device (hipcc, gfx950) and for the host (g++, used by
// tests/host/ols_emul.cpp to check the index algebra without a GPU).
//
// Tile: N = 8192 complex64 points held by 256 threads x 32 points, factored
//   N = 16 (stride 512)  x  16 (stride 32 inside 512)  x  32
// forward = decimation in frequency, inverse = the exact mirror (decimation in time),
// so no bit-reversal pass exists anywhere: the spectrum lives in a scrambled,
// thread-major order and H is stored pre-permuted to match.
//
//   n = 512 a + rho,   rho = 32 b + c,   c = 2 q + e        (a,b,q in [0,16), e in {0,1})
//   k = k1 + 16 k2 + 256 k3                                 (k1,k2 in [0,16), k3 in [0,32))
//
//   pass 1  thread (b,q)  : DFT16 over a  -> k1, times W_8192^(rho k1)
//   xchg 1  (k1,b,q,e) : thread (b,q) -> thread (k1,q)      [workgroup-wide, one barrier]
//   pass 2  thread (k1,q) : DFT16 over b  -> k2, times W_512^(c k2)
//   xchg 2  (k1,k2,q,e): thread (k1,q) -> thread (k1,k2)    [inside 16-lane groups: wave-local]
//   pass 3  thread (k1,k2): DFT32 over c  -> k3
//   multiply by H (pre-permuted, 1/N folded in), then the mirror image back to (b,q).
//
// LDS image: 16 rows (k1) x 16 x 17 float4 (one float4 = the e=0/e=1 pair); the +1
// pad makes every ds_read_b128 / ds_write_b128 pattern used below bank-conflict free
// (MI355X: 64 banks x 4 B, b128 serviced in 16-lane groups).
#pragma once

#include <type_traits>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SK_HD __host__ __device__ __forceinline__
#define SK_UNROLL _Pragma("unroll")
#else
#include <cmath>
#define SK_UNROLL
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#define SK_HD inline
#endif

namespace skdsp {
namespace ols {

typedef float2 cf;

constexpr int kN = 8192;
constexpr int kThreads = 256;
constexpr int kRowPitch = 272;             // float4 units per k1 row (16 x 17)
constexpr int kLdsUnits = 16 * kRowPitch;  // 4352 float4 = 69632 B  (exchange image)
constexpr int kT2Units = 256;              // one 16 x 16 float4 twiddle table = 4 KiB (two copies follow the image)

// cos/sin(2 pi k / 32), k = 0..31, rounded from float64
#define SK_C32                                                                                             \
    {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f,        \
     0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f, 0.0f, -0.19509032201612819f,       \
     -0.38268343236508973f, -0.55557023301960196f, -0.70710678118654746f, -0.83146961230254535f,           \
     -0.92387953251128674f, -0.98078528040323043f, -1.0f, -0.98078528040323043f, -0.92387953251128685f,    \
     -0.83146961230254546f, -0.70710678118654768f, -0.55557023301960218f, -0.38268343236509034f,           \
     -0.19509032201612866f, 0.0f, 0.19509032201612828f, 0.38268343236509f, 0.55557023301960184f,           \
     0.70710678118654735f, 0.83146961230254524f, 0.92387953251128652f, 0.98078528040323032f}
#define SK_S32                                                                                             \
    {0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f,         \
     0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f,         \
     0.92387953251128674f, 0.83146961230254546f, 0.70710678118654757f, 0.55557023301960218f,               \
     0.38268343236508989f, 0.19509032201612861f, 0.0f, -0.19509032201612836f, -0.38268343236508967f,       \
     -0.55557023301960196f, -0.70710678118654746f, -0.83146961230254524f, -0.92387953251128652f,           \
     -0.98078528040323032f, -1.0f, -0.98078528040323043f, -0.92387953251128663f, -0.83146961230254546f,    \
     -0.70710678118654768f, -0.55557023301960218f, -0.38268343236509039f, -0.19509032201612872f}

// ----------------------------------------------------------------------------
// Complex arithmetic.  On the device every operation is ONE or TWO packed-f32
// instructions on the (re,im) register pair: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32
// with op_sel (half swap / broadcast) and neg_lo / neg_hi source modifiers doing the
// swaps and sign flips of complex multiplication for free.  gfx950 issues a wave64
// VALU instruction in 4 cycles whether it is packed or not (the FP32 peak needs packed
// math), and the scalar-f32 version of this kernel was VALU-bound (78 % busy), so
// halving the instruction count is the lever.  hipcc's own SLP packing cannot use the
// modifiers and paid ~800 v_mov per tile instead.  Host build: plain C++ (same maths).
// ----------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef float v2f __attribute__((ext_vector_type(2)));
SK_HD v2f V(cf a) { v2f r; r.x = a.x; r.y = a.y; return r; }
SK_HD cf C(v2f a) { return make_float2(a.x, a.y); }
SK_HD cf cadd(cf a, cf b) { return C(V(a) + V(b)); }
SK_HD cf csub(cf a, cf b) { return C(V(a) - V(b)); }
// a * w   (two instructions in ONE asm statement: hipcc pads every asm boundary with s_nop)
SK_HD cf cmul(cf a, cf w)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(V(a)), "v"(V(w)));
    return C(r);
}
// a * conj(w)
SK_HD cf cmulc(cf a, cf w)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=&v"(r) : "v"(V(a)), "v"(V(w)));
    return C(r);
}
// a * (c - i s) forward, a * (c + i s) inverse; (c, s) compile-time -> one SGPR pair
template <bool INV> SK_HD cf cmul_k(cf a, float c, float s)
{
    v2f K;
    K.x = c; K.y = s;
    v2f r;
    if (INV)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
            : "=&v"(r) : "v"(V(a)), "s"(K));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
            : "=&v"(r) : "v"(V(a)), "s"(K));
    return C(r);
}
// p + (-i) d forward, p + (+i) d inverse
template <bool INV> SK_HD cf madd_mi(cf p, cf d)
{
    v2f r;
    if (INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(V(p)), "v"(V(d)));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(V(p)), "v"(V(d)));
    return C(r);
}
SK_HD cf cneg(cf a) { return C(-V(a)); }
#else
SK_HD cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
SK_HD cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
SK_HD cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
SK_HD cf cmulc(cf a, cf b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
template <bool INV> SK_HD cf cmul_k(cf a, float c, float s)
{
    return INV ? make_float2(a.x * c - a.y * s, a.y * c + a.x * s) : make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
}
template <bool INV> SK_HD cf madd_mi(cf p, cf d)
{
    return INV ? make_float2(p.x - d.y, p.y + d.x) : make_float2(p.x + d.y, p.y - d.x);
}
SK_HD cf cneg(cf a) { return make_float2(-a.x, -a.y); }
#endif
// ----------------------------------------------------------------------------
// Two columns at once.  A thread owns the columns e = 0 / e = 1 of every row; `cf2` holds one complex value of EACH in
// structure-of-arrays form: re = (re of column 0, re of column 1), im likewise.  The butterflies are the same packed
// instructions as on (re, im) pairs -- an add is two v_pk_add_f32 for two columns, a constant twiddle four v_pk_* for two
// columns -- so a DFT costs what two single-column DFTs cost.  Why it exists: a float32 signal rides two REAL tiles
// A / B per complex tile (re = A, im = B), and an 8-byte load of A delivers exactly one `re` pair, of B one `im` pair:
// the loaded registers ARE the cf2 operands, with no 2 x 2 register transposition behind the loads (which hipcc placed
// right behind the prefetch and waited for: the float32 tile ran without a prefetch).  The per-column twiddle multiply
// at the end of pass 1 reads its column through op_sel and writes an ordinary (re, im) pair, so the way back is free.
// ----------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
struct cf2 { v2f re, im; };
SK_HD cf2 make_cf2(cf re_pair, cf im_pair) { return cf2{V(re_pair), V(im_pair)}; }
SK_HD cf2 cadd(cf2 a, cf2 b) { return cf2{a.re + b.re, a.im + b.im}; }
SK_HD cf2 csub(cf2 a, cf2 b) { return cf2{a.re - b.re, a.im - b.im}; }
SK_HD cf2 cneg(cf2 a) { return cf2{-a.re, -a.im}; }
template <bool INV> SK_HD cf2 madd_mi(cf2 p, cf2 d) { return INV ? cf2{p.re - d.im, p.im + d.re} : cf2{p.re + d.im, p.im - d.re}; }
template <bool INV> SK_HD cf2 cmul_k(cf2 a, float c, float s)
{
    return INV ? cf2{a.re * c - a.im * s, a.im * c + a.re * s} : cf2{a.re * c + a.im * s, a.im * c - a.re * s};
}
SK_HD cf column(cf2 a, int e) { return e ? make_float2(a.re.y, a.im.y) : make_float2(a.re.x, a.im.x); }
// column COL of a, times w: an ordinary (re, im) pair out
template <int COL> SK_HD cf soa_cmul(cf2 a, cf w)
{
    v2f r;
    if (COL == 0)
        asm("v_pk_mul_f32 %0, %1, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
            "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]"
            : "=&v"(r) : "v"(a.re), "v"(a.im), "v"(V(w)));
    else
        asm("v_pk_mul_f32 %0, %1, %3 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
            "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
            : "=&v"(r) : "v"(a.re), "v"(a.im), "v"(V(w)));
    return C(r);
}
#else
struct cf2 { float re[2], im[2]; };
SK_HD cf2 make_cf2(cf re_pair, cf im_pair) { return cf2{{re_pair.x, re_pair.y}, {im_pair.x, im_pair.y}}; }
SK_HD cf2 cadd(cf2 a, cf2 b) { return cf2{{a.re[0] + b.re[0], a.re[1] + b.re[1]}, {a.im[0] + b.im[0], a.im[1] + b.im[1]}}; }
SK_HD cf2 csub(cf2 a, cf2 b) { return cf2{{a.re[0] - b.re[0], a.re[1] - b.re[1]}, {a.im[0] - b.im[0], a.im[1] - b.im[1]}}; }
SK_HD cf2 cneg(cf2 a) { return cf2{{-a.re[0], -a.re[1]}, {-a.im[0], -a.im[1]}}; }
SK_HD cf column(cf2 a, int e) { return make_float2(a.re[e], a.im[e]); }
SK_HD cf2 from_columns(cf c0, cf c1) { return cf2{{c0.x, c1.x}, {c0.y, c1.y}}; }
template <bool INV> SK_HD cf2 madd_mi(cf2 p, cf2 d) { return from_columns(madd_mi<INV>(column(p, 0), column(d, 0)), madd_mi<INV>(column(p, 1), column(d, 1))); }
template <bool INV> SK_HD cf2 cmul_k(cf2 a, float c, float s) { return from_columns(cmul_k<INV>(column(a, 0), c, s), cmul_k<INV>(column(a, 1), c, s)); }
template <int COL> SK_HD cf soa_cmul(cf2 a, cf w) { return cmul(column(a, COL), w); }
#endif

// p - (-i) d forward == p + (+i) d
template <bool INV, class E> SK_HD E msub_mi(E p, E d) { return madd_mi<!INV>(p, d); }
// a * (-i) forward, a * (+i) inverse
template <bool INV> SK_HD cf mul_mi(cf a) { return madd_mi<INV>(make_float2(0.f, 0.f), a); }
template <bool INV> SK_HD cf2 mul_mi(cf2 a) { return madd_mi<INV>(make_cf2(make_float2(0.f, 0.f), make_float2(0.f, 0.f)), a); }

// a * W_N^K  (forward, W = exp(-2 pi i / N)) or a * conj(W_N^K) (INV); K compile-time
template <int N, int K, bool INV, class E> SK_HD E twmul(E a)
{
    constexpr int k32 = ((K % N) * (32 / N)) & 31;
    constexpr float Ct[32] = SK_C32;
    constexpr float St[32] = SK_S32;
    if constexpr (k32 == 0) return a;
    else if constexpr (k32 == 8) return mul_mi<INV>(a);
    else if constexpr (k32 == 16) return cneg(a);
    else if constexpr (k32 == 24) return mul_mi<!INV>(a);
    else return cmul_k<INV>(a, Ct[k32], St[k32]);
}

// Xa = E + W_N^K O,  Xb = E - W_N^K O   (the -i / +i cases fold into the add)
template <int N, int K, bool INV, class T> SK_HD void bfly_tw(T E, T O, T &Xa, T &Xb)
{
    constexpr int k32 = ((K % N) * (32 / N)) & 31;
    if constexpr (k32 == 8) {
        Xa = madd_mi<INV>(E, O);
        Xb = msub_mi<INV>(E, O);
    } else if constexpr (k32 == 24) {
        Xa = msub_mi<INV>(E, O);
        Xb = madd_mi<INV>(E, O);
    } else {
        const T t = twmul<N, K, INV>(O);
        Xa = cadd(E, t);
        Xb = csub(E, t);
    }
}

// compile-time loop
template <int I, int E, class F> SK_HD void static_for(F &&f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(static_cast<F &&>(f));
    }
}

// Out-of-place N-point DFT of x[0], x[S], x[2S], ... into X[0..N) (natural order).
// INV = unnormalised inverse.  N in {1,2,4,8,16,32}.  Everything is unrolled at
// compile time, so x/X live in registers.
template <class T, int N, int S, bool INV> struct DftT {
    static SK_HD void run(const T *x, T *X)
    {
        if constexpr (N == 1) {
            X[0] = x[0];
        } else if constexpr (N == 2) {
            X[0] = cadd(x[0], x[S]);
            X[1] = csub(x[0], x[S]);
        } else if constexpr (N == 4) {
            const T s02 = cadd(x[0], x[2 * S]), d02 = csub(x[0], x[2 * S]);
            const T s13 = cadd(x[S], x[3 * S]), d13 = csub(x[S], x[3 * S]);
            X[0] = cadd(s02, s13);
            X[2] = csub(s02, s13);
            X[1] = madd_mi<INV>(d02, d13);
            X[3] = msub_mi<INV>(d02, d13);
        } else if constexpr (N == 8) {
            T E[4], O[4];
            DftT<T, 4, 2 * S, INV>::run(x, E);
            DftT<T, 4, 2 * S, INV>::run(x + S, O);
            static_for<0, 4>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                bfly_tw<8, k, INV>(E[k], O[k], X[k], X[k + 4]);
            });
        } else {
            constexpr int M = N / 4;
            T S0[M], S1[M], S2[M], S3[M];
            DftT<T, M, 4 * S, INV>::run(x, S0);
            DftT<T, M, 4 * S, INV>::run(x + S, S1);
            DftT<T, M, 4 * S, INV>::run(x + 2 * S, S2);
            DftT<T, M, 4 * S, INV>::run(x + 3 * S, S3);
            static_for<0, M>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const T t0 = S0[k];
                const T t1 = twmul<N, k, INV>(S1[k]);
                const T t2 = twmul<N, 2 * k, INV>(S2[k]);
                const T t3 = twmul<N, 3 * k, INV>(S3[k]);
                const T s02 = cadd(t0, t2), d02 = csub(t0, t2);
                const T s13 = cadd(t1, t3), d13 = csub(t1, t3);
                X[k] = cadd(s02, s13);
                X[k + 2 * M] = csub(s02, s13);
                X[k + M] = madd_mi<INV>(d02, d13);
                X[k + 3 * M] = msub_mi<INV>(d02, d13);
            });
        }
    }
};
template <int N, int S, bool INV> using Dft = DftT<cf, N, S, INV>;

// ----------------------------------------------------------------------------
// LDS addressing (float4 units)
// ----------------------------------------------------------------------------
SK_HD int lds_unit(int row_k1, int mid, int q) { return row_k1 * kRowPitch + mid * 17 + q; }

SK_HD cf lo(float4 f) { return make_float2(f.x, f.y); }
SK_HD cf hi(float4 f) { return make_float2(f.z, f.w); }
SK_HD float4 pack(cf a, cf b) { return make_float4(a.x, a.y, b.x, b.y); }

// ----------------------------------------------------------------------------
// Per-thread phases.  v[a*2+e] on entry to fwd_pass1 holds x[512 a + 2 t + e].
// T1[k1*256 + t] = (W_8192^((2t)k1), W_8192^((2t+1)k1));  T2[k2*16 + q] = (W_512^((2q)k2), W_512^((2q+1)k2)).
// ----------------------------------------------------------------------------
// W_8192^K (forward) or its conjugate, K compile-time, from float64-rounded literals
template <int K, bool INV> SK_HD cf mul_w8192(cf a)
{
    // cos/sin(2 pi k / 8192), k = 0..15
    constexpr float C[16] = {1.0f, 0.99999970586288223f, 0.99999882345170188f, 0.99999735276697821f, 0.99999529380957619f, 0.99999264658070719f, 0.9999894110819284f, 0.9999855873151432f, 0.99998117528260111f, 0.99997617498689761f, 0.99997058643097414f, 0.99996440961811828f, 0.9999576445519639f, 0.99995029123649048f, 0.99994234967602391f, 0.999933819875236f};
    constexpr float S[16] = {0.0f, 0.00076699031874270449f, 0.0015339801862847655f, 0.002300969151425805f, 0.0030679567629659761f, 0.0038349425697062275f, 0.0046019261204485705f, 0.0053689069639963425f, 0.0061358846491544753f, 0.0069028587247297558f, 0.007669828739531097f, 0.0084367942423697988f, 0.0092037547820598194f, 0.0099707099074180308f, 0.010737659167264491f, 0.011504602110422714f};
    // plain scalar arithmetic on purpose: the 15 (cos, sin) pairs then travel as 32-bit LITERALS of v_mul_f32 / v_fmamk_f32
    // (4 instructions per product) instead of as 30 more loop-invariant SGPRs next to the DFT twiddles -- with them the
    // kernel needed ~130 SGPRs, hipcc parked the surplus in VGPR lanes and paid two v_readlane per use of ANY spilled
    // constant (208 per tile).
    if constexpr (K == 0) return a;
    else if constexpr (INV) return make_float2(a.x * C[K] - a.y * S[K], a.y * C[K] + a.x * S[K]);
    else return make_float2(a.x * C[K] + a.y * S[K], a.y * C[K] - a.x * S[K]);
}

// pass 1 + twiddle + exchange-1 write.  thread t = 16 b + q.
// tw[k1] = W_4096^(t k1) (k1 = 1..15; tw[0] unused) lives in registers for the whole
// persistent kernel; the odd column's twiddle W_8192^((2t+1)k1) = tw[k1] * W_8192^k1.
SK_HD void fwd_pass1(int t, const cf *v, const cf *tw, float4 *lds)
{
    cf in[16], o0[16], o1[16];
    SK_UNROLL
    for (int a = 0; a < 16; ++a) in[a] = v[2 * a];
    Dft<16, 1, false>::run(in, o0);
    SK_UNROLL
    for (int a = 0; a < 16; ++a) in[a] = v[2 * a + 1];
    Dft<16, 1, false>::run(in, o1);
    const int b = t >> 4, q = t & 15;
    lds[lds_unit(0, b, q)] = pack(o0[0], o1[0]);
    static_for<1, 16>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        lds[lds_unit(k1, b, q)] = pack(cmul(o0[k1], tw[k1]), cmul(mul_w8192<k1, false>(o1[k1]), tw[k1]));
    });
}

// The same for a float32 signal riding two real tiles A / B (re = A, im = B): v[2a] = (A[512a+2t], A[512a+2t+1]) and
// v[2a+1] = (B[512a+2t], B[512a+2t+1]) exactly as the 8-byte loads deliver them -- the (column 0, column 1) pairs of the
// real and of the imaginary parts, i.e. cf2 operands.  Both columns go through ONE two-column DFT16; the twiddle
// multiply picks its column through op_sel and leaves (re, im) pairs for the exchange image.
SK_HD void fwd_pass1_real(int t, const cf *v, const cf *tw, float4 *lds)
{
    cf2 in[16], o[16];
    SK_UNROLL
    for (int a = 0; a < 16; ++a) in[a] = make_cf2(v[2 * a], v[2 * a + 1]);
    DftT<cf2, 16, 1, false>::run(in, o);
    const int b = t >> 4, q = t & 15;
    lds[lds_unit(0, b, q)] = pack(column(o[0], 0), column(o[0], 1));
    static_for<1, 16>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        lds[lds_unit(k1, b, q)] = pack(soa_cmul<0>(o[k1], tw[k1]), mul_w8192<k1, false>(soa_cmul<1>(o[k1], tw[k1])));
    });
}

// exchange-1 read + pass 2 + twiddle + exchange-2 write/read + pass 3.
// thread t = 16 k1 + q for pass 2 and t = 16 k1 + k2 for pass 3.  Z[k3] out (32).
SK_HD void fwd_pass23(int t, const float4 *T2 /* LDS copy [k2][q] */, float4 *lds, cf *Z)
{
    const int k1 = t >> 4, q = t & 15;
    cf in0[16], in1[16], o0[16], o1[16];
    SK_UNROLL
    for (int b = 0; b < 16; ++b) {
        const float4 f = lds[lds_unit(k1, b, q)];
        in0[b] = lo(f);
        in1[b] = hi(f);
    }
    Dft<16, 1, false>::run(in0, o0);
    Dft<16, 1, false>::run(in1, o1);
    lds[lds_unit(k1, 0, q)] = pack(o0[0], o1[0]);
    SK_UNROLL
    for (int k2 = 1; k2 < 16; ++k2) {
        const float4 w = T2[k2 * 16 + q];
        lds[lds_unit(k1, k2, q)] = pack(cmul(o0[k2], lo(w)), cmul(o1[k2], hi(w)));
    }
    // (wave-local: the 16 lanes of this k1 row only read what they wrote)
    const int k2 = q;
    cf z[32];
    SK_UNROLL
    for (int qq = 0; qq < 16; ++qq) {
        const float4 f = lds[lds_unit(k1, k2, qq)];
        z[2 * qq] = lo(f);
        z[2 * qq + 1] = hi(f);
    }
    Dft<32, 1, false>::run(z, Z);
}

// pointwise multiply by the pre-permuted, pre-scaled transfer function.
// Hp[j*256 + t] = (H[k(k3=2j)], H[k(k3=2j+1)]),  k = k1 + 16 k2 + 256 k3, t = 16 k1 + k2.
SK_HD void load_H(int t, const float4 *Hp, float4 *hh)
{
    SK_UNROLL
    for (int j = 0; j < 16; ++j) hh[j] = Hp[j * 256 + t];
}
SK_HD void mul_H(const float4 *hh, cf *Z)
{
    SK_UNROLL
    for (int j = 0; j < 16; ++j) {
        Z[2 * j] = cmul(Z[2 * j], lo(hh[j]));
        Z[2 * j + 1] = cmul(Z[2 * j + 1], hi(hh[j]));
    }
}

// inverse pass 3 + conj twiddle + exchange-2' + inverse pass 2 + exchange-1' write.
SK_HD void inv_pass32(int t, const float4 *T2t /* LDS copy, transposed [qq][k2] */, float4 *lds, const cf *Z)
{
    const int k1 = t >> 4, k2 = t & 15;
    cf z[32];
    Dft<32, 1, true>::run(Z, z);
    SK_UNROLL
    for (int qq = 0; qq < 16; ++qq) {
        const float4 w = T2t[qq * 16 + k2];
        lds[lds_unit(k1, k2, qq)] = pack(cmulc(z[2 * qq], lo(w)), cmulc(z[2 * qq + 1], hi(w)));
    }
    const int q = k2;  // now thread (k1,q)
    cf in0[16], in1[16], o0[16], o1[16];
    SK_UNROLL
    for (int kk = 0; kk < 16; ++kk) {
        const float4 f = lds[lds_unit(k1, kk, q)];
        in0[kk] = lo(f);
        in1[kk] = hi(f);
    }
    Dft<16, 1, true>::run(in0, o0);
    Dft<16, 1, true>::run(in1, o1);
    SK_UNROLL
    for (int b = 0; b < 16; ++b) lds[lds_unit(k1, b, q)] = pack(o0[b], o1[b]);
}

// exchange-1' read + conj twiddle + inverse pass 1.  v[a*2+e] = y[512 a + 2 t + e] out.
SK_HD void inv_pass1(int t, const cf *tw, const float4 *lds, cf *v)
{
    const int b = t >> 4, q = t & 15;
    cf in0[16], in1[16], o0[16], o1[16];
    {
        const float4 f = lds[lds_unit(0, b, q)];
        in0[0] = lo(f);
        in1[0] = hi(f);
    }
    static_for<1, 16>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        const float4 f = lds[lds_unit(k1, b, q)];
        in0[k1] = cmulc(lo(f), tw[k1]);
        in1[k1] = mul_w8192<k1, true>(cmulc(hi(f), tw[k1]));
    });
    Dft<16, 1, true>::run(in0, o0);
    Dft<16, 1, true>::run(in1, o1);
    SK_UNROLL
    for (int a = 0; a < 16; ++a) {
        v[2 * a] = o0[a];
        v[2 * a + 1] = o1[a];
    }
}

// ----------------------------------------------------------------------------
// The inverse transform of a DECIMATING filter (multirate_FIR.dn, M = MF in {2, 4, 8, 16}): only y[j MF] is wanted.
// With n = 512 a + 32 b + c those are the outputs whose c is a multiple of MF, and a DFT32 over k3 evaluated at c = c' MF only is
// the DFT of R = 32 / MF points of the FOLDED spectrum  F[k3'] = sum_m Z[k3' + R m]  -- adds between registers of ONE thread, because
// k3 is the index a thread (k1, k2) holds.  What follows is the full-rate inverse restricted to the columns it still needs: c = 2 q + e
// with e = 0 and q a multiple of MF / 2, i.e. ONE 16-point transform per lane and pass where the full inverse runs two, on every
// (MF / 2)-th lane.  Forward transform, H product, loads: untouched.  Per tile: 1 forward + ~1/4 inverse transform instead of 2.
// ----------------------------------------------------------------------------
template <int MF> SK_HD void inv_pass32_fold(int t, const float4 *T2t /* LDS copy, transposed [qq][k2] */, float4 *lds, const cf *Z)
{
    constexpr int R = 32 / MF, LS = MF / 2;   // points of the folded transform; lane step of the columns in use
    const int k1 = t >> 4, k2 = t & 15;
    cf F[R], z[R];
    SK_UNROLL
    for (int k = 0; k < R; ++k) {
        cf acc = Z[k];
        SK_UNROLL
        for (int m = 1; m < MF; ++m) acc = cadd(acc, Z[k + R * m]);
        F[k] = acc;
    }
    Dft<R, 1, true>::run(F, z);   // z[c'] = the full inverse DFT32 at c = c' MF
    cf *l2 = reinterpret_cast<cf *>(lds);   // (the first half of a float4 unit: column e = 0)
    SK_UNROLL
    for (int cp = 0; cp < R; ++cp) {
        const int qq = cp * LS;
        l2[2 * lds_unit(k1, k2, qq)] = cmulc(z[cp], lo(T2t[qq * 16 + k2]));
    }
    const int q = k2;  // now thread (k1, q): its column exists where LS divides q
    if (q % LS == 0) {
        cf in0[16], o0[16];
        SK_UNROLL
        for (int kk = 0; kk < 16; ++kk) in0[kk] = l2[2 * lds_unit(k1, kk, q)];
        Dft<16, 1, true>::run(in0, o0);
        SK_UNROLL
        for (int b = 0; b < 16; ++b) l2[2 * lds_unit(k1, b, q)] = o0[b];
    }
}

// exchange-1' read + conj twiddle + inverse pass 1 of the same: v[a] = y[512 a + 32 b + 2 q] out (threads whose q is a multiple of MF / 2)
template <int MF> SK_HD void inv_pass1_fold(int t, const cf *tw, const float4 *lds, cf *v)
{
    constexpr int LS = MF / 2;
    const int b = t >> 4, q = t & 15;
    if (q % LS != 0) return;
    const cf *l2 = reinterpret_cast<const cf *>(lds);
    cf in0[16];
    in0[0] = l2[2 * lds_unit(0, b, q)];
    static_for<1, 16>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        in0[k1] = cmulc(l2[2 * lds_unit(k1, b, q)], tw[k1]);
    });
    Dft<16, 1, true>::run(in0, v);
}

// ----------------------------------------------------------------------------
// The forward transform of an INTERPOLATING filter (multirate_FIR.up, L = LF in {2, 4, 8, 16}): the tile holds the zero-stuffed signal, whose
// samples are zero wherever c (n = 512 a + 32 b + c) is not a multiple of LF.  The mirror image of the folded inverse above: passes 1 and 2 run on
// the columns that can be non-zero only (c = 2 q, q a multiple of LF / 2: one 16-point transform per lane where the full forward runs two, on every
// (LF / 2)-th lane), and the DFT32 over c of R = 32 / LF non-zero points is their R-point DFT, REPLICATED LF-fold over k3 -- copies between the
// registers of one thread.  H product, inverse transform, stores: the plain filter's.  Per tile: ~1/4 forward + 1 inverse transform instead of 2.
// ----------------------------------------------------------------------------
// in[a] = x_up[512 a + 32 b + 2 q] (threads whose q is a multiple of LF / 2; the others have nothing to do here)
template <int LF> SK_HD void fwd_pass1_rep(int t, const cf *in, const cf *tw, float4 *lds)
{
    constexpr int LS = LF / 2;
    const int b = t >> 4, q = t & 15;
    if (q % LS != 0) return;
    cf o0[16];
    Dft<16, 1, false>::run(in, o0);
    cf *l2 = reinterpret_cast<cf *>(lds);   // (the first half of a float4 unit: column e = 0)
    l2[2 * lds_unit(0, b, q)] = o0[0];
    static_for<1, 16>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        l2[2 * lds_unit(k1, b, q)] = cmul(o0[k1], tw[k1]);
    });
}
// exchange-1 read + pass 2 + twiddle + exchange-2 + the R-point pass 3, replicated.  Z[k3] out (32).
template <int LF> SK_HD void fwd_pass23_rep(int t, const float4 *T2 /* LDS copy [k2][q] */, float4 *lds, cf *Z)
{
    constexpr int R = 32 / LF, LS = LF / 2;
    const int k1 = t >> 4, q = t & 15;
    cf *l2 = reinterpret_cast<cf *>(lds);
    if (q % LS == 0) {
        cf in0[16], o0[16];
        SK_UNROLL
        for (int b = 0; b < 16; ++b) in0[b] = l2[2 * lds_unit(k1, b, q)];
        Dft<16, 1, false>::run(in0, o0);
        l2[2 * lds_unit(k1, 0, q)] = o0[0];
        SK_UNROLL
        for (int k2 = 1; k2 < 16; ++k2) l2[2 * lds_unit(k1, k2, q)] = cmul(o0[k2], lo(T2[k2 * 16 + q]));
    }
    // (wave-local: the 16 lanes of this k1 row only read what they wrote)
    const int k2 = q;
    cf z[R], F[R];
    SK_UNROLL
    for (int cp = 0; cp < R; ++cp) z[cp] = l2[2 * lds_unit(k1, k2, cp * LS)];
    Dft<R, 1, false>::run(z, F);
    SK_UNROLL
    for (int k3 = 0; k3 < 32; ++k3) Z[k3] = F[k3 % R];
}

}  // namespace ols
}  // namespace skdsp
